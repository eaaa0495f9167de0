// Diagnostic variant of the step-interpreter learner: cycle stamps per step (tools/exp_ddpg_phases.py); results wrong by design.
#define SSC_DDPG_DIAG 1
#include "../../smartstartcontinuous_amd/csrc/ddpg_train.hip"
