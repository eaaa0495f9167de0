// Diagnostic variant of the 64-32 learner: cycle stamps per level (tools/exp_ddpg_phases.py); results wrong by design.
#define SSC_DDPG_DIAG 1
#include "../../smartstartcontinuous_amd/csrc/ddpg_train_fixed.hip"
