// Diagnostic variant of the multi-workgroup learner: cycle stamps per level, 2 pass(es) over the level sequence
// (tools/exp_ddpg_wide_phases.py); results wrong by design.
#define SSC_WIDE_DIAG 2
#include "../../smartstartcontinuous_amd/csrc/ddpg_train_wide.hip"
