// Diagnostic variant of the simulation kernel: in-kernel clock stamps (tools/exp_dyn_clock.py).  The RESULTS of a library built
// with this translation unit are wrong by design (the stamps overwrite S); it is linked into tools/_build/libssc_clk.so only.
#define SSC_DYN_STAMPS 1
#include "../../smartstartcontinuous_amd/csrc/dyn_mfma.hip"
