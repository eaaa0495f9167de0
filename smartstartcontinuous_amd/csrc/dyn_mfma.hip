// dyn_mfma.hip -- bf16-MFMA path of the NND_MB dynamics model (placeholder until the fused
// kernel lands: reports "unsupported" so that callers fall back to SSC_PREC_F32 explicitly).
#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

bool dyn_mfma_supported(const ssc_mlp_desc *, int, int) { return false; }
size_t dyn_mfma_workspace_bytes(const ssc_mlp_desc *) { return 256; }
int dyn_mfma_forward_sim(const ssc_mlp_desc *, const ssc_norm *, int64_t, int32_t, int32_t, int32_t, const float *,
                         int64_t, const float *, float *, void *, hipStream_t) {
    return set_error(SSC_EUNSUPPORTED, "bf16 MFMA dynamics path not built");
}
int dyn_mfma_mlp_forward(const ssc_mlp_desc *, int64_t, const float *, float *, void *, hipStream_t) {
    return set_error(SSC_EUNSUPPORTED, "bf16 MFMA dynamics path not built");
}

}  // namespace ssc
