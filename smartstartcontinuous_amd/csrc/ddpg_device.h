// ddpg_device.h -- pieces shared by the DDPG learner kernels (ddpg_train.hip: single-workgroup step interpreter;
// ddpg_train_fixed.hip: the shipped 64-32 shape compiled straight-line; ddpg_train_wide.hip: any shape, multi-workgroup).
#pragma once

#include "ssc_device.h"

namespace ssc {

constexpr int kB = 64;       // batch size = lanes of a wave
constexpr int kP = kB + 4;   // padded LDS row: 16-B aligned rows (float4 access over 4 samples); a stride of
                             // 68 dwords keeps both ds_read_b128 column gathers and ds_write_b128 conflict-free

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f32x4m __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: exact fp32 products and sums.  Lane l of a wave holds A[row l & 15][k = l >> 4],
// B[k = l >> 4][col l & 15], D[row 4 (l >> 4) + r][col l & 15].
__device__ __forceinline__ f32x4m mfma4(float a, float b, const f32x4m &c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// beta^n for an integer step count, by squaring in f64 (ocml's pow() alone is several thousand instructions)
static __device__ __noinline__ double ipow(double base, int n) {
    double r = 1.0;
    while (n > 0) {
        if (n & 1) r *= base;
        base *= base;
        n >>= 1;
    }
    return r;
}

struct AdamCfg {
    float a, beta1, beta2, eps;   // a = stepsize * sqrt(1 - b2^t) / (1 - b1^t) with t already incremented
};

// ddpg_train_fixed.hip
bool ddpg_fixed_shape(const ssc_ddpg_desc *d);
int ddpg_train_fixed(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx, int32_t n_iters,
                     float *d_losses, hipStream_t stream);

// the shipped shape at batches of 128, 192, ... (multiples of 64): the straight-line kernel on one 64-row tile per workgroup
// (gradients only), then the multi-workgroup apply pass of ddpg_train_wide.hip
bool ddpg_fixed_tiled_shape(const ssc_ddpg_desc *d);
int ddpg_train_fixed_tiled(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx, int32_t n_iters,
                           float *d_losses, void *d_workspace, size_t workspace_bytes, hipStream_t stream);

// ddpg_train_wide.hip: any layer sizes / batch sizes, batch tiled over workgroups
size_t ddpg_wide_workspace_bytes(const ssc_ddpg_desc *d);
// The apply pass over `n_blocks` gradient partials laid out [n_blocks][n_params] at the start of the workspace, loss
// partials [n_blocks][2] behind them (ddpg_wide_partials); iteration `it` of the call; `ddpg_wide_finish` moves the
// MpiAdam step counters once, after the last iteration.
struct WidePartials { float *gpart, *lpart; };
WidePartials ddpg_wide_partials(const ssc_ddpg_desc *d, void *d_workspace, int n_blocks);
void ddpg_wide_apply(const ssc_ddpg_desc *d, void *d_workspace, int n_blocks, int it, float *d_losses_it, hipStream_t stream);
void ddpg_wide_finish(const ssc_ddpg_desc *d, int32_t n_iters, hipStream_t stream);
int ddpg_train_wide(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx, int32_t n_iters,
                    float *d_losses, void *d_workspace, size_t workspace_bytes, hipStream_t stream);

}  // namespace ssc
