// actor.hip -- batched DDPG actor forward, act[m][act_dim] = Actor_Editted(obs[m][obs_dim])
// (DDPG_Baselines_editted/models_editted.py:38-61).  Used by DDPG_Baselines_agent.get_action
// on observation batches and by tests; the rollout hot path uses the same device code
// (actor_device.h) fused into rollout.hip.
#include "actor_device.h"
#include "ssc_host.h"

namespace ssc {

// canonical sizes: one row per lane, fp32 VALU
template <int OBS, int H1, int H2>
__global__ __launch_bounds__(kBlock) void actor_f32_kernel(ActorWeights w, int64_t m, const float *__restrict__ obs,
                                                           float *__restrict__ act) {
    const int64_t gi = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = gi < m;
    const int64_t i = active ? gi : m - 1;
    float o[OBS];
#pragma unroll
    for (int c = 0; c < OBS; ++c) o[c] = clip_obs(obs[i * OBS + c], w.obs_clip);
    ActorF32<OBS, H1, H2> net;
    net.init(w);  // block-cooperative LDS staging: no thread may have exited
    const float a = net.forward(o);
    if (active) act[i] = a;
}

// one wave = 64 rows, hidden GEMM on bf16 MFMA
template <int OBS, int UT, int JT>
__global__ __launch_bounds__(kBlock) void actor_mfma_kernel(ActorWeights w, int64_t m, const float *__restrict__ obs,
                                                            float *__restrict__ act) {
    const int64_t gi = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = gi < m;
    const int64_t i = active ? gi : m - 1;  // whole waves must run the collective forward
    float o[OBS];
#pragma unroll
    for (int c = 0; c < OBS; ++c) o[c] = clip_obs(obs[i * OBS + c], w.obs_clip);
    ActorMfma<OBS, UT, JT> net;
    net.init(w);
    const float a = net.forward(o);
    if (active) act[i] = a;
}

// wide shapes (h1 <= 224, h2 <= 128): W2 fragments staged in LDS
template <int OBS, int UT, int JT>
__global__ __launch_bounds__(kBlock) void actor_mfma_lds_kernel(ActorWeights w, int64_t m, const float *__restrict__ obs,
                                                                float *__restrict__ act) {
    const int64_t gi = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool active = gi < m;
    const int64_t i = active ? gi : m - 1;  // whole waves must run the collective forward
    float o[OBS];
#pragma unroll
    for (int c = 0; c < OBS; ++c) o[c] = clip_obs(obs[i * OBS + c], w.obs_clip);
    ActorMfmaLds<OBS, UT, JT> net;
    net.init(w);
    const float a = net.forward(o);
    if (active) act[i] = a;
}

// any sizes (obs_dim <= SSC_MAX_STATE, act_dim <= SSC_MAX_ACT, h1 <= 512): one row per lane,
// layer-1 activations parked in LDS as [unit][lane] (conflict-free), layer 2 folded into
// layer 3 on the fly.  64-thread blocks; LDS = h1 * 256 B.
__global__ __launch_bounds__(64) void actor_generic_kernel(ActorWeights w, int act_dim, int64_t m,
                                                           const float *__restrict__ obs, float *__restrict__ act) {
    extern __shared__ float h1s[];  // [h1][64] (+ [h2][64] with LayerNorm: layer 2 is normalised over all its units)
    const int lane = threadIdx.x;
    const int64_t gi = (int64_t)blockIdx.x * 64 + lane;
    const bool active = gi < m;
    const int64_t i = active ? gi : m - 1;
    const bool ln = w.ln1_g != nullptr;   // models_editted.py:45-46, 50-51
    float o[SSC_MAX_STATE];
#pragma unroll
    for (int c = 0; c < SSC_MAX_STATE; ++c) o[c] = (c < w.obs_dim) ? clip_obs(obs[i * w.obs_dim + c], w.obs_clip) : 0.0f;
    for (int j = 0; j < w.h1; ++j) {
        float acc = w.b1[j];
#pragma unroll
        for (int c = 0; c < SSC_MAX_STATE; ++c)
            if (c < w.obs_dim) acc = fmaf(o[c], w.W1[c * w.h1 + j], acc);
        h1s[j * 64 + lane] = ln ? acc : fmaxf(acc, 0.0f);
    }
    if (ln) {
        float mean, rstd;
        layer_norm_stats(h1s + lane, w.h1, 64, mean, rstd);
        for (int j = 0; j < w.h1; ++j)
            h1s[j * 64 + lane] = fmaxf(fmaf((h1s[j * 64 + lane] - mean) * rstd, w.ln1_g[j], w.ln1_b[j]), 0.0f);
    }
    float out[SSC_MAX_ACT];
#pragma unroll
    for (int a = 0; a < SSC_MAX_ACT; ++a) out[a] = (a < act_dim) ? w.b3[a] : 0.0f;
    float *h2s = h1s + w.h1 * 64;
    for (int j = 0; j < w.h2; ++j) {
        float acc = w.b2[j];
        for (int k = 0; k < w.h1; ++k) acc = fmaf(h1s[k * 64 + lane], w.W2[k * w.h2 + j], acc);
        if (ln) { h2s[j * 64 + lane] = acc; continue; }
        const float h2 = w.last_layer_tanh ? tanh_fast(acc) : fmaxf(acc, 0.0f);
#pragma unroll
        for (int a = 0; a < SSC_MAX_ACT; ++a)
            if (a < act_dim) out[a] = fmaf(h2, w.W3[j * act_dim + a], out[a]);
    }
    if (ln) {
        float mean, rstd;
        layer_norm_stats(h2s + lane, w.h2, 64, mean, rstd);
        for (int j = 0; j < w.h2; ++j) {
            const float n2 = fmaf((h2s[j * 64 + lane] - mean) * rstd, w.ln2_g[j], w.ln2_b[j]);
            const float h2 = w.last_layer_tanh ? tanh_fast(n2) : fmaxf(n2, 0.0f);
#pragma unroll
            for (int a = 0; a < SSC_MAX_ACT; ++a)
                if (a < act_dim) out[a] = fmaf(h2, w.W3[j * act_dim + a], out[a]);
        }
    }
    if (active)
#pragma unroll
        for (int a = 0; a < SSC_MAX_ACT; ++a)
            if (a < act_dim) act[i * act_dim + a] = tanh_fast(out[a]);
}

// The same arithmetic for A FEW rows (the scalar RLAgent.get_action path: m = 1): one 256-thread block per row, hidden
// units across the threads.  actor_generic_kernel gives a row to ONE lane -- h1 x h2 dependent FMAs in a single thread,
// 0.3-0.5 ms for a 200-100 actor, which was most of a scalar rlTrain step with the wide networks.  Every output unit still
// sums its inputs in index order with fused multiply-adds, so the two kernels return identical bits.
__global__ __launch_bounds__(256) void actor_row_kernel(ActorWeights w, int act_dim, const float *__restrict__ obs,
                                                        float *__restrict__ act) {
    extern __shared__ float rs[];   // h1 activations | h2 activations
    float *h1s = rs, *h2s = rs + w.h1;
    const int64_t i = blockIdx.x;
    float o[SSC_MAX_STATE];
#pragma unroll
    for (int c = 0; c < SSC_MAX_STATE; ++c) o[c] = (c < w.obs_dim) ? clip_obs(obs[i * w.obs_dim + c], w.obs_clip) : 0.0f;
    const bool ln = w.ln1_g != nullptr;
    for (int j = threadIdx.x; j < w.h1; j += blockDim.x) {
        float acc = w.b1[j];
#pragma unroll
        for (int c = 0; c < SSC_MAX_STATE; ++c)
            if (c < w.obs_dim) acc = fmaf(o[c], w.W1[c * w.h1 + j], acc);
        h1s[j] = ln ? acc : fmaxf(acc, 0.0f);
    }
    __syncthreads();
    if (ln) {   // every thread forms the row's statistics itself, in index order: the bits of actor_generic_kernel
        float mean, rstd;
        layer_norm_stats(h1s, w.h1, 1, mean, rstd);
        __syncthreads();
        for (int j = threadIdx.x; j < w.h1; j += blockDim.x) h1s[j] = fmaxf(fmaf((h1s[j] - mean) * rstd, w.ln1_g[j], w.ln1_b[j]), 0.0f);
        __syncthreads();
    }
    for (int j = threadIdx.x; j < w.h2; j += blockDim.x) {
        float acc = w.b2[j];
        for (int k = 0; k < w.h1; ++k) acc = fmaf(h1s[k], w.W2[k * w.h2 + j], acc);
        h2s[j] = ln ? acc : (w.last_layer_tanh ? tanh_fast(acc) : fmaxf(acc, 0.0f));
    }
    __syncthreads();
    if (ln) {
        float mean, rstd;
        layer_norm_stats(h2s, w.h2, 1, mean, rstd);
        __syncthreads();
        for (int j = threadIdx.x; j < w.h2; j += blockDim.x) {
            const float n2 = fmaf((h2s[j] - mean) * rstd, w.ln2_g[j], w.ln2_b[j]);
            h2s[j] = w.last_layer_tanh ? tanh_fast(n2) : fmaxf(n2, 0.0f);
        }
        __syncthreads();
    }
    if ((int)threadIdx.x < act_dim) {
        const int a = threadIdx.x;
        float out = w.b3[a];
        for (int j = 0; j < w.h2; ++j) out = fmaf(h2s[j], w.W3[j * act_dim + a], out);
        act[i * act_dim + a] = tanh_fast(out);
    }
}

}  // namespace ssc

using namespace ssc;

extern "C" int ssc_actor_forward(const ssc_actor_desc *a, int64_t m, const float *d_obs, float *d_act,
                                 ssc_stream_t stream) {
    SSC_REQUIRE(a != nullptr, "ssc_actor_forward: actor NULL");
    SSC_REQUIRE(m >= 0, "ssc_actor_forward: m < 0");
    SSC_REQUIRE(a->obs_dim >= 1 && a->obs_dim <= SSC_MAX_STATE && a->act_dim >= 1 && a->act_dim <= SSC_MAX_ACT,
                "ssc_actor_forward: obs_dim %d / act_dim %d out of range", a->obs_dim, a->act_dim);
    SSC_REQUIRE(a->h1 >= 1 && a->h2 >= 1, "ssc_actor_forward: bad hidden sizes");
    if (m == 0) return SSC_OK;
    SSC_REQUIRE(a->W1 && a->b1 && a->W2 && a->b2 && a->W3 && a->b3 && d_obs && d_act,
                "ssc_actor_forward: NULL device pointer");
    const ActorWeights w{a->W1, a->b1, a->W2, a->b2, a->W3, a->b3, a->obs_dim, a->h1, a->h2, a->last_layer_tanh, a->obs_clip,
                         a->ln1_g, a->ln1_b, a->ln2_g, a->ln2_b};
    const bool ln = a->ln1_g != nullptr;
    SSC_REQUIRE(ln == (a->ln1_b != nullptr) && ln == (a->ln2_g != nullptr) && ln == (a->ln2_b != nullptr),
                "ssc_actor_forward: the four LayerNorm pointers come together");
    if (ln && a->precision != SSC_PREC_F32)
        return set_error(SSC_EUNSUPPORTED, "ssc_actor_forward: LayerNorm networks run on the fp32 kernels (precision SSC_PREC_F32)");
    hipStream_t s = as_stream(stream);
    const dim3 grid(blocks_for(m)), block(kBlock);
    if (a->precision == SSC_PREC_BF16_MFMA) {
        if (a->act_dim != 1 || (a->obs_dim != 2 && a->obs_dim != 3) || a->h1 > 224 || a->h2 > 128)
            return set_error(SSC_EUNSUPPORTED,
                             "ssc_actor_forward: MFMA path needs obs_dim 2|3, act_dim 1, h1 <= 224, h2 <= 128");
        if (a->h1 > 128 || a->h2 > 64) {
            if (a->obs_dim == 2) hipLaunchKernelGGL((actor_mfma_lds_kernel<2, 7, 4>), grid, block, 0, s, w, m, d_obs, d_act);
            else hipLaunchKernelGGL((actor_mfma_lds_kernel<3, 7, 4>), grid, block, 0, s, w, m, d_obs, d_act);
            return check_launch("ssc_actor_forward(mfma, lds)");
        }
        const bool small = a->h1 <= 64 && a->h2 <= 32;
        if (a->obs_dim == 2) {
            if (small) hipLaunchKernelGGL((actor_mfma_kernel<2, 2, 1>), grid, block, 0, s, w, m, d_obs, d_act);
            else hipLaunchKernelGGL((actor_mfma_kernel<2, 4, 2>), grid, block, 0, s, w, m, d_obs, d_act);
        } else {
            if (small) hipLaunchKernelGGL((actor_mfma_kernel<3, 2, 1>), grid, block, 0, s, w, m, d_obs, d_act);
            else hipLaunchKernelGGL((actor_mfma_kernel<3, 4, 2>), grid, block, 0, s, w, m, d_obs, d_act);
        }
        return check_launch("ssc_actor_forward(mfma)");
    }
    if (a->precision != SSC_PREC_F32) return set_error(SSC_EINVAL, "ssc_actor_forward: unknown precision");
    if (!ln && a->act_dim == 1 && a->h1 == 64 && a->h2 == 32 && (a->obs_dim == 2 || a->obs_dim == 3)) {
        if (a->obs_dim == 2) hipLaunchKernelGGL((actor_f32_kernel<2, 64, 32>), grid, block, 0, s, w, m, d_obs, d_act);
        else hipLaunchKernelGGL((actor_f32_kernel<3, 64, 32>), grid, block, 0, s, w, m, d_obs, d_act);
        return check_launch("ssc_actor_forward(f32)");
    }
    if (a->h1 > 512) return set_error(SSC_EUNSUPPORTED, "ssc_actor_forward: h1 %d > 512", a->h1);
    if (m <= 32 && a->h2 <= 4096) {   // a handful of rows (scalar get_action): hidden units across a block's threads
        hipLaunchKernelGGL(actor_row_kernel, dim3((unsigned)m), dim3(256), (size_t)(a->h1 + a->h2) * sizeof(float), s, w,
                           a->act_dim, d_obs, d_act);
        return check_launch("ssc_actor_forward(row)");
    }
    const size_t lds = (size_t)(a->h1 + (ln ? a->h2 : 0)) * 64 * sizeof(float);
    if (lds > 160 * 1024) return set_error(SSC_EUNSUPPORTED, "ssc_actor_forward: h1 %d + h2 %d too wide for the LayerNorm kernel", a->h1, a->h2);
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(actor_generic_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "hipFuncSetAttribute");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(actor_generic_kernel, dim3(blocks_for(m, 64)), dim3(64), lds, s, w, a->act_dim, m, d_obs,
                       d_act);
    return check_launch("ssc_actor_forward(generic)");
}
