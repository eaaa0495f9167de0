// smartstart.hip -- SmartStart selection on the GPU (smartexplorationcontinuous.py:223-305):
// batched critic value V(s) = Q(s, pi(s)), Gaussian kernel-density "visitation count" and the
// UCB1 argmax over the candidate smart-start states.
//
// The KDE is the n_ss x |D| pairwise kernel the reference evaluates through
// scipy.stats.gaussian_kde (2000 x 100 000 in the shipped runs): 2e8 exp() per episode start --
// the latency spike the author timed (:354-360).  Here: one block per 8 query points, the data
// set streamed from L2 (0.8 MB at |D| = 1e5, d = 2), fp32 exp on the transcendental pipe, f64
// block reduction.
#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

struct CriticW {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    int32_t obs_dim, act_dim, h1, h2, last_tanh;
    float obs_clip;
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;   // LayerNorm (models_editted.py:85-86, 91-92); null: none
};

// one row per lane; layer-1 activations (+ the action, models_editted.py:89) parked in LDS as [unit][lane]
__global__ __launch_bounds__(64) void critic_kernel(CriticW w, int64_t m, const float *__restrict__ obs,
                                                    const float *__restrict__ act, float *__restrict__ q) {
    extern __shared__ float hs[];  // [(h1 + act_dim)][64] (+ [h2][64] with LayerNorm)
    const bool ln = w.ln1_g != nullptr;
    const int lane = threadIdx.x;
    const int64_t gi = (int64_t)blockIdx.x * 64 + lane;
    const bool active = gi < m;
    const int64_t i = active ? gi : m - 1;
    float o[SSC_MAX_STATE];
#pragma unroll
    for (int c = 0; c < SSC_MAX_STATE; ++c) {   // ddpg_editted.py:106-109
        const float v = (c < w.obs_dim) ? obs[i * w.obs_dim + c] : 0.0f;
        o[c] = w.obs_clip > 0.0f ? fminf(fmaxf(v, -w.obs_clip), w.obs_clip) : v;
    }
    for (int j = 0; j < w.h1; ++j) {
        float acc = w.b1[j];
#pragma unroll
        for (int c = 0; c < SSC_MAX_STATE; ++c)
            if (c < w.obs_dim) acc = fmaf(o[c], w.W1[c * w.h1 + j], acc);
        hs[j * 64 + lane] = ln ? acc : fmaxf(acc, 0.0f);  // models_editted.py:87
    }
    if (ln) {   // :85-86
        float mean, rstd;
        layer_norm_stats(hs + lane, w.h1, 64, mean, rstd);
        for (int j = 0; j < w.h1; ++j)
            hs[j * 64 + lane] = fmaxf(fmaf((hs[j * 64 + lane] - mean) * rstd, w.ln1_g[j], w.ln1_b[j]), 0.0f);
    }
    for (int a = 0; a < w.act_dim; ++a) hs[(w.h1 + a) * 64 + lane] = act[i * w.act_dim + a];  // :89
    const int in2 = w.h1 + w.act_dim;
    float *h2s = hs + in2 * 64;
    float out = w.b3[0];
    for (int j = 0; j < w.h2; ++j) {
        float acc = w.b2[j];
        for (int k = 0; k < in2; ++k) acc = fmaf(hs[k * 64 + lane], w.W2[k * w.h2 + j], acc);
        if (ln) { h2s[j * 64 + lane] = acc; continue; }
        const float h2 = w.last_tanh ? tanhf(acc) : fmaxf(acc, 0.0f);  // :95-98
        out = fmaf(h2, w.W3[j], out);                                    // :100 (no output tanh)
    }
    if (ln) {   // :91-92
        float mean, rstd;
        layer_norm_stats(h2s + lane, w.h2, 64, mean, rstd);
        for (int j = 0; j < w.h2; ++j) {
            const float n2 = fmaf((h2s[j * 64 + lane] - mean) * rstd, w.ln2_g[j], w.ln2_b[j]);
            out = fmaf(w.last_tanh ? tanhf(n2) : fmaxf(n2, 0.0f), w.W3[j], out);
        }
    }
    if (active) q[i] = out;
}

constexpr int kKdeQ = 8;        // query points per block
constexpr int kKdeThreads = 1024;
constexpr int kKdeU = 4;        // data points per thread and loop trip, loads issued together

struct KdeArgs {
    int32_t d;
    int64_t n, m;
    float wh[SSC_MAX_STATE * SSC_MAX_STATE];
    double norm;
};

// One block = kKdeQ query points against the whole data set.  Round 1 ran this with 256-thread blocks, one data point
// per loop trip and the state dimension as a run-time bound: 2000 / 8 = 250 blocks x 4 waves is ONE wave per SIMD, every
// trip waited for its own L2 round trip, and the d x d whitening plus the per-query distance were 8 x 8 and 8 x 8
// predicated loops for a 2-d state (0.41 ms for 2000 x 100 000, a tenth of the VALU rate).  Now: the dimension is a
// template parameter (D = 0: any d <= SSC_MAX_STATE, guarded), 1024 threads per block (4 waves per SIMD), kKdeU
// independent loads in flight per lane.  fp32 exp on the transcendental pipe, per-thread fp32 sums over <= ~100 terms,
// f64 block reduction in a fixed order (deterministic).
template <int D>
__global__ __launch_bounds__(kKdeThreads) void kde_kernel(KdeArgs a, const float *__restrict__ data,
                                                          const float *__restrict__ points, float *__restrict__ pdf) {
    constexpr int DM = D > 0 ? D : SSC_MAX_STATE;
    __shared__ double red[kKdeThreads / 64][kKdeQ];
    const int d = D > 0 ? D : a.d;
    // whitened query points: y_q = Wh * x_q  (then || Wh (x_q - x_j) || = || y_q - Wh x_j ||).  The whitening matrix is
    // pre-scaled by sqrt(0.5 log2 e), so that exp(-0.5 ||.||^2) = exp2(-||scaled . ||^2): no multiply in front of v_exp_f32
    const float kS = 0.84932180028801904272f;   // sqrt(0.5 * log2(e))
    float yq[kKdeQ][DM];
#pragma unroll
    for (int q = 0; q < kKdeQ; ++q) {
        const int64_t qi = min((int64_t)blockIdx.x * kKdeQ + q, a.m - 1);
#pragma unroll
        for (int r = 0; r < DM; ++r) {
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < DM; ++c)
                if (r < d && c < d) s = fmaf(a.wh[r * d + c] * kS, points[qi * d + c], s);
            yq[q][r] = s;
        }
    }
    float sum[kKdeQ];
#pragma unroll
    for (int q = 0; q < kKdeQ; ++q) sum[q] = 0.0f;
    for (int64_t j0 = threadIdx.x; j0 < a.n; j0 += (int64_t)kKdeThreads * kKdeU) {
        float x[kKdeU][DM];
#pragma unroll
        for (int u = 0; u < kKdeU; ++u) {
            const int64_t j = j0 + (int64_t)u * kKdeThreads;
#pragma unroll
            for (int c = 0; c < DM; ++c) x[u][c] = (c < d && j < a.n) ? data[j * d + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kKdeU; ++u) {
            const bool live = j0 + (int64_t)u * kKdeThreads < a.n;
            float y[DM];
#pragma unroll
            for (int r = 0; r < DM; ++r) {
                float s = 0.0f;
#pragma unroll
                for (int c = 0; c < DM; ++c)
                    if (r < d && c < d) s = fmaf(a.wh[r * d + c] * kS, x[u][c], s);
                y[r] = s;
            }
            // a data point past the end sits infinitely far away: exp2(-inf) = 0, no select per query
            if (!live) y[0] = INFINITY;
#pragma unroll
            for (int q = 0; q < kKdeQ; ++q) {
                float e = 0.0f;
#pragma unroll
                for (int r = 0; r < DM; ++r)
                    if (r < d) {
                        const float t = yq[q][r] - y[r];
                        e = fmaf(t, t, e);
                    }
                sum[q] += __builtin_amdgcn_exp2f(-e);
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < kKdeQ; ++q) {
        double v = (double)sum[q];
#pragma unroll
        for (int msk = 32; msk >= 1; msk >>= 1) v += __shfl_xor(v, msk);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (threadIdx.x < kKdeQ) {
        const int64_t qi = (int64_t)blockIdx.x * kKdeQ + threadIdx.x;
        if (qi < a.m) {
            double v = 0.0;
            for (int w = 0; w < kKdeThreads / 64; ++w) v += red[w][threadIdx.x];
            pdf[qi] = (float)(v * a.norm);
        }
    }
}

// single block: ucb + argmax (m <= a few thousand candidates)
__global__ __launch_bounds__(256) void ucb_kernel(int64_t m, const float *__restrict__ value,
                                                  const float *__restrict__ pdf, float alpha, float beta,
                                                  double buffer_len, double volume, float *__restrict__ ucb,
                                                  int32_t *__restrict__ best) {
    __shared__ float rs[4];
    __shared__ int ri[4];
    float bs = -INFINITY;
    int bi = 0x7fffffff;
    const double num = (double)beta * log(buffer_len);
    for (int64_t i = threadIdx.x; i < m; i += 256) {
        const double c_hat = buffer_len * (double)pdf[i] * volume;              // :276
        const float u = alpha * value[i] + (float)sqrt(num / c_hat);            // :277-279
        if (ucb != nullptr) ucb[i] = u;
        if (u > bs || (u == bs && (int)i < bi)) { bs = u; bi = (int)i; }
    }
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) {
        const float os = __shfl_xor(bs, msk);
        const int oi = __shfl_xor(bi, msk);
        if (os > bs || (os == bs && oi < bi)) { bs = os; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = bs; ri[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (rs[w] > bs || (rs[w] == bs && ri[w] < bi)) { bs = rs[w]; bi = ri[w]; }
        best[0] = (bi == 0x7fffffff) ? 0 : bi;   // :280 np.argmax
    }
}

}  // namespace ssc

using namespace ssc;

extern "C" {

int ssc_critic_forward(const ssc_critic_desc *c, int64_t m, const float *d_obs, const float *d_act, float *d_q,
                       ssc_stream_t stream) {
    SSC_REQUIRE(c != nullptr, "ssc_critic_forward: critic NULL");
    SSC_REQUIRE(m >= 0, "ssc_critic_forward: m < 0");
    SSC_REQUIRE(c->obs_dim >= 1 && c->obs_dim <= SSC_MAX_STATE && c->act_dim >= 1 && c->act_dim <= SSC_MAX_ACT,
                "ssc_critic_forward: obs_dim %d / act_dim %d out of range", c->obs_dim, c->act_dim);
    SSC_REQUIRE(c->h1 >= 1 && c->h2 >= 1 && c->h1 + c->act_dim <= 600, "ssc_critic_forward: bad hidden sizes");
    if (m == 0) return SSC_OK;
    SSC_REQUIRE(c->W1 && c->b1 && c->W2 && c->b2 && c->W3 && c->b3 && d_obs && d_act && d_q,
                "ssc_critic_forward: NULL device pointer");
    const CriticW w{c->W1, c->b1, c->W2, c->b2, c->W3, c->b3, c->obs_dim, c->act_dim, c->h1, c->h2, c->last_layer_tanh, c->obs_clip,
                    c->ln1_g, c->ln1_b, c->ln2_g, c->ln2_b};
    const bool ln = c->ln1_g != nullptr;
    SSC_REQUIRE(ln == (c->ln1_b != nullptr) && ln == (c->ln2_g != nullptr) && ln == (c->ln2_b != nullptr),
                "ssc_critic_forward: the four LayerNorm pointers come together");
    const size_t lds = (size_t)(c->h1 + c->act_dim + (ln ? c->h2 : 0)) * 64 * sizeof(float);
    if (lds > 160 * 1024) return set_error(SSC_EUNSUPPORTED, "ssc_critic_forward: h1 %d + h2 %d too wide for the LayerNorm kernel", c->h1, c->h2);
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(critic_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "hipFuncSetAttribute(critic_kernel)");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(critic_kernel, dim3(blocks_for(m, 64)), dim3(64), lds, as_stream(stream), w, m, d_obs, d_act,
                       d_q);
    return check_launch("ssc_critic_forward");
}

int ssc_kde_evaluate(int32_t d, int64_t n, const float *d_data, int64_t m, const float *d_points,
                     const float *whitening, double norm, float *d_pdf, ssc_stream_t stream) {
    SSC_REQUIRE(d >= 1 && d <= SSC_MAX_STATE, "ssc_kde_evaluate: d = %d out of range", d);
    SSC_REQUIRE(n >= 1 && m >= 0, "ssc_kde_evaluate: need n >= 1, m >= 0");
    SSC_REQUIRE(whitening != nullptr, "ssc_kde_evaluate: whitening NULL");
    if (m == 0) return SSC_OK;
    SSC_REQUIRE(d_data && d_points && d_pdf, "ssc_kde_evaluate: NULL device pointer");
    KdeArgs a{};
    a.d = d; a.n = n; a.m = m; a.norm = norm;
    for (int i = 0; i < d * d; ++i) a.wh[i] = whitening[i];
    const dim3 grid(blocks_for(m, kKdeQ)), block(kKdeThreads);
    hipStream_t s = as_stream(stream);
    if (d == 1) hipLaunchKernelGGL(kde_kernel<1>, grid, block, 0, s, a, d_data, d_points, d_pdf);
    else if (d == 2) hipLaunchKernelGGL(kde_kernel<2>, grid, block, 0, s, a, d_data, d_points, d_pdf);
    else if (d == 3) hipLaunchKernelGGL(kde_kernel<3>, grid, block, 0, s, a, d_data, d_points, d_pdf);
    else hipLaunchKernelGGL(kde_kernel<0>, grid, block, 0, s, a, d_data, d_points, d_pdf);
    return check_launch("ssc_kde_evaluate");
}

int ssc_ucb_argmax(int64_t m, const float *d_value, const float *d_pdf, float alpha, float beta, double buffer_len,
                   double volume, float *d_ucb, int32_t *d_best, ssc_stream_t stream) {
    SSC_REQUIRE(m >= 1, "ssc_ucb_argmax: m < 1");
    SSC_REQUIRE(buffer_len >= 1.0 && volume > 0.0, "ssc_ucb_argmax: bad buffer_len / volume");
    SSC_REQUIRE(d_value && d_pdf && d_best, "ssc_ucb_argmax: NULL device pointer");
    hipLaunchKernelGGL(ucb_kernel, dim3(1), dim3(256), 0, as_stream(stream), m, d_value, d_pdf, alpha, beta, buffer_len,
                       volume, d_ucb, d_best);
    return check_launch("ssc_ucb_argmax");
}

}  // extern "C"
