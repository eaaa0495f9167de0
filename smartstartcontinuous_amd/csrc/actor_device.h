// actor_device.h -- DDPG actor forward as wave-level device code (gfx950).
//
// Actor_Editted.__call__ (DDPG_Baselines_editted/models_editted.py:38-61):
//   x = relu(obs @ W1 + b1); x = tanh|relu(x @ W2 + b2); a = tanh(x @ W3 + b3)
// with TensorFlow weight layout W[in][out].  Two implementations:
//
//  * ActorF32<OBS,H1,H2>   one env per lane, everything in fp32 on the VALU, weights read
//                          through wave-uniform (scalar) loads.  The parity reference path.
//  * ActorMfma<OBS,UT,JT>  one wave = 64 envs.  Layer 1 (K = obs_dim = 2..4) runs on the
//                          exact-f32 MFMA v_mfma_f32_32x32x2_f32; its 32x32 accumulator tile
//                          (hidden unit on the register index, env on the lane) is converted
//                          to bf16 in registers and fed STRAIGHT BACK as the B operand of
//                          v_mfma_f32_32x32x16_bf16 for the [H1]x[H2] hidden GEMM -- no LDS,
//                          no lane movement (cdna_hip_programming.md section 3 "An accumulator
//                          tile as the next MFMA's operand").  Layer 3 (H2 -> 1) is a per-lane
//                          dot over the accumulator registers plus one cross-half add.
#pragma once

#include "ssc_device.h"

namespace ssc {

struct ActorWeights {
    const float *W1, *b1, *W2, *b2, *W3, *b3;
    int32_t obs_dim, h1, h2;
    int32_t last_layer_tanh;
    float obs_clip;  // > 0: clip the observation to [-obs_clip, obs_clip] first (ddpg_editted.py:106-109); 0: no clip
    const float *ln1_g, *ln1_b, *ln2_g, *ln2_b;  // LayerNorm gamma / beta of the two hidden layers (ssc_actor_desc); null: none
};


// tf.clip_by_value(obs, -c, c) of DDPG_editted's network inputs; c <= 0 leaves x alone (wave-uniform select)
__device__ __forceinline__ float clip_obs(float x, float c) { return c > 0.0f ? fminf(fmaxf(x, -c), c) : x; }

// ---------------------------------------------------------------------------------------
// fp32 VALU path
// ---------------------------------------------------------------------------------------
template <int OBS, int H1, int H2>
struct ActorF32 {
    static constexpr int kLanesPerEnv = 1;
    // The weights (9.2 KB for 64-32) are staged ONCE per block into LDS and read back as broadcast
    // ds_read_b128: reading them through the kernel-argument pointers inside the step loop forces
    // hipcc to re-issue ~300 global loads per env-step (the log stores may alias them).
    static constexpr int OFF_B1 = OBS * H1, OFF_W2 = OFF_B1 + H1, OFF_B2 = OFF_W2 + H1 * H2;
    static constexpr int OFF_W3 = OFF_B2 + H2, OFF_B3 = OFF_W3 + H2;
    // LayerNorm gamma / beta of the two hidden layers (models_editted.py:45-46, 50-51), when the network has them
    static constexpr int OFF_G1 = OFF_B3 + 4, OFF_BE1 = OFF_G1 + H1, OFF_G2 = OFF_BE1 + H1, OFF_BE2 = OFF_G2 + H2, TOTAL = OFF_BE2 + H2;
    const float *lw;  // LDS image: W1[OBS][H1] | b1 | W2[H1][H2] | b2 | W3[H2] | b3 | gamma1 | beta1 | gamma2 | beta2
    int last_tanh;
    bool ln;

    // block-cooperative: every thread of the block must call it
    __device__ void init(const ActorWeights &w) {
        __shared__ __attribute__((aligned(16))) float image[TOTAL];
        for (int e = threadIdx.x; e < OBS * H1; e += blockDim.x) image[e] = w.W1[e];
        for (int e = threadIdx.x; e < H1; e += blockDim.x) image[OFF_B1 + e] = w.b1[e];
        for (int e = threadIdx.x; e < H1 * H2; e += blockDim.x) image[OFF_W2 + e] = w.W2[e];
        for (int e = threadIdx.x; e < H2; e += blockDim.x) {
            image[OFF_B2 + e] = w.b2[e];
            image[OFF_W3 + e] = w.W3[e];
        }
        if (threadIdx.x == 0) image[OFF_B3] = w.b3[0];
        ln = w.ln1_g != nullptr;
        if (ln) {
            for (int e = threadIdx.x; e < H1; e += blockDim.x) { image[OFF_G1 + e] = w.ln1_g[e]; image[OFF_BE1 + e] = w.ln1_b[e]; }
            for (int e = threadIdx.x; e < H2; e += blockDim.x) { image[OFF_G2 + e] = w.ln2_g[e]; image[OFF_BE2 + e] = w.ln2_b[e]; }
        }
        __syncthreads();
        lw = image;
        last_tanh = w.last_layer_tanh;
    }

    // tc.layers.layer_norm over the N values of this lane's row (registers): index-order sums like layer_norm_stats
    template <int N>
    static __device__ __forceinline__ void norm_stats(const float (&x)[N], float &mean, float &rstd) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) s += x[i];
        mean = s / (float)N;
        float v = 0.0f;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float d = x[i] - mean;
            v = fmaf(d, d, v);
        }
        rstd = 1.0f / sqrtf(v / (float)N + 1e-12f);
    }

    __device__ float forward(const float (&obs)[OBS]) const {
        static_assert(H1 % 4 == 0 && H2 % 4 == 0, "hidden sizes must be multiples of 4");
        typedef float f4 __attribute__((ext_vector_type(4)));
        float h1[H1];
#pragma unroll
        for (int j = 0; j < H1; j += 4) {
            f4 acc = *reinterpret_cast<const f4 *>(lw + OFF_B1 + j);
#pragma unroll
            for (int i = 0; i < OBS; ++i) {
                const f4 wv = *reinterpret_cast<const f4 *>(lw + i * H1 + j);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(obs[i], wv[q], acc[q]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) h1[j + q] = ln ? acc[q] : fmaxf(acc[q], 0.0f);  // models_editted.py:47
        }
        float out = lw[OFF_B3];
        if (ln) {   // block-uniform: the LayerNorm network (models_editted.py:45-46, 50-51)
            float mean, rstd;
            norm_stats<H1>(h1, mean, rstd);
#pragma unroll
            for (int j = 0; j < H1; ++j) h1[j] = fmaxf(fmaf((h1[j] - mean) * rstd, lw[OFF_G1 + j], lw[OFF_BE1 + j]), 0.0f);
            float z2[H2];
#pragma unroll 2
            for (int j = 0; j < H2; j += 4) {
                f4 acc = *reinterpret_cast<const f4 *>(lw + OFF_B2 + j);
#pragma unroll
                for (int i = 0; i < H1; ++i) {
                    const f4 wv = *reinterpret_cast<const f4 *>(lw + OFF_W2 + i * H2 + j);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(h1[i], wv[q], acc[q]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) z2[j + q] = acc[q];
            }
            norm_stats<H2>(z2, mean, rstd);
#pragma unroll
            for (int j = 0; j < H2; ++j) {
                const float n2 = fmaf((z2[j] - mean) * rstd, lw[OFF_G2 + j], lw[OFF_BE2 + j]);
                out = fmaf(last_tanh ? tanh_fast(n2) : fmaxf(n2, 0.0f), lw[OFF_W3 + j], out);
            }
            return tanh_fast(out);
        }
#pragma unroll 2
        for (int j = 0; j < H2; j += 4) {
            f4 acc = *reinterpret_cast<const f4 *>(lw + OFF_B2 + j);
#pragma unroll
            for (int i = 0; i < H1; ++i) {  // k order 0..H1-1 per output unit: a plain fp32 FMA chain
                const f4 wv = *reinterpret_cast<const f4 *>(lw + OFF_W2 + i * H2 + j);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = fmaf(h1[i], wv[q], acc[q]);
            }
            const f4 w3 = *reinterpret_cast<const f4 *>(lw + OFF_W3 + j);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float h2 = last_tanh ? tanh_fast(acc[q]) : fmaxf(acc[q], 0.0f);  // :53-56
                out = fmaf(h2, w3[q], out);
            }
        }
        return tanh_fast(out);  // :60
    }
};

// ---------------------------------------------------------------------------------------
// MFMA path
// ---------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// row of a 32x32 MFMA accumulator element: (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// ET = number of 32-env tiles per wave:
//   ET = 2  one wave = 64 envs, one env per lane (lane l owns env l);
//   ET = 1  one wave = 32 envs, each env DUPLICATED on lanes l and l+32: twice the waves for the same
//           env count (2 waves per SIMD at 65 536 envs).  Measured SLOWER than ET = 2 (0.322 vs
//           0.293 ms per 65 536 x 256 steps): the kernel is bound by the quarter-rate transcendental
//           pipe (2 per tanh, 66 per 64 env-steps), whose work does not shrink, while the per-env
//           dynamics and noise are computed twice.  Kept selectable; the dispatcher uses ET = 2.
template <int OBS, int UT, int JT, int ET = 2>
struct ActorMfma {
    static constexpr int kLanesPerEnv = (ET == 1) ? 2 : 1;
    static constexpr int KS1 = (OBS + 1) / 2;  // 32x32x2 k-steps of layer 1
    float a1[UT][KS1];     // layer-1 A operand: W1[k = 2ks + half][unit = ut*32 + (lane&31)]
    f32x16 c1[UT];         // b1 broadcast in accumulator layout
    bf16x8 a2[JT][UT][2];  // layer-2 A fragments (W2^T, k order matched to the layer-1 accumulator)
    f32x16 c2[JT];         // b2 in accumulator layout
    float w3[JT][16];      // W3[jt*32 + acc_row(reg)]; premultiplied by -2 when last_tanh (see forward)
    float b3;              // b3 (+ sum_j W3[j] when last_tanh)
    int last_tanh;

    __device__ void init(const ActorWeights &w) {
        const int lane = threadIdx.x & 63;
        const int r = lane & 31, half = lane >> 5;
        const int H1 = w.h1, H2 = w.h2;
        last_tanh = w.last_layer_tanh;
        // tanh(x) = 1 - 2 / (2^(c x) + 1), c = 2 log2(e).  With last_tanh the constant c is folded
        // into W2 / b2 (so the MFMA output is already c*x), and  sum_j w3_j tanh_j  is evaluated as
        // (sum_j w3_j) + sum_j (-2 w3_j) r_j  with r_j = 1 / (2^(c x_j) + 1): per hidden unit that
        // is v_exp_f32, v_add, v_rcp_f32, v_fma -- 4 VALU ops instead of 6.
        const float cs = last_tanh ? 2.88539008177792681472f : 1.0f;
        // Every load below is UNCONDITIONAL (index clamped into the array, value dropped by a select): a guarded load is a
        // branch with a full wait behind it, and this function was ~50 of them in a row -- most of the standalone actor
        // kernel's 11 us at 65 536 rows.
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            const int unit = ut * 32 + r;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int k = 2 * ks + half;
                const float v = w.W1[min(k, OBS - 1) * H1 + min(unit, H1 - 1)];
                a1[ut][ks] = (k < OBS && unit < H1) ? v : 0.0f;
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int u = ut * 32 + acc_row(reg, half);
                const float v = w.b1[min(u, H1 - 1)];
                c1[ut][reg] = (u < H1) ? v : 0.0f;
            }
        }
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const int col = jt * 32 + r;  // layer-2 output unit held by this lane's A row
#pragma unroll
            for (int ut = 0; ut < UT; ++ut)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        // k slot (8*half + j) of k-step (ut, s) carries hidden unit u
                        const int u = ut * 32 + 16 * s + 8 * (j >> 2) + 4 * half + (j & 3);
                        const float x = w.W2[min(u, H1 - 1) * H2 + min(col, H2 - 1)];
                        const float v = (u < H1 && col < H2) ? x * cs : 0.0f;
                        a2[jt][ut][s][j] = (__bf16)v;
                    }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int o = jt * 32 + acc_row(reg, half);
                const float xb = w.b2[min(o, H2 - 1)], xw = w.W3[min(o, H2 - 1)];
                c2[jt][reg] = (o < H2) ? xb * cs : 0.0f;
                w3[jt][reg] = (o < H2) ? xw * (last_tanh ? -2.0f : 1.0f) : 0.0f;
            }
        }
        b3 = w.b3[0];
        if (last_tanh) {     // b3 + sum_j W3[j], added in index order as before; the (uniform) loads go out together
            float w3v[32 * JT];
#pragma unroll
            for (int j = 0; j < 32 * JT; ++j) w3v[j] = w.W3[min(j, H2 - 1)];
#pragma unroll
            for (int j = 0; j < 32 * JT; ++j) b3 += (j < H2) ? w3v[j] : 0.0f;
        }
    }

    // obs: this lane's env observation.  Returns the actor output for this lane's env.
    // Wave-collective: every lane of the wave must call it.
    __device__ float forward(const float (&obs)[OBS]) const {
        const int half = (threadIdx.x & 63) >> 5;
        // layer-1 B operand of tile et, k-step ks: obs component (2ks + half) of env (et*32 + lane&31).
        float bop[ET][KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const float v0 = obs[2 * ks];
            const float v1 = (2 * ks + 1 < OBS) ? obs[2 * ks + 1] : 0.0f;
            if constexpr (ET == 2) {
                // tile 0 needs [v0.lo | v1.lo], tile 1 [v0.hi | v1.hi]: exactly one v_permlane32_swap
                half_swap(v0, v1, bop[0][ks], bop[1][ks]);
            } else {
                bop[0][ks] = half ? v1 : v0;  // both lane halves already hold the env's observation
            }
        }
        f32x16 acc2[JT][ET];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int et = 0; et < ET; ++et) acc2[jt][et] = c2[jt];
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
#pragma unroll
            for (int et = 0; et < ET; ++et) {
                f32x16 d = c1[ut];
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks)
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[ut][ks], bop[et][ks], d, 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    typedef int i32x4 __attribute__((ext_vector_type(4)));
                    i32x4 packed;  // relu (models_editted.py:47) + bf16 convert, two values per 2 VALU ops
#pragma unroll
                    for (int j = 0; j < 4; ++j) packed[j] = relu_pack_bf16(d[8 * s + 2 * j], d[8 * s + 2 * j + 1]);
                    const bf16x8 frag = __builtin_bit_cast(bf16x8, packed);
#pragma unroll
                    for (int jt = 0; jt < JT; ++jt)
                        acc2[jt][et] =
                            __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[jt][ut][s], frag, acc2[jt][et], 0, 0, 0);
                }
            }
        }
        float part[ET];
#pragma unroll
        for (int et = 0; et < ET; ++et) part[et] = 0.0f;
        if (last_tanh) {  // one wave-uniform branch around the whole layer, not one per element
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
#pragma unroll
                    for (int et = 0; et < ET; ++et) {
                        const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc2[jt][et][reg]) + 1.0f);
                        part[et] = fmaf(r, w3[jt][reg], part[et]);
                    }
        } else {
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
#pragma unroll
                    for (int et = 0; et < ET; ++et)
                        part[et] = fmaf(fmaxf(acc2[jt][et][reg], 0.0f), w3[jt][reg], part[et]);
        }
        // ET == 2: lanes 0-31 need part0.lo + part0.hi, lanes 32-63 part1.lo + part1.hi;
        // ET == 1: every lane needs part.lo + part.hi.  One swap + one add either way.
        float s_lo, s_hi;
        half_swap(part[0], part[ET - 1], s_lo, s_hi);
        return tanh_fast(s_lo + s_hi + b3);
    }
};

// ---------------------------------------------------------------------------------------
// ActorMfmaLds<OBS,UT,JT>: the same network for the wide shapes of the reference's grid (actor 200-100 of
// data/ddpg_baselines_summaries/hidden_layer_size_experiment/; anything up to 32 UT x 32 JT units).  ActorMfma keeps the
// W2^T fragments in registers (JT * UT * 8 VGPRs: 224 for 200-100, beside 128 accumulator registers); here they are
// STAGED ONCE PER BLOCK IN LDS as ready-made bf16 A fragments -- [jt][ut][s][lane] x 16 B, one conflict-free
// ds_read_b128 per fragment, each feeding the MFMAs of both 32-env tiles of the wave (56 KB for 200-100; the four waves
// of a block share it) -- together with b1, b2 and W3 in accumulator order.  Layer 1 stays on the exact-f32 MFMA with its
// K <= 4 weights in registers; the data flow from there on is ActorMfma's (accumulator tile -> ReLU -> bf16 -> B operand).
template <int OBS, int UT, int JT, int ET = 2>
struct ActorMfmaLds {
    // ET = 2: one wave = 64 envs; ET = 1: one wave = 32 envs, each on lanes l and l + 32 (twice the waves: two per SIMD at
    // 65 536 envs, so that one wave's tanh layer runs under the other's MFMAs -- see ActorMfma)
    static constexpr int kLanesPerEnv = (ET == 1) ? 2 : 1;
    static constexpr int KS1 = (OBS + 1) / 2;
    float a1[UT][KS1];
    const unsigned char *l_a2;   // bf16 [JT][UT][2][64 lanes][8]
    const float *l_b1, *l_b2, *l_w3;   // [UT * 32], [JT * 32] (x cs), [JT * 32] (x -2 when last_tanh)
    float b3;
    int last_tanh;

    // block-cooperative: every thread of the block must call it
    __device__ void init(const ActorWeights &w) {
        __shared__ __attribute__((aligned(16))) unsigned char s_a2[JT * UT * 2 * 1024];
        __shared__ __attribute__((aligned(16))) float s_b1[UT * 32], s_b2[JT * 32], s_w3[JT * 32];
        const int lane = threadIdx.x & 63;
        const int r = lane & 31, half = lane >> 5;
        const int H1 = w.h1, H2 = w.h2;
        last_tanh = w.last_layer_tanh;
        const float cs = last_tanh ? 2.88539008177792681472f : 1.0f;   // see ActorMfma::init
        for (int e = threadIdx.x; e < JT * UT * 2 * 64 * 8; e += blockDim.x) {
            const int j = e & 7, ln = (e >> 3) & 63, sq = (e >> 9) & 1, ut = (e >> 10) % UT, jt = (e >> 10) / UT;
            const int col = jt * 32 + (ln & 31);
            const int u = ut * 32 + 16 * sq + 8 * (j >> 2) + 4 * (ln >> 5) + (j & 3);   // k slot -> hidden unit (ActorMfma::init)
            const float v = (u < H1 && col < H2) ? w.W2[u * H2 + col] * cs : 0.0f;
            reinterpret_cast<__bf16 *>(s_a2)[e] = (__bf16)v;
        }
        for (int e = threadIdx.x; e < UT * 32; e += blockDim.x) s_b1[e] = (e < H1) ? w.b1[e] : 0.0f;
        for (int e = threadIdx.x; e < JT * 32; e += blockDim.x) {
            s_b2[e] = (e < H2) ? w.b2[e] * cs : 0.0f;
            s_w3[e] = (e < H2) ? w.W3[e] * (last_tanh ? -2.0f : 1.0f) : 0.0f;
        }
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            const int unit = ut * 32 + r;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int k = 2 * ks + half;
                a1[ut][ks] = (k < OBS && unit < H1) ? w.W1[k * H1 + unit] : 0.0f;
            }
        }
        b3 = w.b3[0];
        if (last_tanh)
            for (int j = 0; j < H2; ++j) b3 += w.W3[j];
        __syncthreads();
        l_a2 = s_a2 + lane * 16;
        l_b1 = s_b1 + 4 * half;
        l_b2 = s_b2 + 4 * half;
        l_w3 = s_w3 + 4 * half;
    }

    // 16 consecutive accumulator registers of unit tile `tile` from an LDS vector in unit order: register reg holds unit
    // tile * 32 + acc_row(reg, half) = 4 floats at tile * 32 + 8 * (reg >> 2) (+ 4 * half, folded into the pointer)
    static __device__ __forceinline__ f32x16 acc_layout(const float *v, int tile) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f32x16 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4 x = *reinterpret_cast<const f4 *>(v + tile * 32 + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) o[4 * q + i] = x[i];
        }
        return o;
    }

    __device__ float forward(const float (&obs)[OBS]) const {
        float bop[ET][KS1];
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const float v0 = obs[2 * ks];
            const float v1 = (2 * ks + 1 < OBS) ? obs[2 * ks + 1] : 0.0f;
            if constexpr (ET == 2) half_swap(v0, v1, bop[0][ks], bop[1][ks]);
            else bop[0][ks] = ((threadIdx.x & 63) >> 5) ? v1 : v0;   // both lane halves already hold the env's observation
        }
        f32x16 acc2[JT][ET];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            acc2[jt][0] = acc_layout(l_b2, jt);
            if constexpr (ET == 2) acc2[jt][1] = acc2[jt][0];
        }
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            bf16x8 frag[ET][2];
            const f32x16 c1 = acc_layout(l_b1, ut);
#pragma unroll
            for (int et = 0; et < ET; ++et) {
                f32x16 d = c1;
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks)
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[ut][ks], bop[et][ks], d, 0, 0, 0);
#pragma unroll
                for (int sq = 0; sq < 2; ++sq) {
                    typedef int i32x4 __attribute__((ext_vector_type(4)));
                    i32x4 packed;  // relu (models_editted.py:47) + bf16 convert
#pragma unroll
                    for (int j = 0; j < 4; ++j) packed[j] = relu_pack_bf16(d[8 * sq + 2 * j], d[8 * sq + 2 * j + 1]);
                    frag[et][sq] = __builtin_bit_cast(bf16x8, packed);
                }
            }
            // the 2 JT fragments of this unit tile, each read one fragment ahead of its two MFMAs (the scheduler otherwise
            // puts every read right in front of its wait: an LDS round trip per fragment with an idle matrix pipe)
            bf16x8 a = *reinterpret_cast<const bf16x8 *>(l_a2 + ((0 * UT + ut) * 2 + 0) * 1024);
            __builtin_amdgcn_sched_barrier(0);   // the first read stays in front of the groups below
#pragma unroll
            for (int q = 0; q < 2 * JT; ++q) {
                const int sq = q / JT, jt = q % JT;
                const int qn = (q + 1 < 2 * JT) ? q + 1 : q;
                const bf16x8 an = *reinterpret_cast<const bf16x8 *>(l_a2 + (((qn % JT) * UT + ut) * 2 + qn / JT) * 1024);
                acc2[jt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag[0][sq], acc2[jt][0], 0, 0, 0);
                if constexpr (ET == 2) acc2[jt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, frag[1][sq], acc2[jt][1], 0, 0, 0);
                a = an;
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one LDS read ...
                __builtin_amdgcn_sched_group_barrier(0x008, ET, 0);  // ... then the MFMAs of the fragment before it
            }
        }
        float part[ET];
#pragma unroll
        for (int et = 0; et < ET; ++et) part[et] = 0.0f;
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) {
            const f32x16 w3 = acc_layout(l_w3, jt);
            if (last_tanh) {   // wave-uniform
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
#pragma unroll
                    for (int et = 0; et < ET; ++et) {
                        const float rr = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc2[jt][et][reg]) + 1.0f);
                        part[et] = fmaf(rr, w3[reg], part[et]);
                    }
            } else {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg)
#pragma unroll
                    for (int et = 0; et < ET; ++et) part[et] = fmaf(fmaxf(acc2[jt][et][reg], 0.0f), w3[reg], part[et]);
            }
        }
        float s_lo, s_hi;
        half_swap(part[0], part[ET - 1], s_lo, s_hi);
        return tanh_fast(s_lo + s_hi + b3);
    }
};

// ---------------------------------------------------------------------------------------
// ActorMfma2<LAST_TANH>: the shipped shape (obs_dim 2, h1 <= 64, h2 <= 32; BASELINE config 3) with BOTH
// contractions on the bf16 MFMA and no lane movement on the way in.
//
//  * Layer 1 at fp32-level accuracy on v_mfma_f32_32x32x16_bf16 (32 cycles per tile instead of the 64 of the
//    exact-f32 shape): x = xh + xl, w = wh + wl (bf16 head + residual), x*w ~= xh*wh + xl*wh + xh*wl (the dropped
//    xl*wl is <= 2^-16 |x w|; products of bf16 pairs are exact in the fp32 accumulator), and the bias rides in two
//    more k slots against a constant 1 -- 2 inputs x 3 + 2 = 8 k slots, the accumulator starts from the inline
//    constant 0.
//  * The K = 16 slots of the shape split into the half held by lanes 0-31 (k 0..7) and by lanes 32-63 (k 8..15).
//    The A operand carries the SAME 8 weight slots in both halves; the B operand of env tile 0 is {own slots in
//    lanes 0-31, zeros in lanes 32-63}, that of env tile 1 {zeros, own slots}: lane l contributes its OWN env to
//    the tile it belongs to, so the observation never changes lanes (the exact-f32 path needs a
//    v_permlane32_swap per k-step).  Building both operands is 10 VALU ops per wave-step.
//  * Everything from the hidden contraction on is ActorMfma<2,2,1,2>'s code; LAST_TANH is a template
//    parameter so that the forward pass is ONE basic block (the scheduler can then run the caller's
//    independent work -- noise generation -- under the MFMAs).
template <bool LAST_TANH>
struct ActorMfma2 {
    static constexpr int kLanesPerEnv = 1;
    static constexpr int UT = 2;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    bf16x8 a1[UT];        // layer-1 A fragments: 8 k slots of unit ut*32 + (lane & 31), identical in both lane halves
    bf16x8 a2[UT][2];     // layer-2 A fragments (W2^T, k order matched to the layer-1 accumulator)
    f32x16 c2;            // b2 in accumulator layout
    float w3[16];         // W3[acc_row(reg)] (x -2 when LAST_TANH, see ActorMfma::init)
    float b3;
    uint32_t m0, m1;      // lane masks: all-ones in lanes 0-31 / 32-63
    uint32_t one0, one1;  // the constant-1 slot pair (bias carrier) masked the same way

    static __device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
        const ssc_f32x2 v = {a, b};
        return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ssc_bf16x2));
    }
    static __device__ __forceinline__ float bf16_head_f(float v) { return (float)(__bf16)v; }

    __device__ void init(const ActorWeights &w) {
        const int lane = threadIdx.x & 63;
        const int r = lane & 31, half = lane >> 5;
        const int H1 = w.h1, H2 = w.h2;
        const float cs = LAST_TANH ? 2.88539008177792681472f : 1.0f;  // see ActorMfma::init
        m0 = half ? 0u : 0xFFFFFFFFu;
        m1 = ~m0;
        one0 = 0x3F803F80u & m0;
        one1 = 0x3F803F80u & m1;
#pragma unroll
        for (int ut = 0; ut < UT; ++ut) {
            const int unit = ut * 32 + r;
            const bool ok = unit < H1;           // (unconditional loads, clamped index + select: see ActorMfma::init)
            const int uc = min(unit, H1 - 1);
            const float x0 = w.W1[0 * H1 + uc], x1 = w.W1[1 * H1 + uc], xb = w.b1[uc];
            const float w0 = ok ? x0 : 0.0f, w1 = ok ? x1 : 0.0f;
            const float bb = ok ? xb : 0.0f;
            const float w0h = bf16_head_f(w0), w1h = bf16_head_f(w1), bh = bf16_head_f(bb);
            i32x4 p;
            p[0] = (int)pack_bf16(w0h, w1h);            // x (xh0, xh1)
            p[1] = p[0];                                // x (xl0, xl1)
            p[2] = (int)pack_bf16(w0 - w0h, w1 - w1h);  // x (xh0, xh1)
            p[3] = (int)pack_bf16(bh, bb - bh);         // x (1, 1)
            a1[ut] = __builtin_bit_cast(bf16x8, p);
        }
        const int col = r;  // layer-2 output unit held by this lane's A row
#pragma unroll
        for (int ut = 0; ut < UT; ++ut)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int u = ut * 32 + 16 * s + 8 * (j >> 2) + 4 * half + (j & 3);
                    const float x = w.W2[min(u, H1 - 1) * H2 + min(col, H2 - 1)];
                    const float v = (u < H1 && col < H2) ? x * cs : 0.0f;
                    a2[ut][s][j] = (__bf16)v;
                }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int o = acc_row(reg, half);
            const float xb = w.b2[min(o, H2 - 1)], xw = w.W3[min(o, H2 - 1)];
            c2[reg] = (o < H2) ? xb * cs : 0.0f;
            w3[reg] = (o < H2) ? xw * (LAST_TANH ? -2.0f : 1.0f) : 0.0f;
        }
        b3 = w.b3[0];
        if (LAST_TANH) {     // same order of additions as before, the loads together
            float w3v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) w3v[j] = w.W3[min(j, H2 - 1)];
#pragma unroll
            for (int j = 0; j < 32; ++j) b3 += (j < H2) ? w3v[j] : 0.0f;
        }
    }

    // Wave-collective.  Returns the PRE-tanh output sum_j w3_j h2_j + b3 of this lane's env (the caller applies
    // the final tanh, so that it can place it in its own dependent chain).
    __device__ __forceinline__ float forward_pre(float x0, float x1) const {
        const uint32_t H = pack_bf16(x0, x1);
        const float h0 = __builtin_bit_cast(float, H << 16), h1 = __builtin_bit_cast(float, H & 0xFFFF0000u);
        const uint32_t L = pack_bf16(x0 - h0, x1 - h1);
        bf16x8 bop[2];
        {
            i32x4 q0 = {(int)(H & m0), (int)(L & m0), (int)(H & m0), (int)one0};
            i32x4 q1 = {(int)(H & m1), (int)(L & m1), (int)(H & m1), (int)one1};
            bop[0] = __builtin_bit_cast(bf16x8, q0);
            bop[1] = __builtin_bit_cast(bf16x8, q1);
        }
        const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // all four layer-1 tiles first (independent: 4 x 32 cycles of matrix pipe), then tile by tile ReLU + convert
        // and the two hidden k-steps that consume it -- env tile 0's chain first, so that its tanh layer can start
        // while env tile 1's MFMAs are still in the pipe
        f32x16 d[2][UT];
#pragma unroll
        for (int et = 0; et < 2; ++et)
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) d[et][ut] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[ut], bop[et], zero, 0, 0, 0);
        f32x16 acc2[2] = {c2, c2};
#pragma unroll
        for (int et = 0; et < 2; ++et) {
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    i32x4 packed;  // relu (models_editted.py:47) + bf16 convert
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        packed[j] = relu_pack_bf16(d[et][ut][8 * s + 2 * j], d[et][ut][8 * s + 2 * j + 1]);
                    acc2[et] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[ut][s], __builtin_bit_cast(bf16x8, packed),
                                                                       acc2[et], 0, 0, 0);
                }
            }
        }
        float part[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};  // two partial sums per env tile: shorter dependent chains
        if (LAST_TANH) {
            // the "+ 1" and the w3-weighted sum as packed fp32 ops (v_pk_add_f32 / v_pk_fma_f32: two values per issue
            // slot); the partial sums keep their order (even registers in .x, odd ones in .y): bit-identical results
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 p2[2] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
#pragma unroll
            for (int reg = 0; reg < 16; reg += 2)
#pragma unroll
                for (int et = 0; et < 2; ++et) {
                    f32x2 e = {__builtin_amdgcn_exp2f(acc2[et][reg]), __builtin_amdgcn_exp2f(acc2[et][reg + 1])};
                    e += f32x2{1.0f, 1.0f};
                    const f32x2 r = {__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
                    p2[et] = __builtin_elementwise_fma(r, f32x2{w3[reg], w3[reg + 1]}, p2[et]);
                }
            part[0][0] = p2[0].x; part[0][1] = p2[0].y;
            part[1][0] = p2[1].x; part[1][1] = p2[1].y;
        } else
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
#pragma unroll
            for (int et = 0; et < 2; ++et) {
                const float v = acc2[et][reg];
                const float h = LAST_TANH ? __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v) + 1.0f) : fmaxf(v, 0.0f);
                part[et][reg & 1] = fmaf(h, w3[reg], part[et][reg & 1]);
            }
        // lanes 0-31 need tile 0's lo + hi halves, lanes 32-63 tile 1's: one swap + one add
        float s_lo, s_hi;
        half_swap(part[0][0] + part[0][1], part[1][0] + part[1][1], s_lo, s_hi);
        return s_lo + s_hi + b3;
    }

    __device__ float forward(const float (&obs)[2]) const { return tanh_fast(forward_pre(obs[0], obs[1])); }
};

}  // namespace ssc
