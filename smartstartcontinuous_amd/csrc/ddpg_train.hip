// ddpg_train.hip -- the DDPG learner step on the GPU: DDPG_editted.train() + update_target_net()
// (DDPG_Baselines_editted/ddpg_editted.py:287-339; graph :127-133,168-199) for batch 64 and the
// 64-32 actor / critic of the shipped runs.
//
// The iterations form a serial chain through the parameters (the reference runs one per env step),
// so ONE workgroup executes `n_iters` of them per launch: 4 waves, lane = sample, every activation and
// every back-propagated delta of the batch lives in LDS as [unit][68]; in the dense passes a lane owns a
// unit (its weights arrive as one coalesced load per input row), a wave owns 16 samples whose
// activations are broadcast float4 LDS reads.  fp32, plain
// FMA chains in k order; the Adam moments, step counters and target networks are updated in place.
#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

constexpr int kB = 64;       // batch size = lanes of a wave
constexpr int kP = kB + 4;   // padded LDS row: 16-B aligned rows (float4 access over 4 samples); a stride of
                             // 68 dwords keeps both ds_read_b128 column gathers and ds_write_b128 conflict-free
constexpr int kTrainThreads = 256;

struct NetDims {
    int in, h1, h2, out;     // actor: obs -> h1 -> h2 -> act ; critic: obs -> h1 (+act) -> h2 -> 1
    int extra;               // rows concatenated to the first hidden layer (critic: act_dim, actor: 0)
    __device__ int oW1() const { return 0; }
    __device__ int ob1() const { return in * h1; }
    __device__ int oW2() const { return ob1() + h1; }
    __device__ int ob2() const { return oW2() + (h1 + extra) * h2; }
    __device__ int oW3() const { return ob2() + h2; }
    __device__ int ob3() const { return oW3() + h2 * out; }
    __device__ int total() const { return ob3() + out; }
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kSPW = kB / (kTrainThreads / 64);   // samples per wave = 16

// Z[j][b] = act(bias[j] + sum_i W[i][j] X[i][b]).  Lane = output unit j (W[i][j] is one coalesced load
// per input row i), wave w = samples [16w, 16w+16) (X rows are broadcast float4 reads).
__device__ __forceinline__ void dense_fwd(const float *__restrict__ W, const float *__restrict__ bias, int in, int out,
                                          const float *X, float *Z, int act) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b0 = wave * kSPW;
    for (int j0 = 0; j0 < out; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < out;
        const int jc = valid ? j : out - 1;
        f4 acc[kSPW / 4];
        const float bj = bias[jc];
#pragma unroll
        for (int q = 0; q < kSPW / 4; ++q) acc[q] = (f4)(bj);
#pragma unroll 4
        for (int i = 0; i < in; ++i) {
            const float w = W[i * out + jc];
#pragma unroll
            for (int q = 0; q < kSPW / 4; ++q) acc[q] += w * *reinterpret_cast<const f4 *>(X + i * kP + b0 + 4 * q);
        }
        if (valid) {
#pragma unroll
            for (int q = 0; q < kSPW / 4; ++q) {
                f4 v = acc[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (act == ACT_RELU) v[e] = fmaxf(v[e], 0.0f);
                    else if (act == ACT_TANH) v[e] = tanhf(v[e]);
                }
                *reinterpret_cast<f4 *>(Z + j * kP + b0 + 4 * q) = v;
            }
        }
    }
}

// dX[i - i0][b] = sum_j W[i][j] dZ[j][b] for input rows i in [i0, i1).  Lane = input row.
__device__ __forceinline__ void dense_bwd_in(const float *__restrict__ W, int out, int i0, int i1, const float *dZ,
                                             float *dX) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, b0 = wave * kSPW;
    for (int r0 = i0; r0 < i1; r0 += 64) {
        const int i = r0 + lane;
        const bool valid = i < i1;
        const int ic = valid ? i : i1 - 1;
        f4 acc[kSPW / 4];
#pragma unroll
        for (int q = 0; q < kSPW / 4; ++q) acc[q] = (f4)(0.0f);
#pragma unroll 4
        for (int j = 0; j < out; ++j) {
            const float w = W[ic * out + j];
#pragma unroll
            for (int q = 0; q < kSPW / 4; ++q) acc[q] += w * *reinterpret_cast<const f4 *>(dZ + j * kP + b0 + 4 * q);
        }
        if (valid) {
#pragma unroll
            for (int q = 0; q < kSPW / 4; ++q) *reinterpret_cast<f4 *>(dX + (i - i0) * kP + b0 + 4 * q) = acc[q];
        }
    }
}

struct AdamCfg {
    float a, beta1, beta2, eps;   // a = stepsize * sqrt(1 - b2^t) / (1 - b1^t) with t already incremented
};

// MpiAdam.update (baselines common/mpi_adam.py [third-party], ddpg_editted.py:326-327) on one element
__device__ __forceinline__ void adam_apply(float *theta, float *m, float *v, int idx, float g, const AdamCfg &c) {
    const float mi = c.beta1 * m[idx] + (1.0f - c.beta1) * g;
    const float vi = c.beta2 * v[idx] + (1.0f - c.beta2) * (g * g);
    m[idx] = mi;
    v[idx] = vi;
    theta[idx] += (-c.a) * mi / (sqrtf(vi) + c.eps);
}

// dW[i][j] = sum_b X[i][b] dZ[j][b]; db[j] = sum_b dZ[j][b]; applied straight into Adam.
__device__ __forceinline__ void weight_grad_adam(const float *X, const float *dZ, int in, int out, float *theta,
                                                 float *m, float *v, int offW, int offb, const AdamCfg &c) {
    for (int idx = threadIdx.x; idx < in * out; idx += kTrainThreads) {
        const int i = idx / out, j = idx - i * out;
        f4 g4 = (f4)(0.0f);
#pragma unroll 4
        for (int q = 0; q < kB / 4; ++q)
            g4 += *reinterpret_cast<const f4 *>(X + i * kP + 4 * q) * *reinterpret_cast<const f4 *>(dZ + j * kP + 4 * q);
        adam_apply(theta, m, v, offW + idx, (g4[0] + g4[1]) + (g4[2] + g4[3]), c);
    }
    for (int j = threadIdx.x; j < out; j += kTrainThreads) {
        f4 g4 = (f4)(0.0f);
        for (int q = 0; q < kB / 4; ++q) g4 += *reinterpret_cast<const f4 *>(dZ + j * kP + 4 * q);
        adam_apply(theta, m, v, offb + j, (g4[0] + g4[1]) + (g4[2] + g4[3]), c);
    }
}

struct TrainArgs {
    ssc_ddpg_desc d;
    ssc_replay_view rp;
    const int32_t *batch_idx;
    int32_t n_iters;
    float *losses;
};

__global__ __launch_bounds__(kTrainThreads) void ddpg_train_kernel(TrainArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ssc_ddpg_desc &d = g.d;
    const NetDims A{d.obs_dim, d.actor_h1, d.actor_h2, d.act_dim, 0};
    const NetDims C{d.obs_dim, d.critic_h1, d.critic_h2, 1, d.act_dim};
    const bool llt = d.last_layer_tanh != 0;
    const int act2 = llt ? ACT_TANH : ACT_RELU;
    const int tid = threadIdx.x, b = tid & 63;

    // ---- LDS carve (rows of kP floats) ------------------------------------------------------
    float *p = lds;
    auto take = [&](int rows) { float *q = p; p += rows * kP; return q; };
    float *S = take(d.obs_dim), *S2 = take(d.obs_dim);
    float *RT = take(3);                                  // r, terminal, y (target Q)
    float *X2 = take(C.h1 + d.act_dim);                   // critic: relu(layer 1) rows, then the action rows
    float *CA2 = take(C.h2), *DQ = take(1), *DZ2 = take(C.h2), *DZ1 = take(C.h1);
    float *U1 = take(A.h1), *U2 = take(A.h2), *PI = take(d.act_dim);
    float *DZ3A = take(d.act_dim), *DZ2A = take(A.h2), *DZ1A = take(A.h1);
    float *X2B = take(C.h1 + d.act_dim), *CB2 = take(C.h2), *DZB2 = take(C.h2);   // also the target-pass scratch
    __shared__ float red[2][kTrainThreads / 64];

    for (int it = 0; it < g.n_iters; ++it) {
        // ---- gather the batch (ReplayBuffer.sample_batch rows) -------------------------------------
        if (tid < kB) {
            const int64_t rec = g.batch_idx[(int64_t)it * kB + tid];
            for (int c = 0; c < d.obs_dim; ++c) {
                S[c * kP + tid] = g.rp.s[rec * d.obs_dim + c];
                S2[c * kP + tid] = g.rp.s2[rec * d.obs_dim + c];
            }
            for (int c = 0; c < d.act_dim; ++c) X2[(C.h1 + c) * kP + tid] = g.rp.a[rec * d.act_dim + c];
            RT[0 * kP + tid] = g.rp.r[rec];
            RT[1 * kP + tid] = g.rp.t[rec] ? 1.0f : 0.0f;
        }
        __syncthreads();
        // ---- target_Q = r + (1 - terminal) * gamma * Q'(s2, pi'(s2))      (ddpg_editted.py:132-133) ----
        {
            const float *ta = d.target_actor, *tc = d.target_critic;
            dense_fwd(ta + A.oW1(), ta + A.ob1(), A.in, A.h1, S2, DZ1A, ACT_RELU);
            dense_fwd(tc + C.oW1(), tc + C.ob1(), C.in, C.h1, S2, X2B, ACT_RELU);
            __syncthreads();
            dense_fwd(ta + A.oW2(), ta + A.ob2(), A.h1, A.h2, DZ1A, DZ2A, act2);
            __syncthreads();
            dense_fwd(ta + A.oW3(), ta + A.ob3(), A.h2, A.out, DZ2A, X2B + C.h1 * kP, ACT_TANH);
            __syncthreads();
            dense_fwd(tc + C.oW2(), tc + C.ob2(), C.h1 + d.act_dim, C.h2, X2B, CB2, act2);
            __syncthreads();
            dense_fwd(tc + C.oW3(), tc + C.ob3(), C.h2, 1, CB2, DZB2, ACT_NONE);
            __syncthreads();
            if (tid < kB) RT[2 * kP + tid] = RT[tid] + (1.0f - RT[kP + tid]) * d.gamma * DZB2[tid];
        }
        // ---- critic on (s, a): loss = mean((Q - y)^2)                              (:181) --------------
        dense_fwd(d.critic + C.oW1(), d.critic + C.ob1(), C.in, C.h1, S, X2, ACT_RELU);
        // ---- actor on s (independent of the critic pass)                           (:127) --------------
        dense_fwd(d.actor + A.oW1(), d.actor + A.ob1(), A.in, A.h1, S, U1, ACT_RELU);
        __syncthreads();
        dense_fwd(d.critic + C.oW2(), d.critic + C.ob2(), C.h1 + d.act_dim, C.h2, X2, CA2, act2);
        dense_fwd(d.actor + A.oW2(), d.actor + A.ob2(), A.h1, A.h2, U1, U2, act2);
        __syncthreads();
        dense_fwd(d.critic + C.oW3(), d.critic + C.ob3(), C.h2, 1, CA2, DQ, ACT_NONE);
        dense_fwd(d.actor + A.oW3(), d.actor + A.ob3(), A.h2, A.out, U2, PI, ACT_TANH);
        __syncthreads();
        float closs = 0.0f;
        if (tid < kB) {
            const float e = DQ[tid] - RT[2 * kP + tid];
            closs = e * e;
            DQ[tid] = 2.0f * e / (float)kB;                 // d loss / d q
            // critic input for the actor loss: relu(layer 1) is the same, the action is pi(s)
            for (int c = 0; c < d.act_dim; ++c) X2B[(C.h1 + c) * kP + tid] = PI[c * kP + tid];
        }
        for (int e = tid; e < C.h1 * kP; e += kTrainThreads) X2B[e] = X2[e];
        __syncthreads();
        // ---- critic backward (weights' deltas kept for the gradient pass) ---------------------------------
        {   // dz2 = (W3 dq) * act'(z2)
            const int wave = tid >> 6;
            for (int j = wave; j < C.h2; j += 4) {
                const float a2 = CA2[j * kP + b];
                const float da2 = d.critic[C.oW3() + j] * DQ[b];
                DZ2[j * kP + b] = da2 * (llt ? (1.0f - a2 * a2) : (a2 > 0.0f ? 1.0f : 0.0f));
            }
        }
        // ---- critic forward on (s, pi(s)): actor loss = -mean(Q)                    (:168) --------------
        dense_fwd(d.critic + C.oW2(), d.critic + C.ob2(), C.h1 + d.act_dim, C.h2, X2B, CB2, act2);
        __syncthreads();
        dense_bwd_in(d.critic + C.oW2(), C.h2, 0, C.h1, DZ2, DZ1);
        dense_fwd(d.critic + C.oW3(), d.critic + C.ob3(), C.h2, 1, CB2, DZ3A, ACT_NONE);   // q(s, pi) -> DZ3A row 0 (temp)
        __syncthreads();
        float aloss = 0.0f;
        if (tid < kB) aloss = -DZ3A[tid];
        {   // relu mask of critic layer 1; dzb2 for the action gradient with dq = -1/B
            for (int e = tid; e < C.h1 * kB; e += kTrainThreads) {
                const int i = e >> 6, bb = e & 63;
                if (!(X2[i * kP + bb] > 0.0f)) DZ1[i * kP + bb] = 0.0f;
            }
            const int wave = tid >> 6;
            for (int j = wave; j < C.h2; j += 4) {
                const float a2 = CB2[j * kP + b];
                const float da2 = d.critic[C.oW3() + j] * (-1.0f / (float)kB);
                DZB2[j * kP + b] = da2 * (llt ? (1.0f - a2 * a2) : (a2 > 0.0f ? 1.0f : 0.0f));
            }
        }
        __syncthreads();
        // d(-mean Q)/d(action) = rows h1.. of W2 dzb2, then through the actor's output tanh
        dense_bwd_in(d.critic + C.oW2(), C.h2, C.h1, C.h1 + d.act_dim, DZB2, DZ3A);
        __syncthreads();
        for (int e = tid; e < d.act_dim * kB; e += kTrainThreads) {
            const int c = e >> 6, bb = e & 63;
            const float pi = PI[c * kP + bb];
            DZ3A[c * kP + bb] *= (1.0f - pi * pi);
        }
        __syncthreads();
        dense_bwd_in(d.actor + A.oW3(), A.out, 0, A.h2, DZ3A, DZ2A);
        __syncthreads();
        for (int e = tid; e < A.h2 * kB; e += kTrainThreads) {
            const int j = e >> 6, bb = e & 63;
            const float u2 = U2[j * kP + bb];
            DZ2A[j * kP + bb] *= llt ? (1.0f - u2 * u2) : (u2 > 0.0f ? 1.0f : 0.0f);
        }
        __syncthreads();
        dense_bwd_in(d.actor + A.oW2(), A.h2, 0, A.h1, DZ2A, DZ1A);
        __syncthreads();
        for (int e = tid; e < A.h1 * kB; e += kTrainThreads) {
            const int i = e >> 6, bb = e & 63;
            if (!(U1[i * kP + bb] > 0.0f)) DZ1A[i * kP + bb] = 0.0f;
        }
        __syncthreads();
        // ---- losses (means over the batch) -----------------------------------------------------------
        {
            float v0 = closs, v1 = aloss;
#pragma unroll
            for (int msk = 32; msk >= 1; msk >>= 1) { v0 += __shfl_xor(v0, msk); v1 += __shfl_xor(v1, msk); }
            if ((tid & 63) == 0) { red[0][tid >> 6] = v0; red[1][tid >> 6] = v1; }
        }
        // ---- gradients + MpiAdam, all from the OLD parameters' deltas               (:326-327) ------------
        const int tA = g.d.adam_t[0] + 1, tC = g.d.adam_t[1] + 1;
        // bias-correction factors in f64: 1 - 0.999^t loses 5 digits in fp32 for small t
        const AdamCfg ca{(float)((double)d.actor_lr * sqrt(1.0 - pow((double)d.beta2, (double)tA)) /
                                 (1.0 - pow((double)d.beta1, (double)tA))), d.beta1, d.beta2, d.epsilon};
        const AdamCfg cc{(float)((double)d.critic_lr * sqrt(1.0 - pow((double)d.beta2, (double)tC)) /
                                 (1.0 - pow((double)d.beta1, (double)tC))), d.beta1, d.beta2, d.epsilon};
        __syncthreads();   // every read of the old parameters is done
        weight_grad_adam(S, DZ1, C.in, C.h1, d.critic, d.adam_m_critic, d.adam_v_critic, C.oW1(), C.ob1(), cc);
        weight_grad_adam(X2, DZ2, C.h1 + d.act_dim, C.h2, d.critic, d.adam_m_critic, d.adam_v_critic, C.oW2(), C.ob2(), cc);
        weight_grad_adam(CA2, DQ, C.h2, 1, d.critic, d.adam_m_critic, d.adam_v_critic, C.oW3(), C.ob3(), cc);
        weight_grad_adam(S, DZ1A, A.in, A.h1, d.actor, d.adam_m_actor, d.adam_v_actor, A.oW1(), A.ob1(), ca);
        weight_grad_adam(U1, DZ2A, A.h1, A.h2, d.actor, d.adam_m_actor, d.adam_v_actor, A.oW2(), A.ob2(), ca);
        weight_grad_adam(U2, DZ3A, A.h2, A.out, d.actor, d.adam_m_actor, d.adam_v_actor, A.oW3(), A.ob3(), ca);
        __syncthreads();   // the block's own global writes are visible to itself after the barrier
        // ---- update_target_net: theta' <- (1 - tau) theta' + tau theta              (:338-339) ------------
        for (int e = tid; e < A.total(); e += kTrainThreads)
            d.target_actor[e] = (1.0f - d.tau) * d.target_actor[e] + d.tau * d.actor[e];
        for (int e = tid; e < C.total(); e += kTrainThreads)
            d.target_critic[e] = (1.0f - d.tau) * d.target_critic[e] + d.tau * d.critic[e];
        if (tid == 0) {
            g.d.adam_t[0] = tA;
            g.d.adam_t[1] = tC;
            if (g.losses != nullptr) {
                g.losses[2 * it + 0] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)kB;
                g.losses[2 * it + 1] = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)kB;
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

}  // namespace ssc

using namespace ssc;

extern "C" int ssc_ddpg_train(const ssc_ddpg_desc *d, const ssc_replay_view *rp, const int32_t *d_batch_idx,
                              int32_t n_iters, float *d_losses, ssc_stream_t stream) {
    SSC_REQUIRE(d && rp, "ssc_ddpg_train: NULL descriptor");
    SSC_REQUIRE(n_iters >= 0, "ssc_ddpg_train: n_iters < 0");
    if (d->batch_size != kB)
        return set_error(SSC_EUNSUPPORTED, "ssc_ddpg_train: batch_size %d (only 64, the value of every shipped run)",
                         d->batch_size);
    SSC_REQUIRE(d->obs_dim >= 1 && d->obs_dim <= SSC_MAX_STATE && d->act_dim >= 1 && d->act_dim <= SSC_MAX_ACT,
                "ssc_ddpg_train: obs_dim/act_dim out of range");
    SSC_REQUIRE(d->actor_h1 >= 1 && d->actor_h2 >= 1 && d->critic_h1 >= 1 && d->critic_h2 >= 1,
                "ssc_ddpg_train: bad hidden sizes");
    if (n_iters == 0) return SSC_OK;
    SSC_REQUIRE(d->actor && d->critic && d->target_actor && d->target_critic && d->adam_m_actor && d->adam_v_actor &&
                    d->adam_m_critic && d->adam_v_critic && d->adam_t,
                "ssc_ddpg_train: NULL parameter / optimiser pointer");
    SSC_REQUIRE(rp->s && rp->a && rp->r && rp->t && rp->s2 && rp->capacity > 0 && d_batch_idx,
                "ssc_ddpg_train: NULL replay pointer");
    const int rows = 2 * d->obs_dim + 3 + (d->critic_h1 + d->act_dim) + d->critic_h2 + 1 + d->critic_h2 + d->critic_h1 +
                     d->actor_h1 + d->actor_h2 + d->act_dim + d->act_dim + d->actor_h2 + d->actor_h1 +
                     (d->critic_h1 + d->act_dim) + 2 * d->critic_h2;
    const size_t lds = (size_t)rows * kP * sizeof(float);
    if (lds > 150 * 1024)
        return set_error(SSC_EUNSUPPORTED, "ssc_ddpg_train: hidden sizes need %zu B of LDS (> 150 KB)", lds);
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(ddpg_train_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "hipFuncSetAttribute(ddpg_train_kernel)");
        if (rc) return rc;
    }
    TrainArgs g{*d, *rp, d_batch_idx, n_iters, d_losses};
    hipLaunchKernelGGL(ddpg_train_kernel, dim3(1), dim3(kTrainThreads), lds, as_stream(stream), g);
    return check_launch("ssc_ddpg_train");
}
