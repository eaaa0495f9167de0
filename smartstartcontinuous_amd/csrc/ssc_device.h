// ssc_device.h -- per-env device math shared by every kernel of libssc.so (gfx950).
//
// The functions are __host__ __device__ so that tests/host_harness can run the SAME source
// on the CPU (no GPU in the build container) to validate indexing / RNG / step ordering
// before a kernel ever reaches the GPU box.  The product only ever calls them from kernels.
//
// Citations are relative to the reference repository root.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssc.h"

#define SSC_HD __host__ __device__ __forceinline__

namespace ssc {

// ---------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. SC'11, Random123 constants).  Integer only => bit-exact
// with oracle/ssc_oracle.py:philox4x32_10.
// ---------------------------------------------------------------------------------------
struct u32x4 {
    uint32_t x, y, z, w;
};

SSC_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return u32x4{c0, c1, c2, c3};
}

// stream tags -- oracle/ssc_oracle.py TAG_*
enum : uint32_t { TAG_ACTION = 0, TAG_RESET = 1, TAG_OU = 2, TAG_MPC = 3, TAG_MPC_NOISE = 4, TAG_SS_EPISODE = 8 };

// key = (seed lo, seed hi); counter = (env lo, env hi, t lo, (t hi << 8) | tag)
SSC_HD u32x4 rng_words(uint64_t seed, uint64_t env_id, uint64_t t, uint32_t tag) {
    return philox4x32_10((uint32_t)env_id, (uint32_t)(env_id >> 32), (uint32_t)t,
                         (uint32_t)(((t >> 32) << 8) | tag), (uint32_t)seed, (uint32_t)(seed >> 32));
}

SSC_HD uint32_t pick(const u32x4 &w, uint32_t i) {
    return i == 0 ? w.x : (i == 1 ? w.y : (i == 2 ? w.z : w.w));
}

// uint32 -> [0,1) with 24 bits, then one fused multiply-add: low + span*u.
SSC_HD float uniform_f32(uint32_t x, float low, float span) {
    const float u = (float)(x >> 8) * (1.0f / 16777216.0f);
    return fmaf(u, span, low);
}

// Box-Muller.  u1 in (0,1], u2 in [0,1); second = false -> r cos(2 pi u2), true -> r sin(2 pi u2).  On the device
// the transcendental pipe is used directly: v_log_f32, v_sqrt_f32 and v_cos_f32 / v_sin_f32 (whose input is in
// revolutions, i.e. they evaluate cos(2*pi*u2) with no range reduction).
SSC_HD float gaussian_f32(uint32_t x0, uint32_t x1, bool second = false) {
    const float u1 = ((float)(x0 >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(x1 >> 8) * (1.0f / 16777216.0f);
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_sqrtf(-2.0f * __logf(u1));
    return r * (second ? __builtin_amdgcn_sinf(u2) : __builtin_amdgcn_cosf(u2));
#else
    const float r = sqrtf(-2.0f * logf(u1));
    return r * (second ? sinf(6.28318530717958647692f * u2) : cosf(6.28318530717958647692f * u2));
#endif
}

// The OU-noise gaussian of global step t (oracle/ssc_oracle.py:ou_gaussian): ONE Philox evaluation (counter
// t >> 2, TAG_OU) serves four consecutive steps -- word pair (x, y) for t & 2 == 0, (z, w) otherwise; each pair is
// one Box-Muller transform whose cos output goes to the even and whose sin output to the odd step.
SSC_HD float ou_gaussian_from_words(const u32x4 &w, uint64_t t) {
    const bool hi = (t & 2) != 0;
    return gaussian_f32(hi ? w.z : w.x, hi ? w.w : w.y, (t & 1) != 0);
}

// ---------------------------------------------------------------------------------------
// cos(x) for |x| <= 3*pi/2 (MountainCar needs cos(3*pos), pos in [-1.2, 0.6]).
// Even minimax polynomial (degree 10) on [-pi/2, pi/2] after the reflection
// cos(y) = -cos(pi - y); approximation error 2.5e-10, fp32 evaluation error < 1e-7
// (fit + check: DESIGN.md "cos").  ~14 VALU ops, no transcendental-pipe instruction.
// ---------------------------------------------------------------------------------------
SSC_HD float cos_bounded(float x) {
    const float y = fabsf(x);
    const float PI_HI = 3.14159274101257324f;      // float(pi)
    const float PI_LO = -8.74227765734758577e-8f;  // pi - float(pi)
    const bool refl = y > 1.57079637050628662f;
    const float z = refl ? ((PI_HI - y) + PI_LO) : y;  // PI_HI - y exact (Sterbenz) for y in [pi/2, 2pi]
    const float w = z * z;
    float p = -2.6022024712801795e-07f;
    p = fmaf(p, w, 2.475851033523213e-05f);
    p = fmaf(p, w, -0.001388832926750183f);
    p = fmaf(p, w, 0.04166663438081741f);
    p = fmaf(p, w, -0.5f);
    p = fmaf(p, w, 1.0f);
    return refl ? -p : p;
}

// ---------------------------------------------------------------------------------------
// MountainCar: Continuous_MountainCarEnv_Editted.step
// (smartstart/environments/continuous_mountain_car_editted.py:60-82), fp32.
// ---------------------------------------------------------------------------------------
struct McConst {
    float min_action, max_action, min_position, max_position, max_speed, goal_position, power;
    float reset_low, reset_span;
    int32_t max_episode_steps;  // <=0: none
};

SSC_HD McConst make_mc_const(const ssc_env_params &p) {
    McConst c;
    c.min_action = p.min_action; c.max_action = p.max_action;
    c.min_position = p.min_position; c.max_position = p.max_position;
    c.max_speed = p.max_speed; c.goal_position = p.goal_position; c.power = p.power;
    c.reset_low = p.reset_low; c.reset_span = p.reset_high - p.reset_low;
    c.max_episode_steps = p.max_episode_steps;
    return c;
}

// a is the RAW action: force uses the clipped action (:64), reward the raw one (:79).
SSC_HD void mc_step_one(const McConst &c, float &pos, float &vel, float a, float &rew, bool &goal) {
    const float force = fminf(fmaxf(a, c.min_action), c.max_action);          // :64
    float v = vel + fmaf(force, c.power, -0.0025f * cos_bounded(3.0f * pos));  // :66
    v = fminf(fmaxf(v, -c.max_speed), c.max_speed);                            // :67-68
    float p = pos + v;                                                         // :69
    p = fminf(fmaxf(p, c.min_position), c.max_position);                       // :70-71
    if (p == c.min_position && v < 0.0f) v = 0.0f;                             // :72
    goal = p >= c.goal_position;                                               // :74
    rew = (goal ? 100.0f : 0.0f) - a * a * 0.1f;                               // :76-79
    pos = p;
    vel = v;
}

SSC_HD void mc_reset_one(const McConst &c, const u32x4 &w, float &pos, float &vel) {
    pos = uniform_f32(w.x, c.reset_low, c.reset_span);  // :85
    vel = 0.0f;
}

// ---------------------------------------------------------------------------------------
// Pendulum-v0 (gym 0.10.5 PendulumEnv.step) [third-party; parity unpinned], fp32.
// ---------------------------------------------------------------------------------------
struct PendConst {
    float max_torque, max_speed, dt, g, m, l;
    int32_t v1_order;
    int32_t max_episode_steps;
};

SSC_HD PendConst make_pend_const(const ssc_env_params &p) {
    PendConst c;
    c.max_torque = p.max_torque; c.max_speed = p.pend_max_speed; c.dt = p.dt;
    c.g = p.g; c.m = p.m; c.l = p.l; c.v1_order = p.pend_v1_order;
    c.max_episode_steps = p.max_episode_steps;
    return c;
}

// The Pendulum arithmetic spells out every fused multiply-add and forbids implicit contraction: hipcc otherwise
// decides per call site which a*b+c become v_fma, and two kernels that inline the same source (single-step API,
// fused rollout, MPC rollout step) would round differently.
// ((x + pi) mod 2pi) - pi with python-modulo semantics (result of mod in [0, 2pi)).
SSC_HD float angle_normalize(float x) {
#pragma clang fp contract(off)
    const float TWO_PI = 6.28318530717958647692f;
    const float y = x + 3.14159265358979323846f;
    float r = fmaf(-TWO_PI, floorf(y * (1.0f / TWO_PI)), y);
    // floorf rounding can leave r marginally outside [0, 2pi)
    if (r < 0.0f) r += TWO_PI;
    if (r >= TWO_PI) r -= TWO_PI;
    return r - 3.14159265358979323846f;
}

SSC_HD void pend_step_one(const PendConst &c, float &th, float &thdot, float a, float &rew) {
#pragma clang fp contract(off)
    const float u = fminf(fmaxf(a, -c.max_torque), c.max_torque);
    const float an = angle_normalize(th);
    const float cost = fmaf(an, an, fmaf(0.1f * thdot, thdot, 0.001f * (u * u)));
    // sin(th + pi) = -sin(th)
    const float k1 = 3.0f * c.g / (2.0f * c.l), k2 = 3.0f / (c.m * c.l * c.l);
    float nd = fmaf(fmaf(k1, sinf(th), k2 * u), c.dt, thdot);
    float nt;
    if (c.v1_order) {
        nd = fminf(fmaxf(nd, -c.max_speed), c.max_speed);
        nt = fmaf(nd, c.dt, th);
    } else {
        nt = fmaf(nd, c.dt, th);
        nd = fminf(fmaxf(nd, -c.max_speed), c.max_speed);
    }
    th = nt;
    thdot = nd;
    rew = -cost;
}

// PendulumEnv._get_obs [third-party gym 0.10.5]: (cos th, sin th, thdot).  One definition for every kernel, so
// the single-step API, the fused rollouts and the MPC rollout step report the same bits.
SSC_HD void pend_observe_one(float th, float thdot, float &c, float &s, float &td) {
#ifdef __HIP_DEVICE_COMPILE__
    sincosf(th, &s, &c);
#else
    s = sinf(th);
    c = cosf(th);
#endif
    td = thdot;
}

SSC_HD void pend_reset_one(const u32x4 &w, float &th, float &thdot) {
    const float PI_F = 3.14159274101257324f;
    th = uniform_f32(w.x, -PI_F, PI_F - (-PI_F));
    thdot = uniform_f32(w.y, -1.0f, 2.0f);
}

// fast tanh on the transcendental pipe: 1 - 2/(2^(2x*log2 e) + 1); |err| ~ 1e-7.  No clamp is
// needed: 2^big = +inf -> rcp = 0 -> 1;  2^-big = 0 -> rcp(1) = 1 -> -1.  5 VALU ops (v_mul,
// v_exp_f32, v_add, v_rcp_f32, v_fma).
SSC_HD float tanh_fast(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float e = __builtin_amdgcn_exp2f(x * 2.88539008177792681472f);
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
#else
    const float e = exp2f(x * 2.88539008177792681472f);
    return 1.0f - 2.0f / (e + 1.0f);
#endif
}

// ReLU + fp32 -> bf16 for a PAIR of values in two instructions: v_cvt_pk_bf16_f32 then
// v_pk_max_i16(x, 0) -- a negative bf16 has its sign bit set, i.e. is a negative int16, so the
// integer max zeroes exactly the negative halves (and -0.0).  Half the VALU work of max+max+cvt.
typedef __bf16 ssc_bf16x2 __attribute__((ext_vector_type(2)));
typedef short ssc_s16x2 __attribute__((ext_vector_type(2)));
typedef float ssc_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int relu_pack_bf16(float a, float b) {
    const ssc_f32x2 v = {a, b};
    const ssc_bf16x2 p = __builtin_convertvector(v, ssc_bf16x2);
    ssc_s16x2 s = __builtin_bit_cast(ssc_s16x2, p);
    s = __builtin_elementwise_max(s, (ssc_s16x2)(0));
    return __builtin_bit_cast(int, s);
}

// The two cross-half exchanges of a 64-lane wave in ONE VALU op (v_permlane32_swap, gfx950):
// given a, b returns lo_lo = [a.lo | b.lo] and hi_hi = [a.hi | b.hi] (lo = lanes 0-31, hi = 32-63).
// Inline asm on purpose: ROCm 7.2's __builtin_amdgcn_permlane32_swap returns its FIRST result for
// both elements of the pair (the second output register is dropped in instruction selection;
// reproducer in DESIGN.md).  hipcc pads no hazards around asm, hence the s_nop brackets.
__device__ __forceinline__ void half_swap(float a, float b, float &lo_lo, float &hi_hi) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    lo_lo = a;
    hi_hi = b;
}

// tc.layers.layer_norm(center=True, scale=True) [third-party TF 1.5] over n values x[i * stride]: mean and biased variance
// (tf.nn.moments) summed in index order, variance epsilon 1e-12 -> (mean, 1 / sqrt(var + eps))
__device__ __forceinline__ void layer_norm_stats(const float *x, int n, int stride, float &mean, float &rstd) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s += x[i * stride];
    mean = s / (float)n;
    float v = 0.0f;
    for (int i = 0; i < n; ++i) {
        const float d = x[i * stride] - mean;
        v = fmaf(d, d, v);
    }
    rstd = 1.0f / sqrtf(v / (float)n + 1e-12f);
}

}  // namespace ssc
