// Host-side harness: runs the __host__ __device__ per-env functions of
// smartstartcontinuous_amd/csrc/ssc_device.h on the CPU so that `-m "not gpu"` tests can check
// the device math (RNG keying, step ordering, fp32 error budget) against the oracle without a GPU.
// TEST INFRASTRUCTURE: never loaded by the product.
#include "ssc_device.h"

using namespace ssc;

extern "C" {

void h_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
    const u32x4 r = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

void h_rng_words(uint64_t seed, uint64_t env_id, uint64_t t, uint32_t tag, uint32_t *out) {
    const u32x4 r = rng_words(seed, env_id, t, tag);
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}

void h_uniform(int64_t n, const uint32_t *x, float low, float span, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = uniform_f32(x[i], low, span);
}

void h_cos_bounded(int64_t n, const float *x, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = cos_bounded(x[i]);
}

void h_mc_step(const ssc_env_params *p, int64_t n, float *pos, float *vel, const float *act, float *rew,
               uint8_t *goal) {
    const McConst c = make_mc_const(*p);
    for (int64_t i = 0; i < n; ++i) {
        bool g;
        mc_step_one(c, pos[i], vel[i], act[i], rew[i], g);
        goal[i] = g;
    }
}

void h_mc_reset(const ssc_env_params *p, int64_t n, uint64_t seed, uint64_t env_id0, uint64_t t, float *pos,
                float *vel) {
    const McConst c = make_mc_const(*p);
    for (int64_t i = 0; i < n; ++i) mc_reset_one(c, rng_words(seed, env_id0 + i, t, TAG_RESET), pos[i], vel[i]);
}

void h_pend_step(const ssc_env_params *p, int64_t n, float *th, float *thdot, const float *act, float *rew) {
    const PendConst c = make_pend_const(*p);
    for (int64_t i = 0; i < n; ++i) pend_step_one(c, th[i], thdot[i], act[i], rew[i]);
}

void h_pend_reset(int64_t n, uint64_t seed, uint64_t env_id0, uint64_t t, float *th, float *thdot) {
    for (int64_t i = 0; i < n; ++i) pend_reset_one(rng_words(seed, env_id0 + i, t, TAG_RESET), th[i], thdot[i]);
}

void h_angle_normalize(int64_t n, const float *x, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = angle_normalize(x[i]);
}

void h_tanh_fast(int64_t n, const float *x, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = tanh_fast(x[i]);
}

void h_gaussian(int64_t n, const uint32_t *x0, const uint32_t *x1, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = gaussian_f32(x0[i], x1[i]);
}

}  // extern "C"
