// rollout.hip -- the fused K-step rollout kernel: the rlTrain inner loop
// (smartstart/reinforcementLearningCore/rlTrain.py:75-100) for n independent envs.
//
//   for k in range(K):  action = agent.get_action(obs)        (:81)
//                       obs2, r, done = env.step(action)      (:84)  [+ gym TimeLimit]
//                       record (obs, a, r, done, obs2)        (:91-94, replay_buffer.py:53)
//                       if done: Summary.append((len, ret)); env.reset()   (:97-114)
//
// MI355X mapping: one thread owns one env; pos/vel/elapsed/return (and the OU state) live
// in VGPRs for all K steps, so the only HBM traffic is the transition log itself --
// [K][n] SoA columns, each wave-instruction storing 256 contiguous bytes (1 dword/lane).
// 25 B per MountainCar env-step, 33 B per Pendulum env-step: the
// kernel is HBM-write-bound by construction.  Episode records go to a ring through a
// wave-aggregated atomic cursor; chunk statistics are reduced wave -> block -> one f64
// atomic per block.
#include "actor_device.h"
#ifndef SSC_WIDE_ACTOR_ET
#define SSC_WIDE_ACTOR_ET 2   // env tiles per wave of the LDS-staged wide actor (1: two waves per SIMD at 65 536 envs)
#endif
#include "mpc_device.h"
#include "ssc_device.h"
#include "ssc_host.h"


namespace ssc {

struct RolloutArgs {
    int64_t n;
    int32_t K;
    ssc_rollout_state st;
    ssc_transition_log log;
    ssc_episode_ring ring;
    double *stats;
    uint64_t seed, env_id0, step0;
    int32_t has_log, has_ring;
    // one_base: every fp32 column of the log starts within 4 GB above obs[0] (a packed chunk: column c at + c * n floats),
    // so a step's stores share ONE running row pointer and differ in a 32-bit per-lane offset (col_off[c] + 4 * env)
    // computed once -- instead of seven 64-bit row pointers rebuilt from k * row_stride in every step (35 of the ~100
    // instructions of a random-policy step were that scalar address arithmetic)
    int32_t one_base;
    uint32_t col_off[2 * SSC_MAX_OBS + 2];   // bytes: obs[0..OBS), act, rew, obs2[0..OBS)
};

struct PolicyArgs {
    float act_low, act_span, act_high;
    ActorWeights actor;
    float ou_mu, ou_sigma, ou_theta, ou_dt, ou_eps;
    const float *d_eps;  // ssc_ou_desc::d_epsilon: the device value wins over ou_eps and the noise path always runs
};

// ------------------------------------------------------------------------------- envs --
struct McEnv {
    using Const = McConst;
    static constexpr int OBS = 2;
    float pos, vel;
    __device__ void load(float a, float b) { pos = a; vel = b; }
    __device__ void observe(float (&o)[OBS]) const { o[0] = pos; o[1] = vel; }
    __device__ void step(const Const &c, float a, float &rew, bool &goal) { mc_step_one(c, pos, vel, a, rew, goal); }
    __device__ void reset(const Const &c, const u32x4 &w) { mc_reset_one(c, w, pos, vel); }
    __device__ float s0() const { return pos; }
    __device__ float s1() const { return vel; }
};

struct PendEnv {
    using Const = PendConst;
    static constexpr int OBS = 3;
    float th, thdot;
    __device__ void load(float a, float b) { th = a; thdot = b; }
    __device__ void observe(float (&o)[OBS]) const { pend_observe_one(th, thdot, o[0], o[1], o[2]); }
    __device__ void step(const Const &c, float a, float &rew, bool &goal) {
        pend_step_one(c, th, thdot, a, rew);
        goal = false;
    }
    __device__ void reset(const Const &, const u32x4 &w) { pend_reset_one(w, th, thdot); }
    __device__ float s0() const { return th; }
    __device__ float s1() const { return thdot; }
};

// --------------------------------------------------------------------------- policies --
// A Philox4x32-10 evaluation that can be advanced a few rounds at a time, so that the
// (independent) RNG chain of the NEXT 4-step group is interleaved with the serial dynamics
// chain of the current steps: at 65 536 envs there is one wave per SIMD and instruction-level
// parallelism is the only latency hiding available (DESIGN.md "rollout kernel").
struct PhiloxPipe {
    uint32_t c0, c1, c2, c3, k0, k1;
    __device__ __forceinline__ void start(uint64_t seed, uint64_t env_id, uint64_t t, uint32_t tag) {
        c0 = (uint32_t)env_id; c1 = (uint32_t)(env_id >> 32);
        c2 = (uint32_t)t; c3 = (uint32_t)(((t >> 32) << 8) | tag);
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
    }
    template <int R>
    __device__ __forceinline__ void rounds() {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
            const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
            const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
            c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
        }
    }
    // Keeps the rounds issued so far in the basic block where they were written (the optimiser
    // would otherwise sink the whole evaluation to its single use at the end of the group).
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)); }
    __device__ __forceinline__ u32x4 get() const { return u32x4{c0, c1, c2, c3}; }
};

// Policy_Random.get_action = U(low, high)  (NN_Dynamics_Model/policy_random.py:14-15).
// One Philox call serves 4 consecutive steps: counter t>>2, word t&3.
template <int OBS>
struct RandomPolicy {
    static constexpr bool kPipelined = true;
    static constexpr bool kFusedActor = false;
    static constexpr int kLanesPerEnv = 1;
    u32x4 cache;
    float low, span;
    __device__ void init(const PolicyArgs &pa, const RolloutArgs &, int64_t) { low = pa.act_low; span = pa.act_span; }
    __device__ float act(const float (&)[OBS], uint64_t seed, uint64_t env_id, uint64_t t, bool first) {
        if (first || (t & 3) == 0) cache = rng_words(seed, env_id, t >> 2, TAG_ACTION);  // wave-uniform branch
        return uniform_f32(pick(cache, (uint32_t)(t & 3)), low, span);
    }
    __device__ float from_word(uint32_t w) const { return uniform_f32(w, low, span); }
    __device__ void on_reset() {}
    __device__ void store(const RolloutArgs &, int64_t) const {}
};

// DDPG_Baselines_agent.get_action (smartstart/RLAgents/DDPG_Baselines_agent.py:206-234):
//   a = actor(obs) + epsilon * OU();  a = clip(a, -1, 1)      (ddpg_editted.py:262-271)
//   a = scale(scale(a))                                        (DDPG_Baselines_agent.py:236-240)
// OU [third-party baselines 0.1.5 ddpg/noise.py]: x += theta*(mu-x)*dt + sigma*sqrt(dt)*N(0,1).
// One Philox call serves the gaussians of 4 consecutive steps (counter t >> 2, both Box-Muller outputs of both
// word pairs: ou_gaussian_from_words).
template <class Net, int OBS>
struct ActorPolicy {
    static constexpr bool kPipelined = false;
    static constexpr bool kFusedActor = false;
    static constexpr int kLanesPerEnv = Net::kLanesPerEnv;
    Net net;
    u32x4 cache;
    float ou_x, mu, sig_sqrt_dt, theta_dt, eps;
    float low, high, obs_clip;
    bool identity_scale, noisy;

    __device__ void init(const PolicyArgs &pa, const RolloutArgs &ra, int64_t i) {
        net.init(pa.actor);
        mu = pa.ou_mu;
        sig_sqrt_dt = pa.ou_sigma * sqrtf(pa.ou_dt);
        theta_dt = pa.ou_theta * pa.ou_dt;
        eps = fmaxf(pa.d_eps != nullptr ? *pa.d_eps : pa.ou_eps, 0.0f);  // DDPG_Baselines_agent.py:74
        noisy = pa.d_eps != nullptr || eps > 0.0f;
        ou_x = (ra.st.ou_x != nullptr) ? ra.st.ou_x[i] : 0.0f;
        low = pa.act_low;
        high = pa.act_high;
        obs_clip = pa.actor.obs_clip;
        identity_scale = (low == -1.0f && high == 1.0f);
    }
    __device__ float scale(float a) const {  // DDPG_Baselines_agent.py:236-240
        a = fminf(fmaxf(a, -1.0f), 1.0f);
        return identity_scale ? a : fmaf((a + 1.0f) * 0.5f, high - low, low);
    }
    __device__ float act(const float (&obs)[OBS], uint64_t seed, uint64_t env_id, uint64_t t, bool first) {
        float oc[OBS];   // the network sees the clipped observation (ddpg_editted.py:106-109); the log keeps the raw one
#pragma unroll
        for (int c = 0; c < OBS; ++c) oc[c] = clip_obs(obs[c], obs_clip);
        float a = net.forward(oc);
        if (noisy) {  // wave-uniform
            if (first || (t & 3) == 0) cache = rng_words(seed, env_id, t >> 2, TAG_OU);
            const float g = ou_gaussian_from_words(cache, t);
            ou_x = fmaf(sig_sqrt_dt, g, fmaf(theta_dt, mu - ou_x, ou_x));
            a = fmaf(ou_x, eps, a);  // ddpg_editted.py:267-270
        }
        a = fminf(fmaxf(a, -1.0f), 1.0f);  // ddpg_editted.py:271 (action_range = (-1, 1))
        return scale(scale(a));
    }
    __device__ float from_word(uint32_t) const { return 0.0f; }
    __device__ void on_reset() { ou_x = 0.0f; }  // DDPG_Baselines_agent.end_episode -> noise reset (:255-258)
    __device__ void store(const RolloutArgs &ra, int64_t i) const {
        if (ra.st.ou_x != nullptr) ra.st.ou_x[i] = ou_x;
    }
};

// The same policy for the shape BASELINE config 3 runs (MountainCar: 2 observations, actor h1 <= 64 / h2 <= 32 on the
// bf16 MFMA, action bounds [-1, 1], epsilon > 0), written so that one env-step is ONE basic block: every
// wave-uniform choice of ActorPolicy (last_layer_tanh, epsilon > 0, the identity scale, the even/odd Philox word
// pair) is a template parameter or resolved by the caller's two-step loop.  With one wave per SIMD the only thing
// that can run under the MFMAs and in the bubbles of the dependent action -> env.step chain is independent work of
// the same wave, and the scheduler only moves code inside a basic block: the noise generation (half a Philox
// evaluation + Box-Muller per step) is that work.
template <bool LAST_TANH>
struct ActorPolicyFused {
    static constexpr bool kPipelined = false;
    static constexpr bool kFusedActor = true;
    static constexpr int kLanesPerEnv = 1;
    ActorMfma2<LAST_TANH> net;
    float ou_x, mu, sig_sqrt_dt, theta_dt, eps;

    __device__ void init(const PolicyArgs &pa, const RolloutArgs &ra, int64_t i) {
        net.init(pa.actor);
        mu = pa.ou_mu;
        sig_sqrt_dt = pa.ou_sigma * sqrtf(pa.ou_dt);
        theta_dt = pa.ou_theta * pa.ou_dt;
        eps = pa.d_eps != nullptr ? fmaxf(*pa.d_eps, 0.0f) : pa.ou_eps;  // host value: > 0 (dispatch)
        ou_x = ra.st.ou_x[i];
    }
    // Box-Muller like gaussian_f32 in three pieces, so that the radius and the sin output of a word pair computed in
    // the even step are reused by the odd one.  -2 ln(u1) = (-2 ln 2) log2(u1) on the bare v_log_f32: u1 >= 2^-24 is
    // never denormal, so the range handling logf() wraps around the instruction (10 more VALU ops) is dead weight.
    static __device__ __forceinline__ float bm_radius(uint32_t x0) {
        const float u1 = ((float)(x0 >> 8) + 1.0f) * (1.0f / 16777216.0f);
        return __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));
    }
    static __device__ __forceinline__ float bm_cos(uint32_t x1) { return __builtin_amdgcn_cosf((float)(x1 >> 8) * (1.0f / 16777216.0f)); }
    static __device__ __forceinline__ float bm_sin(uint32_t x1) { return __builtin_amdgcn_sinf((float)(x1 >> 8) * (1.0f / 16777216.0f)); }
    static __device__ __forceinline__ float gaussian(const u32x4 &w, uint64_t t) {  // = ou_gaussian_from_words
        const bool hi = (t & 2) != 0;
        const uint32_t x0 = hi ? w.z : w.x, x1 = hi ? w.w : w.y;
        return bm_radius(x0) * ((t & 1) ? bm_sin(x1) : bm_cos(x1));
    }
    // DDPG_Baselines_agent.get_action for this lane's env; g = the step's N(0,1) draw.  Every multiply-add is
    // spelled out: the step body exists in several copies (head, the four bodies of the main loop, tail) and
    // hipcc decides contraction per copy -- a rollout must not depend on how it was cut into chunks.
    __device__ __forceinline__ float act(const float (&obs)[2], float g) {
        const float pre = net.forward_pre(obs[0], obs[1]);
        ou_x = fmaf(sig_sqrt_dt, g, fmaf(theta_dt, mu - ou_x, ou_x));
        const float a = fmaf(ou_x, eps, tanh_fast(pre));  // ddpg_editted.py:267-270
        // clip (:271); scale(scale(.)) (DDPG_Baselines_agent.py:236-240) is the identity on a clipped action for
        // bounds [-1, 1]
        return fminf(fmaxf(a, -1.0f), 1.0f);
    }
    __device__ float from_word(uint32_t) const { return 0.0f; }
    __device__ void on_reset() { ou_x = 0.0f; }
    __device__ void store(const RolloutArgs &ra, int64_t i) const { ra.st.ou_x[i] = ou_x; }
};

// (wave-uniform row pointer) + (32-bit per-lane byte offset) as the SGPR-base form of global_store: the row pointer
// is pinned to SGPRs with readfirstlane (free for a value that already is uniform; otherwise the optimiser
// re-associates base + (row + lane) and every store pays a 64-bit VALU add), and the result is an explicit
// global-address-space pointer (the integer round trip hides the kernel-argument provenance, and a generic
// pointer would turn the stores into flat_store).
#define SSC_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ SSC_GLOBAL T *lane_ptr(T *row_uniform, uint32_t lane_byte_off) {
    const uint64_t v = reinterpret_cast<uint64_t>(row_uniform);
    // (the builtin returns a SIGNED int: without the casts the low half is sign-extended into the high one)
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    const uint64_t u = ((uint64_t)hi << 32) | (uint64_t)lo;
    return (SSC_GLOBAL T *)((SSC_GLOBAL char *)u + lane_byte_off);
}

// ----------------------------------------------------------------------------- kernel --
// Per-thread rollout state + the one-step body shared by the generic and the pipelined loop.
template <class EnvT, class PolT, bool LOG, bool ONEBASE = false>
struct Rollout {
    static constexpr int OBS = EnvT::OBS;
    const float *rowp;          // ONEBASE: obs[0] row of the current step (wave-uniform, advanced by the step)
    uint8_t *donep;             // ... and the done row
    uint32_t coff[2 * OBS + 2]; // ... per-lane byte offsets of the fp32 columns: col_off[c] + 4 * env
    const typename EnvT::Const &ec;
    const RolloutArgs &ra;
    EnvT env;
    PolT pol;
    float obs[OBS];
    int32_t el;
    float ep_ret, sum_r;
    int32_t n_goal, n_eps;
    uint64_t env_id;
    uint32_t voff;  // byte offset of this env inside a log row (fp32 columns); row bases are wave-uniform
    bool active;    // this lane's env exists (ragged last wave) ...
    bool owner;     // ... and this lane is the one that reports it (kLanesPerEnv == 2: lanes 0-31 only)

    __device__ __forceinline__ Rollout(const typename EnvT::Const &ec_, const RolloutArgs &ra_) : ec(ec_), ra(ra_) {}

    // Everything after the action is chosen: env.step, bookkeeping, log row k, rare reset path.
    // Inactive lanes of a ragged last wave shadow env n-1 and store the same values to the same
    // addresses as its owner -- benign, and it keeps the hot loop free of exec-mask branches.
    __device__ __forceinline__ void step(float a, int32_t k) {
        float rew;
        bool goal;
        float obs2[OBS];
        env.step(ec, a, rew, goal);
        env.observe(obs2);
        el += 1;
        // bitwise, not short-circuit: no exec-mask region in the hot loop
        const bool done = goal | ((ec.max_episode_steps > 0) & (el >= ec.max_episode_steps));
        ep_ret += rew;
        sum_r += rew;
        n_goal += goal ? 1 : 0;
        if constexpr (LOG && ONEBASE) {
            if (PolT::kLanesPerEnv == 1 || owner) {
                float *rp = const_cast<float *>(rowp);
                uint32_t vo[2 * OBS + 2];
#pragma unroll
                for (int c = 0; c < 2 * OBS + 2; ++c) { vo[c] = coff[c]; asm volatile("" : "+v"(vo[c])); }   // (32-bit offsets: see below)
#pragma unroll
                for (int c = 0; c < OBS; ++c) __builtin_nontemporal_store(obs[c], lane_ptr(rp, vo[c]));
                __builtin_nontemporal_store(a, lane_ptr(rp, vo[OBS]));
                __builtin_nontemporal_store(rew, lane_ptr(rp, vo[OBS + 1]));
                uint32_t vd = voff >> 2;
                asm volatile("" : "+v"(vd));
                *lane_ptr(donep, vd) = (uint8_t)(done ? 1 : 0);            // plain store: see the generic branch
#pragma unroll
                for (int c = 0; c < OBS; ++c) __builtin_nontemporal_store(obs2[c], lane_ptr(rp, vo[OBS + 2 + c]));
            }
            rowp += ra.log.row_stride;          // scalar: one 64-bit add per pointer and step
            donep += ra.log.done_row_stride;
        } else if (LOG && (PolT::kLanesPerEnv == 1 || owner)) {
            const int64_t row = (int64_t)k * ra.log.row_stride;  // wave-uniform
            const int64_t drow = (int64_t)k * ra.log.done_row_stride;
            // The lane offset is re-materialised as a 32-bit value in every step: hoisted out of the loop it becomes a
            // 64-bit VGPR pair and every store then pays a v_lshl_add_u64 (2 passes) for base + offset; a 32-bit
            // VGPR offset against the wave-uniform row base is the SGPR-base form of global_store (no VALU op).
            uint32_t vo = voff;
            asm volatile("" : "+v"(vo));
#define SSC_STORE_F32(val, ptr) __builtin_nontemporal_store((val), (ptr))
#pragma unroll
            for (int c = 0; c < OBS; ++c) SSC_STORE_F32(obs[c], lane_ptr(ra.log.obs[c] + row, vo));
            SSC_STORE_F32(a, lane_ptr(ra.log.act + row, vo));
            SSC_STORE_F32(rew, lane_ptr(ra.log.rew + row, vo));
            // a PLAIN store for the byte column: a wave writes 64 B of it, half a 128-byte line; as a non-temporal
            // store that half line goes to memory on its own, as a plain one the L2 merges it with the neighbour
            // wave's half first.  tools/membw.hip `series .. rows` (profiles/r02/membw_rows.txt): fp32 columns nt +
            // byte column plain 6.0 TB/s, all nt 5.6, all plain 5.4
            *lane_ptr(ra.log.done + drow, vo >> 2) = (uint8_t)(done ? 1 : 0);
#pragma unroll
            for (int c = 0; c < OBS; ++c) SSC_STORE_F32(obs2[c], lane_ptr(ra.log.obs2[c] + row, vo));
        }
#pragma unroll
        for (int c = 0; c < OBS; ++c) obs[c] = obs2[c];
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(done) != 0, 0)) reset_path(done, k);  // wave-uniform, rare
    }

    __device__ __forceinline__ void reset_path(bool done, int32_t k) {
        {
            if (done) {
                if (ra.has_ring && active && owner) {
                    const uint32_t slot = atomicAdd(ra.ring.cursor, 1u);
                    if (slot < (uint32_t)ra.ring.capacity) {
                        ra.ring.env_id[slot] = (int64_t)env_id;
                        ra.ring.length[slot] = el;
                        ra.ring.ret[slot] = ep_ret;
                    }
                }
                n_eps += 1;
                env.reset(ec, rng_words(ra.seed, env_id, ra.step0 + (uint64_t)k, TAG_RESET));
                el = 0;
                ep_ret = 0.0f;
                pol.on_reset();
                env.observe(obs);
            }
        }
    }
};

template <class EnvT, class PolT, bool LOG, bool ONEBASE = false>
__global__ __launch_bounds__(kBlock) void rollout_kernel(typename EnvT::Const ec, PolicyArgs pa, RolloutArgs ra) {
    constexpr int LPE = PolT::kLanesPerEnv;
    constexpr int kEnvsPerBlock = kBlock / LPE;
    // LPE == 2: a wave covers 32 envs, env j of the wave lives on lanes j and j+32
    const int in_block = (LPE == 1) ? (int)threadIdx.x : (int)((threadIdx.x >> 6) * 32 + (threadIdx.x & 31));
    const int64_t gi = (int64_t)blockIdx.x * kEnvsPerBlock + in_block;
    Rollout<EnvT, PolT, LOG, ONEBASE> r(ec, ra);
    r.active = gi < ra.n;
    r.owner = (LPE == 1) || ((threadIdx.x & 63) < 32);
    const int64_t i = r.active ? gi : ra.n - 1;
    r.env_id = ra.env_id0 + (uint64_t)i;
    r.voff = (uint32_t)i * 4u;
    if constexpr (LOG && ONEBASE) {
        r.rowp = ra.log.obs[0];
        r.donep = ra.log.done;
#pragma unroll
        for (int c = 0; c < 2 * EnvT::OBS + 2; ++c) r.coff[c] = ra.col_off[c] + r.voff;
    }

    r.env.load(ra.st.s0[i], ra.st.s1[i]);
    r.el = ra.st.steps[i];
    r.ep_ret = ra.st.ep_ret[i];
    r.pol.init(pa, ra, i);
    r.env.observe(r.obs);
    r.sum_r = 0.0f;
    r.n_goal = 0;
    r.n_eps = 0;

    // Retire the state loads HERE.  Otherwise hipcc places their s_waitcnt vmcnt(N) inside the
    // loop, where vmcnt also counts the log stores: every iteration would then wait for the
    // previous iteration's stores to land (0x0F70 = vmcnt(0), expcnt/lgkmcnt untouched).
    __builtin_amdgcn_s_waitcnt(0x0F70);

    int32_t k = 0;
    if constexpr (PolT::kPipelined) {
        // head: single steps until the global step index is a multiple of 4
        for (; k < ra.K && ((ra.step0 + (uint64_t)k) & 3) != 0; ++k)
            r.step(r.pol.act(r.obs, ra.seed, r.env_id, ra.step0 + (uint64_t)k, k == 0), k);
        if (k + 4 <= ra.K) {
            u32x4 cur = rng_words(ra.seed, r.env_id, (ra.step0 + (uint64_t)k) >> 2, TAG_ACTION);
            for (; k + 4 <= ra.K; k += 4) {
                PhiloxPipe nx;
                nx.start(ra.seed, r.env_id, ((ra.step0 + (uint64_t)k) >> 2) + 1, TAG_ACTION);
                nx.rounds<3>(); nx.pin();
                r.step(r.pol.from_word(cur.x), k);
                nx.rounds<3>(); nx.pin();
                r.step(r.pol.from_word(cur.y), k + 1);
                nx.rounds<2>(); nx.pin();
                r.step(r.pol.from_word(cur.z), k + 2);
                nx.rounds<2>(); nx.pin();
                r.step(r.pol.from_word(cur.w), k + 3);
                cur = nx.get();
            }
        }
        // tail (< 4 steps)
        const int32_t k_tail = k;
        for (; k < ra.K; ++k)
            r.step(r.pol.act(r.obs, ra.seed, r.env_id, ra.step0 + (uint64_t)k, k == k_tail), k);
    } else if constexpr (PolT::kFusedActor) {
        // Four steps per iteration: one Philox evaluation (counter t >> 2) serves them (ou_gaussian_from_words),
        // and the evaluation for the NEXT four is split 3 + 3 + 2 + 2 rounds over the four step bodies.
        uint64_t t = ra.step0;
        u32x4 cur = rng_words(ra.seed, r.env_id, t >> 2, TAG_OU);
        for (; k < ra.K && (t & 3) != 0; ++k, ++t) r.step(r.pol.act(r.obs, PolT::gaussian(cur, t)), k);  // head
        if (k + 4 <= ra.K) {
            if (k != 0) cur = rng_words(ra.seed, r.env_id, t >> 2, TAG_OU);
            for (; k + 4 <= ra.K; k += 4, t += 4) {
                PhiloxPipe nx;
                nx.start(ra.seed, r.env_id, (t >> 2) + 1, TAG_OU);
                nx.rounds<3>(); nx.pin();
                const float r0 = PolT::bm_radius(cur.x), s0 = PolT::bm_sin(cur.y);
                r.step(r.pol.act(r.obs, r0 * PolT::bm_cos(cur.y)), k);
                nx.rounds<3>(); nx.pin();
                r.step(r.pol.act(r.obs, r0 * s0), k + 1);
                nx.rounds<2>(); nx.pin();
                const float r1 = PolT::bm_radius(cur.z), s1 = PolT::bm_sin(cur.w);
                r.step(r.pol.act(r.obs, r1 * PolT::bm_cos(cur.w)), k + 2);
                nx.rounds<2>(); nx.pin();
                r.step(r.pol.act(r.obs, r1 * s1), k + 3);
                cur = nx.get();
            }
        } else if (k < ra.K && k != 0) {
            cur = rng_words(ra.seed, r.env_id, t >> 2, TAG_OU);
        }
        for (; k < ra.K; ++k, ++t) r.step(r.pol.act(r.obs, PolT::gaussian(cur, t)), k);  // tail (< 4 steps, t & 3 == 0 at its start)
    } else {
        for (; k < ra.K; ++k)
            r.step(r.pol.act(r.obs, ra.seed, r.env_id, ra.step0 + (uint64_t)k, k == 0), k);
    }

    if (r.active && r.owner) {
        ra.st.s0[i] = r.env.s0();
        ra.st.s1[i] = r.env.s1();
        ra.st.steps[i] = r.el;
        ra.st.ep_ret[i] = r.ep_ret;
        r.pol.store(ra, i);
    }

    if (ra.stats != nullptr) {
        const bool cnt = r.active && r.owner;
        double v[4] = {cnt ? (double)r.sum_r : 0.0, cnt ? (double)r.n_goal : 0.0, cnt ? (double)ra.K : 0.0,
                       cnt ? (double)r.n_eps : 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v[q] += __shfl_xor(v[q], m);
        __shared__ double red[kBlock / 64][4];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave][q] = v[q];
        __syncthreads();
        if (threadIdx.x < 4) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kBlock / 64; ++w) s += red[w][threadIdx.x];
            atomicAdd(ra.stats + threadIdx.x, s);
        }
    }
}

// Can the fp32 columns of the log be addressed from obs[0] with 32-bit offsets?  (always for a packed chunk)
static void set_one_base(RolloutArgs &ra, int obs_dim) {
    const float *cols[2 * SSC_MAX_OBS + 2];
    int nc = 0;
    for (int c = 0; c < obs_dim; ++c) cols[nc++] = ra.log.obs[c];
    cols[nc++] = ra.log.act;
    cols[nc++] = ra.log.rew;
    for (int c = 0; c < obs_dim; ++c) cols[nc++] = ra.log.obs2[c];
    ra.one_base = 1;
    const uintptr_t b0 = reinterpret_cast<uintptr_t>(cols[0]);
    for (int c = 0; c < nc; ++c) {
        const uintptr_t p = reinterpret_cast<uintptr_t>(cols[c]);
        if (p < b0 || (p - b0) + (uint64_t)ra.n * 4u > 0xFFFFFFFFull) { ra.one_base = 0; break; }
        ra.col_off[c] = (uint32_t)(p - b0);
    }
}

template <class EnvT, class PolT>
static int launch_rollout(const typename EnvT::Const &ec, const PolicyArgs &pa, const RolloutArgs &ra,
                          hipStream_t stream) {
    const dim3 grid(blocks_for(ra.n, kBlock / PolT::kLanesPerEnv));
    // One running row pointer + per-column lane offsets (ONEBASE) takes ~30 scalar address instructions out of every
    // step: 12 % faster for the actor policies, whose step is instruction-bound (0.254 -> 0.224 ms per 65 536 x 256).
    // The random policy is HBM-bound and gets SLOWER with it (0.242 -> 0.27-0.28 ms per 65 536 x 1024, median; the minimum
    // stays at 0.237): with the stores in one burst AND with the stores spread over the step by scheduling fences -- so
    // it is not the burst.  The launch-time distribution widens, as if the 1024 waves, freed of ~30 scalar instructions
    // of per-step jitter, fell into step with each other.  It keeps the per-column pointers.
    constexpr bool kOneBase = !PolT::kPipelined;
    if (ra.has_log && ra.one_base && kOneBase)
        hipLaunchKernelGGL((rollout_kernel<EnvT, PolT, true, kOneBase>), grid, dim3(kBlock), 0, stream, ec, pa, ra);
    else if (ra.has_log)
        hipLaunchKernelGGL((rollout_kernel<EnvT, PolT, true>), grid, dim3(kBlock), 0, stream, ec, pa, ra);
    else
        hipLaunchKernelGGL((rollout_kernel<EnvT, PolT, false>), grid, dim3(kBlock), 0, stream, ec, pa, ra);
    return check_launch("ssc_rollout");
}

int validate_mc_params(const ssc_env_params *p, const char *who);  // env_step.hip

template <class EnvT>
static int dispatch_policy(const typename EnvT::Const &ec, const ssc_policy_desc *pol, const PolicyArgs &pa,
                           const RolloutArgs &ra, hipStream_t stream) {
    constexpr int OBS = EnvT::OBS;
    if (pol->kind == SSC_POLICY_RANDOM) return launch_rollout<EnvT, RandomPolicy<OBS>>(ec, pa, ra, stream);
    const ssc_actor_desc &a = pol->actor;
    if (a.obs_dim != OBS)
        return set_error(SSC_EINVAL, "ssc_rollout: actor obs_dim %d != env obs_dim %d", a.obs_dim, OBS);
    if (a.act_dim != 1) return set_error(SSC_EUNSUPPORTED, "ssc_rollout: act_dim %d (only 1)", a.act_dim);
    if (a.precision == SSC_PREC_F32) {
        if (a.h1 == 64 && a.h2 == 32)
            return launch_rollout<EnvT, ActorPolicy<ActorF32<OBS, 64, 32>, OBS>>(ec, pa, ra, stream);
        if (a.h1 == 64 && a.h2 == 64)      // Actor_Editted's own default sizes (models_editted.py:23)
            return launch_rollout<EnvT, ActorPolicy<ActorF32<OBS, 64, 64>, OBS>>(ec, pa, ra, stream);
        return set_error(SSC_EUNSUPPORTED,
                         "ssc_rollout: fused fp32 actor supports h1-h2 = 64-32 and 64-64 (got %d-%d); use the bf16 MFMA "
                         "path or step the env with ssc_actor_forward + ssc_*_step",
                         a.h1, a.h2);
    }
    if (a.precision == SSC_PREC_BF16_MFMA) {
        if constexpr (OBS == 2) {
            // the shipped shape on unit action bounds with exploration noise on: the straight-line fused policy
            // (MountainCar observations are bounded by the env's own clamps: an observation clip at or beyond them
            // -- DDPG's (-5, 5) against |pos| <= 1.2, |vel| <= 0.07 -- never acts, so the fused policy skips it)
            const float obs_bound = fmaxf(fmaxf(fabsf(ec.min_position), fabsf(ec.max_position)), fabsf(ec.max_speed));
            const bool clip_inert = !(a.obs_clip > 0.0f) || a.obs_clip >= obs_bound;
            if (a.h1 <= 64 && a.h2 <= 32 && pa.act_low == -1.0f && pa.act_high == 1.0f && (pa.ou_eps > 0.0f || pa.d_eps != nullptr) && clip_inert) {
                if (a.last_layer_tanh) return launch_rollout<EnvT, ActorPolicyFused<true>>(ec, pa, ra, stream);
                return launch_rollout<EnvT, ActorPolicyFused<false>>(ec, pa, ra, stream);
            }
        }
        if (a.h1 <= 64 && a.h2 <= 32)
            return launch_rollout<EnvT, ActorPolicy<ActorMfma<OBS, 2, 1, 2>, OBS>>(ec, pa, ra, stream);
        if (a.h1 <= 128 && a.h2 <= 64)
            return launch_rollout<EnvT, ActorPolicy<ActorMfma<OBS, 4, 2, 2>, OBS>>(ec, pa, ra, stream);
        // the wide shapes of the reference's grid (200-100): W2 fragments staged in LDS (ActorMfmaLds)
        if (a.h1 <= 224 && a.h2 <= 128)
            return launch_rollout<EnvT, ActorPolicy<ActorMfmaLds<OBS, 7, 4, SSC_WIDE_ACTOR_ET>, OBS>>(ec, pa, ra, stream);
        return set_error(SSC_EUNSUPPORTED, "ssc_rollout: MFMA actor supports h1 <= 224, h2 <= 128 (got %d-%d)",
                         a.h1, a.h2);
    }
    return set_error(SSC_EINVAL, "ssc_rollout: unknown precision %d", a.precision);
}


// ---------------------------------------------------------------- MPC-policy rollout step --
// Everything of one rollout(K, 'mpc') step that follows the scoring, for P envs (one navigation problem each)
// in one launch: executed action, env.step, log row, statistics, episode record, navigator bookkeeping,
// auto-reset, planning state of the next step.  The step index t and the log row k come from device counters
// that the launch advances itself (the block that finishes last), so the whole step is HIP-graph replayable.
struct MpcStepArgs {
    MpcArgs nav;
    int32_t *cur_idx;
    const int32_t *start_idx;
    int32_t *actions_done;
    uint8_t *at_goal;
    int32_t give_up, final_steps;
    const float *A;
    const int32_t *best_idx;
    float noise;
    uint64_t noise_seed, problem_id0;
    uint64_t *d_t;
    int32_t *d_k, *ticket;
    float *plan_state;
};

// the vectorised SmartStartContinuous step (ssc_smartstart_rollout_step): per-env mode on top of the MPC step
struct SsStepArgs {
    uint8_t *mode;
    int32_t *plan_of;
    const float *actor_out, *d_eta, *d_eps;
    const int32_t *d_pool;
    float ou_mu, ou_sig_sqrt_dt, ou_theta_dt, act_low, act_high;
    uint8_t *mode_log;
    int64_t mode_log_stride;
    int32_t *n_live;   // ssc_smartstart_step::d_n_live: zeroed here for the next step's ssc_nav_compact
};

template <class EnvT, bool SS = false>
__global__ __launch_bounds__(kBlock) void mpc_rollout_step_kernel(typename EnvT::Const ec, RolloutArgs ra, MpcStepArgs ma, SsStepArgs sa) {
    constexpr int OBS = EnvT::OBS;
    // the navigator plans in observation space (checked on the host): with the dimension a constant the waypoint / radii loads of
    // the bookkeeping below are straight-line and go out together -- as a run-time bound every one of them was a branch with a
    // load and a full wait behind it, a dozen dependent round trips in a kernel that is one wave per SIMD
    ma.nav.d = OBS;
    const int64_t gi = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (SS && gi == 0 && sa.n_live != nullptr) *sa.n_live = 0;   // simulation and scoring of this step are behind us in the stream
    const bool active = gi < ra.n;
    const int64_t i = active ? gi : ra.n - 1;
    const uint64_t t = *ma.d_t;
    const int32_t k = *ma.d_k;
    float rew = 0.0f;
    bool goal = false, done = false;
    if (active) {
        EnvT env;
        env.load(ra.st.s0[i], ra.st.s1[i]);
        int32_t el = ra.st.steps[i];
        float ep_ret = ra.st.ep_ret[i];
        float obs[OBS], obs2[OBS];
        env.observe(obs);
        // the navigator's bookkeeping state and plan: requested here, with the env's state, not behind the log stores below
        // (which the compiler cannot move loads across)
        int idx = ma.cur_idx[i];
        int done_act = ma.actions_done[i];
        const NavPlan plan = nav_plan_load(ma.nav, (int)i);
        // SmartStartContinuous.get_action (smartexplorationcontinuous.py:307-317): the navigator while smart_start_pathing,
        // the base agent otherwise
        bool navigating = true;
        float ou_x = 0.0f;
        if (SS) {
            navigating = sa.mode[i] != 0;
            ou_x = ra.st.ou_x[i];
            if (sa.mode_log != nullptr) sa.mode_log[(int64_t)k * sa.mode_log_stride + i] = navigating ? 1 : 0;
        }
        float a;
        if (navigating) {
            // action = best_sequence[0] + noise * N(0,1), no clip (NND_MB_agent.py:353-356; ssc_mpc_select_action's draw)
            const int64_t row = i * ma.nav.N + ma.best_idx[i];
            a = ma.A[row * ma.nav.H];   // act_dim == 1 (the envs of this engine)
            if (ma.noise != 0.0f) {
                const u32x4 w = rng_words(ma.noise_seed, ma.problem_id0 + (uint64_t)i, t, TAG_MPC_NOISE);
                a += ma.noise * gaussian_f32(w.x, w.y);
            }
        } else {
            // DDPG_Baselines_agent.get_action (:206-240): actor + epsilon * OU, clip, scale twice -- ActorPolicy::act's
            // arithmetic and noise stream; the OU state moves on these steps only (the agent is not asked while navigating)
            a = sa.actor_out[i];
            const float eps = fmaxf(*sa.d_eps, 0.0f);
            if (eps > 0.0f) {
                const u32x4 w = rng_words(ra.seed, ra.env_id0 + (uint64_t)i, t >> 2, TAG_OU);
                const float g = ou_gaussian_from_words(w, t);
                ou_x = fmaf(sa.ou_sig_sqrt_dt, g, fmaf(sa.ou_theta_dt, sa.ou_mu - ou_x, ou_x));
                a = fmaf(ou_x, eps, a);
            }
            a = fminf(fmaxf(a, -1.0f), 1.0f);
            const bool identity = sa.act_low == -1.0f && sa.act_high == 1.0f;
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                a = fminf(fmaxf(a, -1.0f), 1.0f);
                a = identity ? a : fmaf((a + 1.0f) * 0.5f, sa.act_high - sa.act_low, sa.act_low);
            }
        }
        env.step(ec, a, rew, goal);
        env.observe(obs2);
        el += 1;
        done = goal || (ec.max_episode_steps > 0 && el >= ec.max_episode_steps);
        ep_ret += rew;
        if (ra.has_log) {
            const int64_t lr = (int64_t)k * ra.log.row_stride + i, ld = (int64_t)k * ra.log.done_row_stride + i;
#pragma unroll
            for (int c = 0; c < OBS; ++c) {
                ra.log.obs[c][lr] = obs[c];
                ra.log.obs2[c][lr] = obs2[c];
            }
            ra.log.act[lr] = a;
            ra.log.rew[lr] = rew;
            ra.log.done[ld] = done ? 1 : 0;
        }
        // NND_MB_agent.get_action counts the action (:340), observe advances the waypoint (:360-373)
        float x[SSC_MAX_STATE];
#pragma unroll
        for (int c = 0; c < SSC_MAX_STATE; ++c) x[c] = (c < OBS) ? obs2[c < OBS ? c : 0] : 0.0f;
        bool at_goal = false;
        if (navigating) {
            done_act += 1;
            at_goal = nav_observe_plan(ma.nav, plan, x, idx, done_act, ma.give_up, ma.final_steps);
        }
        if (ma.at_goal != nullptr) ma.at_goal[i] = at_goal ? 1 : 0;
        if (SS && navigating && at_goal) navigating = false;     // :336-339: the base agent takes over
        if (done) {
            if (ra.has_ring) {
                const uint32_t slot = atomicAdd(ra.ring.cursor, 1u);
                if (slot < (uint32_t)ra.ring.capacity) {
                    ra.ring.env_id[slot] = (int64_t)(ra.env_id0 + (uint64_t)i);
                    ra.ring.length[slot] = el;
                    ra.ring.ret[slot] = ep_ret;
                }
            }
            env.reset(ec, rng_words(ra.seed, ra.env_id0 + (uint64_t)i, t, TAG_RESET));
            el = 0;
            ep_ret = 0.0f;
            idx = ma.start_idx[i];    // start_new_episode_plan (:383-384)
            done_act = 0;
            if (!SS && ra.st.ou_x != nullptr) ra.st.ou_x[i] = 0.0f;
            if (SS) {
                ou_x = 0.0f;              // DDPG_Baselines_agent.end_episode: noise reset (:255-258)
                // SmartStartContinuous.start_new_episode (:341-370) on the reset state
                navigating = false;
                const u32x4 w = rng_words(ra.seed, ra.env_id0 + (uint64_t)i, t, TAG_SS_EPISODE);
                const int first = sa.d_pool[0], count = sa.d_pool[1], slots = sa.d_pool[2];
                if (count > 0 && (float)(w.x >> 8) * (1.0f / 16777216.0f) <= *sa.d_eta) {   // np.random.rand() <= eta (:350)
                    const int q = (first + (int)(w.y % (uint32_t)count)) % slots;
                    sa.plan_of[i] = q;    // read by the scorer of the next step (same stream)
                    idx = 0;
                    // close_enough_to_goal(state) with a fresh plan (:425-432): the goal test alone
                    float xr[SSC_MAX_STATE], obs_r[OBS];
                    env.observe(obs_r);
#pragma unroll
                    for (int c = 0; c < SSC_MAX_STATE; ++c) xr[c] = (c < OBS) ? obs_r[c < OBS ? c : 0] : 0.0f;
                    const int W = ma.nav.wp_len[q], d = ma.nav.d;
                    float inv_r[SSC_MAX_STATE];
#pragma unroll
                    for (int c = 0; c < SSC_MAX_STATE; ++c) inv_r[c] = (c < d) ? 1.0f / ma.nav.radii[q * d + c] : 0.0f;
                    const float *goal_wp = ma.nav.wp + ((int64_t)ma.nav.wp_off[q] + W - 1) * d;
                    navigating = !(ell_dist(xr, goal_wp, inv_r, d) <= ma.nav.theta);
                }
            }
        }
        if (SS) {
            sa.mode[i] = navigating ? 1 : 0;
            ra.st.ou_x[i] = ou_x;
        }
        ra.st.s0[i] = env.s0();
        ra.st.s1[i] = env.s1();
        ra.st.steps[i] = el;
        ra.st.ep_ret[i] = ep_ret;
        ma.cur_idx[i] = idx;
        ma.actions_done[i] = done_act;
        env.observe(obs);
#pragma unroll
        for (int c = 0; c < OBS; ++c) ma.plan_state[i * OBS + c] = obs[c];
    }
    __shared__ double red[kBlock / 64][4];
    if (ra.stats != nullptr) {
        double v[4] = {active ? (double)rew : 0.0, (active && goal) ? 1.0 : 0.0, active ? 1.0 : 0.0, (active && done) ? 1.0 : 0.0};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) v[q] += __shfl_xor(v[q], m);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave][q] = v[q];
    }
    __syncthreads();
    if (ra.stats != nullptr && threadIdx.x < 4) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) s += red[w][threadIdx.x];
        atomicAdd(ra.stats + threadIdx.x, s);
    }
    // every block has read *d_t / *d_k before it gets here; the last one to arrive advances them.  The ticket orders
    // nothing but those two READS (this lane's have returned after the wait; the other lanes consumed theirs before the
    // barrier), so it is a relaxed agent-scope add: a __threadfence() here is an L2 write-back + invalidate per block --
    // it was 15 of this kernel's 37 us at 65 536 envs.  What the step wrote is published by the kernel boundary.
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (__hip_atomic_fetch_add(ma.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
            *ma.ticket = 0;
            *ma.d_t = t + 1;
            *ma.d_k = k + 1;
        }
    }
}

}  // namespace ssc

using namespace ssc;

extern "C" int ssc_rollout(const ssc_env_params *p, const ssc_policy_desc *policy, int64_t n, int32_t K,
                           const ssc_rollout_state *state, const ssc_transition_log *log,
                           const ssc_episode_ring *ring, double *d_stats, uint64_t seed, uint64_t env_id0,
                           uint64_t step0, ssc_stream_t stream) {
    SSC_REQUIRE(p && policy && state, "ssc_rollout: NULL descriptor");
    SSC_REQUIRE(n >= 0 && K >= 0, "ssc_rollout: n = %lld, K = %d", (long long)n, K);
    SSC_REQUIRE(n <= ((int64_t)1 << 30), "ssc_rollout: n = %lld > 2^30 envs per launch", (long long)n);
    SSC_REQUIRE(policy->kind == SSC_POLICY_RANDOM || policy->kind == SSC_POLICY_ACTOR,
                "ssc_rollout: unknown policy kind %d", policy->kind);
    if (n == 0 || K == 0) return SSC_OK;
    SSC_REQUIRE(state->s0 && state->s1 && state->steps && state->ep_ret, "ssc_rollout: NULL state column");
    SSC_REQUIRE(policy->act_low <= policy->act_high, "ssc_rollout: act_low > act_high");
    const int obs_dim = (p->kind == SSC_ENV_MOUNTAINCAR) ? 2 : 3;
    RolloutArgs ra;
    ra.n = n;
    ra.K = K;
    ra.st = *state;
    ra.has_log = log != nullptr;
    ra.has_ring = ring != nullptr;
    ra.one_base = 0;
    if (log) {
        ra.log = *log;
        for (int c = 0; c < obs_dim; ++c)
            SSC_REQUIRE(log->obs[c] && log->obs2[c], "ssc_rollout: NULL log obs column %d", c);
        SSC_REQUIRE(log->act && log->rew && log->done, "ssc_rollout: NULL log column");
        SSC_REQUIRE(log->row_stride >= 0 && log->done_row_stride >= 0, "ssc_rollout: negative row stride");
        if (ra.log.row_stride == 0) ra.log.row_stride = n;
        if (ra.log.done_row_stride == 0) ra.log.done_row_stride = n;
        SSC_REQUIRE(ra.log.row_stride >= n && ra.log.done_row_stride >= n, "ssc_rollout: row stride < n");
        set_one_base(ra, obs_dim);
    } else {
        ra.log = ssc_transition_log{};
    }
    if (ring) {
        ra.ring = *ring;
        SSC_REQUIRE(ring->env_id && ring->length && ring->ret && ring->cursor && ring->capacity >= 0,
                    "ssc_rollout: bad episode ring");
    } else {
        ra.ring = ssc_episode_ring{};
    }
    ra.stats = d_stats;
    ra.seed = seed;
    ra.env_id0 = env_id0;
    ra.step0 = step0;

    PolicyArgs pa{};
    pa.act_low = policy->act_low;
    pa.act_high = policy->act_high;
    pa.act_span = policy->act_high - policy->act_low;
    if (policy->kind == SSC_POLICY_ACTOR) {
        const ssc_actor_desc &a = policy->actor;
        SSC_REQUIRE(a.W1 && a.b1 && a.W2 && a.b2 && a.W3 && a.b3, "ssc_rollout: NULL actor weight pointer");
        SSC_REQUIRE(a.h1 > 0 && a.h2 > 0, "ssc_rollout: bad actor sizes");
        pa.actor = ActorWeights{a.W1, a.b1, a.W2, a.b2, a.W3, a.b3, a.obs_dim, a.h1, a.h2, a.last_layer_tanh, a.obs_clip,
                                a.ln1_g, a.ln1_b, a.ln2_g, a.ln2_b};
        const bool ln = a.ln1_g != nullptr;
        SSC_REQUIRE(ln == (a.ln1_b != nullptr) && ln == (a.ln2_g != nullptr) && ln == (a.ln2_b != nullptr),
                    "ssc_rollout: the four LayerNorm pointers come together");
        if (ln && a.precision != SSC_PREC_F32)
            return set_error(SSC_EUNSUPPORTED, "ssc_rollout: LayerNorm actors run on the fp32 policy (precision SSC_PREC_F32)");
        pa.ou_mu = policy->ou.mu;
        pa.ou_sigma = policy->ou.sigma;
        pa.ou_theta = policy->ou.theta;
        pa.ou_dt = policy->ou.dt;
        pa.ou_eps = policy->ou.epsilon;
        pa.d_eps = policy->ou.d_epsilon;
        SSC_REQUIRE(!(pa.ou_eps > 0.0f || pa.d_eps != nullptr) || state->ou_x != nullptr, "ssc_rollout: OU noise needs state->ou_x");
    }
    if (p->kind == SSC_ENV_MOUNTAINCAR) {
        if (int rc = validate_mc_params(p, "ssc_rollout")) return rc;
        return dispatch_policy<McEnv>(make_mc_const(*p), policy, pa, ra, as_stream(stream));
    }
    if (p->kind == SSC_ENV_PENDULUM)
        return dispatch_policy<PendEnv>(make_pend_const(*p), policy, pa, ra, as_stream(stream));
    return set_error(SSC_EINVAL, "ssc_rollout: unknown env kind %d", p->kind);
}

// argument checks and packing shared by ssc_mpc_rollout_step and ssc_smartstart_rollout_step
static int build_mpc_step_args(const char *fn, const ssc_env_params *p, const ssc_mpc_problems *pr, const ssc_mpc_nav_state *nav,
                               const float *d_A, const int32_t *d_best_idx, float noise_amount, uint64_t noise_seed,
                               uint64_t problem_id0, const ssc_rollout_state *state, const ssc_transition_log *log,
                               const ssc_episode_ring *ring, double *d_stats, uint64_t env_seed, uint64_t env_id0,
                               uint64_t *d_t, int32_t *d_k, int32_t *d_ticket, float *d_plan_state, RolloutArgs &ra, MpcStepArgs &ma) {
    SSC_REQUIRE(p && pr && nav && state, "%s: NULL descriptor", fn);
    const int64_t n = pr->n_problems;
    SSC_REQUIRE(n >= 0 && pr->n_samples >= 1 && pr->horizon >= 1, "%s: bad problem sizes", fn);
    const int obs_dim = (p->kind == SSC_ENV_MOUNTAINCAR) ? 2 : 3;
    SSC_REQUIRE(pr->state_dim == obs_dim, "%s: the navigator plans in observation space (%d dims), "
                                          "problems have %d", fn, obs_dim, pr->state_dim);
    SSC_REQUIRE(pr->wp && pr->wp_off && pr->radii && nav->cur_idx && nav->start_idx && nav->actions_done,
                "%s: NULL navigator pointer", fn);
    SSC_REQUIRE((pr->plan_of == nullptr) == (pr->wp_len == nullptr), "%s: plan_of and wp_len go together", fn);
    SSC_REQUIRE(state->s0 && state->s1 && state->steps && state->ep_ret, "%s: NULL state column", fn);
    SSC_REQUIRE(d_A && d_best_idx && d_t && d_k && d_ticket && d_plan_state, "%s: NULL device pointer", fn);
    ra.n = n;
    ra.K = 1;
    ra.st = *state;
    ra.has_log = log != nullptr;
    ra.has_ring = ring != nullptr;
    ra.one_base = 0;
    if (log) {
        ra.log = *log;
        for (int c = 0; c < obs_dim; ++c)
            SSC_REQUIRE(log->obs[c] && log->obs2[c], "%s: NULL log obs column %d", fn, c);
        SSC_REQUIRE(log->act && log->rew && log->done, "%s: NULL log column", fn);
        if (ra.log.row_stride == 0) ra.log.row_stride = n;
        if (ra.log.done_row_stride == 0) ra.log.done_row_stride = n;
        SSC_REQUIRE(ra.log.row_stride >= n && ra.log.done_row_stride >= n, "%s: row stride < n", fn);
    } else {
        ra.log = ssc_transition_log{};
    }
    if (ring) {
        ra.ring = *ring;
        SSC_REQUIRE(ring->env_id && ring->length && ring->ret && ring->cursor && ring->capacity >= 0,
                    "%s: bad episode ring", fn);
    } else {
        ra.ring = ssc_episode_ring{};
    }
    ra.stats = d_stats;
    ra.seed = env_seed;
    ra.env_id0 = env_id0;
    ra.step0 = 0;
    ma = MpcStepArgs{};
    ma.nav.P = pr->n_problems; ma.nav.N = pr->n_samples; ma.nav.H = pr->horizon; ma.nav.d = pr->state_dim;
    ma.nav.wp = pr->wp; ma.nav.left = pr->left; ma.nav.radii = pr->radii; ma.nav.wp_off = pr->wp_off;
    ma.nav.cur_idx = pr->cur_idx; ma.nav.theta = pr->theta; ma.nav.plan_of = pr->plan_of; ma.nav.wp_len = pr->wp_len;
    ma.cur_idx = nav->cur_idx; ma.start_idx = nav->start_idx; ma.actions_done = nav->actions_done; ma.at_goal = nav->at_goal;
    ma.give_up = nav->give_up_after; ma.final_steps = nav->final_steps;
    ma.A = d_A; ma.best_idx = d_best_idx; ma.noise = noise_amount; ma.noise_seed = noise_seed; ma.problem_id0 = problem_id0;
    ma.d_t = d_t; ma.d_k = d_k; ma.ticket = d_ticket; ma.plan_state = d_plan_state;
    return SSC_OK;
}

extern "C" int ssc_mpc_rollout_step(const ssc_env_params *p, const ssc_mpc_problems *pr, const ssc_mpc_nav_state *nav,
                                    const float *d_A, const int32_t *d_best_idx, float noise_amount, uint64_t noise_seed,
                                    uint64_t problem_id0, const ssc_rollout_state *state, const ssc_transition_log *log,
                                    const ssc_episode_ring *ring, double *d_stats, uint64_t env_seed, uint64_t env_id0,
                                    uint64_t *d_t, int32_t *d_k, int32_t *d_ticket, float *d_plan_state,
                                    ssc_stream_t stream) {
    RolloutArgs ra;
    MpcStepArgs ma;
    if (int rc = build_mpc_step_args("ssc_mpc_rollout_step", p, pr, nav, d_A, d_best_idx, noise_amount, noise_seed, problem_id0, state,
                                     log, ring, d_stats, env_seed, env_id0, d_t, d_k, d_ticket, d_plan_state, ra, ma))
        return rc;
    const int64_t n = ra.n;
    if (n == 0) return SSC_OK;
    hipStream_t s = as_stream(stream);
    if (p->kind == SSC_ENV_MOUNTAINCAR) {
        if (int rc = validate_mc_params(p, "ssc_mpc_rollout_step")) return rc;
        hipLaunchKernelGGL(mpc_rollout_step_kernel<McEnv>, dim3(blocks_for(n)), dim3(kBlock), 0, s, make_mc_const(*p), ra, ma, SsStepArgs{});
    } else if (p->kind == SSC_ENV_PENDULUM) {
        hipLaunchKernelGGL(mpc_rollout_step_kernel<PendEnv>, dim3(blocks_for(n)), dim3(kBlock), 0, s, make_pend_const(*p), ra, ma, SsStepArgs{});
    } else {
        return set_error(SSC_EINVAL, "ssc_mpc_rollout_step: unknown env kind %d", p->kind);
    }
    return check_launch("ssc_mpc_rollout_step");
}

extern "C" int ssc_smartstart_rollout_step(const ssc_env_params *p, const ssc_mpc_problems *pr, const ssc_mpc_nav_state *nav,
                                           const ssc_smartstart_step *ss, const float *d_A, const int32_t *d_best_idx,
                                           float noise_amount, uint64_t noise_seed, uint64_t problem_id0,
                                           const ssc_rollout_state *state, const ssc_transition_log *log,
                                           const ssc_episode_ring *ring, double *d_stats, uint64_t env_seed, uint64_t env_id0,
                                           uint64_t *d_t, int32_t *d_k, int32_t *d_ticket, float *d_plan_state,
                                           ssc_stream_t stream) {
    RolloutArgs ra;
    MpcStepArgs ma;
    if (int rc = build_mpc_step_args("ssc_smartstart_rollout_step", p, pr, nav, d_A, d_best_idx, noise_amount, noise_seed, problem_id0,
                                     state, log, ring, d_stats, env_seed, env_id0, d_t, d_k, d_ticket, d_plan_state, ra, ma))
        return rc;
    SSC_REQUIRE(ss != nullptr, "ssc_smartstart_rollout_step: NULL step descriptor");
    SSC_REQUIRE(ss->mode && ss->plan_of && ss->d_actor_out && ss->d_eta && ss->d_ou_epsilon && ss->d_pool,
                "ssc_smartstart_rollout_step: NULL device pointer in the step descriptor");
    SSC_REQUIRE(pr->plan_of != nullptr && pr->plan_of == ss->plan_of && pr->wp_len != nullptr,
                "ssc_smartstart_rollout_step: the problem set must use the plan pool (plan_of == ss->plan_of, wp_len)");
    SSC_REQUIRE(state->ou_x != nullptr, "ssc_smartstart_rollout_step: the base agent's OU noise needs state->ou_x");
    SSC_REQUIRE(ss->act_low <= ss->act_high && ss->mode_log_stride >= 0, "ssc_smartstart_rollout_step: bad action bounds / stride");
    const int64_t n = ra.n;
    if (n == 0) return SSC_OK;
    SsStepArgs sa{};
    sa.mode = ss->mode; sa.plan_of = ss->plan_of; sa.actor_out = ss->d_actor_out; sa.d_eta = ss->d_eta; sa.d_eps = ss->d_ou_epsilon;
    sa.d_pool = ss->d_pool;
    sa.ou_mu = ss->ou.mu; sa.ou_sig_sqrt_dt = ss->ou.sigma * sqrtf(ss->ou.dt); sa.ou_theta_dt = ss->ou.theta * ss->ou.dt;
    sa.act_low = ss->act_low; sa.act_high = ss->act_high;
    sa.mode_log = ss->d_mode_log; sa.mode_log_stride = ss->mode_log_stride ? ss->mode_log_stride : n;
    sa.n_live = ss->d_n_live;
    SSC_REQUIRE(sa.mode_log_stride >= n, "ssc_smartstart_rollout_step: mode log stride < n");
    hipStream_t s = as_stream(stream);
    if (p->kind == SSC_ENV_MOUNTAINCAR) {
        if (int rc = validate_mc_params(p, "ssc_smartstart_rollout_step")) return rc;
        hipLaunchKernelGGL((mpc_rollout_step_kernel<McEnv, true>), dim3(blocks_for(n)), dim3(kBlock), 0, s, make_mc_const(*p), ra, ma, sa);
    } else if (p->kind == SSC_ENV_PENDULUM) {
        hipLaunchKernelGGL((mpc_rollout_step_kernel<PendEnv, true>), dim3(blocks_for(n)), dim3(kBlock), 0, s, make_pend_const(*p), ra, ma, sa);
    } else {
        return set_error(SSC_EINVAL, "ssc_smartstart_rollout_step: unknown env kind %d", p->kind);
    }
    return check_launch("ssc_smartstart_rollout_step");
}
