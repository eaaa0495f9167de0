// replay.hip -- device-resident replay ring: the storage behind
// ReplayBuffer.add / sample_batch (smartstart/RLAgents/replay_buffer.py:49-74, 79-91) for the vectorised
// loop, so that rollout chunk -> replay -> DDPG minibatches -> ssc_ddpg_train never leaves HBM.
//
//   ssc_replay_append   scatters the (s, a, r, t, s2) records of a rollout chunk (SoA columns [K][n], the
//                       layout ssc_rollout writes) into row-major ring arrays; FIFO like the reference's
//                       deque: record number j lives at j % capacity, the oldest records are overwritten.
//                       HBM-bound: 25 B read + 25 B written per record (MountainCar).
//   ssc_replay_sample   minibatch indices, uniform WITHOUT replacement inside a batch (random.sample,
//                       replay_buffer.py:79-83), one wave per batch, Philox-keyed (bit-exact with the oracle).
#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

enum : uint32_t { TAG_REPLAY = 5, TAG_SMART_START = 7 };

struct ReplayAppendArgs {
    ssc_replay_ring ring;
    ssc_transition_log log;
    int64_t n, row_stride, done_row_stride;
    int64_t first, count, start;  // records [first, first + count) of the chunk go to ring slots (start + j) % capacity
    int64_t n_total, env_off;     // ssc_replay_append_shard: the chunk is envs [env_off, env_off + n) of n_total per step
    float reward_scale;
};

// One record's reads, all requested before the first store of it (the ring and the log may alias as far as the compiler knows:
// written load -> store -> load -> store, every column was a round trip of its own).  OD = 0: any obs_dim, guarded.
template <int OD>
struct AppendRow {
    float s[OD > 0 ? OD : SSC_MAX_OBS], s2[OD > 0 ? OD : SSC_MAX_OBS], a, r;
    uint8_t t;
    __device__ __forceinline__ void load(const ReplayAppendArgs &g, int64_t src, int64_t done_src) {
        const int od = OD > 0 ? OD : g.ring.obs_dim;
#pragma unroll
        for (int c = 0; c < (OD > 0 ? OD : SSC_MAX_OBS); ++c) {
            const int cc = c < od ? c : od - 1;      // (clamped column: no guard around the load)
            s[c] = g.log.obs[cc][src];
            s2[c] = g.log.obs2[cc][src];
        }
        a = g.log.act[src];
        r = g.log.rew[src];
        t = g.log.done[done_src];
    }
    __device__ __forceinline__ void store(const ReplayAppendArgs &g, int64_t pos) const {
        const int od = OD > 0 ? OD : g.ring.obs_dim;
#pragma unroll
        for (int c = 0; c < (OD > 0 ? OD : SSC_MAX_OBS); ++c)
            if (c < od) {
                g.ring.s[pos * od + c] = s[c];
                g.ring.s2[pos * od + c] = s2[c];
            }
        g.ring.a[pos] = a;
        g.ring.r[pos] = r * g.reward_scale;   // DDPG_Baselines_agent.observe scales the reward (:238-240)
        g.ring.t[pos] = t;
    }
};

template <int OD>
__global__ __launch_bounds__(kBlock) void replay_append_kernel(ReplayAppendArgs g) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= g.count) return;
    const int64_t j = g.first + i;           // record number inside the chunk: step-major, then env
    const int64_t k = j / g.n, e = j - k * g.n;
    const int64_t pos = (g.start + k * g.n_total + g.env_off + e) % g.ring.capacity;
    AppendRow<OD> row;
    row.load(g, k * g.row_stride + e, k * g.done_row_stride + e);
    row.store(g, pos);
}

// The same append with the episode index (ssc_replay_ring::ep_steps / ep_run): one thread per env walks its column
// of the chunk in step order (coalesced across envs on both sides), counting the steps of its running episode -- four
// steps' reads in flight at a time.
template <int OD>
__global__ __launch_bounds__(kBlock) void replay_append_indexed_kernel(ReplayAppendArgs g, int32_t K) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= g.n) return;
    int32_t run = g.ring.ep_run[g.env_off + e];
    constexpr int kU = 4;
    for (int32_t k0 = 0; k0 < K; k0 += kU) {
        AppendRow<OD> rows[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int32_t k = min(k0 + u, K - 1);      // (a step past the chunk re-reads the last one and is dropped below)
            rows[u].load(g, (int64_t)k * g.row_stride + e, (int64_t)k * g.done_row_stride + e);
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int32_t k = k0 + u;
            if (k >= K) break;
            run += 1;
            const int64_t j = (int64_t)k * g.n + e;
            if (j >= g.first) {
                const int64_t pos = (g.start + (int64_t)k * g.n_total + g.env_off + e) % g.ring.capacity;
                rows[u].store(g, pos);
                g.ring.ep_steps[pos] = run;
            }
            if (rows[u].t) run = 0;  // rlTrain breaks on done and the next step opens a new episode (rlTrain.py:97, replay_buffer.py:109-115)
        }
    }
    g.ring.ep_run[g.env_off + e] = run;
}

// ---- smart-start index queries --------------------------------------------------------------------------------
struct SmartStartArgs {
    const int32_t *ep_steps;
    int64_t capacity, count, size, n;
    int32_t n_ss;
    uint64_t seed, counter;
    int32_t *idx, *n_out;
    int32_t tsize;     // hash-set cells: power of two >= 4 * n_ss
};

// buffer index i (0 = oldest) is a possible smart start iff the first recorded step of its episode is still in the ring
__device__ __forceinline__ bool smart_start_valid(const SmartStartArgs &a, int64_t i) {
    const int64_t rec = a.count - a.size + i;
    const int32_t L = a.ep_steps[rec % a.capacity];
    return L >= 1 && rec - (int64_t)(L - 1) * a.n >= a.count - a.size;
}

constexpr int kSmartBlock = 1024, kSmartMaxRounds = 64;

// ONE workgroup; slot s (< n_ss) is served by thread s % 1024.  An open-addressing hash set in the workspace:
// keys[h] = buffer index + 1 (0 = empty, claimed by atomicCAS), owners[h] = (round << 12 | slot) of the slot that
// holds the key, lowered with atomicMin -- so of all the slots that want a key in a round the LOWEST owns it whatever
// the thread order, and a key taken in an earlier round (smaller round field) can never be taken over.
__global__ __launch_bounds__(kSmartBlock) void smart_start_indices_kernel(SmartStartArgs a, uint32_t *keys, uint32_t *owners) {
    constexpr int SPT = 4;  // slots per thread: n_ss <= 4096
    __shared__ int n_unresolved, n_done;
    const uint32_t mask = (uint32_t)a.tsize - 1u;
    for (int i = threadIdx.x; i < a.tsize; i += kSmartBlock) {
        keys[i] = 0u;
        owners[i] = 0xFFFFFFFFu;
    }
    int32_t val[SPT];
    bool done[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        val[q] = -1;
        done[q] = (int)(threadIdx.x + q * kSmartBlock) >= a.n_ss;
    }
    __syncthreads();
    for (int round = 0; round < kSmartMaxRounds; ++round) {
        if (threadIdx.x == 0) n_unresolved = 0;
        __syncthreads();
        uint32_t cand[SPT];
        bool ok[SPT];
        // phase 1: draw, and bid for the candidate's key
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const uint32_t slot = threadIdx.x + q * kSmartBlock;
            ok[q] = false;
            cand[q] = 0;
            if (done[q]) continue;
            const u32x4 w = rng_words(a.seed, a.counter, ((uint64_t)round << 20) | slot, TAG_SMART_START);
            cand[q] = (uint32_t)(((uint64_t)w.x * (uint64_t)a.size) >> 32);
            if (!smart_start_valid(a, cand[q])) continue;
            ok[q] = true;
            uint32_t h = (cand[q] * 2654435761u) & mask;
            for (int probe = 0; probe < a.tsize; ++probe) {
                const uint32_t prev = atomicCAS(&keys[h], 0u, cand[q] + 1u);
                if (prev == 0u || prev == cand[q] + 1u) {
                    atomicMin(&owners[h], ((uint32_t)round << 12) | slot);
                    break;
                }
                h = (h + 1u) & mask;
            }
        }
        __syncthreads();
        // phase 2: whoever owns its key keeps it
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const uint32_t slot = threadIdx.x + q * kSmartBlock;
            if (done[q]) continue;
            if (ok[q]) {
                uint32_t h = (cand[q] * 2654435761u) & mask;
                for (int probe = 0; probe < a.tsize; ++probe) {
                    if (keys[h] == cand[q] + 1u) {
                        if (owners[h] == (((uint32_t)round << 12) | slot)) { val[q] = (int32_t)cand[q]; done[q] = true; }
                        break;
                    }
                    h = (h + 1u) & mask;
                }
            }
            if (!done[q]) atomicAdd(&n_unresolved, 1);
        }
        __syncthreads();
        if (n_unresolved == 0) break;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_done = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int slot = threadIdx.x + q * kSmartBlock;
        if (slot < a.n_ss) {
            a.idx[slot] = val[q];
            if (val[q] >= 0) atomicAdd(&n_done, 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) *a.n_out = n_done;
}

struct EpisodePathArgs {
    ssc_replay_ring ring;
    int64_t count, size, n;
    const int32_t *buffer_index;
    int32_t max_len;
    float *path;
    int32_t *len;
};

__global__ __launch_bounds__(kBlock) void episode_path_kernel(EpisodePathArgs a) {
    const int64_t i = *a.buffer_index;
    const int od = a.ring.obs_dim;
    int32_t rows = 0;
    int64_t rec = 0;
    int32_t L = 0;
    if (i >= 0 && i < a.size) {
        rec = a.count - a.size + i;
        L = a.ring.ep_steps[rec % a.ring.capacity];
        if (L >= 1 && rec - (int64_t)(L - 1) * a.n >= a.count - a.size) rows = (L < a.max_len ? L : a.max_len) + 1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.len = rows;
    if (rows == 0) return;
    const int32_t steps = rows - 1;   // the newest `steps` states of the prefix, oldest first, then s2 of the record
    for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < (int64_t)rows * od; e += (int64_t)gridDim.x * kBlock) {
        const int32_t m = (int32_t)(e / od), c = (int32_t)(e % od);
        if (m < steps) {
            const int64_t r = rec - (int64_t)(steps - 1 - m) * a.n;
            a.path[e] = a.ring.s[(r % a.ring.capacity) * od + c];
        } else {
            a.path[e] = a.ring.s2[(rec % a.ring.capacity) * od + c];
        }
    }
}

// One wave per batch.  Slot i draws candidate (word0(Philox(seed; batch id, attempt << 8 | i)) * size) >> 32 in
// rounds: in a round every unresolved slot draws its next attempt and accepts unless the candidate equals an
// already accepted index or the candidate of a lower-numbered unresolved slot of the same round
// (oracle: replay_sample_indices).
__global__ __launch_bounds__(64) void replay_sample_kernel(uint64_t seed, uint64_t counter0, int64_t size, int32_t batch,
                                                            int32_t *__restrict__ idx) {
    const int lane = threadIdx.x;
    const uint64_t bid = counter0 + blockIdx.x;
    const bool slot = lane < batch;
    bool done = !slot;
    uint32_t val = 0xFFFFFFFFu;
    for (uint32_t attempt = 0; __ballot(!done) != 0; ++attempt) {
        uint32_t cand = 0xFFFFFFFFu;
        if (!done) cand = (uint32_t)(((uint64_t)rng_words(seed, bid, ((uint64_t)attempt << 8) | (uint64_t)lane, TAG_REPLAY).x * (uint64_t)size) >> 32);
        bool clash = false;
        for (int j = 0; j < 64; ++j) {
            const uint32_t vj = __shfl(val, j), cj = __shfl(cand, j);
            const bool dj = __shfl((int)done, j) != 0;
            const bool is_slot = j < batch;
            clash |= is_slot && dj && vj == cand;              // taken in an earlier round
            clash |= is_slot && !dj && j < lane && cj == cand; // a lower slot wants it in this round
        }
        if (!done && !clash) { val = cand; done = true; }
    }
    if (slot) idx[(int64_t)blockIdx.x * batch + lane] = (int32_t)val;
}

// Batches of 65 .. 4096 indices (the multi-workgroup learner's batch sizes): ONE workgroup per batch, slot s served by
// thread s % 1024, the same accept / reject rule as above decided through an open-addressing hash set in LDS
// (keys[h] = index + 1 claimed by atomicCAS; owners[h] = (round << 12 | slot) lowered with atomicMin -- of all the slots
// that want an index in a round the LOWEST keeps it whatever the thread order, and an index accepted in an earlier
// round cannot be taken over).  The slot rides in 16 bits of the Philox counter here (attempt << 16 | slot): batches of
// at most 64 keep the 8-bit form of the one-wave kernel, so their streams are unchanged.  Oracle: replay_sample_indices.
constexpr int kSampleBlock = 1024;
__global__ __launch_bounds__(kSampleBlock) void replay_sample_big_kernel(uint64_t seed, uint64_t counter0, int64_t size, int32_t batch,
                                                                         int32_t tsize, int32_t *__restrict__ idx) {
    constexpr int SPT = 4;
    extern __shared__ uint32_t table[];   // keys [tsize] | owners [tsize]
    uint32_t *keys = table, *owners = table + tsize;
    __shared__ int n_unresolved;
    const uint32_t mask = (uint32_t)tsize - 1u;
    const uint64_t bid = counter0 + blockIdx.x;
    for (int i = threadIdx.x; i < tsize; i += kSampleBlock) {
        keys[i] = 0u;
        owners[i] = 0xFFFFFFFFu;
    }
    int32_t val[SPT];
    bool done[SPT];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        val[q] = -1;
        done[q] = (int)(threadIdx.x + q * kSampleBlock) >= batch;
    }
    __syncthreads();
    for (uint32_t round = 0;; ++round) {
        if (threadIdx.x == 0) n_unresolved = 0;
        __syncthreads();
        uint32_t cand[SPT];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const uint32_t slot = threadIdx.x + q * kSampleBlock;
            cand[q] = 0;
            if (done[q]) continue;
            cand[q] = (uint32_t)(((uint64_t)rng_words(seed, bid, ((uint64_t)round << 16) | slot, TAG_REPLAY).x * (uint64_t)size) >> 32);
            uint32_t h = (cand[q] * 2654435761u) & mask;
            for (int probe = 0; probe < tsize; ++probe) {
                const uint32_t prev = atomicCAS(&keys[h], 0u, cand[q] + 1u);
                if (prev == 0u || prev == cand[q] + 1u) {
                    atomicMin(&owners[h], (round << 12) | slot);
                    break;
                }
                h = (h + 1u) & mask;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const uint32_t slot = threadIdx.x + q * kSampleBlock;
            if (done[q]) continue;
            uint32_t h = (cand[q] * 2654435761u) & mask;
            for (int probe = 0; probe < tsize; ++probe) {
                if (keys[h] == cand[q] + 1u) {
                    if (owners[h] == ((round << 12) | slot)) { val[q] = (int32_t)cand[q]; done[q] = true; }
                    break;
                }
                h = (h + 1u) & mask;
            }
            if (!done[q]) atomicAdd(&n_unresolved, 1);
        }
        __syncthreads();
        const bool finished = n_unresolved == 0 || round >= (1u << 19);   // (size >= batch: every round resolves at least one slot)
        __syncthreads();
        if (finished) break;
    }
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const int slot = threadIdx.x + q * kSampleBlock;
        if (slot < batch) idx[(int64_t)blockIdx.x * batch + slot] = val[q];
    }
}

}  // namespace ssc

using namespace ssc;

extern "C" {

static int replay_append_impl(const char *fn, const ssc_replay_ring *ring, const ssc_transition_log *log, int32_t K, int64_t n,
                              int64_t start, int64_t n_total, int64_t env_off, float reward_scale, ssc_stream_t stream) {
    SSC_REQUIRE(ring != nullptr && log != nullptr, "%s: NULL descriptor", fn);
    SSC_REQUIRE(ring->capacity > 0 && ring->obs_dim >= 1 && ring->obs_dim <= SSC_MAX_OBS && ring->act_dim == 1,
                "%s: capacity %lld / obs_dim %d / act_dim %d not supported", fn, (long long)ring->capacity,
                ring->obs_dim, ring->act_dim);
    SSC_REQUIRE(K >= 0 && n >= 0 && start >= 0, "%s: negative size", fn);
    SSC_REQUIRE(env_off >= 0 && env_off + n <= n_total, "%s: shard [%lld, %lld) outside the %lld envs of a step", fn,
                (long long)env_off, (long long)(env_off + n), (long long)n_total);
    if ((int64_t)K * n == 0) return SSC_OK;
    SSC_REQUIRE(ring->s && ring->a && ring->r && ring->t && ring->s2, "%s: NULL ring array", fn);
    SSC_REQUIRE(log->act && log->rew && log->done, "%s: NULL log column", fn);
    for (int c = 0; c < ring->obs_dim; ++c)
        SSC_REQUIRE(log->obs[c] && log->obs2[c], "%s: NULL obs column %d", fn, c);
    ReplayAppendArgs g;
    g.ring = *ring; g.log = *log; g.n = n; g.n_total = n_total; g.env_off = env_off;
    g.row_stride = log->row_stride ? log->row_stride : n;
    g.done_row_stride = log->done_row_stride ? log->done_row_stride : n;
    const int64_t total = (int64_t)K * n;
    if (n_total == n) {
        g.count = total < ring->capacity ? total : ring->capacity;  // older records of the chunk would be overwritten anyway
    } else {
        SSC_REQUIRE((int64_t)K * n_total <= ring->capacity, "%s: %d steps of %lld envs do not fit a ring of %lld records", fn, K,
                    (long long)n_total, (long long)ring->capacity);
        g.count = total;
    }
    g.first = total - g.count;
    g.start = start; g.reward_scale = reward_scale;
    SSC_REQUIRE((ring->ep_steps == nullptr) == (ring->ep_run == nullptr), "%s: ep_steps and ep_run come together", fn);
    if (ring->ep_steps != nullptr) {
        SSC_REQUIRE(start % n_total == 0, "%s: an indexed ring takes whole steps of n = %lld envs (start = %lld)", fn,
                    (long long)n_total, (long long)start);
        if (ring->obs_dim == 2) hipLaunchKernelGGL(replay_append_indexed_kernel<2>, dim3(blocks_for(n)), dim3(kBlock), 0, as_stream(stream), g, K);
        else if (ring->obs_dim == 3) hipLaunchKernelGGL(replay_append_indexed_kernel<3>, dim3(blocks_for(n)), dim3(kBlock), 0, as_stream(stream), g, K);
        else hipLaunchKernelGGL(replay_append_indexed_kernel<0>, dim3(blocks_for(n)), dim3(kBlock), 0, as_stream(stream), g, K);
        return check_launch(fn);
    }
    if (ring->obs_dim == 2) hipLaunchKernelGGL(replay_append_kernel<2>, dim3(blocks_for(g.count)), dim3(kBlock), 0, as_stream(stream), g);
    else if (ring->obs_dim == 3) hipLaunchKernelGGL(replay_append_kernel<3>, dim3(blocks_for(g.count)), dim3(kBlock), 0, as_stream(stream), g);
    else hipLaunchKernelGGL(replay_append_kernel<0>, dim3(blocks_for(g.count)), dim3(kBlock), 0, as_stream(stream), g);
    return check_launch(fn);
}

int ssc_replay_append(const ssc_replay_ring *ring, const ssc_transition_log *log, int32_t K, int64_t n,
                      int64_t start, float reward_scale, ssc_stream_t stream) {
    return replay_append_impl("ssc_replay_append", ring, log, K, n, start, n, 0, reward_scale, stream);
}

int ssc_replay_append_shard(const ssc_replay_ring *ring, const ssc_transition_log *log, int32_t K, int64_t n,
                            int64_t start, int64_t n_total, int64_t env_off, float reward_scale, ssc_stream_t stream) {
    SSC_REQUIRE(n_total >= 1, "ssc_replay_append_shard: n_total = %lld", (long long)n_total);
    return replay_append_impl("ssc_replay_append_shard", ring, log, K, n, start, n_total, env_off, reward_scale, stream);
}

static int smart_table_size(int32_t n_ss) {
    int t = 1024;
    while (t < 4 * n_ss) t <<= 1;
    return t;
}

size_t ssc_replay_smart_start_workspace_bytes(int32_t n_ss) {
    return n_ss > 0 ? (size_t)smart_table_size(n_ss) * 8 : 256;
}

int ssc_replay_smart_start_indices(const ssc_replay_ring *ring, int64_t count, int64_t n, int32_t n_ss, uint64_t seed,
                                   uint64_t counter, int32_t *d_idx, int32_t *d_n, void *d_workspace,
                                   size_t workspace_bytes, ssc_stream_t stream) {
    SSC_REQUIRE(ring != nullptr, "ssc_replay_smart_start_indices: NULL ring");
    SSC_REQUIRE(ring->ep_steps != nullptr, "ssc_replay_smart_start_indices: the ring keeps no episode index (ep_steps NULL)");
    SSC_REQUIRE(ring->capacity > 0 && count >= 0 && n >= 1, "ssc_replay_smart_start_indices: bad sizes");
    SSC_REQUIRE(n_ss >= 0 && n_ss <= 4096, "ssc_replay_smart_start_indices: n_ss %d not in 0..4096", n_ss);
    SSC_REQUIRE(d_n != nullptr, "ssc_replay_smart_start_indices: d_n NULL");
    const int64_t size = count < ring->capacity ? count : ring->capacity;
    SSC_REQUIRE(size <= 0x7fffffffLL, "ssc_replay_smart_start_indices: ring too large for int32 buffer indices");
    if (n_ss == 0 || size == 0) {
        return check_hip(hipMemsetAsync(d_n, 0, sizeof(int32_t), as_stream(stream)), "hipMemsetAsync(d_n)");
    }
    SSC_REQUIRE(d_idx != nullptr, "ssc_replay_smart_start_indices: d_idx NULL");
    SSC_REQUIRE(d_workspace && workspace_bytes >= ssc_replay_smart_start_workspace_bytes(n_ss),
                "ssc_replay_smart_start_indices: workspace too small");
    SmartStartArgs a;
    a.ep_steps = ring->ep_steps; a.capacity = ring->capacity; a.count = count; a.size = size; a.n = n;
    a.n_ss = n_ss; a.seed = seed; a.counter = counter; a.idx = d_idx; a.n_out = d_n;
    a.tsize = smart_table_size(n_ss);
    uint32_t *keys = static_cast<uint32_t *>(d_workspace);
    hipLaunchKernelGGL(smart_start_indices_kernel, dim3(1), dim3(kSmartBlock), 0, as_stream(stream), a, keys, keys + a.tsize);
    return check_launch("ssc_replay_smart_start_indices");
}

int ssc_replay_episode_path(const ssc_replay_ring *ring, int64_t count, int64_t n, const int32_t *d_buffer_index,
                            int32_t max_len, float *d_path, int32_t *d_len, ssc_stream_t stream) {
    SSC_REQUIRE(ring != nullptr, "ssc_replay_episode_path: NULL ring");
    SSC_REQUIRE(ring->ep_steps != nullptr, "ssc_replay_episode_path: the ring keeps no episode index (ep_steps NULL)");
    SSC_REQUIRE(ring->capacity > 0 && count >= 0 && n >= 1 && max_len >= 1, "ssc_replay_episode_path: bad sizes");
    SSC_REQUIRE(ring->obs_dim >= 1 && ring->obs_dim <= SSC_MAX_OBS, "ssc_replay_episode_path: obs_dim out of range");
    SSC_REQUIRE(ring->s && ring->s2 && d_buffer_index && d_path && d_len, "ssc_replay_episode_path: NULL device pointer");
    EpisodePathArgs a;
    a.ring = *ring; a.count = count; a.size = count < ring->capacity ? count : ring->capacity; a.n = n;
    a.buffer_index = d_buffer_index; a.max_len = max_len; a.path = d_path; a.len = d_len;
    const int64_t elems = ((int64_t)max_len + 1) * ring->obs_dim;
    int64_t blocks = blocks_for(elems);
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(episode_path_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), a);
    return check_launch("ssc_replay_episode_path");
}

int ssc_replay_sample(uint64_t seed, uint64_t counter0, int64_t size, int32_t n_batches, int32_t batch_size,
                      int32_t *d_idx, ssc_stream_t stream) {
    SSC_REQUIRE(n_batches >= 0 && batch_size >= 1 && batch_size <= 4096, "ssc_replay_sample: batch_size %d not in 1..4096",
                batch_size);
    SSC_REQUIRE(size >= batch_size && size <= 0x7fffffffLL,
                "ssc_replay_sample: %lld records cannot give %d distinct indices", (long long)size, batch_size);
    if (n_batches == 0) return SSC_OK;
    SSC_REQUIRE(d_idx != nullptr, "ssc_replay_sample: d_idx NULL");
    if (batch_size > 64) {
        int tsize = 1024;
        while (tsize < 2 * batch_size) tsize <<= 1;
        if ((size_t)tsize * 8 + 256 > 64 * 1024) {   // dynamic table + the kernel's static LDS pass the default 64 KB limit (batch > 2048)
            if (int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(replay_sample_big_kernel),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, tsize * 8),
                                   "hipFuncSetAttribute(replay_sample_big_kernel)"))
                return rc;
        }
        hipLaunchKernelGGL(replay_sample_big_kernel, dim3(n_batches), dim3(kSampleBlock), (size_t)tsize * 8, as_stream(stream), seed,
                           counter0, size, batch_size, tsize, d_idx);
        return check_launch("ssc_replay_sample");
    }
    hipLaunchKernelGGL(replay_sample_kernel, dim3(n_batches), dim3(64), 0, as_stream(stream), seed, counter0, size,
                       batch_size, d_idx);
    return check_launch("ssc_replay_sample");
}

}  // extern "C"
