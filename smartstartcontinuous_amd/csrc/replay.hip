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

enum : uint32_t { TAG_REPLAY = 5 };

struct ReplayAppendArgs {
    ssc_replay_ring ring;
    ssc_transition_log log;
    int64_t n, row_stride, done_row_stride;
    int64_t first, count, start;  // records [first, first + count) of the chunk go to ring slots (start + j) % capacity
    float reward_scale;
};

__global__ __launch_bounds__(kBlock) void replay_append_kernel(ReplayAppendArgs g) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= g.count) return;
    const int64_t j = g.first + i;           // record number inside the chunk: step-major, then env
    const int64_t k = j / g.n, e = j - k * g.n;
    const int64_t pos = (g.start + j) % g.ring.capacity;
    const int64_t src = k * g.row_stride + e;
    const int od = g.ring.obs_dim;
#pragma unroll
    for (int c = 0; c < SSC_MAX_OBS; ++c)
        if (c < od) {
            g.ring.s[pos * od + c] = g.log.obs[c][src];
            g.ring.s2[pos * od + c] = g.log.obs2[c][src];
        }
    g.ring.a[pos] = g.log.act[src];
    g.ring.r[pos] = g.log.rew[src] * g.reward_scale;   // DDPG_Baselines_agent.observe scales the reward (:238-240)
    g.ring.t[pos] = g.log.done[k * g.done_row_stride + e];
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

}  // namespace ssc

using namespace ssc;

extern "C" {

int ssc_replay_append(const ssc_replay_ring *ring, const ssc_transition_log *log, int32_t K, int64_t n,
                      int64_t start, float reward_scale, ssc_stream_t stream) {
    SSC_REQUIRE(ring != nullptr && log != nullptr, "ssc_replay_append: NULL descriptor");
    SSC_REQUIRE(ring->capacity > 0 && ring->obs_dim >= 1 && ring->obs_dim <= SSC_MAX_OBS && ring->act_dim == 1,
                "ssc_replay_append: capacity %lld / obs_dim %d / act_dim %d not supported", (long long)ring->capacity,
                ring->obs_dim, ring->act_dim);
    SSC_REQUIRE(K >= 0 && n >= 0 && start >= 0, "ssc_replay_append: negative size");
    if ((int64_t)K * n == 0) return SSC_OK;
    SSC_REQUIRE(ring->s && ring->a && ring->r && ring->t && ring->s2, "ssc_replay_append: NULL ring array");
    SSC_REQUIRE(log->act && log->rew && log->done, "ssc_replay_append: NULL log column");
    for (int c = 0; c < ring->obs_dim; ++c)
        SSC_REQUIRE(log->obs[c] && log->obs2[c], "ssc_replay_append: NULL obs column %d", c);
    ReplayAppendArgs g;
    g.ring = *ring; g.log = *log; g.n = n;
    g.row_stride = log->row_stride ? log->row_stride : n;
    g.done_row_stride = log->done_row_stride ? log->done_row_stride : n;
    const int64_t total = (int64_t)K * n;
    g.count = total < ring->capacity ? total : ring->capacity;  // older records of the chunk would be overwritten anyway
    g.first = total - g.count;
    g.start = start; g.reward_scale = reward_scale;
    hipLaunchKernelGGL(replay_append_kernel, dim3(blocks_for(g.count)), dim3(kBlock), 0, as_stream(stream), g);
    return check_launch("ssc_replay_append");
}

int ssc_replay_sample(uint64_t seed, uint64_t counter0, int64_t size, int32_t n_batches, int32_t batch_size,
                      int32_t *d_idx, ssc_stream_t stream) {
    SSC_REQUIRE(n_batches >= 0 && batch_size >= 1 && batch_size <= 64, "ssc_replay_sample: batch_size %d not in 1..64",
                batch_size);
    SSC_REQUIRE(size >= batch_size && size <= 0x7fffffffLL,
                "ssc_replay_sample: %lld records cannot give %d distinct indices", (long long)size, batch_size);
    if (n_batches == 0) return SSC_OK;
    SSC_REQUIRE(d_idx != nullptr, "ssc_replay_sample: d_idx NULL");
    hipLaunchKernelGGL(replay_sample_kernel, dim3(n_batches), dim3(64), 0, as_stream(stream), seed, counter0, size,
                       batch_size, d_idx);
    return check_launch("ssc_replay_sample");
}

}  // extern "C"
