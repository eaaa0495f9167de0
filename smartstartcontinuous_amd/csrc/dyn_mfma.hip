// dyn_mfma.hip -- bf16-MFMA path of the NND_MB dynamics model: the H-step forward simulation
// Dyn_Model.do_forward_sim (NN_Dynamics_Model/dynamics_model.py:204-240) over
// feedforward_network (feedforward_network.py:3-23) as ONE persistent kernel.
//
// Shape of the work (BASELINE config 4: in 4, 2x500 hidden, out 3): 507 kflop per row-step, of
// which 500 k are the [rows,500]x[500,500] hidden contraction -> MFMA-bound (bf16 dense peak
// ~2.5 PFLOP/s).  The state feeds back every step, so the H loop lives inside the kernel and
// NOTHING but the [H+1][m][d] trajectory (the result) and the per-step actions touch HBM.
//
// Mapping to CDNA4:
//   * a block = 512 threads = 8 waves x 32 rows; waves w and w+4 share a SIMD (2 waves per SIMD, 256
//     VGPRs each).  A wave holds the first hidden layer of its 32 rows as bf16 MFMA fragments in
//     registers (128 VGPRs at depth 512); the waves of a block share the weight fragments through LDS.
//   * every layer is computed TRANSPOSED: D[unit][row] = W^T[unit][k] * H[k][row], i.e. the weights
//     are the MFMA A operand and the activations the B operand.  All contractions run on
//     v_mfma_f32_16x16x32_bf16 (on this chip the 16x16x32 shape holds a ~6.5 % higher clock than
//     32x32x16 at equal cycles: tools/exp_dyn_clock.py).  Two 16x16 accumulator tiles of a layer (units
//     32p..32p+15 and 32p+16..32p+31 of 16 rows) are -- after ReLU + v_cvt_pk_bf16_f32 in registers --
//     exactly ONE B fragment (32 k x 16 rows) of the next layer under a fixed permutation of k that is
//     folded into the packed weights: activations never touch LDS and no lane moves
//     (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand").
//   * layer 1 (K = state+action <= 12) also runs on the bf16 MFMA at fp32-level accuracy by splitting
//     x and W1 into bf16 head + bf16 residual (3 products per input, bias in two more k slots).
//   * layer-2 output tiles (32 units) are consumed immediately by the output layer (one more MFMA per
//     16 rows), so the second hidden activation is never materialised.
//   * W2 is packed once per call into fragment order (bf16) and streamed L2 -> LDS by LDS-DMA one
//     32-unit output tile (32 KiB at depth 512) at a time through a ring of 3 slots, one barrier per
//     tile, the two waves of a SIMD half a tile out of phase (details at tile_body below).
#include <float.h>
#include <type_traits>

#include "ssc_device.h"
#include "ssc_host.h"

// SSC_DYN_STAMPS (diagnostic builds only: tools/variants/dyn_mfma_clk.hip -> tools/_build/libssc_clk.so, read by
// tools/exp_dyn_clock.py; never defined in libssc.so): every block overwrites S[48 * block ..] with {s_memtime, s_memrealtime}
// deltas of its step loop and the cycles of its phases, for one wave of each group -- the RESULTS ARE WRONG in such a build.
#ifndef SSC_DYN_STAMPS
#define SSC_DYN_STAMPS 0
#endif

namespace ssc {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t vu32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kDynRows = 256;     // rows per block: 8 waves x two 16-row MFMA column tiles
constexpr int kDynThreads = 512;
constexpr int kNW = kDynThreads / 64;
constexpr int kMaxIn = 12;        // network inputs (state + action)
constexpr int kMaxKS1 = 2;        // layer-1 k-steps of 32 slots: 2 bias slots + 3 per input
// LAG kernel tuning, each the winner of a same-box interleaved A/B (NOTEBOOK sections 9.1, 9.7; the variants are in git history):
constexpr int kLagBarrierPair = 1;   // layer-1 unit pair behind which group 0 takes its phase barrier
constexpr int kLagX1 = 11;           // fragment at which group 1 takes its tile barrier
constexpr int kLagPrio = 3;          // issue priority of a wave inside its per-step phase

// LDS-DMA: one wave-instruction copies 64 x 16 B = 1 KiB global -> LDS with no VGPR staging
// (buffer_load_dwordx4 ... lds).  LDS destination = wave-uniform base + lane*16; source = buffer base +
// wave-uniform soffset + lane*16.  Completion is tracked by vmcnt.  The MUBUF form on purpose: hipcc models
// the FLAT-encoded global_load_lds as "may touch LDS and memory" and from then on turns every counted
// s_waitcnt lgkmcnt(N) into lgkmcnt(0), which serialises the fragment ring behind the LDS latency.
__device__ __forceinline__ void lds_dma_1k(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int soffset, unsigned char *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(
        rsrc,
        reinterpret_cast<__attribute__((address_space(3))) void *>(
            static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_wave_base))),
        16, lane_off, soffset, 0, 0);
}

// v_mfma_f32_16x16x32_bf16: lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15], j < 8, and
// D[row 4(l>>4)+r][col l&15], r < 4.  Two D tiles (units 32p + 4g + r and 32p + 16 + 4g + r, g = l>>4) packed
// to bf16 are the 8 k values of lane group g in k-step p of the next layer: k slot (8g + j) carries unit
__host__ __device__ __forceinline__ int frag_unit(int p, int g, int j) { return 32 * p + 16 * (j >> 2) + 4 * g + (j & 3); }

// Layer 1 on the bf16 MFMA at fp32-level accuracy: x = xh + xl, w = wh + wl (bf16 head + bf16 residual),
// x*w ~= xh*wh + xl*wh + xh*wl (dropped xl*wl <= 2^-16 |x w|; products of bf16 pairs are exact in the
// fp32 accumulator).  K slots of the layer-1 contraction (32 per k-step):
//   slot 0: 1 * bf16(b1)   slot 1: 1 * residual(b1)   slot 2+3i+{0,1,2}: {xh_i*wh_i, xl_i*wh_i, xh_i*wl_i}
__host__ __device__ __forceinline__ int l1_ksteps(int in) { return (2 + 3 * in + 31) / 32; }

// Packed weight image in the workspace (all offsets in bytes, 256-aligned); UT = 32-unit tiles of a hidden layer
struct DynPack {
    size_t a2;   // bf16 [UT jt][2*UT f = 2p+mh][64 lane][8]  W2^T fragments: units 32jt+16mh+(lane&15), k-step p (NFC == 2)
    size_t a3;   // bf16 [UT p][4 g][8 o][8]                  Wout^T fragments: MFMA rows m and m+8 share entry o = m & 7
    size_t a1;   // bf16 [KS1][2*UT mt][64 lane][8]           layer-1 fragments (bias + split W1, see above)
    size_t b2;   // f32  [UT jt][2 mh][4 g][4 r]              b2 in accumulator layout (when not in the k slots)
    size_t b3;   // f32  [4 g][4 r]
    size_t nm;   // f32  [6][8]  mean_x 1/std_x mean_y 1/std_y mean_z std_z
    size_t ctr;  // i32  [2]     row-tile counter of walking blocks (DynSimArgs::tile_ctr), zero at rest
    size_t total;
};

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static DynPack make_pack(int UT, int nfc) {
    DynPack p;
    size_t o = 0;
    p.a2 = o; o += al256(nfc == 2 ? (size_t)UT * UT * 2 * 64 * 8 * 2 : 0);
    p.a3 = o; o += al256((size_t)UT * 4 * 8 * 8 * 2);
    p.a1 = o; o += al256((size_t)kMaxKS1 * 2 * UT * 64 * 8 * 2);
    p.b2 = o; o += al256((size_t)UT * 32 * 4);
    p.b3 = o; o += al256(16 * 4);
    p.nm = o; o += al256(48 * 4);
    p.ctr = o; o += 256;
    p.total = o;
    return p;
}

struct DynNet {
    const float *W1, *b1, *W2, *b2, *W3, *b3;  // W3/b3 = output layer; W2/b2 unused when nfc == 1
    int in, depth, out, nfc;
    bool biask;  // b2 in the spare k slots depth, depth+1 of the hidden contraction (needs depth + 2 <= 32*UT)
    bool compact1;  // layer-1 fragments in the compact lane-group layout of the KIN = 4 kernels (see l1_compact below)
};

// Compact layer-1 layout (networks with <= 4 inputs and outputs, i.e. every kernel compiled with KIN = 4): k group g = lane >> 4 of the MFMA carries
// INPUT g alone -- slots 8g + {0, 1, 2} = {xh * wh, xl * wh, xh * wl}, slot 8g + 3 = 1 * (bf16 head of b1 in group 0, its
// residual in group 1), slots 8g + 4..7 unused.  A lane then splits ONE input per step instead of selecting its eight
// slots out of all of them (the per-step input code drops from ~120 to ~16 VALU ops per 16 rows), and a weight fragment
// is 8 bytes per lane instead of 16: the layer-1 image is 16 KB instead of 32 KB, which is what buys the W2 ring its
// fourth 32 KB slot.
__host__ __device__ __forceinline__ bool l1_compact(int in, int out) { return in <= 4 && out <= 4; }   // == the KIN = 4 kernels

__device__ __forceinline__ __bf16 bf16_head(float v) { return (__bf16)v; }
__device__ __forceinline__ __bf16 bf16_resid(float v) { return (__bf16)(v - (float)(__bf16)v); }

__global__ __launch_bounds__(256) void dyn_pack_kernel(DynNet n, int UT, DynPack pk, ssc_norm nm,
                                                       unsigned char *__restrict__ ws) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid < 48) {
        const int q = (int)gid >> 3, k = (int)gid & 7;
        float v = 0.0f;
        // 1/std (std == 0 -> inf) so that the z-score is one multiply in the kernel (zscore())
        if (q == 0) v = nm.mean_x[k]; else if (q == 1) v = 1.0f / nm.std_x[k];
        else if (q == 2) v = nm.mean_y[k & 3]; else if (q == 3) v = 1.0f / nm.std_y[k & 3];
        else if (q == 4) v = nm.mean_z[k]; else v = nm.std_z[k];
        reinterpret_cast<float *>(ws + pk.nm)[gid] = v;
        if (gid < 2) reinterpret_cast<int32_t *>(ws + pk.ctr)[gid] = 0;
    }
    const int64_t n_a2 = (n.nfc == 2) ? (int64_t)UT * UT * 2 * 64 * 8 : 0;
    const int64_t n_a3 = (int64_t)UT * 4 * 8 * 8;
    const int64_t n_a1 = (int64_t)kMaxKS1 * 2 * UT * 64 * 8;
    const int64_t n_b = (int64_t)UT * 32;
    int64_t e = gid;
    if (e < n_a2) {
        const int j = e & 7, lane = (e >> 3) & 63;
        const int f = (int)((e >> 9) % (2 * UT)), jt = (int)((e >> 9) / (2 * UT));
        const int u = frag_unit(f >> 1, lane >> 4, j), col = jt * 32 + 16 * (f & 1) + (lane & 15);
        __bf16 v = (__bf16)0.0f;
        if (col < n.depth) {
            if (u < n.depth) v = (__bf16)n.W2[(int64_t)u * n.depth + col];
            else if (n.biask && u == n.depth) v = bf16_head(n.b2[col]);
            else if (n.biask && u == n.depth + 1) v = bf16_resid(n.b2[col]);
        }
        reinterpret_cast<__bf16 *>(ws + pk.a2)[e] = v;
        return;
    }
    e -= n_a2;
    if (e < n_a3) {
        // MFMA row m = 0..15 of the output layer computes output (m & omask): with <= 4 outputs every k-group lane
        // of a batch row then ends up holding ALL outputs in its 4 accumulator registers (no lane exchange)
        const int j = e & 7, o8 = (e >> 3) & 7, g = (e >> 6) & 3, p = (int)(e >> 8);
        const int u = frag_unit(p, g, j), o = o8 & (n.out <= 4 ? 3 : 7);
        const float v = (u < n.depth && o < n.out) ? n.W3[(int64_t)u * n.out + o] : 0.0f;
        reinterpret_cast<__bf16 *>(ws + pk.a3)[e] = (__bf16)v;
        return;
    }
    e -= n_a3;
    if (e < n_a1 && n.compact1) {   // [mt][64 lane][4]: slots 0..3 of the lane's k group (l1_compact)
        if (e >= (int64_t)2 * UT * 64 * 4) return;
        const int j = e & 3, lane = (e >> 2) & 63, mt = (int)(e >> 8);
        const int i = lane >> 4, unit = mt * 16 + (lane & 15);
        __bf16 v = (__bf16)0.0f;
        if (unit < n.depth) {
            if (j == 3) v = (i == 0) ? bf16_head(n.b1[unit]) : (i == 1 ? bf16_resid(n.b1[unit]) : (__bf16)0.0f);
            else if (i < n.in) {
                const float w = n.W1[(int64_t)i * n.depth + unit];
                v = (j == 2) ? bf16_resid(w) : bf16_head(w);
            }
        } else if (n.biask && unit < n.depth + 2 && i == 0 && j == 3) {
            v = (__bf16)1.0f;  // the carriers of b2 (see below)
        }
        reinterpret_cast<__bf16 *>(ws + pk.a1)[e] = v;
        return;
    }
    if (e < n_a1) {
        const int j = e & 7, lane = (e >> 3) & 63, mt = (int)((e >> 9) % (2 * UT)), ks = (int)((e >> 9) / (2 * UT));
        const int q = 32 * ks + 8 * (lane >> 4) + j, unit = mt * 16 + (lane & 15);
        __bf16 v = (__bf16)0.0f;
        if (unit < n.depth) {
            if (q == 0) v = bf16_head(n.b1[unit]);
            else if (q == 1) v = bf16_resid(n.b1[unit]);
            else {
                const int i = (q - 2) / 3, c = (q - 2) % 3;
                if (i < n.in) {
                    const float w = n.W1[(int64_t)i * n.depth + unit];
                    v = (c == 2) ? bf16_resid(w) : bf16_head(w);
                }
            }
        } else if (n.biask && unit < n.depth + 2 && q == 0) {
            v = (__bf16)1.0f;  // hidden units depth, depth+1 == ReLU(1 * 1) == 1: the carriers of b2
        }
        reinterpret_cast<__bf16 *>(ws + pk.a1)[e] = v;
        return;
    }
    e -= n_a1;
    if (e < n_b) {  // b2: [jt][mh][g][r]
        const int r = e & 3, g = (e >> 2) & 3, mh = (e >> 4) & 1, jt = (int)(e >> 5);
        const int u = jt * 32 + 16 * mh + 4 * g + r;
        reinterpret_cast<float *>(ws + pk.b2)[e] = (n.nfc == 2 && !n.biask && u < n.depth) ? n.b2[u] : 0.0f;
        return;
    }
    e -= n_b;
    if (e < 16) {  // b3: [g][r], MFMA row 4g + r computes output (4g + r) & omask
        const int o = (int)e & (n.out <= 4 ? 3 : 7);
        reinterpret_cast<float *>(ws + pk.b3)[e] = (o < n.out) ? n.b3[o] : 0.0f;
    }
}

struct DynSimArgs {
    int64_t m;
    int32_t H, d, a;        // horizon, state dim, action dim (forward mode: H = 1)
    int32_t in, out;        // network input / output width
    int32_t fwd_mode;       // 1: plain y = net(x) (ssc_mlp_forward); 0: forward simulation
    const float *s0;
    int64_t s0_rows;        // consecutive rows sharing one start state (m / number of start states)
    int32_t walk;           // streamed W2: a block walks over row tiles -- blockIdx.x, blockIdx.x + gridDim.x, then whatever the
                            // shared counter hands out -- and fetches the next tile's start states by LDS-DMA under the last
                            // step of the current one (launch_sim decides)
    int32_t *tile_ctr;      // walk: device int32[2], zero between launches: {row tiles handed out beyond the first two rounds,
                            // blocks that have finished}; the last block to finish zeroes both again
    const float *A;         // sim: [m][H][a]; fwd: x [m][in]
    float *S;               // sim: [H+1][m][d]; fwd: y [m][out]
    const unsigned char *a1, *a2, *a3;  // packed weight image (workspace)
    const float *b2, *b3, *nm;
    // candidate action sequences drawn INSIDE the kernel (ssc_mpc_forward_sim): A == nullptr, row r is sample r % N of
    // problem r / N, Philox stream of ssc_mpc_sample_actions (oracle: mpc_action_samples); A_out (may be null) receives them
    int32_t sample, N;
    uint64_t seed, pid0, t;
    const uint64_t *t_base;
    float low[SSC_MAX_ACT], span[SSC_MAX_ACT];
    float *A_out;
    const uint8_t *active;   // sampling mode: problems (row / N) whose byte is 0 need not be simulated; may be NULL
};

// np.nan_to_num((x - mean) / std) (dynamics_model.py:228-229) with inv = 1/std: 0/0 = 0 * inf = NaN -> 0;
// x/0 = +-inf is clamped to the largest bf16 by split_bf16 (nan_to_num's +-max, as far as bf16 reaches).
__device__ __forceinline__ float zscore(float x, float mean, float inv) {
    const float v = (x - mean) * inv;
    return (v != v) ? 0.0f : v;
}

// ReLU (feedforward_network.py:19) fused with the bf16 convert: two accumulator tiles -> one B fragment
__device__ __forceinline__ bf16x8 relu_to_frag(const f32x4 &lo, const f32x4 &hi) {
    i32x4 p;
    p[0] = relu_pack_bf16(lo[0], lo[1]);
    p[1] = relu_pack_bf16(lo[2], lo[3]);
    p[2] = relu_pack_bf16(hi[0], hi[1]);
    p[3] = relu_pack_bf16(hi[2], hi[3]);
    return __builtin_bit_cast(bf16x8, p);
}

// bf16 head / residual of x as 16-bit patterns (round to nearest even, like the weights)
__device__ __forceinline__ void split_bf16(float x, uint32_t &hi, uint32_t &lo) {
    x = fminf(fmaxf(x, -3.3895314e38f), 3.3895314e38f);  // +-FLT_MAX of nan_to_num would round to bf16 inf
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}

// a block-uniform value, moved to an SGPR
__device__ __forceinline__ float uniform_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ f32x4 mfma16(const bf16x8 &a, const bf16x8 &b, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// LDS carve (bytes): [a2: NBUF x UT*2048] [a1: KS1*UT*2048] [a3: UT*512] [b2 UT*128] [b3 64]
// W2 (NFC == 2): resident when it fits (UT <= 4: all UT tiles), otherwise streamed through a ring of 3.
template <int UT, int NFC>
__host__ __device__ constexpr int dyn_a2_bufs() { return NFC == 2 ? (UT <= 4 ? UT : 3) : 0; }

// BIASK: b2 rides in two spare k slots of the hidden contraction (hidden units depth, depth+1 are the constant
// 1; their W2^T rows hold bf16 head and residual of b2) instead of being the accumulator's C-in.
// KIN: compile-time bound on the network inputs AND outputs (4 covers MountainCar 3->2 and Pendulum 4->3; 10 is
// what one layer-1 k-step holds): the per-step input/normalisation/state code is unrolled KIN times with
// block-uniform guards, and layer 1 runs KS1(KIN) k-steps (unused slots carry zero weights).
//
// LAG (streamed W2, <= 4 inputs: the BASELINE shape): the two wave groups of a SIMD run 1.5 tiles apart instead of half
// a tile, so that one group's per-step phase (state update, input code, layer 1: VALU work, ~2 tiles long) runs while
// the OTHER group still has W2 tiles to multiply, instead of both leaving the matrix pipe idle together:
//   * the W2 ring has FOUR 32 KB slots (slot = tile & 3); the room comes from the compact layer-1 image (l1_compact);
//   * barrier #k (k counts tiles over all steps) is taken by group 0 inside tile k at fragment X0 and by group 1 inside
//     tile k - 1 at fragment X1; passing it retires every read of tile k - 2 and publishes tile k + 1, and after it a
//     wave issues its pieces of tile k + 2 (group 0: at the start of tile k + 1; group 1: right behind the barrier);
//   * group 0 takes the barrier of a step's tile 0 in the MIDDLE OF ITS PHASE instead (where group 1, inside tile 15 of
//     the step before, arrives at about the same time) and runs tile 0 without one: its phase never makes group 1 wait
//     for a barrier, and group 1's phase falls into group 0's tiles 0 and 1, which need none from it until X0 of tile 1;
//   * group 0 alone feeds the ring (it is the group that WAITS at the tile barriers -- the oldest wave wins the issue
//     arbitration and runs ahead -- so the LDS-DMA issue cost comes out of its slack instead of the critical group's time);
//   * every wave takes 16 H + 1 barriers: group 0 one per phase + tiles 1..15 + one behind its last tile, group 1 one in
//     its first phase + one in every tile.
// MODE (compile-time when >= 0): 0 simulation with the actions read from memory, 1 simulation drawing its candidate
// actions itself, 2 plain forward; -1: decided by the run-time flags.  With the mode known the per-step input code holds
// no load (mode 1), so no s_waitcnt vmcnt(0) in front of it has to sit out the LDS-DMA pieces still in flight.
// WALK (streamed W2 only): the block walks over row tiles (next_tile below); compiled as its own instantiation so that the launch
// with one block per row tile -- the 65 536-row BASELINE shape -- carries none of the boundary code or its registers.
template <int UT, int NFC, bool BIASK, int KIN, bool LAG = false, int MODE = -1, bool WALK = false>
__global__ __launch_bounds__(kDynThreads, 2) void dyn_mfma_sim_kernel(DynSimArgs g) {
    static_assert(!LAG || (UT == 16 && NFC == 2 && KIN == 4), "LAG: streamed W2 with the compact layer-1 layout");
    static_assert(!WALK || ((NFC == 2) && (UT > 4) && MODE != 2), "WALK: streamed W2, simulation modes");
    constexpr int DS = KIN < SSC_MAX_STATE ? KIN : SSC_MAX_STATE;  // state / output registers per row
    constexpr int KS1 = (2 + 3 * KIN + 31) / 32;                   // layer-1 k-steps compiled in
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const bool fwd_mode = MODE < 0 ? g.fwd_mode != 0 : MODE == 2;
    const bool sample = MODE < 0 ? g.sample != 0 : MODE == 1;
    const uint64_t stamp_entry = SSC_DYN_STAMPS ? __builtin_amdgcn_s_memrealtime() : 0;
    constexpr int MT = 2 * UT;          // 16-unit tiles of a hidden layer
    constexpr int A2_TILE = UT * 2048;  // W2^T fragments of one 32-unit output tile: NF fragments of 1 KiB
    constexpr int NBUF = LAG ? 4 : dyn_a2_bufs<UT, NFC>();
    constexpr bool STREAM = (NFC == 2) && (UT > 4);
    constexpr bool CMP = (KIN == 4);                            // compact lane-group layer 1 (l1_compact)
    constexpr bool G0DMA = LAG;                                  // LAG: wave group 0 issues every LDS-DMA piece
    constexpr int A1_BYTES = CMP ? MT * 512 : KS1 * MT * 1024;   // layer-1 fragments: 8 B per lane when compact
    constexpr int A2_CHUNKS = A2_TILE / 1024;           // LDS-DMA pieces per tile
    constexpr int PPW = (A2_CHUNKS + kNW - 1) / kNW;    // pieces per wave per tile
    unsigned char *l_a2 = lds;
    unsigned char *l_a1 = l_a2 + NBUF * A2_TILE;
    unsigned char *l_a3 = l_a1 + A1_BYTES;
    float *l_b2 = reinterpret_cast<float *>(l_a3 + UT * 512);
    float *l_b3 = l_b2 + UT * 32;
    // STREAM: start states of the NEXT row tile of a walking block, [wave][column tile][k group = state component][16 rows]
    float *l_s0 = l_b3 + 16;
    int32_t *l_nn = reinterpret_cast<int32_t *>(l_s0 + kNW * 2 * 64);   // ... and the row tile after that one (see next_tile below)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kg = lane >> 4;  // MFMA column (row of the batch tile) and k group of this lane
    // Waves 0-3 (group 0) and 4-7 (group 1) pair up on the SIMDs; the groups take the per-tile barrier half a
    // tile apart (tile_body), so the two waves of a SIMD stay out of phase and the tile-boundary work of one
    // (ReLU, output MFMAs) runs under the other's MFMA stream instead of both idling the matrix pipe together.
    const int group = wave >> 2;

    // ---- stage the resident weights ------------------------------------------------------------
    // Everything goes out before anything is waited for: the layer-1 and output-layer fragments and the first W2^T
    // tiles by LDS-DMA (1 KiB pieces dealt round-robin to the 8 waves, no VGPR staging), the small bias arrays through
    // registers, then -- below -- this lane's start state, first actions and the normalisation constants; ONE wait +
    // barrier in front of layer 1 covers all of it.
    const __amdgpu_buffer_rsrc_t a2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(g.a2), 0, NFC == 2 ? UT * A2_TILE : 0, 0x00020000);   // reads past the end return 0
    {
        constexpr int A1_PIECES = A1_BYTES / 1024;       // 1 KiB each
        constexpr int A3_BYTES = UT * 512;
        const __amdgpu_buffer_rsrc_t a1_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned char *>(g.a1), 0, A1_PIECES * 1024, 0x00020000);
#pragma unroll
        for (int ch = wave; ch < A1_PIECES; ch += kNW) lds_dma_1k(a1_rsrc, lane * 16, ch * 1024, l_a1 + ch * 1024);
        if constexpr (A3_BYTES % 1024 == 0) {
            const __amdgpu_buffer_rsrc_t a3_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned char *>(g.a3), 0, A3_BYTES, 0x00020000);
#pragma unroll
            for (int ch = wave; ch < A3_BYTES / 1024; ch += kNW) lds_dma_1k(a3_rsrc, lane * 16, ch * 1024, l_a3 + ch * 1024);
        } else {   // a piece is 1 KiB: a smaller image goes through registers
            const f32x4 *s3 = reinterpret_cast<const f32x4 *>(g.a3);
            f32x4 *d3 = reinterpret_cast<f32x4 *>(l_a3);
            for (int e = tid; e < A3_BYTES / 16; e += kDynThreads) d3[e] = s3[e];
        }
        if (NFC == 2) {  // first W2^T tiles: everything when resident, tiles 0 and 1 of the ring otherwise
            constexpr int PRE = STREAM ? 2 : NBUF;
#pragma unroll
            for (int ch = wave; ch < PRE * A2_CHUNKS; ch += kNW) lds_dma_1k(a2_rsrc, lane * 16, ch * 1024, l_a2 + ch * 1024);
        }
        for (int e = tid; e < UT * 32; e += kDynThreads) l_b2[e] = g.b2[e];
        if (tid < 16) l_b3[tid] = g.b3[tid];
    }

    // output-layer A fragments: MFMA row (lane & 15) reads entry (lane & 7) of its k group
    const unsigned char *a3_lane = l_a3 + (kg * 8 + (c & 7)) * 16;
    constexpr int a3_step = 512;

    // ---- this lane's rows: one per 16-row column tile; the 4 k-group lanes of a row carry it redundantly ----
    // A block of a kernel whose weights are RESIDENT in LDS (no W2 stream) can walk over row tiles tile, tile + gridDim.x, ...
    // (setup_rows below); the launcher gives every tile its own block, see launch_sim.
    int32_t rowc[2];   // this lane's row of each column tile, clamped to m - 1 (run_mfma: m < 2^31); == the row when valid[]
    bool valid[2];
    float st[2][DS];

    // the actions of step t+1 are fetched during step t (a global load costs ~1-2 k cycles even from L2) -- or, in
    // sampling mode, drawn from Philox here: candidate sequence of sample n of problem p = words of
    // Philox(seed; (problem_id0 + p) << 32 | n, t * ceil(H a / 4) + c, TAG_MPC), flat index h * a + ai = 4 c + word
    // (NND_MB_agent.py:500-501; bit-exact with ssc_mpc_sample_actions / oracle mpc_action_samples), so the [m][H][a]
    // matrix never has to exist in memory
    constexpr int AMAX = KIN < 4 ? KIN : 4;
    float act[2][AMAX];
    u32x4 wcache[2] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    int wc = -1;  // Philox call the cached words belong to (block-uniform)
    uint64_t tt = 0;
    if (sample) tt = (g.t + (g.t_base != nullptr ? *g.t_base : 0)) * (uint64_t)((g.H * g.a + 3) / 4);
    // (ar / av: the rows the actions belong to -- this tile's, or, from the last step of a walking block's tile, the next tile's)
    auto fetch_actions = [&](int ts, int32_t ar0, int32_t ar1, bool av0, bool av1) __attribute__((always_inline)) {
        const int32_t ar[2] = {ar0, ar1};
        const bool av[2] = {av0, av1};
        if (sample) {
#pragma unroll
            for (int ai = 0; ai < AMAX; ++ai)
                if (ai < g.a) {
                    const int f = ts * g.a + ai, c4 = f >> 2;
                    if (c4 != wc) {  // block-uniform
                        wc = c4;
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            // sample id of the row: (problem, sample within the problem) -- derived here, once per Philox call,
                            // rather than kept in four registers across the step loop
                            const uint32_t r32 = (uint32_t)ar[nt], q = r32 / (uint32_t)g.N;
                            const uint64_t sidv = ((g.pid0 + (uint64_t)q) << 32) + (uint64_t)(r32 - q * (uint32_t)g.N);
                            wcache[nt] = rng_words(g.seed, sidv, tt + (uint64_t)c4, TAG_MPC);
                        }
                    }
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        act[nt][ai] = uniform_f32(pick(wcache[nt], (uint32_t)(f & 3)), g.low[ai], g.span[ai]);
                        if (g.A_out != nullptr && av[nt] && kg == 0) g.A_out[((int64_t)ar[nt] * g.H + ts) * g.a + ai] = act[nt][ai];
                    }
                }
        } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const float *ap = g.A + ((int64_t)ar[nt] * g.H + ts) * g.a;
#pragma unroll
                for (int ai = 0; ai < AMAX; ++ai)
                    if (ai < g.a) act[nt][ai] = ap[ai];
            }
        }
    };
    bool skip_tile = false;
    // (rows fit 32 bits, run_mfma: the per-tile index arithmetic of a walking block runs inside the phase the other wave
    // group has to bridge, and a 64-bit division is ~10 x a 32-bit one)
    // `ga`: the kernel arguments the row set-up reads.  At the row-tile boundary of a walking block they are RE-READ from the
    // kernarg segment (scalar loads, once per row tile) instead of being kept in ~16 SGPRs across the step loop, where
    // the allocator has none to spare (SGPR spills cost VGPR lanes, and the VGPR file is exactly full)
    auto setup_rows = [&](const DynSimArgs &ga, int64_t tile, auto staged_tag) __attribute__((always_inline)) {   // (a call would put the row state in scratch memory)
        constexpr bool staged = decltype(staged_tag)::value;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int64_t r64 = tile * kDynRows + wave * 32 + nt * 16 + c;
            valid[nt] = r64 < ga.m;
            rowc[nt] = (int32_t)(valid[nt] ? r64 : ga.m - 1);
            // staged: a walking block at a row-tile boundary -- the start states were fetched by stage_next_tile under the
            // previous tile's last step, where the sample ids were switched and the first actions drawn as well (step loop).
            // (The state is assigned ONCE, behind the branch: with a store to st[] in one arm and one to act[] in the other
            // the optimiser merges the two into a store through a pointer phi, and both arrays end up in scratch memory.)
            float s_new[DS];
            if (STREAM && staged) {
                int cc = c;
                asm volatile("" : "+v"(cc));   // (keeps the lane's LDS address from being hoisted out of the step loop -- and spilled)
#pragma unroll
                for (int k = 0; k < DS; ++k) s_new[k] = l_s0[((wave * 2 + nt) * 4 + (k & 3)) * 16 + cc];
            } else {
#pragma unroll
                for (int k = 0; k < DS; ++k)
                    s_new[k] = (!fwd_mode && k < ga.d) ? ga.s0[((int64_t)rowc[nt] / ga.s0_rows) * ga.d + k] : 0.0f;   // s0_rows: rows per start state
            }
#pragma unroll
            for (int k = 0; k < DS; ++k) st[nt][k] = (k < ga.d && (k < 4 || !(STREAM && staged))) ? s_new[k] : 0.0f;
            if (!(STREAM && staged)) {
#pragma unroll
                for (int ai = 0; ai < AMAX; ++ai) act[nt][ai] = 0.0f;
            }
        }
        if (STREAM && staged) return;
        wc = -1;
        // a wave none of whose rows belongs to a live problem (ssc_mpc_sampling.d_problem_active: e.g. the envs of the
        // vectorised SmartStart loop that are not navigating) skips the tile -- only where no ring barrier needs it
        skip_tile = false;
        if (!STREAM && sample && ga.active != nullptr) {
            const bool mine = (valid[0] && ga.active[rowc[0] / ga.N] != 0) || (valid[1] && ga.active[rowc[1] / ga.N] != 0);
            skip_tile = __builtin_amdgcn_ballot_w64(mine) == 0;
        }
        if (!fwd_mode && g.H > 0 && !skip_tile) fetch_actions(0, rowc[0], rowc[1], valid[0], valid[1]);
    };
    int64_t tile = blockIdx.x;
    setup_rows(g, tile, std::false_type{});
    // STREAM, walking block: lane (c, kg) fetches component kg of the start state of row c of each of its two column tiles
    // of row tile `nxt` -- global -> LDS with no register in between (the register file is full while the hidden tiles
    // run); the pieces are tracked by vmcnt like the W2 pieces and every tile barrier waits for vmcnt(0)
    // (kernel arguments read by the boundary code only are RE-READ from the kernarg segment there -- scalar loads, once per row
    // tile -- instead of living in SGPRs across the step loop, where the allocator has none to spare: SGPR spills cost VGPR
    // lanes, and the VGPR file is exactly full)
    typedef const __attribute__((address_space(4))) DynSimArgs *KernArgs;
    auto reread_args = [&]() __attribute__((always_inline)) -> KernArgs {
        KernArgs gp = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();  // `g` is the only argument
        asm volatile("" : "+s"(gp));   // opaque: the loads below are not merged with the kernel's own argument loads
        return gp;
    };
    auto stage_next_tile = [&](int64_t nxt, int32_t &nr0, int32_t &nr1, bool &nv0, bool &nv1) __attribute__((always_inline)) {
        KernArgs gp = reread_args();
        struct { const float *s0; int64_t s0_rows, m; int32_t d; } ga = {gp->s0, gp->s0_rows, g.m, g.d};
        // (the descriptor is built here, once per row tile, instead of living in four SGPRs across the step loop)
        const __amdgpu_buffer_rsrc_t s0_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(ga.s0), 0, (int)(((ga.m + ga.s0_rows - 1) / ga.s0_rows) * ga.d * 4), 0x00020000);
        auto one = [&](auto nt_tag, int32_t &nr, bool &nv) __attribute__((always_inline)) {
            constexpr int nt = decltype(nt_tag)::value;
            const int64_t r = nxt * kDynRows + wave * 32 + nt * 16 + c;
            nv = r < ga.m;
            const uint32_t rc = (uint32_t)(nv ? r : ga.m - 1);
            nr = (int32_t)rc;
            const uint32_t voff = ((rc / (uint32_t)ga.s0_rows) * (uint32_t)ga.d + (uint32_t)(kg < ga.d ? kg : 0)) * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                s0_rsrc,
                reinterpret_cast<__attribute__((address_space(3))) void *>(
                    static_cast<uint32_t>(reinterpret_cast<uintptr_t>(l_s0 + (wave * 2 + nt) * 64))),
                4, (int)voff, 0, 0, 0);
        };
        one(std::integral_constant<int, 0>{}, nr0, nv0);
        one(std::integral_constant<int, 1>{}, nr1, nv1);
        wc = -1;
    };

    // Normalisation constants per network input (input k = state k for k < d, else action k-d) and per state
    // delta, read once through scalar loads: block-uniform, so they live in SGPRs and the per-step input code
    // below is straight-line VALU code with no branch and no LDS access.  Inputs >= in get inv = 0 -> x = 0.
    float n_mean[KIN], n_inv[KIN], z_mean[DS], z_std[DS];
#pragma unroll
    for (int k = 0; k < KIN; ++k) {
        const bool is_state = k < g.d;
        const int ai = (k - g.d) & 3;
        n_mean[k] = uniform_f32(g.nm[is_state ? 0 * 8 + (k & 7) : 2 * 8 + ai]);
        n_inv[k] = (k < g.in) ? uniform_f32(g.nm[is_state ? 1 * 8 + (k & 7) : 3 * 8 + ai]) : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < DS; ++k) {
        z_mean[k] = (k < g.d) ? uniform_f32(g.nm[4 * 8 + k]) : 0.0f;
        z_std[k] = (k < g.d) ? uniform_f32(g.nm[5 * 8 + k]) : 0.0f;
    }
    // CMP: this lane's k group carries network input kg alone (l1_compact): its mean / 1/std, and the constant-1 slot
    // (bias carrier) in k groups 0 and 1
    float my_mean = 0.0f, my_inv = 0.0f;
    uint32_t my_one = 0;
    if (CMP) {
#pragma unroll
        for (int k = 0; k < KIN; ++k) {
            my_mean = (kg == k) ? n_mean[k] : my_mean;
            my_inv = (kg == k) ? n_inv[k] : my_inv;
        }
        my_one = (kg < 2) ? 0x3F800000u : 0u;
    }

    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the LDS-DMA pieces (and every load above)
    __syncthreads();

    int bsel = 0;  // LDS slot of the current W2 tile (STREAM)
    constexpr int NF = 2 * UT;                // W2^T fragments (ring entries) per hidden tile: k-step p = f>>1, half f&1
    constexpr int RING = (NF >= 4) ? 4 : 2;   // fragment reads in flight per wave
    bf16x8 ring[RING];
    if (NFC == 2) {
#pragma unroll
        for (int q = 0; q < RING; ++q) ring[q] = *reinterpret_cast<const bf16x8 *>(l_a2 + (q * 64 + lane) * 16);
    }
    const uint64_t stamp_c0 = SSC_DYN_STAMPS ? __builtin_amdgcn_s_memtime() : 0;
    const uint64_t stamp_r0 = SSC_DYN_STAMPS ? __builtin_amdgcn_s_memrealtime() : 0;
    uint64_t ph[6] = {0, 0, 0, 0, 0, 0};  // diagnostic: cycles in input code / layer 1 / hidden tiles / step tail / phase barrier / tile barriers
#define SSC_STAMP(var) uint64_t var = 0; if (SSC_DYN_STAMPS) { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    int vt = 0;  // steps this block has run over all its row tiles: the W2 ring and its barriers go on across row tiles
    // Row tiles of a walking block: its first two are blockIdx.x and blockIdx.x + gridDim.x, every later one comes from a
    // counter all blocks share (the CUs do not run at one clock -- XCDs differ by several per cent under this load -- so
    // a fixed stride would make every launch wait for the slowest CU).  The block always knows its NEXT tile: during the
    // last step of tile i wave 0 draws the index of tile i + 2 (the atomic is issued at the top of the step and its
    // result parked in LDS behind layer 1), and everybody picks it up at the boundary into tile i + 1, many barriers later.
    const int64_t n_tiles = (g.m + kDynRows - 1) / kDynRows;
    int64_t next_tile = WALK ? tile + gridDim.x : n_tiles;
    int round = 0;   // row tiles this block has finished; the LDS slot alternates (H = 1: wave 0 parks the next draw while the
                     // slowest wave -- 1.5 hidden tiles behind -- may not have read the previous one yet)
    for (;;) {   // row tiles of this block
    if (!skip_tile) {
    for (int t = 0; t < g.H; ++t, ++vt) {
        SSC_STAMP(stamp_a)
        const bool draw = WALK && t + 1 == g.H && next_tile < n_tiles;   // block-uniform
        int32_t drawn = 0;
        if (draw && wave == 0 && lane == 0)
            drawn = __hip_atomic_fetch_add(g.tile_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- inputs: record S[t]; layer-1 B fragments of x = normalised (state, action) ---------------
        bf16x8 xf[2][KS1];
        s16x4 xc[2];   // CMP: the compact layer-1 B fragments (four k slots per lane)
        if constexpr (CMP) {
            // The phase is what the other wave of this SIMD has to bridge with W2 tiles: it runs at raised issue priority
            // (set behind the last hidden tile, dropped again in front of the first one) ...
            if (LAG && vt == 0) __builtin_amdgcn_s_setprio(kLagPrio);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float xv;
                if (fwd_mode) {
                    xv = (kg < g.in) ? g.A[(int64_t)rowc[nt] * g.in + kg] : 0.0f;
                } else {
                    xv = st[nt][0];
#pragma unroll
                    for (int k = 1; k < DS; ++k) xv = (kg == k) ? st[nt][k] : xv;       // (k >= d: overwritten below or unused)
#pragma unroll
                    for (int ai = 0; ai < AMAX; ++ai) xv = (kg - g.d == ai) ? act[nt][ai] : xv;
                }
                uint32_t xh, xl;
                split_bf16(fwd_mode ? xv : zscore(xv, my_mean, my_inv), xh, xl);   // plain forward: x is fed as it is
                typedef uint32_t vu32x2 __attribute__((ext_vector_type(2)));
                // slots 0, 1: xh * wh, xl * wh; slots 2, 3: xh * wl, 1 * bias part
                xc[nt] = __builtin_bit_cast(s16x4, vu32x2{xh | (xl << 16), xh | my_one});
            }
            // ... and everything that ENTERS the memory queue comes behind the input code: the actions above may come
            // from global loads, so the compiler waits for vmcnt(0) in front of their first use -- which would also wait
            // for stores and LDS-DMA issued a few instructions earlier (a full L2 round trip at the head of the phase)
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (!fwd_mode) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    if (valid[nt] && kg == 0) {
                        float *sp = g.S + ((int64_t)t * g.m + rowc[nt]) * g.d;
#pragma unroll
                        for (int k = 0; k < DS; ++k)
                            if (k < g.d) sp[k] = st[nt][k];  // dynamics_model.py:225
                    }
            }
        } else {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float xs[KIN];
            if (fwd_mode) {
#pragma unroll
                for (int k = 0; k < KIN; ++k) xs[k] = (k < g.in) ? g.A[(int64_t)rowc[nt] * g.in + k] : 0.0f;
            } else {
                if (valid[nt] && kg == 0) {
                    float *sp = g.S + ((int64_t)t * g.m + rowc[nt]) * g.d;
#pragma unroll
                    for (int k = 0; k < DS; ++k)
                        if (k < g.d) sp[k] = st[nt][k];  // dynamics_model.py:225
                }
#pragma unroll
                for (int k = 0; k < KIN; ++k) {
                    float av = act[nt][0];  // action component k - d (block-uniform index)
#pragma unroll
                    for (int ai = 1; ai < AMAX; ++ai) av = (k - g.d == ai) ? act[nt][ai] : av;
                    xs[k] = zscore((k < g.d) ? st[nt][k < DS ? k : 0] : av, n_mean[k], n_inv[k]);
                }
            }
            uint32_t xh[KIN], xl[KIN];
#pragma unroll
            for (int k = 0; k < KIN; ++k) split_bf16(xs[k], xh[k], xl[k]);
            // this lane's 8 k slots of k-step ks are slots 32 ks + 8 kg + j (slot layout at l1_ksteps)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                vu32x4 w;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    uint32_t dw[4];
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        uint32_t v16[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int q = 32 * ks + 8 * gq + 2 * jj + e;
                            const int i = (q - 2) / 3, c3 = (q - 2) % 3;
                            v16[e] = (q < 2) ? 0x3F80u : (i < KIN ? (c3 == 1 ? xl[i] : xh[i]) : 0u);
                        }
                        dw[gq] = v16[0] | (v16[1] << 16);
                    }
                    w[jj] = (kg & 1) ? ((kg & 2) ? dw[3] : dw[1]) : ((kg & 2) ? dw[2] : dw[0]);
                }
                xf[nt][ks] = __builtin_bit_cast(bf16x8, w);
            }
        }
        }
        SSC_STAMP(stamp_l)
        // ---- layer 1: D[unit][row] = W1^T x + b1 -> ReLU -> bf16 B fragments of the next layer ------
        bf16x8 h1f[2][UT];
        struct L1 { f32x4 d[2][2]; };  // [16-unit half][column tile]
        // LAG: the weight fragments of a unit pair are requested kL1Ahead pairs before their MFMAs (the registers of the
        // previous step's hidden fragments are free here): under the other wave's W2 stream an LDS read takes longer
        // than the pair in front of it
        constexpr int kL1Ahead = 4;
        s16x4 a1q[CMP ? UT : 1][2];
        auto l1_request = [&](int p) {
            if constexpr (CMP) {
                a1q[p][0] = *reinterpret_cast<const s16x4 *>(l_a1 + ((2 * p + 0) * 64 + lane) * 8);
                a1q[p][1] = *reinterpret_cast<const s16x4 *>(l_a1 + ((2 * p + 1) * 64 + lane) * 8);
            }
        };
        if constexpr (CMP) {
#pragma unroll
            for (int q = 0; q < (kL1Ahead < UT ? kL1Ahead : UT); ++q) l1_request(q);
        }
        auto layer1_pair = [&](int p) {
            L1 o;
#pragma unroll
            for (int mh = 0; mh < 2; ++mh) {
                o.d[mh][0] = f32x4{0, 0, 0, 0};
                o.d[mh][1] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {  // no runtime bound: a branch here would serialise every LDS read
                    if constexpr (CMP) {
                        // compact image: 8 B per lane = the four k slots of this lane's k group, which is exactly the
                        // operand of v_mfma_f32_16x16x16_bf16 (same 16 cycles as the K = 32 shape, same D layout, but
                        // two-register operands: no zero padding to build)
                        const s16x4 a = a1q[p][mh];
                        o.d[mh][0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, xc[0], o.d[mh][0], 0, 0, 0);
                        o.d[mh][1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, xc[1], o.d[mh][1], 0, 0, 0);
                    } else {
                        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(l_a1 + ((ks * MT + 2 * p + mh) * 64 + lane) * 16);
                        o.d[mh][0] = mfma16(a, xf[0][ks], o.d[mh][0]);
                        o.d[mh][1] = mfma16(a, xf[1][ks], o.d[mh][1]);
                    }
                }
            }
            return o;
        };
        {
            L1 pp[2];  // ping-pong: pair p+1's MFMAs run under pair p's ReLU/convert
            pp[0] = layer1_pair(0);
#pragma unroll
            for (int p = 0; p < UT; ++p) {
                if (CMP && p + kL1Ahead < UT) l1_request(p + kL1Ahead);
                if (p + 1 < UT) pp[(p + 1) & 1] = layer1_pair(p + 1);
                h1f[0][p] = relu_to_frag(pp[p & 1].d[0][0], pp[p & 1].d[1][0]);
                h1f[1][p] = relu_to_frag(pp[p & 1].d[0][1], pp[p & 1].d[1][1]);
                // Pin the conversion HERE.  The fragments are first used inside the jt loop, so the
                // optimiser otherwise sinks ReLU+convert down to that loop's preheader and keeps all
                // fp32 accumulator tiles alive until then -> hundreds of spills.
                asm volatile("" : "+v"(h1f[0][p]), "+v"(h1f[1][p]));
                __builtin_amdgcn_sched_barrier(0);  // and one unit pair at a time in the machine scheduler
                if constexpr (LAG) {
                    // barrier #16 t: group 0 takes the barrier of this step's tile 0 HERE, about where group 1 (inside
                    // tile 15 of the step before) reaches it; in step 0 group 1 takes its first one here as well and
                    // then issues its pieces of tile 2 (later steps: in tile 15 of the step before)
                    if (p == kLagBarrierPair && (group == 0 || vt == 0)) {
                        {
                            SSC_STAMP(stamp_p0)
                            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of tile 1 landed
                            asm volatile("" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                            SSC_STAMP(stamp_p1)
                            ph[4] += stamp_p1 - stamp_p0;
                        }
                        if (group == 1 && !G0DMA) {
#pragma unroll
                            for (int q = 0; q < PPW; ++q)
                                lds_dma_1k(a2_rsrc, lane * 16, 2 * A2_TILE + (wave + q * kNW) * 1024, l_a2 + 2 * A2_TILE + (wave + q * kNW) * 1024);
                        }
                    }
                }
            }
        }
        if constexpr (LAG) {   // tiles: group 0 at priority 0; group 1 (the one the barriers wait for) may be given more
            __builtin_amdgcn_s_setprio(0);
        }
        // the actions of the next step: issued here, behind layer 1's scheduling fences, so that the loads fly
        // under the hidden tiles (hoisted to the top of the step they were waited for at once)
        if (!fwd_mode) {
            int ts = t + 1;
            int32_t ar0 = rowc[0], ar1 = rowc[1];
            bool av0 = valid[0], av1 = valid[1];
            bool more = ts < g.H;
            if (draw) {   // the last step of a row tile that has a successor
                if (wave == 0 && lane == 0) {
                    l_nn[round & 1] = 2 * (int32_t)gridDim.x + drawn;
                    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): in LDS before this wave's next barrier (the ring barriers are bare s_barriers)
                }
                stage_next_tile(next_tile, ar0, ar1, av0, av1);
                ts = 0;
                more = true;
            }
            if (more) fetch_actions(ts, ar0, ar1, av0, av1);
        }
        // ---- hidden layer 2 (+ output layer fused per tile) -------------------------------------
        f32x4 acc3[2];
        acc3[0] = acc3[1] = *reinterpret_cast<const f32x4 *>(l_b3 + kg * 4);
        if (NFC == 2) {
            // One 32-unit output tile jt of hidden layer 2: NF fragments (k-step p = f>>1, 16-unit half f&1)
            // from the current LDS slot, two MFMAs each (the wave's two 16-row column tiles), then ReLU and
            // the output-layer MFMAs.
            //  * The body is ONE basic block (no branch between the k-steps), so hipcc's s_waitcnt insertion
            //    keeps counted lgkmcnt waits and the fragment ring really stays RING deep.
            //  * The fragment ring runs on ACROSS tiles: the last RING reads of a tile already fetch the
            //    first fragments of the next one (and of the next step's first tile), so a tile starts with
            //    its operands in registers; with BIASK the accumulators start from the inline constant 0.
            //  * STREAM: 3 LDS slots hold tiles tl (being read), tl+1 (landed) and the one in flight.
            //    Barrier number tl -- taken in tile tl after fragment X0 by group 0 and X1 = X0 - NF/2 by
            //    group 1, which keeps the two waves of a SIMD half a tile out of phase -- retires every read
            //    of tile tl-1 and publishes tile tl+1; after it a wave issues its PPW pieces of tile tl+2 into
            //    the slot of tile tl-1, a few fragments apart so that their issue cost hides under MFMAs.
            //    The issue is unconditional: past the last tile it re-loads a tile nobody reads, and group 0
            //    in tile 0 re-loads the (identical) bytes of tile 1.
            auto tile_body = [&](auto group_tag, auto first_tag, int jt) {
                constexpr int GROUP = decltype(group_tag)::value;
                constexpr bool FIRST = decltype(first_tag)::value == 1;   // LAG, group 0, tile 0: no barrier, nothing to issue
                // LAG, group 0, tile 15: behind its barrier (#15) the pieces of the NEXT step's tile 1 go out as well --
                // the phase that follows takes barrier #16, which wants them landed (step 0: loaded by the prologue)
                constexpr bool LAST = decltype(first_tag)::value == 2;
                // barrier fragment of group 0 / group 1 (LAG: group 1's may sit anywhere behind the end of the tile before)
                constexpr int X0 = NF - RING - 1, X1 = LAG ? kLagX1 : X0 - NF / 2;
                static_assert(!STREAM || (UT == 16 && X1 + 2 + 4 * (PPW - 1) < NF), "LDS-DMA issue slots");
                const int nsel = LAG ? ((jt + 1) & 3) : STREAM ? (bsel == 2 ? 0 : bsel + 1) : ((jt + 1) & (UT - 1));
                const unsigned char *buf = l_a2 + (LAG ? (jt & 3) : STREAM ? bsel : jt) * A2_TILE + lane * 16;
                const unsigned char *nbuf = l_a2 + nsel * A2_TILE + lane * 16;
                // group 0 is past barrier tl-1 -> its pieces of tile tl+1; group 1 passes barrier tl at X1 ->
                // pieces of tile tl+2.  LAG: group 1 passes barrier jt+1 inside tile jt -> pieces of tile jt+3
                const int dma_jn = (jt + 1 + (LAG ? 2 : 1) * GROUP) & (UT - 1);
                const int dma_slot = LAG ? (dma_jn & 3) : (bsel + 1 + GROUP) % 3;
                const int dma_src = dma_jn * A2_TILE + wave * 1024;
                unsigned char *dma_dst = l_a2 + dma_slot * A2_TILE + wave * 1024;
                f32x4 acc[2][2];  // [16-unit half][column tile]
#pragma unroll
                for (int mh = 0; mh < 2; ++mh) {
                    acc[mh][0] = f32x4{0, 0, 0, 0};
                    if (!BIASK) acc[mh][0] = *reinterpret_cast<const f32x4 *>(l_b2 + ((jt * 2 + mh) * 4 + kg) * 4);
                    acc[mh][1] = acc[mh][0];
                }
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const bf16x8 a = ring[f % RING];
                    ring[f % RING] = (f + RING < NF) ? *reinterpret_cast<const bf16x8 *>(buf + (f + RING) * 1024)
                                                     : *reinterpret_cast<const bf16x8 *>(nbuf + (f + RING - NF) * 1024);
                    acc[f & 1][0] = mfma16(a, h1f[0][f >> 1], acc[f & 1][0]);
                    acc[f & 1][1] = mfma16(a, h1f[1][f >> 1], acc[f & 1][1]);
                    if (STREAM) {
                        const int f0 = GROUP ? X1 + 2 : 1;  // first issue slot after this group's barrier
                        if constexpr (G0DMA) {
                            // group 0 feeds the ring alone (it is the group that waits at the barriers: the issue cost comes out of
                            // its slack): all 32 pieces of tile jt + 1 in tile jt, eight per wave at fragments 1, 3, .. 15
                            if (GROUP == 0 && !FIRST && f >= 1 && f <= 15 && (f & 1)) {
                                const int piece = (wave + 4 * (f >> 1)) * 1024;
                                lds_dma_1k(a2_rsrc, lane * 16, ((jt + 1) & (UT - 1)) * A2_TILE + piece, l_a2 + ((jt + 1) & 3) * A2_TILE + piece);
                            }
                            if (LAST && f > X0) {   // the next step's tile 1: two pieces per fragment
#pragma unroll
                                for (int h = 0; h < 2; ++h) {
                                    const int piece = (wave + 4 * (2 * (f - X0 - 1) + h)) * 1024;
                                    lds_dma_1k(a2_rsrc, lane * 16, 1 * A2_TILE + piece, l_a2 + 1 * A2_TILE + piece);
                                }
                            }
                        } else {
                        if (!FIRST && f >= f0 && (f - f0) % 4 == 0 && (f - f0) / 4 < PPW) {
                            const int p = (f - f0) / 4;
                            lds_dma_1k(a2_rsrc, lane * 16, dma_src + p * kNW * 1024, dma_dst + p * kNW * 1024);
                        }
                        if (LAST && f > X0 && f - X0 - 1 < PPW) {
                            const int p = f - X0 - 1;
                            lds_dma_1k(a2_rsrc, lane * 16, 1 * A2_TILE + (wave + p * kNW) * 1024, l_a2 + 1 * A2_TILE + (wave + p * kNW) * 1024);
                        }
                        }
                        if (!FIRST && f == (GROUP ? X1 : X0)) {  // barrier tl
                            SSC_STAMP(stamp_t0)
                            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of tile tl+1 landed
                            // A bare s_barrier: __syncthreads() also waits for lgkmcnt(0), i.e. drains the fragment ring
                            // (the read issued one instruction ago included) in every tile.  Nothing here needs that: the
                            // reads of tile tl-1 have all been consumed by MFMAs (LDS reads return in order), the reads in
                            // flight belong to tile tl, and tile tl+1 is only read after the barrier.  The empty asms keep
                            // the compiler from moving LDS accesses across it.
                            asm volatile("" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                            SSC_STAMP(stamp_t1)
                            ph[5] += stamp_t1 - stamp_t0;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const bf16x8 a3f = *reinterpret_cast<const bf16x8 *>(a3_lane + jt * a3_step);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc3[nt] = mfma16(a3f, relu_to_frag(acc[0][nt], acc[1][nt]), acc3[nt]);
                if (STREAM) bsel = nsel;
            };
            SSC_STAMP(stamp_b)
            if (!STREAM || group == 0) {
                if constexpr (LAG) tile_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, 0);
#pragma unroll 1
                for (int jt = LAG ? 1 : 0; jt < (LAG ? UT - 1 : UT); ++jt) tile_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, jt);
                if constexpr (LAG) tile_body(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, UT - 1);
            } else {
#pragma unroll 1
                for (int jt = 0; jt < UT; ++jt) tile_body(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, jt);
            }
            // the phase ahead (state update, input code, layer 1) is what the other wave of this SIMD has to bridge
            if constexpr (LAG) __builtin_amdgcn_s_setprio(kLagPrio);
            SSC_STAMP(stamp_d)
            if (SSC_DYN_STAMPS) { ph[0] += stamp_l - stamp_a; ph[1] += stamp_b - stamp_l; ph[2] += stamp_d - stamp_b; ph[3] -= stamp_d; }
        } else {
#pragma unroll
            for (int p = 0; p < UT; ++p) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(a3_lane + p * a3_step);
                acc3[0] = mfma16(a, h1f[0][p], acc3[0]);
                acc3[1] = mfma16(a, h1f[1][p], acc3[1]);
                if ((p & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- z[o]: register r of every lane holds output r (<= 4 outputs) or output (4 kg + r) & 7 -----------
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float z[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                z[r] = acc3[nt][r];
                z[4 + r] = 0.0f;
                if (DS > 4 && g.out > 4) {  // block-uniform; k groups 0/2 hold outputs 0..3, 1/3 hold 4..7
                    const float other = __shfl_xor(acc3[nt][r], 16);
                    z[r] = (kg & 1) ? other : acc3[nt][r];
                    z[4 + r] = (kg & 1) ? acc3[nt][r] : other;
                }
            }
            if (fwd_mode) {
                if (valid[nt] && kg == 0) {
#pragma unroll
                    for (int o = 0; o < DS; ++o)
                        if (o < g.out) g.S[(int64_t)rowc[nt] * g.out + o] = z[o];
                }
            } else {
#pragma unroll
                for (int k = 0; k < DS; ++k) st[nt][k] = st[nt][k] + (z[k] * z_std[k] + z_mean[k]);  // :234-237 (0 for k >= d)
            }
        }
        if (SSC_DYN_STAMPS && NFC == 2) { SSC_STAMP(stamp_e) ph[3] += stamp_e; }
    }
    if (!fwd_mode) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            if (valid[nt] && kg == 0) {
#pragma unroll
                for (int k = 0; k < DS; ++k)
                    if (k < g.d) g.S[((int64_t)g.H * g.m + rowc[nt]) * g.d + k] = st[nt][k];  // :240
            }
    }
    }
    if constexpr (STREAM && !WALK) break;
    if constexpr (STREAM) {
        tile = next_tile;
        if (tile >= n_tiles) break;
        next_tile = __builtin_amdgcn_readfirstlane(l_nn[round & 1]);
        ++round;
    } else {
        tile += gridDim.x;
        if (tile >= n_tiles) break;
    }
    // (a streamed-W2 block gets here only when it walks -- launch_sim: simulation modes, H > 0 -- with the next tile staged)
    if constexpr (STREAM) setup_rows(g, tile, std::true_type{});
    else setup_rows(g, tile, std::false_type{});
    }
    // LAG: group 1's last tile holds barrier #16 H; group 0 meets it here (no phase follows its last tile)
    if (LAG && group == 0 && g.H > 0) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();
    }
    // group 1 issued LDS-DMA after its last barrier: it must land before this workgroup's LDS is released
    if (STREAM) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    if (WALK && tid == 0) {   // every draw of this block is behind it: the last block out leaves the counters at zero
        if (__hip_atomic_fetch_add(g.tile_ctr + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
            __hip_atomic_store(g.tile_ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(g.tile_ctr + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (SSC_DYN_STAMPS) {
        const uint64_t dc = __builtin_amdgcn_s_memtime() - stamp_c0, dr = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        if (lane == 0 && (wave == 0 || wave == 4)) {  // per block 2 x 24 dwords (wave 0, wave 4): {dc, dr}, 4 phase sums, entry and loop-start realtime, 2 barrier sums
            uint32_t *o = reinterpret_cast<uint32_t *>(g.S) + 48 * blockIdx.x + 24 * group;
            o[0] = (uint32_t)dc; o[1] = (uint32_t)(dc >> 32); o[2] = (uint32_t)dr; o[3] = (uint32_t)(dr >> 32);
#pragma unroll
            for (int k = 0; k < 4; ++k) { o[4 + 2 * k] = (uint32_t)ph[k]; o[5 + 2 * k] = (uint32_t)(ph[k] >> 32); }
            o[12] = (uint32_t)stamp_entry; o[13] = (uint32_t)(stamp_entry >> 32); o[14] = (uint32_t)stamp_r0; o[15] = (uint32_t)(stamp_r0 >> 32);
            o[16] = (uint32_t)ph[4]; o[17] = (uint32_t)(ph[4] >> 32); o[18] = (uint32_t)ph[5]; o[19] = (uint32_t)(ph[5] >> 32);
        }
    }
}

static int tiles_for(int depth) { return depth <= 32 ? 1 : (depth <= 128 ? 4 : 16); }

bool dyn_mfma_supported(const ssc_mlp_desc *mlp, int state_dim, int act_dim) {
    (void)state_dim; (void)act_dim;
    const int nfc = mlp->n_layers - 1;
    if (nfc != 1 && nfc != 2) return false;
    const int depth = mlp->dims[1];
    if (depth > 512 || (nfc == 2 && mlp->dims[2] != depth)) return false;
    if (mlp->dims[0] > kMaxIn || mlp->dims[mlp->n_layers] > SSC_MAX_STATE) return false;
    // streamed W2 (depth > 128) leaves LDS room for one layer-1 k-step only: 2 + 3*in <= 32
    if (nfc == 2 && tiles_for(depth) > 4 && l1_ksteps(mlp->dims[0]) > 1) return false;
    return true;
}

size_t dyn_mfma_workspace_bytes(const ssc_mlp_desc *mlp) {
    const int nfc = mlp->n_layers - 1;
    if (nfc != 1 && nfc != 2) return 256;
    return make_pack(tiles_for(mlp->dims[1]), nfc).total;
}

template <int UT, int NFC, bool BIASK, int KIN, bool LAG = false, int MODE = -1>
static int launch_sim(DynSimArgs g, hipStream_t s) {
    constexpr bool STREAM = (NFC == 2) && (UT > 4);
    size_t lds = (size_t)(LAG ? 4 : dyn_a2_bufs<UT, NFC>()) * UT * 2048 +
                 (KIN == 4 ? (size_t)UT * 1024 : (size_t)l1_ksteps(KIN) * UT * 2048) + (size_t)UT * 512 + (size_t)UT * 128 + 64;
    const int64_t tiles = (g.m + kDynRows - 1) / kDynRows;
    unsigned grid = (unsigned)tiles;
    g.walk = 0;
    const void *kern = reinterpret_cast<const void *>(dyn_mfma_sim_kernel<UT, NFC, BIASK, KIN, LAG, MODE>);
    // Only the 4-slot LAG kernel of the BASELINE shape walks (the other streamed shapes keep a block per tile).
    if constexpr (STREAM && LAG && MODE != 2) {
        // Streamed W2 (one block per CU: 8 waves x 256 VGPRs, ~158 KB of LDS): with more row tiles than CUs a block WALKS
        // over row tiles -- the W2 ring, its barriers and the two wave groups' 1.5-tile lag run on across the row tiles
        // as if the next tile's first step were the next step of the same rows, so the prologue (weight images + the
        // first two W2 tiles: ~4.5 us of a ~92 us tile at H = 4) is paid once per block instead of once per 256 rows.
        // The next tile's start states come in by LDS-DMA under the current tile's last step (stage_next_tile), the
        // tiles beyond a block's first two are handed out by a shared counter (next_tile in the kernel).
        static int n_cu = 0;
        if (n_cu == 0) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
                n_cu = v;
            else
                n_cu = 256;
        }
        const int64_t n_s0 = g.s0_rows > 0 ? (g.m + g.s0_rows - 1) / g.s0_rows : 0;
        const bool can_walk = !g.fwd_mode && g.H > 0 && g.d <= 4 && g.m <= 0x7fffffffLL && n_s0 * g.d * 4 <= 0x7fffffffLL;
        if (can_walk && tiles > n_cu && g.tile_ctr != nullptr) {
            grid = (unsigned)n_cu;
            g.walk = 1;
            lds += (size_t)kNW * 2 * 64 * 4 + 64;   // the start-state staging area and the next-tile slots
            kern = reinterpret_cast<const void *>(dyn_mfma_sim_kernel<UT, NFC, BIASK, KIN, LAG, MODE, true>);
        }
    }
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "hipFuncSetAttribute(dyn_mfma_sim_kernel)");
        if (rc) return rc;
    }
    void *params[] = {&g};
    if (int rc = check_hip(hipLaunchKernel(kern, dim3(grid), dim3(kDynThreads), params, lds, s), "hipLaunchKernel(dyn_mfma_sim_kernel)"))
        return rc;
    return check_launch("dyn_mfma_sim_kernel");
}

static DynNet make_net(const ssc_mlp_desc *mlp, int UT) {
    const int nfc = mlp->n_layers - 1;
    DynNet n;
    n.in = mlp->dims[0]; n.depth = mlp->dims[1]; n.out = mlp->dims[mlp->n_layers]; n.nfc = nfc;
    n.biask = (nfc == 2) && (n.depth + 2 <= 32 * UT);
    n.compact1 = l1_compact(n.in, n.out);
    n.W1 = mlp->W[0]; n.b1 = mlp->b[0];
    n.W2 = (nfc == 2) ? mlp->W[1] : nullptr; n.b2 = (nfc == 2) ? mlp->b[1] : nullptr;
    n.W3 = mlp->W[nfc]; n.b3 = mlp->b[nfc];
    return n;
}

// write the packed weight image (ssc_dyn_prepare, or the first half of an unprepared call)
int dyn_mfma_prepare(const ssc_mlp_desc *mlp, const ssc_norm *norm, void *wsv, hipStream_t s) {
    const int nfc = mlp->n_layers - 1;
    const int UT = tiles_for(mlp->dims[1]);
    const DynPack pk = make_pack(UT, nfc);
    const DynNet n = make_net(mlp, UT);
    const int64_t n_pack = (nfc == 2 ? (int64_t)UT * UT * 1024 : 0) + (int64_t)UT * 256 + (int64_t)kMaxKS1 * UT * 1024 +
                           (int64_t)UT * 32 + 16;
    ssc_norm nm{};
    if (norm) nm = *norm;
    hipLaunchKernelGGL(dyn_pack_kernel, dim3(blocks_for(n_pack)), dim3(256), 0, s, n, UT, pk, nm,
                       static_cast<unsigned char *>(wsv));
    return check_launch("dyn_pack_kernel");
}

static int run_mfma(const ssc_mlp_desc *mlp, DynSimArgs &g, void *wsv, hipStream_t s) {
    const int nfc = mlp->n_layers - 1;
    const int UT = tiles_for(mlp->dims[1]);
    const DynPack pk = make_pack(UT, nfc);
    const DynNet n = make_net(mlp, UT);
    unsigned char *ws = static_cast<unsigned char *>(wsv);
    SSC_REQUIRE(g.m <= 0x7fffffffLL, "dyn_mfma: m = %lld rows per launch (the kernel indexes rows with 32 bits)", (long long)g.m);
    g.in = n.in; g.out = n.out;
    g.a1 = ws + pk.a1; g.a2 = ws + pk.a2; g.a3 = ws + pk.a3;
    g.b2 = reinterpret_cast<const float *>(ws + pk.b2); g.b3 = reinterpret_cast<const float *>(ws + pk.b3);
    g.nm = reinterpret_cast<const float *>(ws + pk.nm);
    g.tile_ctr = reinterpret_cast<int32_t *>(ws + pk.ctr);
    const int kin = (n.in <= 4 && n.out <= 4) ? 4 : (n.in <= 10 ? 10 : kMaxIn);
#define SSC_DYN_CASE(U, F, B)                                                                 \
    if (UT == U && nfc == F && n.biask == B)                                                  \
        return kin == 4 ? launch_sim<U, F, B, 4>(g, s) : kin == 10 ? launch_sim<U, F, B, 10>(g, s) : launch_sim<U, F, B, kMaxIn>(g, s)
    if (n.compact1 && UT == 16 && nfc == 2) {   // streamed W2, <= 4 inputs and outputs (the BASELINE shape): the 4-slot, 1.5-tile-lag kernel
        const int mode = g.fwd_mode ? 2 : (g.sample ? 1 : 0);
        if (n.biask)
            return mode == 2 ? launch_sim<16, 2, true, 4, true, 2>(g, s) : mode == 1 ? launch_sim<16, 2, true, 4, true, 1>(g, s)
                                                                                     : launch_sim<16, 2, true, 4, true, 0>(g, s);
        return mode == 2 ? launch_sim<16, 2, false, 4, true, 2>(g, s) : mode == 1 ? launch_sim<16, 2, false, 4, true, 1>(g, s)
                                                                                  : launch_sim<16, 2, false, 4, true, 0>(g, s);
    }
    SSC_DYN_CASE(1, 1, false); SSC_DYN_CASE(4, 1, false); SSC_DYN_CASE(16, 1, false);
    SSC_DYN_CASE(1, 2, false); SSC_DYN_CASE(4, 2, false); SSC_DYN_CASE(16, 2, false);
    SSC_DYN_CASE(1, 2, true); SSC_DYN_CASE(4, 2, true); SSC_DYN_CASE(16, 2, true);
#undef SSC_DYN_CASE
    return set_error(SSC_EUNSUPPORTED, "dyn_mfma: no kernel for UT=%d nfc=%d", UT, nfc);
}

int dyn_mfma_forward_sim(const ssc_mlp_desc *mlp, const ssc_norm *norm, int64_t m, int32_t H, int32_t state_dim,
                         int32_t act_dim, const float *d_s0, int64_t s0_rows, const float *d_A, float *d_S,
                         void *ws, bool prepared, hipStream_t s) {
    if (!prepared)
        if (int rc = dyn_mfma_prepare(mlp, norm, ws, s)) return rc;
    DynSimArgs g{};
    g.m = m; g.H = H; g.d = state_dim; g.a = act_dim; g.fwd_mode = 0;
    g.s0 = d_s0; g.s0_rows = m / s0_rows; g.A = d_A; g.S = d_S;
    return run_mfma(mlp, g, ws, s);
}

// the same simulation with the candidate actions drawn inside the kernel (ssc_mpc_forward_sim)
int dyn_mfma_forward_sim_sampled(const ssc_mlp_desc *mlp, const ssc_norm *norm, const ssc_mpc_sampling *sp, int64_t m,
                                 int32_t H, int32_t state_dim, int32_t act_dim, const float *d_s0, int64_t s0_rows,
                                 float *d_A_out, float *d_S, void *ws, bool prepared, hipStream_t s) {
    if (!prepared)
        if (int rc = dyn_mfma_prepare(mlp, norm, ws, s)) return rc;
    DynSimArgs g{};
    g.m = m; g.H = H; g.d = state_dim; g.a = act_dim; g.fwd_mode = 0;
    g.s0 = d_s0; g.s0_rows = m / s0_rows; g.A = nullptr; g.S = d_S;
    g.sample = 1; g.N = sp->n_samples; g.seed = sp->seed; g.pid0 = sp->problem_id0; g.t = sp->t; g.t_base = sp->d_t_base;
    for (int a = 0; a < SSC_MAX_ACT; ++a) {
        g.low[a] = a < act_dim ? sp->low[a] : 0.0f;
        g.span[a] = a < act_dim ? sp->high[a] - sp->low[a] : 0.0f;
    }
    g.A_out = d_A_out;
    g.active = sp->d_problem_active;
    return run_mfma(mlp, g, ws, s);
}

int dyn_mfma_mlp_forward(const ssc_mlp_desc *mlp, int64_t m, const float *d_x, float *d_y, void *ws, bool prepared,
                         hipStream_t s) {
    if (!prepared)
        if (int rc = dyn_mfma_prepare(mlp, nullptr, ws, s)) return rc;
    DynSimArgs g{};
    g.m = m; g.H = 1; g.d = 0; g.a = 0; g.fwd_mode = 1;
    g.s0 = nullptr; g.s0_rows = 1; g.A = d_x; g.S = d_y;
    return run_mfma(mlp, g, ws, s);
}

}  // namespace ssc
