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
//   * one wave owns two 32-row tiles (64 rows); a block = 4 waves = 256 rows = one CU's worth at
//     one wave per SIMD (the 512-entry register file holds the first hidden layer of both tiles as
//     bf16 fragments); the waves of a block share the weight fragments through LDS and every
//     fragment read from LDS feeds two MFMAs.
//   * every layer is computed TRANSPOSED: D[unit][row] = W^T[unit][k] * H[k][row], i.e. the
//     weights are the MFMA A operand and the activations the B operand.  The 32x32 accumulator
//     tile of a layer (unit on the register index, row on the lane) is therefore already in the
//     B-operand layout of the next layer: ReLU + v_cvt_pk_bf16_f32 in registers and it is fed
//     straight back -- no LDS round trip for activations, no lane movement
//     (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand").
//   * layer 1 (K = state+action <= 12) runs on the exact-fp32 MFMA v_mfma_f32_32x32x2_f32 with the
//     bias as C-in; hidden and output layers on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//   * layer-2 output tiles are consumed immediately by the output layer (2 more MFMAs per tile),
//     so the second hidden activation is never materialised: live registers = first hidden layer
//     as bf16 B fragments (UT*8 VGPRs) + two accumulator tiles.
//   * W2 is pre-packed once per call into fragment order (bf16, k order matched to the accumulator
//     layout) and streamed L2 -> registers -> LDS one 32-unit output tile (UT*2 KB) at a time,
//     double buffered, one barrier per tile.
#include <float.h>
#include <type_traits>

#include "ssc_device.h"
#include "ssc_host.h"

namespace ssc {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kDynRows = 256;     // rows per block: 8 waves x one 32-row tile
constexpr int kDynThreads = 512;  // waves w and w+4 share a SIMD (2 waves per SIMD, 256 VGPRs each)
constexpr int kNW = kDynThreads / 64;
constexpr int kMaxIn = 12;        // network inputs (state + action)
constexpr int kMaxKS1 = 3;        // layer-1 bf16 k-steps of 16 slots: 2 bias slots + 3 per input
// Diagnostic builds for tools/exp_dyn_clock.py (results are WRONG when set; never in libssc.so):
// 32 hidden k-steps on the 16x16x32 MFMA shape (clock experiment, wrong results);
// 16 clock stamps: every block overwrites S[4*block .. +3] with {d_memtime, d_memrealtime} of its step loop
#ifndef SSC_DYN_ABLATE
#define SSC_DYN_ABLATE 0
#endif

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

__device__ __forceinline__ int acc_row32(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }
// hidden unit carried by k slot (8*half + j) of bf16 k-step (ut, s) when the B operand is a converted
// 32x32 accumulator tile: registers 8s..8s+7 of unit tile ut
__device__ __forceinline__ int frag_unit(int ut, int s, int half, int j) {
    return ut * 32 + 16 * s + 8 * (j >> 2) + 4 * half + (j & 3);
}

// Layer 1 on the bf16 MFMA at fp32-level accuracy: x = xh + xl, w = wh + wl (bf16 head + bf16 residual),
// x*w ~= xh*wh + xl*wh + xh*wl (dropped xl*wl <= 2^-16 |x w|; products of bf16 pairs are exact in the
// fp32 accumulator).  K slots of the layer-1 contraction (16 per k-step):
//   slot 0: 1 * bf16(b1)   slot 1: 1 * residual(b1)   slot 2+3i+{0,1,2}: {xh_i*wh_i, xl_i*wh_i, xh_i*wl_i}
__host__ __device__ __forceinline__ int l1_ksteps(int in) { return (2 + 3 * in + 15) / 16; }

// Packed weight image in the workspace (all offsets in bytes, 256-aligned)
struct DynPack {
    size_t a2;   // bf16 [UT jt][UT ut][2 s][64 lane][8]   W2^T fragments (NFC == 2)
    size_t a3;   // bf16 [UT ut][2 s][2 half][8 o][8]       Wout^T fragments, only the 8 possible output rows
    size_t a1;   // bf16 [KS1][UT ut][64 lane][8]           layer-1 fragments (bias + split W1, see above)
    size_t b2;   // f32  [UT][2 half][16]                   b2 in accumulator layout
    size_t b3;   // f32  [2][16]
    size_t nm;   // f32  [6][8]  mean_x std_x mean_y std_y mean_z std_z
    size_t total;
};

static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static DynPack make_pack(int UT, int nfc) {
    DynPack p;
    size_t o = 0;
    p.a2 = o; o += al256(nfc == 2 ? (size_t)UT * UT * 2 * 64 * 8 * 2 : 0);
    p.a3 = o; o += al256((size_t)UT * 2 * 2 * 8 * 8 * 2);
    p.a1 = o; o += al256((size_t)kMaxKS1 * UT * 64 * 8 * 2);
    p.b2 = o; o += al256((size_t)UT * 32 * 4);
    p.b3 = o; o += al256(32 * 4);
    p.nm = o; o += al256(48 * 4);
    p.total = o;
    return p;
}

struct DynNet {
    const float *W1, *b1, *W2, *b2, *W3, *b3;  // W3/b3 = output layer; W2/b2 unused when nfc == 1
    int in, depth, out, nfc;
    bool biask;  // b2 in the spare k slots depth, depth+1 of the hidden contraction (needs depth + 2 <= 32*UT)
};

__device__ __forceinline__ __bf16 bf16_head(float v) { return (__bf16)v; }
__device__ __forceinline__ __bf16 bf16_resid(float v) { return (__bf16)(v - (float)(__bf16)v); }

__global__ __launch_bounds__(256) void dyn_pack_kernel(DynNet n, int UT, DynPack pk, ssc_norm nm,
                                                       unsigned char *__restrict__ ws) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid < 48) {
        const int q = (int)gid >> 3, k = (int)gid & 7;
        float v = 0.0f;
        if (q == 0) v = nm.mean_x[k]; else if (q == 1) v = nm.std_x[k];
        else if (q == 2) v = nm.mean_y[k & 3]; else if (q == 3) v = nm.std_y[k & 3];
        else if (q == 4) v = nm.mean_z[k]; else v = nm.std_z[k];
        reinterpret_cast<float *>(ws + pk.nm)[gid] = v;
    }
    const int64_t n_a2 = (n.nfc == 2) ? (int64_t)UT * UT * 2 * 64 * 8 : 0;
    const int64_t n_a3 = (int64_t)UT * 2 * 2 * 8 * 8;
    const int64_t n_a1 = (int64_t)kMaxKS1 * UT * 64 * 8;
    const int64_t n_b = (int64_t)UT * 32;
    int64_t e = gid;
    if (e < n_a2) {
        const int j = e & 7, lane = (e >> 3) & 63, s = (e >> 9) & 1;
        const int ut = (int)((e >> 10) % UT), jt = (int)((e >> 10) / UT);
        const int u = frag_unit(ut, s, lane >> 5, j), col = jt * 32 + (lane & 31);
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
        const int j = e & 7, o = (e >> 3) & 7, half = (e >> 6) & 1, s = (e >> 7) & 1, ut = (int)(e >> 8);
        const int u = frag_unit(ut, s, half, j);
        const float v = (u < n.depth && o < n.out) ? n.W3[(int64_t)u * n.out + o] : 0.0f;
        reinterpret_cast<__bf16 *>(ws + pk.a3)[e] = (__bf16)v;
        return;
    }
    e -= n_a3;
    if (e < n_a1) {
        const int j = e & 7, lane = (e >> 3) & 63, ut = (int)((e >> 9) % UT), ks = (int)((e >> 9) / UT);
        const int q = 16 * ks + 8 * (lane >> 5) + j, unit = ut * 32 + (lane & 31);
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
    if (e < n_b) {  // b2: [ut][half][reg]
        const int reg = e & 15, half = (e >> 4) & 1, ut = (int)(e >> 5);
        const int u = ut * 32 + acc_row32(reg, half);
        reinterpret_cast<float *>(ws + pk.b2)[e] = (n.nfc == 2 && !n.biask && u < n.depth) ? n.b2[u] : 0.0f;
        return;
    }
    e -= n_b;
    if (e < 32) {
        const int reg = e & 15, half = (e >> 4) & 1;
        const int o = acc_row32(reg, half);
        reinterpret_cast<float *>(ws + pk.b3)[e] = (o < n.out) ? n.b3[o] : 0.0f;
    }
}

struct DynSimArgs {
    int64_t m;
    int32_t H, d, a;        // horizon, state dim, action dim (forward mode: H = 1)
    int32_t in, out, ks1;   // network input / output width, layer-1 k-steps
    int32_t fwd_mode;       // 1: plain y = net(x) (ssc_mlp_forward); 0: forward simulation
    const float *s0;
    int64_t s0_rows;
    const float *A;         // sim: [m][H][a]; fwd: x [m][in]
    float *S;               // sim: [H+1][m][d]; fwd: y [m][out]
    const unsigned char *a1, *a2, *a3;  // packed weight image (workspace)
    const float *b2, *b3, *nm;
};

__device__ __forceinline__ float nan_to_num_div(float x, float mean, float stdv) {
    const float v = (x - mean) / stdv;  // dynamics_model.py:228-229
    if (isnan(v)) return 0.0f;
    if (isinf(v)) return v > 0.0f ? FLT_MAX : -FLT_MAX;
    return v;
}

__device__ __forceinline__ f32x16 lds_tile16(const float *p) {  // 16 consecutive floats -> accumulator init
    const f32x4 a = *reinterpret_cast<const f32x4 *>(p), b = *reinterpret_cast<const f32x4 *>(p + 4);
    const f32x4 c = *reinterpret_cast<const f32x4 *>(p + 8), d = *reinterpret_cast<const f32x4 *>(p + 12);
    f32x16 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    r[8] = c[0]; r[9] = c[1]; r[10] = c[2]; r[11] = c[3]; r[12] = d[0]; r[13] = d[1]; r[14] = d[2]; r[15] = d[3];
    return r;
}

__device__ __forceinline__ void relu_to_frags(const f32x16 &acc, bf16x8 &f0, bf16x8 &f1) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 p0, p1;  // feedforward_network.py:19 (ReLU) fused with the bf16 convert: 2 values per 2 VALU ops
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        p0[j] = relu_pack_bf16(acc[2 * j], acc[2 * j + 1]);
        p1[j] = relu_pack_bf16(acc[8 + 2 * j], acc[8 + 2 * j + 1]);
    }
    f0 = __builtin_bit_cast(bf16x8, p0);
    f1 = __builtin_bit_cast(bf16x8, p1);
}

// bf16 head / residual of x as 16-bit patterns (round to nearest even, like the weights)
__device__ __forceinline__ void split_bf16(float x, uint32_t &hi, uint32_t &lo) {
    x = fminf(fmaxf(x, -3.3895314e38f), 3.3895314e38f);  // +-FLT_MAX of nan_to_num would round to bf16 inf
    const __bf16 h = (__bf16)x;
    const __bf16 l = (__bf16)(x - (float)h);
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, l);
}

// LDS carve (bytes): [a2: NBUF x UT*2048] [a1: ks1*UT*1024] [a3: UT*512] [zero 16] [b2 UT*128] [b3 128] [nm 192]
// W2 (NFC == 2): resident when it fits (UT <= 4: all UT tiles), otherwise streamed through a ring of 3.
template <int UT, int NFC>
__host__ __device__ constexpr int dyn_a2_bufs() { return NFC == 2 ? (UT <= 4 ? UT : 3) : 0; }

// BIASK: b2 rides in two spare k slots of the hidden contraction (hidden units depth, depth+1 are the constant
// 1; their W2^T rows hold bf16 head and residual of b2) instead of being the accumulator's C-in.
template <int UT, int NFC, bool BIASK>
__global__ __launch_bounds__(kDynThreads, 2) void dyn_mfma_sim_kernel(DynSimArgs g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int A2_TILE = UT * 2048;  // one 32-unit output tile of W2^T fragments
    constexpr int NBUF = dyn_a2_bufs<UT, NFC>();
    constexpr bool STREAM = (NFC == 2) && (UT > 4);
    constexpr int A2_CHUNKS = A2_TILE / 1024;           // LDS-DMA pieces per tile
    constexpr int PPW = (A2_CHUNKS + kNW - 1) / kNW;    // pieces per wave per tile
    unsigned char *l_a2 = lds;
    unsigned char *l_a1 = l_a2 + NBUF * A2_TILE;
    unsigned char *l_a3 = l_a1 + g.ks1 * UT * 1024;
    unsigned char *l_zero = l_a3 + UT * 512;
    float *l_b2 = reinterpret_cast<float *>(l_zero + 16);
    float *l_b3 = l_b2 + UT * 32;
    float *l_nm = l_b3 + 32;  // [6][8]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, half = lane >> 5;
    // Waves 0-3 (group 0) and 4-7 (group 1) pair up on the SIMDs.  Group 1 takes the per-tile barrier in
    // the MIDDLE of its tile, group 0 at the end: the two waves of a SIMD are then half a tile out of
    // phase, and the tile-boundary work of one (ReLU, output MFMAs, bias/fragment refill) runs under the
    // other's MFMA stream instead of both idling the matrix pipe together.
    const int group = wave >> 2;

    // ---- stage the resident weights ------------------------------------------------------------
    {
        const f32x4 *s3 = reinterpret_cast<const f32x4 *>(g.a3);
        f32x4 *d3 = reinterpret_cast<f32x4 *>(l_a3);
        for (int e = tid; e < UT * 512 / 16; e += kDynThreads) d3[e] = s3[e];
        const f32x4 *s1 = reinterpret_cast<const f32x4 *>(g.a1);
        f32x4 *d1 = reinterpret_cast<f32x4 *>(l_a1);
        for (int e = tid; e < g.ks1 * UT * 1024 / 16; e += kDynThreads) d1[e] = s1[e];
        for (int e = tid; e < UT * 32; e += kDynThreads) l_b2[e] = g.b2[e];
        if (tid < 32) l_b3[tid] = g.b3[tid];
        if (tid < 48) l_nm[tid] = g.nm[tid];
        if (tid < 4) reinterpret_cast<float *>(l_zero)[tid] = 0.0f;
    }
    // raw buffer over the W2^T fragment image (reads past the end return 0)
    const __amdgpu_buffer_rsrc_t a2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char *>(g.a2), 0, NFC == 2 ? UT * A2_TILE : 0, 0x00020000);
    if (NFC == 2) {  // first W2^T tiles by LDS-DMA: everything when resident, tiles 0 and 1 of the ring otherwise
        constexpr int PRE = STREAM ? 2 : NBUF;
#pragma unroll
        for (int c = wave; c < PRE * A2_CHUNKS; c += kNW) lds_dma_1k(a2_rsrc, lane * 16, c * 1024, l_a2 + c * 1024);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the LDS-DMA pieces above
    __syncthreads();

    // output-layer A fragments: output row o = lane & 31 < 8 reads its 16 B, the others a zero line
    const unsigned char *a3_lane = (r < 8) ? l_a3 + (half * 8 + r) * 16 : l_zero;
    const int a3_step = (r < 8) ? 256 : 0;

    // ---- this lane's row ---------------------------------------------------------------------
    const int64_t row = (int64_t)blockIdx.x * kDynRows + wave * 32 + r;
    const bool valid = row < g.m;
    const int64_t rowc = valid ? row : g.m - 1;
    float st[SSC_MAX_STATE];
    if (!g.fwd_mode) {
#pragma unroll
        for (int k = 0; k < SSC_MAX_STATE; ++k) st[k] = (k < g.d) ? g.s0[(g.s0_rows == 1 ? 0 : rowc) * g.d + k] : 0.0f;
    }

    int bsel = 0;  // LDS slot of the current W2 tile (STREAM)
    constexpr int NK = UT * 2;                // k-steps per hidden tile
    constexpr int RING = (NK >= 4) ? 4 : 2;   // W2^T fragment reads in flight per wave
    bf16x8 ring[RING];
    if (NFC == 2) {
#pragma unroll
        for (int q = 0; q < RING; ++q) ring[q] = *reinterpret_cast<const bf16x8 *>(l_a2 + (q * 64 + lane) * 16);
    }
    const uint64_t stamp_c0 = (SSC_DYN_ABLATE & 16) ? __builtin_amdgcn_s_memtime() : 0;
    const uint64_t stamp_r0 = (SSC_DYN_ABLATE & 16) ? __builtin_amdgcn_s_memrealtime() : 0;
    for (int t = 0; t < g.H; ++t) {
        // ---- inputs: record S[t]; x = normalised (state, action) ------------------------------
        float xs[kMaxIn];
        if (g.fwd_mode) {
#pragma unroll
            for (int k = 0; k < kMaxIn; ++k) xs[k] = (k < g.in) ? g.A[rowc * g.in + k] : 0.0f;
        } else {
            if (valid && half == 0) {
#pragma unroll
                for (int k = 0; k < SSC_MAX_STATE; ++k)
                    if (k < g.d) g.S[((int64_t)t * g.m + row) * g.d + k] = st[k];  // dynamics_model.py:225
            }
#pragma unroll
            for (int k = 0; k < kMaxIn; ++k) {
                float v = 0.0f;
                if (k < g.d) {
                    v = nan_to_num_div(st[k < SSC_MAX_STATE ? k : 0], l_nm[0 * 8 + (k & 7)], l_nm[1 * 8 + (k & 7)]);
                } else if (k < g.in) {
                    const int ai = (k - g.d) & 3;
                    v = nan_to_num_div(g.A[(rowc * g.H + t) * g.a + ai], l_nm[2 * 8 + ai], l_nm[3 * 8 + ai]);
                }
                xs[k] = v;
            }
        }
        // ---- layer-1 B fragments: this lane's 8 k slots per k-step (slot layout above) -------------
        bf16x8 xf[kMaxKS1];
        {
            uint32_t xh[kMaxIn], xl[kMaxIn];
#pragma unroll
            for (int k = 0; k < kMaxIn; ++k) split_bf16(xs[k], xh[k], xl[k]);
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int ks = 0; ks < kMaxKS1; ++ks) {
                u32x4 w;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    uint32_t dw[2];
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        uint32_t v16[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int q = 16 * ks + 8 * hf + 2 * jj + e;
                            const int i = (q - 2) / 3, c = (q - 2) % 3;
                            v16[e] = (q < 2) ? 0x3F80u : (i < kMaxIn ? (c == 1 ? xl[i] : xh[i]) : 0u);
                        }
                        dw[hf] = v16[0] | (v16[1] << 16);
                    }
                    w[jj] = half ? dw[1] : dw[0];
                }
                xf[ks] = __builtin_bit_cast(bf16x8, w);
            }
        }
        // ---- layer 1: D[unit][row] = W1^T x + b1 -> ReLU -> bf16 B fragments of the next layer ------
        bf16x8 h1f[UT][2];
        auto layer1_tile = [&](int ut) {
            f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < kMaxKS1; ++ks)
                if (ks < g.ks1) {  // block-uniform
                    const bf16x8 a = *reinterpret_cast<const bf16x8 *>(l_a1 + ((ks * UT + ut) * 64 + lane) * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[ks], acc, 0, 0, 0);
                }
            return acc;
        };
        {
            f32x16 accn = layer1_tile(0);
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
                const f32x16 acc = accn;
                if (ut + 1 < UT) accn = layer1_tile(ut + 1);  // its MFMA runs under this tile's ReLU/convert
                relu_to_frags(acc, h1f[ut][0], h1f[ut][1]);
                // Pin the conversion HERE.  The fragments are first used inside the jt loop, so the
                // optimiser otherwise sinks ReLU+convert down to that loop's preheader and keeps all
                // UT fp32 accumulator tiles (16 VGPRs each) alive until then -> hundreds of spills.
                asm volatile("" : "+v"(h1f[ut][0]), "+v"(h1f[ut][1]));
                __builtin_amdgcn_sched_barrier(0);  // and one unit tile at a time in the machine scheduler
            }
        }
        // ---- hidden layer 2 (+ output layer fused per tile) -------------------------------------
        f32x16 acc3 = lds_tile16(l_b3 + half * 16);
        if (NFC == 2) {
            // One 32-unit output tile jt of hidden layer 2: NK k-steps over the W2^T fragments of the current
            // LDS slot, then ReLU and the two output-layer MFMAs.
            //  * The body is ONE basic block (no branch between the k-steps), so hipcc's s_waitcnt insertion
            //    keeps counted lgkmcnt waits and the fragment ring really stays RING deep.
            //  * The fragment ring runs on ACROSS tiles: the last RING k-steps of a tile already fetch the
            //    first fragments of the next one (and of the next step's first tile), so a tile starts with
            //    its operands in registers; with BIASK the accumulator starts from the inline constant 0.
            //  * STREAM: 3 LDS slots hold tiles tl (being read), tl+1 (landed) and the one in flight.
            //    Barrier number tl -- taken in tile tl after k-step X0 by group 0 and X1 = X0 - NK/2 by group
            //    1, which keeps the two waves of a SIMD half a tile out of phase -- retires every read of tile
            //    tl-1 and publishes tile tl+1; after it a wave issues its PPW pieces of tile tl+2 into the
            //    slot of tile tl-1, a few k-steps apart so that their issue cost hides under MFMAs.  The
            //    issue is unconditional: past the last tile it re-loads a tile nobody reads, and group 0 in
            //    tile 0 re-loads the (identical) bytes of tile 1.
            auto tile_body = [&](auto group_tag, int jt) {
                constexpr int GROUP = decltype(group_tag)::value;
                constexpr int X0 = NK - RING - 1, X1 = X0 - NK / 2;   // barrier k-step of group 0 / group 1
                static_assert(!STREAM || (UT == 16 && X1 + 2 + 4 * (PPW - 1) < NK), "LDS-DMA issue slots");
                const int nsel = STREAM ? (bsel == 2 ? 0 : bsel + 1) : ((jt + 1) & (UT - 1));
                const unsigned char *buf = l_a2 + (STREAM ? bsel : jt) * A2_TILE + lane * 16;
                const unsigned char *nbuf = l_a2 + nsel * A2_TILE + lane * 16;
                // group 0 is past barrier tl-1 -> its pieces of tile tl+1; group 1 passes barrier tl at X1 ->
                // pieces of tile tl+2
                const int dma_jn = (jt + 1 + GROUP) & (UT - 1);
                const int dma_slot = (bsel + 1 + GROUP) % 3;
                const int dma_src = dma_jn * A2_TILE + wave * 1024;
                unsigned char *dma_dst = l_a2 + dma_slot * A2_TILE + wave * 1024;
                f32x16 acc2 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (!BIASK) acc2 = lds_tile16(l_b2 + (jt * 2 + half) * 16);
#pragma unroll
                for (int i = 0; i < NK; ++i) {
                    const bf16x8 a = ring[i % RING];
                    ring[i % RING] = (i + RING < NK) ? *reinterpret_cast<const bf16x8 *>(buf + (i + RING) * 1024)
                                                     : *reinterpret_cast<const bf16x8 *>(nbuf + (i + RING - NK) * 1024);
#if SSC_DYN_ABLATE & 32   // timing only: the same k-step as two v_mfma_f32_16x16x32_bf16 (operand layout NOT adapted)
                    {
                        f32x4 q0 = {acc2[0], acc2[1], acc2[2], acc2[3]}, q1 = {acc2[4], acc2[5], acc2[6], acc2[7]};
                        f32x4 q2 = {acc2[8], acc2[9], acc2[10], acc2[11]}, q3 = {acc2[12], acc2[13], acc2[14], acc2[15]};
                        if (i & 1) {
                            q2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, h1f[i >> 1][1], q2, 0, 0, 0);
                            q3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, h1f[(i >> 1) ^ 1][1], q3, 0, 0, 0);
                        } else {
                            q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, h1f[i >> 1][0], q0, 0, 0, 0);
                            q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, h1f[(i >> 1) ^ 1][0], q1, 0, 0, 0);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) { acc2[e] = q0[e]; acc2[4 + e] = q1[e]; acc2[8 + e] = q2[e]; acc2[12 + e] = q3[e]; }
                    }
#else
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, h1f[i >> 1][i & 1], acc2, 0, 0, 0);
#endif
                    if (STREAM) {
                        const int i0 = GROUP ? X1 + 2 : 1;  // first issue slot after this group's barrier
                        if (i >= i0 && (i - i0) % 4 == 0 && (i - i0) / 4 < PPW) {
                            const int p = (i - i0) / 4;
                            lds_dma_1k(a2_rsrc, lane * 16, dma_src + p * kNW * 1024, dma_dst + p * kNW * 1024);
                        }
                        if (i == (GROUP ? X1 : X0)) {  // barrier tl
                            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's pieces of tile tl+1 landed
                            __syncthreads();
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const bf16x8 a30 = *reinterpret_cast<const bf16x8 *>(a3_lane + (jt * 2 + 0) * a3_step);
                const bf16x8 a31 = *reinterpret_cast<const bf16x8 *>(a3_lane + (jt * 2 + 1) * a3_step);
                bf16x8 f0, f1;
                relu_to_frags(acc2, f0, f1);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a30, f0, acc3, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a31, f1, acc3, 0, 0, 0);
                if (STREAM) bsel = nsel;
            };
            if (!STREAM || group == 0) {
#pragma unroll 1
                for (int jt = 0; jt < UT; ++jt) tile_body(std::integral_constant<int, 0>{}, jt);
            } else {
#pragma unroll 1
                for (int jt = 0; jt < UT; ++jt) tile_body(std::integral_constant<int, 1>{}, jt);
            }
        } else {
#pragma unroll
            for (int i = 0; i < UT * 2; ++i) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(a3_lane + i * a3_step);
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, h1f[i >> 1][i & 1], acc3, 0, 0, 0);
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- z[o]: rows 0..3 sit in regs 0..3 of half 0, rows 4..7 in regs 0..3 of half 1 ------------
        float z[SSC_MAX_STATE];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const float mine = acc3[o], other = __shfl_xor(acc3[o], 32);
            z[o] = half ? other : mine;
            z[4 + o] = half ? mine : other;
        }
        if (g.fwd_mode) {
            if (valid && half == 0) {
#pragma unroll
                for (int o = 0; o < SSC_MAX_STATE; ++o)
                    if (o < g.out) g.S[row * g.out + o] = z[o];
            }
        } else {
#pragma unroll
            for (int k = 0; k < SSC_MAX_STATE; ++k)
                if (k < g.d) st[k] = st[k] + (z[k] * l_nm[5 * 8 + k] + l_nm[4 * 8 + k]);  // :234-237
        }
    }
    if (!g.fwd_mode) {
        if (valid && half == 0) {
#pragma unroll
            for (int k = 0; k < SSC_MAX_STATE; ++k)
                if (k < g.d) g.S[((int64_t)g.H * g.m + row) * g.d + k] = st[k];  // :240
        }
    }
    // group 1 issued LDS-DMA after its last barrier: it must land before this workgroup's LDS is released
    if (STREAM) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    if (SSC_DYN_ABLATE & 16) {
        const uint64_t dc = __builtin_amdgcn_s_memtime() - stamp_c0, dr = __builtin_amdgcn_s_memrealtime() - stamp_r0;
        if (tid == 0) {
            uint32_t *o = reinterpret_cast<uint32_t *>(g.S) + 4 * blockIdx.x;
            o[0] = (uint32_t)dc; o[1] = (uint32_t)(dc >> 32); o[2] = (uint32_t)dr; o[3] = (uint32_t)(dr >> 32);
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
    return true;
}

size_t dyn_mfma_workspace_bytes(const ssc_mlp_desc *mlp) {
    const int nfc = mlp->n_layers - 1;
    if (nfc != 1 && nfc != 2) return 256;
    return make_pack(tiles_for(mlp->dims[1]), nfc).total;
}

template <int UT, int NFC, bool BIASK>
static int launch_sim(const DynSimArgs &g, hipStream_t s) {
    const size_t lds = (size_t)dyn_a2_bufs<UT, NFC>() * UT * 2048 + (size_t)g.ks1 * UT * 1024 + (size_t)UT * 512 + 16 +
                       (size_t)UT * 128 + 128 + 192;
    auto kern = dyn_mfma_sim_kernel<UT, NFC, BIASK>;
    if (lds > 64 * 1024) {
        int rc = check_hip(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                           "hipFuncSetAttribute(dyn_mfma_sim_kernel)");
        if (rc) return rc;
    }
    const unsigned grid = (unsigned)((g.m + kDynRows - 1) / kDynRows);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kDynThreads), lds, s, g);
    return check_launch("dyn_mfma_sim_kernel");
}

static int run_mfma(const ssc_mlp_desc *mlp, const ssc_norm *norm, DynSimArgs &g, void *wsv, hipStream_t s) {
    const int nfc = mlp->n_layers - 1;
    const int depth = mlp->dims[1];
    const int UT = tiles_for(depth);
    const DynPack pk = make_pack(UT, nfc);
    DynNet n;
    n.in = mlp->dims[0]; n.depth = depth; n.out = mlp->dims[mlp->n_layers]; n.nfc = nfc;
    n.biask = (nfc == 2) && (depth + 2 <= 32 * UT);
    n.W1 = mlp->W[0]; n.b1 = mlp->b[0];
    n.W2 = (nfc == 2) ? mlp->W[1] : nullptr; n.b2 = (nfc == 2) ? mlp->b[1] : nullptr;
    n.W3 = mlp->W[nfc]; n.b3 = mlp->b[nfc];
    unsigned char *ws = static_cast<unsigned char *>(wsv);
    const int64_t n_pack = (nfc == 2 ? (int64_t)UT * UT * 1024 : 0) + (int64_t)UT * 256 + (int64_t)kMaxKS1 * UT * 512 +
                           (int64_t)UT * 32 + 32;
    ssc_norm nm{};
    if (norm) nm = *norm;
    hipLaunchKernelGGL(dyn_pack_kernel, dim3(blocks_for(n_pack)), dim3(256), 0, s, n, UT, pk, nm, ws);
    g.in = n.in; g.out = n.out; g.ks1 = l1_ksteps(n.in);
    g.a1 = ws + pk.a1; g.a2 = ws + pk.a2; g.a3 = ws + pk.a3;
    g.b2 = reinterpret_cast<const float *>(ws + pk.b2); g.b3 = reinterpret_cast<const float *>(ws + pk.b3);
    g.nm = reinterpret_cast<const float *>(ws + pk.nm);
#define SSC_DYN_CASE(U, F, B) if (UT == U && nfc == F && n.biask == B) return launch_sim<U, F, B>(g, s)
    SSC_DYN_CASE(1, 1, false); SSC_DYN_CASE(4, 1, false); SSC_DYN_CASE(16, 1, false);
    SSC_DYN_CASE(1, 2, false); SSC_DYN_CASE(4, 2, false); SSC_DYN_CASE(16, 2, false);
    SSC_DYN_CASE(1, 2, true); SSC_DYN_CASE(4, 2, true); SSC_DYN_CASE(16, 2, true);
#undef SSC_DYN_CASE
    return set_error(SSC_EUNSUPPORTED, "dyn_mfma: no kernel for UT=%d nfc=%d", UT, nfc);
}

int dyn_mfma_forward_sim(const ssc_mlp_desc *mlp, const ssc_norm *norm, int64_t m, int32_t H, int32_t state_dim,
                         int32_t act_dim, const float *d_s0, int64_t s0_rows, const float *d_A, float *d_S,
                         void *ws, hipStream_t s) {
    DynSimArgs g{};
    g.m = m; g.H = H; g.d = state_dim; g.a = act_dim; g.fwd_mode = 0;
    g.s0 = d_s0; g.s0_rows = s0_rows; g.A = d_A; g.S = d_S;
    return run_mfma(mlp, norm, g, ws, s);
}

int dyn_mfma_mlp_forward(const ssc_mlp_desc *mlp, int64_t m, const float *d_x, float *d_y, void *ws, hipStream_t s) {
    DynSimArgs g{};
    g.m = m; g.H = 1; g.d = 0; g.a = 0; g.fwd_mode = 1;
    g.s0 = nullptr; g.s0_rows = 1; g.A = d_x; g.S = d_y;
    return run_mfma(mlp, nullptr, g, ws, s);
}

}  // namespace ssc
