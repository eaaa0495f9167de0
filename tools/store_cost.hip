// What does a global store cost the ISSUING wave?  (fused actor rollout: 7 dword stores per wave-step cost ~330 of
// ~2300 cycles although the chip moves only 1.3 TB/s -- tools/exp_actor_abl.py log=1 vs log=0.)
// One wave per SIMD (256 blocks x 256 threads), each iteration = NV independent v_fma (4 cycles each) followed by
// NS stores of W dwords per lane (SGPR-base form, row-major [iter][store][lane]); cycles per iteration from
// s_memtime.  The byte rate is kept far below HBM, so what is measured is issue cost, not bandwidth.
// build: hipcc -O3 --offload-arch=gfx950 tools/store_cost.hip -o tools/_build/store_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define GLOBAL __attribute__((address_space(1)))
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

template <int NV, int NS, int W, int NT>
__global__ __launch_bounds__(256) void k(float *buf, uint64_t *cyc, int iters, float seed) {
    const int lane_global = blockIdx.x * 256 + threadIdx.x;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
    // one contiguous region per (iteration, store): [iter][NS][n][W]
    const int64_t n = (int64_t)gridDim.x * 256;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NV / 16; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(v[i]));
        float *row = buf + ((int64_t)it * NS) * n * W;      // wave-uniform
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float *p = row + (int64_t)s * n * W;
            uint64_t u = (uint64_t)p;
            u = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u);
            GLOBAL char *g = (GLOBAL char *)u + (uint32_t)lane_global * (4u * W);
            if constexpr (W == 1) { if (NT) __builtin_nontemporal_store(v[s & 15], (GLOBAL float *)g); else *(GLOBAL float *)g = v[s & 15]; }
            else if constexpr (W == 2) { f2 x = {v[s & 15], v[(s + 1) & 15]}; if (NT) __builtin_nontemporal_store(x, (GLOBAL f2 *)g); else *(GLOBAL f2 *)g = x; }
            else { f4 x = {v[s & 15], v[(s + 1) & 15], v[(s + 2) & 15], v[(s + 3) & 15]}; if (NT) __builtin_nontemporal_store(x, (GLOBAL f4 *)g); else *(GLOBAL f4 *)g = x; }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (v[0] == 123.456f) buf[0] = v[1];
}

template <int NV, int NS, int W, int NT>
int run(const char *name, float *buf, uint64_t *dcyc, int iters) {
    std::vector<uint64_t> h(1024);
    for (int rep = 0; rep < 3; ++rep) {
        k<NV, NS, W, NT><<<256, 256>>>(buf, dcyc, iters, 1.0f);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), dcyc, 1024 * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-44s cycles/iter median %.0f  (min %.0f max %.0f)  bytes/iter/wave %d\n", name, (double)h[512] / iters, (double)h[0] / iters,
           (double)h[1023] / iters, NS * W * 256);
    return 0;
}


// MODE 1: raw buffer store, per-lane VGPR byte offset (offen).  MODE 2: ADD_TID_ENABLE descriptor (stride 4, no address
// VGPR at all: the hardware adds lane * stride), wave position in soffset.  A wrong descriptor field drops stores (range
// check) instead of faulting; the host counts what landed in the LAST iteration's rows.
template <int NV, int NS, int MODE>
__global__ __launch_bounds__(256) void kb(float *buf, uint64_t *cyc, int iters, float seed, unsigned total_bytes) {
    const int lane_global = blockIdx.x * 256 + threadIdx.x;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
    const unsigned n = gridDim.x * 256;
    const __amdgpu_buffer_rsrc_t rs = MODE == 2 ? __builtin_amdgcn_make_buffer_rsrc(buf, 4, 64, 0x00800000)
                                                : __builtin_amdgcn_make_buffer_rsrc(buf, 0, total_bytes, 0x00020000);
    const unsigned wave_off = __builtin_amdgcn_readfirstlane((lane_global & ~63) * 4);
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NV / 16; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(v[i]));
        const unsigned row = __builtin_amdgcn_readfirstlane((unsigned)it * NS * n * 4u);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const unsigned so = row + (unsigned)s * n * 4u;     // wave-uniform
            if constexpr (MODE == 2)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[s & 15]), rs, 0, so + wave_off, 2 /*slc*/);
            else {
                unsigned vo = (unsigned)lane_global * 4u;
                asm volatile("" : "+v"(vo));
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[s & 15]), rs, vo, so, 2);
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    if (v[0] == 123.456f) buf[0] = v[1];
}

template <int NV, int NS, int MODE>
int runb(const char *name, float *buf, uint64_t *dcyc, int iters, unsigned total_bytes) {
    std::vector<uint64_t> h(1024);
    CK(hipMemset(buf, 0, total_bytes));
    for (int rep = 0; rep < 3; ++rep) {
        kb<NV, NS, MODE><<<256, 256>>>(buf, dcyc, iters, 1.0f, total_bytes);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), dcyc, 1024 * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const size_t n = 65536, cnt = (size_t)NS * n;
    std::vector<float> last(cnt + 1);
    if (cnt) CK(hipMemcpy(last.data(), buf + (size_t)(iters - 1) * NS * n, cnt * 4, hipMemcpyDeviceToHost));
    size_t nz = 0;
    for (size_t i = 0; i < cnt; ++i) nz += last[i] != 0.0f;
    printf("%-44s cycles/iter median %.0f  (min %.0f max %.0f)  landed %zu of %zu\n", name, (double)h[512] / iters, (double)h[0] / iters,
           (double)h[1023] / iters, nz, cnt);
    return 0;
}

int main() {
    const int iters = 256;
    float *buf; uint64_t *dcyc;
    CK(hipMalloc(&buf, (size_t)iters * 28 * 65536 * 4 * 4 + (1 << 20)));
    CK(hipMalloc(&dcyc, 1024 * 8));
#define R(NV, NS, W, NT) if (run<NV, NS, W, NT>("valu " #NV " + " #NS " x " #W "-dword stores, nt=" #NT, buf, dcyc, iters)) return 1;
    R(512, 0, 1, 1)
    R(512, 7, 1, 1) R(512, 7, 1, 0) R(512, 7, 4, 1) R(512, 7, 4, 0) R(512, 7, 2, 1)
    R(512, 2, 4, 1) R(512, 14, 1, 1) R(512, 28, 1, 1)
    R(128, 0, 1, 1) R(128, 7, 1, 1) R(128, 7, 4, 1) R(128, 2, 4, 1)
#define RB(NV, NS, MODE) if (runb<NV, NS, MODE>("valu " #NV " + " #NS " buffer stores, mode " #MODE, buf, dcyc, iters, (unsigned)((size_t)iters * 28 * 65536 * 4))) return 1;
    RB(512, 7, 1) RB(512, 7, 2) RB(128, 7, 1) RB(128, 7, 2) RB(128, 0, 2)
    return 0;
}
