// Issue rate of v_mfma_f32_16x16x16_bf16 (legacy K = 16) vs v_mfma_f32_16x16x32_bf16 on gfx950: cycles per MFMA for one
// wave per SIMD issuing back to back on 4 accumulators.  build: hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void rate(float *out, uint64_t *cyc, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 a8, b8;
    s16x4 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(float)(threadIdx.x + i); b8[i] = (__bf16)(float)(i + 1); }
    for (int i = 0; i < 4; ++i) { a4[i] = (short)(0x3f80 + threadIdx.x + i); b4[i] = (short)(0x3f80 + i); }
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (SHAPE == 32) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[q], 0, 0, 0);
            else acc[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[q], 0, 0, 0);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float *out; uint64_t *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int shape : {32, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            if (shape == 32) hipLaunchKernelGGL(rate<32>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(rate<16>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        // s_memtime counts at 100 MHz on gfx9; report per MFMA in memtime ticks and let the ratio speak
        printf("16x16x%d bf16: %.3f memtime ticks per MFMA (one wave per SIMD, 4 accumulators)\n", shape, (double)c / (iters * 4.0));
    }
    return 0;
}
