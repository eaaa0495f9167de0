// A one-wave sampler of the shader clock, run on a side stream BESIDE the kernel under study:
// every ~50 us it stores (s_memrealtime [100 MHz, constant], s_memtime [shader cycles]); the slope of the
// second against the first is the clock the chip holds (MI355X_MICROARCH.md "DVFS give-back" item 6).
// The loop is bounded (nsamples x sleeps), so the kernel always exits by itself.
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/clock_probe.hip -o tools/_build/libclockprobe.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t *out, int nsamples, int sleeps) {
    if (threadIdx.x != 0) return;
    for (int s = 0; s < nsamples; ++s) {
        const uint64_t rt = __builtin_amdgcn_s_memrealtime();
        const uint64_t ct = __builtin_amdgcn_s_memtime();
        out[2 * s] = rt;
        out[2 * s + 1] = ct;
        for (int j = 0; j < sleeps; ++j) __builtin_amdgcn_s_sleep(127);
    }
}

extern "C" int clock_probe_launch(void *d_out, int nsamples, int sleeps, void *stream) {
    if (nsamples <= 0 || nsamples > (1 << 20) || sleeps < 1 || sleeps > 64) return -1;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint64_t *)d_out, nsamples, sleeps);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
