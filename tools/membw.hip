// Write-bandwidth ceilings on MI355X for the store shapes the rollout kernel can use.
// build: hipcc -O3 --offload-arch=gfx950 tools/membw.hip -o gpurun_out/membw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>
#include <ctime>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each thread = one column i of a [K][n] array, writes K rows (the rollout's access pattern)
template <int NT, typename T>
__global__ __launch_bounds__(256) void col_writer(T *__restrict__ p, int64_t n, int K, int ncols) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < ncols; ++c) {
            T v;
            if constexpr (sizeof(T) == 4) v = (T)(k + c);
            else if constexpr (sizeof(T) == 1) v = (T)(k);
            else { v = T{(float)k, (float)c, 0.f, 1.f}; }
            T *dst = p + ((int64_t)c * K + k) * n + i;
            if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
}

// rollout-like mix: 6 float columns + 1 byte column
template <int NT, int BYTE>
__global__ __launch_bounds__(256) void mix_writer(float *__restrict__ p, uint8_t *__restrict__ b, int64_t n, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < K; ++k) {
        for (int c = 0; c < 6; ++c) {
            float *dst = p + ((int64_t)c * K + k) * n + i;
            if (NT) __builtin_nontemporal_store((float)(k + c), dst); else *dst = (float)(k + c);
        }
        if (BYTE) {
            uint8_t *d = b + (int64_t)k * n + i;
            if (NT) __builtin_nontemporal_store((uint8_t)k, d); else *d = (uint8_t)k;
        }
    }
}

// packed rows: [K][6][n] floats (one contiguous 6*n*4-byte region per step) + done [K][n]
template <int NT, int BYTE, int DONE_PACK>
__global__ __launch_bounds__(256) void packed_writer(float *__restrict__ p, uint8_t *__restrict__ b, int64_t n, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < K; ++k) {
        for (int c = 0; c < 6; ++c) {
            float *dst = p + ((int64_t)k * 6 + c) * n + i;
            if (NT) __builtin_nontemporal_store((float)(k + c), dst); else *dst = (float)(k + c);
        }
        if (BYTE) {
            if (DONE_PACK) {
                // one dword store per 4 steps: lane L writes bytes of envs 4*(L&15)..+3 of step (k&~3)+(L>>4)
                if ((k & 3) == 3) {
                    const int lane = threadIdx.x & 63;
                    const int64_t wave_base = i - lane;
                    uint32_t *d = (uint32_t *)(b + (int64_t)((k & ~3) + (lane >> 4)) * n + wave_base) + (lane & 15);
                    if (NT) __builtin_nontemporal_store((uint32_t)k, d); else *d = (uint32_t)k;
                }
            } else {
                uint8_t *d = b + (int64_t)k * n + i;
                if (NT) __builtin_nontemporal_store((uint8_t)k, d); else *d = (uint8_t)k;
            }
        }
    }
}

// rollout-shaped rows with switches: NTF / NTB = non-temporal float / byte stores; DONE: 0 none, 1 byte per step,
// 2 one dword per 4 steps (4 rows x 16 lanes), 3 byte per step INSIDE the packed row (row = 25 n bytes, done last)
template <int NTF, int NTB, int DONE>
__global__ __launch_bounds__(256) void row_writer(float *__restrict__ p, uint8_t *__restrict__ b, int64_t n, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t row_f = (DONE == 3) ? (25 * n) / 4 : 6 * n;   // floats per row
    for (int k = 0; k < K; ++k) {
        float *row = p + (int64_t)k * row_f;
        for (int c = 0; c < 6; ++c) {
            float *dst = row + c * n + i;
            if (NTF) __builtin_nontemporal_store((float)(k + c), dst); else *dst = (float)(k + c);
        }
        if (DONE == 1 || DONE == 3) {
            uint8_t *d = (DONE == 3) ? reinterpret_cast<uint8_t *>(row + 6 * n) + i : b + (int64_t)k * n + i;
            if (NTB) __builtin_nontemporal_store((uint8_t)k, d); else *d = (uint8_t)k;
        } else if (DONE == 2) {
            if ((k & 3) == 3) {
                const int lane = threadIdx.x & 63;
                const int64_t wave_base = i - lane;
                uint32_t *d = (uint32_t *)(b + (int64_t)((k & ~3) + (lane >> 4)) * n + wave_base) + (lane & 15);
                if (NTB) __builtin_nontemporal_store((uint32_t)k, d); else *d = (uint32_t)k;
            }
        }
    }
}

// wave-tiled records: [K][n/64][6 fp32 columns + 64 done bytes][64 envs] -- the 7 stores of a wave-step land in ONE
// contiguous 1600-byte block, and a step of all waves is one contiguous sweep
template <int NT>
__global__ __launch_bounds__(256) void tile_writer(float *__restrict__ p, int64_t n, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t tiles = n / 64, tile = i / 64;
    const int lane = (int)(i % 64);
    for (int k = 0; k < K; ++k) {
        float *base = p + ((int64_t)k * tiles + tile) * 400;     // 6 * 64 floats + 64 bytes = 1600 B = 400 floats
        for (int c = 0; c < 6; ++c) {
            if (NT) __builtin_nontemporal_store((float)(k + c), base + c * 64 + lane); else base[c * 64 + lane] = (float)(k + c);
        }
        uint8_t *d = reinterpret_cast<uint8_t *>(base + 384) + lane;
        if (NT) __builtin_nontemporal_store((uint8_t)k, d); else *d = (uint8_t)k;
    }
}

// grid-stride f4 fill (the classic streaming-store ceiling)
template <int NT>
__global__ __launch_bounds__(256) void fill4(f4 *__restrict__ p, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f4 v = {1.f, 2.f, 3.f, 4.f};
        if (NT) __builtin_nontemporal_store(v, p + i); else p[i] = v;
    }
}

template <typename F>
float time_ms(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

// `membw series N`: N back-to-back launches of one shape, every launch bracketed by its own pair of events and
// no host sync inside the series -- the per-launch durations as a TIME SERIES (one JSON line per shape), so that a
// "ceiling" is a distribution under sustained load and not the median of ten launches after an idle gap.
template <typename F>
int series(const char *name, double bytes_per_launch, int n_launch, F f) {
    std::vector<hipEvent_t> ev(2 * (size_t)n_launch);
    for (auto &evt : ev) CK(hipEventCreate(&evt));
    f(); f(); CK(hipDeviceSynchronize());
    timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    const double t0 = ts.tv_sec + 1e-9 * ts.tv_nsec;
    for (int i = 0; i < n_launch; ++i) { CK(hipEventRecord(ev[2 * i])); f(); CK(hipEventRecord(ev[2 * i + 1])); }
    CK(hipDeviceSynchronize());
    printf("{\"shape\": \"%s\", \"bytes_per_launch\": %.0f, \"t0_monotonic\": %.6f, \"start_us\": [", name, bytes_per_launch, t0);
    for (int i = 0; i < n_launch; ++i) { float ms; hipEventElapsedTime(&ms, ev[0], ev[2 * i]); printf("%s%.1f", i ? "," : "", ms * 1e3); }
    printf("], \"dur_us\": [");
    for (int i = 0; i < n_launch; ++i) { float ms; hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]); printf("%s%.1f", i ? "," : "", ms * 1e3); }
    printf("]}\n");
    fflush(stdout);
    for (auto &evt : ev) hipEventDestroy(evt);
    return 0;
}

int main(int argc, char **argv) {
    const size_t bytes = (size_t)4 << 30;
    float *p; uint8_t *b;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&b, (size_t)256 << 20));
    const int64_t n4 = bytes / 16;
    float ms;
    if (argc > 2 && std::string(argv[1]) == "series") {
        const int nl = atoi(argv[2]);
        const int64_t n = 65536; const int K = 1024;
        const double gb6 = 6.0 * 4 * n * K, gb = gb6 + (double)n * K;
        if (argc > 3 && std::string(argv[3]) == "rows") {
            // the store-flavour / done-column matrix of the rollout's row shape, two buffers alternating like bench.py
            float *p2 = p + (bytes / 2) / 4; uint8_t *b2 = b + ((size_t)128 << 20);
            int flip = 0;
#define ROWS(NAME, NTF, NTB, DONE, BYTES) \
            if (series(NAME, BYTES, nl, [&] { flip ^= 1; row_writer<NTF, NTB, DONE><<<n / 256, 256>>>(flip ? p : p2, flip ? b : b2, n, K); })) return 1;
            ROWS("rows f:nt    done:none", 1, 0, 0, gb6)
            ROWS("rows f:plain done:none", 0, 0, 0, gb6)
            ROWS("rows f:nt    done:byte-nt", 1, 1, 1, gb)
            ROWS("rows f:plain done:byte-plain", 0, 0, 1, gb)
            ROWS("rows f:plain done:byte-nt", 0, 1, 1, gb)
            ROWS("rows f:nt    done:byte-plain", 1, 0, 1, gb)
            ROWS("rows f:nt    done:dword/4steps-nt", 1, 1, 2, gb)
            ROWS("rows f:plain done:dword/4steps-plain", 0, 0, 2, gb)
            ROWS("rows f:nt    done:dword/4steps-plain", 1, 0, 2, gb)
            ROWS("rows f:nt    done:in-row byte-plain", 1, 0, 3, gb)
            ROWS("rows f:nt    done:in-row byte-nt", 1, 1, 3, gb)
            ROWS("rows f:plain done:in-row byte-plain", 0, 0, 3, gb)
            if (series("rows f:nt done:byte-nt SAME buffer every launch", gb, nl, [&] { row_writer<1, 1, 1><<<n / 256, 256>>>(p, b, n, K); })) return 1;
            if (series("rows f:plain done:none SAME buffer every launch", gb6, nl, [&] { row_writer<0, 0, 0><<<n / 256, 256>>>(p, b, n, K); })) return 1;
            return 0;
        }
        if (series("fill4 nt grid256 (2 GiB)", (double)bytes, nl, [&] { fill4<1><<<256, 256>>>((f4 *)p, n4); })) return 1;
        if (series("packed rows + byte nt n=65536 K=1024", gb, nl, [&] { packed_writer<1, 1, 0><<<n / 256, 256>>>(p, b, n, K); })) return 1;
        if (series("6xdword plain n=65536 K=1024", gb6, nl, [&] { col_writer<0, float><<<n / 256, 256>>>(p, n, K, 6); })) return 1;
        return 0;
    }
    ms = time_ms([&] { fill4<0><<<2048, 256>>>((f4 *)p, n4); }); printf("fill4 plain grid2048 : %.1f GB/s\n", bytes / ms / 1e6);
    ms = time_ms([&] { fill4<1><<<2048, 256>>>((f4 *)p, n4); }); printf("fill4 nt    grid2048 : %.1f GB/s\n", bytes / ms / 1e6);
    ms = time_ms([&] { fill4<1><<<256, 256>>>((f4 *)p, n4); });  printf("fill4 nt    grid256  : %.1f GB/s\n", bytes / ms / 1e6);
    for (int64_t n : {65536, 262144}) {
        const int K = (int)(1024 * 65536 / n);
        const double gb6 = 6.0 * 4 * n * K, gb = gb6 + (double)n * K;
        ms = time_ms([&] { col_writer<0, float><<<n / 256, 256>>>(p, n, K, 6); }); printf("n=%ld 6xdword plain : %.1f GB/s\n", (long)n, gb6 / ms / 1e6);
        ms = time_ms([&] { col_writer<1, float><<<n / 256, 256>>>(p, n, K, 6); }); printf("n=%ld 6xdword nt    : %.1f GB/s\n", (long)n, gb6 / ms / 1e6);
        ms = time_ms([&] { mix_writer<1, 1><<<n / 256, 256>>>(p, b, n, K); }); printf("n=%ld 6xdword+byte nt : %.1f GB/s\n", (long)n, gb / ms / 1e6);
        ms = time_ms([&] { mix_writer<0, 1><<<n / 256, 256>>>(p, b, n, K); }); printf("n=%ld 6xdword+byte plain : %.1f GB/s\n", (long)n, gb / ms / 1e6);
        ms = time_ms([&] { packed_writer<1, 0, 0><<<n / 256, 256>>>(p, b, n, K); }); printf("n=%ld packed rows 6xdword nt : %.1f GB/s\n", (long)n, gb6 / ms / 1e6);
        ms = time_ms([&] { packed_writer<0, 0, 0><<<n / 256, 256>>>(p, b, n, K); }); printf("n=%ld packed rows 6xdword plain : %.1f GB/s\n", (long)n, gb6 / ms / 1e6);
        ms = time_ms([&] { packed_writer<1, 1, 0><<<n / 256, 256>>>(p, b, n, K); }); printf("n=%ld packed rows + byte nt : %.1f GB/s\n", (long)n, gb / ms / 1e6);
        ms = time_ms([&] { packed_writer<0, 1, 0><<<n / 256, 256>>>(p, b, n, K); }); printf("n=%ld packed rows + byte plain : %.1f GB/s\n", (long)n, gb / ms / 1e6);
        ms = time_ms([&] { packed_writer<1, 1, 1><<<n / 256, 256>>>(p, b, n, K); }); printf("n=%ld packed rows + done-as-dword/4steps nt : %.1f GB/s\n", (long)n, gb / ms / 1e6);
        ms = time_ms([&] { packed_writer<0, 1, 1><<<n / 256, 256>>>(p, b, n, K); }); printf("n=%ld packed rows + done-as-dword/4steps plain : %.1f GB/s\n", (long)n, gb / ms / 1e6);
        ms = time_ms([&] { tile_writer<1><<<n / 256, 256>>>(p, n, K); }); printf("n=%ld wave-tiled records nt : %.1f GB/s\n", (long)n, gb / ms / 1e6);
        ms = time_ms([&] { tile_writer<0><<<n / 256, 256>>>(p, n, K); }); printf("n=%ld wave-tiled records plain : %.1f GB/s\n", (long)n, gb / ms / 1e6);
        // same bytes as f4 columns: n/4 threads... emulate "4 envs per lane" = 16 B per lane, 1.5 columns of f4
        const int64_t nq = n / 4;
        ms = time_ms([&] { col_writer<1, f4><<<nq / 256, 256>>>((f4 *)p, nq, K, 6); }); printf("n=%ld 6xdwordx4 nt (n/4 threads) : %.1f GB/s\n", (long)n, gb6 / ms / 1e6);
        ms = time_ms([&] { col_writer<1, f4><<<n / 256, 256>>>((f4 *)p, n, K / 4, 6); }); printf("n=%ld 6xdwordx4 nt (n threads, K/4 rows) : %.1f GB/s\n", (long)n, gb6 / ms / 1e6);
    }
    return 0;
}
