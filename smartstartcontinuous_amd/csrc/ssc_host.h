// ssc_host.h -- host-side helpers shared by the C-ABI launchers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ssc.h"

namespace ssc {

// records a thread-local message and returns `code`
int set_error(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
// converts a HIP error into SSC_EHIP (+ message); returns SSC_OK on hipSuccess
int check_hip(hipError_t e, const char *what);
// call after every kernel launch
int check_launch(const char *what);

static inline hipStream_t as_stream(ssc_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// launch geometry for one-thread-per-env kernels
constexpr int kBlock = 256;
static inline unsigned blocks_for(int64_t n, int per_block = kBlock) {
    return (unsigned)((n + per_block - 1) / per_block);
}

}  // namespace ssc

#define SSC_REQUIRE(cond, ...)                                              \
    do {                                                                    \
        if (!(cond)) return ::ssc::set_error(SSC_EINVAL, __VA_ARGS__);      \
    } while (0)
