// Shared host-side helpers of libnic_hip.so: error reporting and launch checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/nic_rollout.h"

namespace nic {

char* last_error_buffer();  // thread-local, 512 bytes (nic_abi.hip)

inline int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buffer(), 512, fmt, ap);
    va_end(ap);
    return 1;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

}  // namespace nic

#define NIC_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) return nic::fail(__VA_ARGS__); \
    } while (0)
