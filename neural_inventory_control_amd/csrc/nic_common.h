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

// Name of the kernel the calling thread's most recent nic_* call launched (nic_last_kernel()).  bench.py reads it so that its
// roofline object names the kernel that RAN for a shape rather than a table it keeps by hand.
const char** last_kernel_slot();  // thread-local (nic_abi.hip)
char* last_kernel_buffer();       // thread-local, 160 bytes
inline void note_kernel(const char* literal) { *last_kernel_slot() = literal; }
inline void note_kernelf(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_kernel_buffer(), 160, fmt, ap);
    va_end(ap);
    *last_kernel_slot() = last_kernel_buffer();
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Compute units of the current device (256 on MI355X), asked once: the launch heuristics (tile choice, split counts) are written
// in multiples of it instead of a literal 256.
inline int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
        (void)hipGetLastError();   // (a CPU-only process asking for split counts: keep the default, clear the error)
    }
    return n;
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

#if defined(__HIPCC__)
// Workgroup barrier that publishes LDS only.  __syncthreads() is a workgroup-scope release / acquire: in front of the s_barrier it
// waits for every outstanding GLOBAL load and store of the wave (s_waitcnt vmcnt(0)) - a full round trip to L2 / HBM whenever the
// wave has just written results.  Where the lanes of a workgroup exchange data through LDS alone (quad partial sums, layer-to-layer
// activations) only the LDS counter has to drain.  NOT a substitute where one lane reads global memory another lane wrote.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

}  // namespace nic

#define NIC_REQUIRE(cond, ...) \
    do {                       \
        if (!(cond)) return nic::fail(__VA_ARGS__); \
    } while (0)
