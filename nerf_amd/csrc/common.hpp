// Shared host-side helpers for the C-ABI entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/nerf_hip.h"

namespace nerfhip {

char* err_buf();  // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NERF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), issued once per (device, kernel, size)
// instead of before every launch: later calls are a table lookup under a mutex (no runtime call besides hipGetDevice),
// so an entry point that has run once on a device makes no attribute call when it runs again, e.g. under graph capture.
hipError_t lds_limit_once(const void* kernel, size_t bytes);
// compute units of the current device, asked of the runtime once per device (256 when the query fails)
int compute_units();

#define NERF_REQUIRE(cond, ...) \
    do { if (!(cond)) return ::nerfhip::fail(NERF_E_ARG, __VA_ARGS__); } while (0)

constexpr int WAVE = 64;

}  // namespace nerfhip
