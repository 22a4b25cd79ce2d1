// Shared host-side helpers for libvqae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/vqae_hip.h"

namespace vqae {

char* last_error_buf();   // thread-local, 512 bytes

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define VQAE_HIP_CHECK(expr)                                                                     \
    do {                                                                                         \
        hipError_t e__ = (expr);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return ::vqae::fail(VQAE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                __FILE__, __LINE__);                                             \
    } while (0)

#define VQAE_LAUNCH_CHECK() VQAE_HIP_CHECK(hipGetLastError())

#define VQAE_REQUIRE(cond, code, ...)                 \
    do {                                              \
        if (!(cond)) return ::vqae::fail(code, __VA_ARGS__); \
    } while (0)

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

}  // namespace vqae
