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

// ---- optional per-kernel-class timing with HIP events on the launch stream (bench.py roofline) ----
enum { PROF_NONE = 0, PROF_CONV3X3_TRUNK = 1, PROF_CONV1X1_TRUNK = 2, PROF_VQ_TIER1 = 3 };
struct ProfState {
    int cls = 0;
    int used = 0;
    int cap = 0;
    hipEvent_t* ev = nullptr;
    double work = 0.0;            // algorithmic flops (or ops) of the timed launches
};
ProfState& prof_state();
struct ProfScope {
    bool on;
    hipStream_t st;
    ProfScope(int cls, hipStream_t stream, double work = 0.0) : st(stream) {
        ProfState& p = prof_state();
        on = cls != 0 && p.cls == cls && p.used + 2 <= p.cap;
        if (on) {
            p.work += work;
            (void)hipEventRecord(p.ev[p.used], st);
        }
    }
    void done() {
        if (!on) return;
        ProfState& p = prof_state();
        (void)hipEventRecord(p.ev[p.used + 1], st);
        p.used += 2;
    }
};

// autocast rounding: fp32 value -> nearest bf16 / f16 (RNE) -> fp32
__device__ __forceinline__ float round_dt(float v, int dt) {
    if (dt == VQAE_DT_BF16) return (float)(__bf16)v;
    if (dt == VQAE_DT_F16) return (float)(_Float16)v;
    return v;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

}  // namespace vqae
