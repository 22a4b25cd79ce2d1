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

// ELU(alpha = 1) = max(v, 0) + min(exp(v) - 1, 0) on the hardware exponential (v_exp_f32, <= 1 ulp): 5 VALU
// instructions, branch-free.  fp32 MFMA does not co-execute with VALU on gfx950, so every instruction here is paid in
// matrix-pipe time; the previous range-reduced expm1 polynomial (19 instructions, <= 1.5 ulp relative) cost 2x more.
// Absolute error <= ~1.2e-7 (one ulp of 1.0) -- the size of the fp32 rounding of the O(1) activations around it;
// relative error near v -> 0- is not preserved, which no consumer of this path needs (the next op adds an O(0.05) bias).
// VQAE_ELU_POLY=1 at compile time restores the polynomial.
__device__ __forceinline__ float elu_act(float v) {
#ifdef VQAE_ELU_POLY
    const float x = fmaxf(fminf(v, 0.f), -88.f);
    const float k = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(k, -0.693145751953125f, x);
    r = __builtin_fmaf(k, -1.42860682030941723e-06f, r);
    float p = 1.98412698e-04f;
    p = __builtin_fmaf(p, r, 1.38888889e-03f);
    p = __builtin_fmaf(p, r, 8.33333333e-03f);
    p = __builtin_fmaf(p, r, 4.16666667e-02f);
    p = __builtin_fmaf(p, r, 1.66666667e-01f);
    p = __builtin_fmaf(p, r, 0.5f);
    const float em = __builtin_fmaf(p * r, r, r);
    const float sc = __builtin_ldexpf(1.0f, (int)k);
    const float e = __builtin_fmaf(sc, em, sc - 1.0f);
    return v > 0.f ? v : e;
#else
    // ELU(v) = median(v, e^v - 1, 0): in exact arithmetic 0 <= v <= e^v - 1 for v >= 0 and v < e^v - 1 < 0 for v < 0.  Four
    // instructions (v_mul, v_exp, v_add, v_med3) instead of five for the select form.  In fp32 `e - 1` is a multiple of
    // ulp(1) = 1.19e-7, so for 0 < v < ~3.5e-4 it can round BELOW v and the median returns it instead of v (e.g.
    // ELU(1.0003e-4) -> 1.000166e-4, ELU(3e-8) -> 0): the positive branch is NOT the identity there, unlike torch's ELU.
    // Bound: |elu_act(v) - ELU(v)| <= 1.2e-7 absolute on both branches (the same bound the negative branch has anyway;
    // recorded in DESIGN.md section 4).  The activation arithmetic is what bounds the small-channel 16-bit block kernels
    // (PMC: vector issue), hence the shorter form.
    const float e = __builtin_amdgcn_exp2f(v * 1.44269504088896341f);
    return __builtin_amdgcn_fmed3f(v, e - 1.0f, 0.0f);
#endif
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter
// (s_waitcnt vmcnt(0)): every global load in flight -- prefetched weight fragments, residual rows, the next tile's input
// rows -- has to land before each phase boundary.  For kernels whose waves exchange data through LDS only (each lane
// re-reads / overwrites only its own global elements; outputs are consumed by the next launch).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

}  // namespace vqae
