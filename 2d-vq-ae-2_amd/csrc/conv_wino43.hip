// Winograd F(4x4, 3x3) form of the C = 128 trunk Fixup block on the 32-wide code grid, fp32 (round 3):
//   conv2 (3x3 circular, conv_block.py:208)  as  Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A   on 6 x 6 input / 4 x 4 output tiles,
// followed in the same workgroup by conv3 (+ scale / bias4 / residual) and the NEXT block's conv1 -- the contract of
// wino_trunk_kernel (conv_wino.hip), whose 1x1 tails this kernel repeats on two 128-pixel halves.
//
// Why: on gfx950 the fp32 matrix instruction runs at the vector rate and does not co-execute with vector work, so the trunk
// kernel is bound by MFMA + VALU issue cycles.  F(4x4, 3x3) needs 36 multiplies per 16 outputs and channel pair
// (2.25 per output) where F(2x2, 3x3) needs 4: conv2's matrix work drops 1.78x, the block's (conv2 + two 1x1) 1.41x, for ~1.3x the
// transform / fold vector work.  The price is numerical: B^T / A^T carry entries up to 8, and the fp32 result is ~4x further
// from the exact value than the direct form (5e-7 instead of 1.4e-7 of the feature range through the 68-block encoder;
// tools/dbg/wino43_numerics.py) -- indices stay bit-exact on every fixture.  VQAE_WINO43=0 keeps F(2x2, 3x3).
//
// Work split.  A 256-thread workgroup owns 8 image rows x 32 columns = 256 output pixels = 16 tiles (2 tile rows x 8 tile
// columns).  A wave owns 32 output channels (two 16-channel MFMA row blocks) x all 16 tiles, on v_mfma_f32_16x16x4_f32 with the
// weights as the ROW operand: lane (li, q) receives tile li, channels 4 q .. 4 q + 3 of each block -- four consecutive channels,
// so every LDS / global access of the epilogue is 128 bits wide.  The 6 x 6 transformed domain is walked one row xi at a time:
//   transform  V_xi[nu][tile][c] = (B^T d B)[xi][nu], nu = 0..5, all 128 channels -> LDS (6 x 16 x 132 floats); thread = (tile column,
//              channel quad), two tile rows; input rows straight from global / L2 in batches of 2 columns, two batches in flight
//   GEMM       acc (2 blocks x 4 registers) = U[xi, nu] (32 ch x 128) x V_xi[nu] (128 x 16 tiles): one ds_read_b128 + two 1 KiB
//              weight-fragment loads feed 8 MFMAs
//   fold       Z[b] += A^T[b][nu] acc  (pairs nu = 1, 2 and 3, 4 share their sum / difference);  after nu = 5:  Y[a][b] += A^T[a][xi] Z[b]
// Live: Y 128 + Z 32 + 2 x 8 accumulator registers.  The passes xi = 1..4 share one loop body (their row combinations
// d4 + c1 d1 + c2 d2 + c3 d3 and folds Y_a += a_a Z differ in coefficients only); xi = 0 and 5 are specialised (three rows, one Y row).
// Tails: output rows {0, 1} of every tile (image rows {0, 1, 4, 5} of the 8) form the first 128-pixel half, rows {2, 3} the second;
// each half is t2 -> LDS [128 px][132], conv3, epilogue, next conv1 exactly as in wino_trunk_kernel.
// The same kernel serves the levels above the trunk (C = 64 on the 64-wide grid: 2 slices x 2 tile groups; C = 32 on the 128-wide
// grid: 1 slice x 4 tile groups) and C = 256 on the code grid (8 slices, 512 threads, one workgroup per CU).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
using vqae::elu_act;
using vqae::lds_barrier;

struct W43K {
    const float* __restrict__ t1;        // [M][C] conv2 input
    const float* __restrict__ U;         // G g G^T, [36 pos][C / 32 slices][C / 16 k-groups][2 blocks][64 lanes][4]
    const float* __restrict__ w3;        // [C][C], fragment order of conv_wino.hip (k-slice 8)
    const float* __restrict__ w1n;       // same, the next block's conv1 (TAIL == 2)
    float* xio;                          // [M][C] residual stream, updated in place
    float* y2;                           // [M][C] next block's t1 (TAIL == 2)
    int H, M;
    int stag, first_gen;                 // developer experiment (VQAE_W43_STAG, default 0 = off): delay (x 1024 cycles) of the odd-slot workgroups among the first first_gen
    float act_a, act_b, t_scale, t_b4, n_b1a, n_b1b, n_b2a, n_b2b;
};

// (C, grid width): (256, 32) [512 threads, one workgroup per CU], (128, 32), (64, 64), (32, 128): 8 image rows x the whole grid width
template <int C_> struct W43Cfg {
    static constexpr int C = C_;
    static constexpr int W = C >= 128 ? 32 : (C == 64 ? 64 : 128);
    static constexpr int NT = C == 256 ? 512 : 256;
    static constexpr int NWV = NT / 64;
    static constexpr int NS = C / 32;                  // 32-channel slices
    static constexpr int TG = NWV / NS;                // 16-tile groups (a wave = one slice x one group)
    static constexpr int TILES = 16 * TG;              // = 2 tile rows x W / 4 tile columns
    static constexpr int TC = W / 4;
    static constexpr int PXH = 4 * W;                  // pixels of one half (4 of the 8 rows)
    static constexpr int C4 = C / 4;
    static constexpr int RP = NT / C4;                 // pixels one sweep of the workgroup's threads covers (= TC)
    static constexpr int LDT = C + 4;
    static constexpr int KG = C / 16;                  // k-groups (16 channels = 4 MFMAs per block) per position
    static constexpr int KS = C / 8;                   // tails: k-slices
    static constexpr int NI = C >= 64 ? 2 : 1;         // tails: 32-channel tiles per wave
    static constexpr int WN = C / (32 * NI);           //        waves along the channels
    static constexpr int MI = PXH / ((NWV / WN) * 32); //        32-pixel tiles per wave (MI * NI = 4 accumulators)
    static constexpr int V_BYTES = 6 * TILES * LDT * 4, T_BYTES = PXH * LDT * 4;
    static constexpr int LDS_BYTES = V_BYTES > T_BYTES ? V_BYTES : T_BYTES;
    static_assert(TILES == 2 * TC && RP == TC && PXH / RP == 16 && MI * NI == 4, "geometry");
};
#ifndef W43_RD
#define W43_RD 4
#endif
#ifndef W43_NRES0
#define W43_NRES0 4
#endif
#ifndef W43_EARLY
#define W43_EARLY 1
#endif
constexpr int EARLY = W43_EARLY;                      // input batches of the next pass requested before the fold over xi (0, 1, 2)
constexpr int RD = W43_RD;                                // weight-fragment ring depth (k-groups of 8 MFMAs = 256 MFMA cycles each)
constexpr int RT = 3;                                // tails: ring depth in k-slices (16 MFMAs = 1024 cycles each)

__device__ __forceinline__ f32x4 fma4(const f32x4& a, float s, const f32x4& c) {
    return __builtin_elementwise_fma(a, f32x4{s, s, s, s}, c);
}

// uniform base (SGPR pair) + 32-bit per-lane byte offset: the `global_load ... v_off, s[base]` form.  (A per-lane 64-bit pointer plus
// a large constant offset per load makes hipcc materialise -- and, hoisted out of the xi loop, spill -- one address pair per load.)
__device__ __forceinline__ f32x4 ldg(const float* sbase, unsigned voff_bytes) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(sbase) + voff_bytes);
}

// Developer aid (off by default; tools/dbg/w43_trace.py): per-phase s_memtime stamps of every wave of the chained (TAIL == 2) launches.
#ifdef W43_TRACE
__device__ unsigned long long* g_w43_trace = nullptr;
#define STAMP(i) do { if (TAIL == 2 && lane == 0 && g_w43_trace) g_w43_trace[((int64_t)blockIdx.x * 8 + wave) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

template <int N> struct IC { static constexpr int value = N; };

template <int C, int TAIL>
__global__ __launch_bounds__((W43Cfg<C>::NT), (W43Cfg<C>::NT == 512 ? 1 : 2))
void wino43_trunk_kernel(const W43K p) {
    using K = W43Cfg<C>;
    constexpr int W = K::W, LDT = K::LDT, KG = K::KG, KS = K::KS, TILES = K::TILES, TC = K::TC, NS = K::NS, C4 = K::C4, RP = K::RP;
    constexpr int MI = K::MI, NI = K::NI, WN = K::WN;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, q = lane >> 4;                         // main phase: tile of the group, channel quad of the 16-channel block
    const int ns = wave % NS, tg = wave / NS;                        // main phase: 32-channel slice, 16-tile group

    if (p.stag > 0 && (int)blockIdx.x < p.first_gen) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);       // HW_ID.wave_id
        if (slot & 1) for (int i = 0; i < p.stag; ++i) __builtin_amdgcn_s_sleep(16);
    }
#ifdef W43_TRACE
    if (TAIL == 2 && lane == 0 && g_w43_trace) {
        const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);           // HW_ID[15:0]
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);          // XCC_ID[3:0]
        g_w43_trace[((int64_t)blockIdx.x * 8 + wave) * 64 + 63] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    STAMP(0);
    int tile_m;                                                      // XCD-contiguous order (conv_wino.hip)
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int qq = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile_m = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (bid >> 3);
    }
    const int tpi = p.H / 8;
    const int img = tile_m / tpi;
    const int row0 = 8 * (tile_m - img * tpi);
    const float* const xim = p.t1 + (int64_t)img * p.H * W * C;

    // ---- transform geometry: thread -> channel quad cg, tile column tj, both tile rows ------------------------------------
    const int cg = tid % C4, tj = tid / C4;
    unsigned coff[6];                                                // per-lane byte offsets of the unit's 6 columns
    int roff[2][6];                                                  // uniform float offsets of its 6 rows
#pragma unroll
    for (int j = 0; j < 6; ++j) coff[j] = (unsigned)((((4 * tj - 1 + j) & (W - 1)) * C + 4 * cg) * 4);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int r = row0 + 4 * s - 1 + i;
            r = r < 0 ? r + p.H : (r >= p.H ? r - p.H : r);
            roff[s][i] = r * W * C;
        }

    // One pass's input: per tile row s, column pair cp: rows (0, 2, 4) [KIND 0], (1, 3, 5) [KIND 5] or (1, 2, 3, 4) [KIND 1].
    f32x4 d[2][4][2];                                                // two batches in flight
    f32x4 w[6];                                                      // row-combined columns of the unit in work
    auto issue = [&](auto kind_c, int k) __attribute__((always_inline)) {                           // batch k = 3 s + cp (compile-time after unrolling)
        constexpr int KIND = decltype(kind_c)::value;
        const int s = k / 3, cp = k % 3, bf = k & 1;
#pragma unroll
        for (int i = 0; i < (KIND == 1 ? 4 : 3); ++i) {
            const int row = KIND == 0 ? 2 * i : (KIND == 5 ? 2 * i + 1 : i + 1);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#ifdef W43_DBG_NOLOAD
                d[bf][i][jj] = f32x4{1.f, 2.f, (float)i, (float)tid};
#else
                d[bf][i][jj] = ldg(xim + roff[s][row], coff[2 * cp + jj]);
#endif
        }
    };
    auto combine = [&](auto kind_c, int k, float c1, float c2, float c3) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_c)::value;
        const int cp = k % 3, bf = k & 1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if (KIND == 1) w[2 * cp + jj] = fma4(d[bf][0][jj], c1, fma4(d[bf][1][jj], c2, fma4(d[bf][2][jj], c3, d[bf][3][jj])));
            else w[2 * cp + jj] = fma4(d[bf][0][jj], 4.f, fma4(d[bf][1][jj], -5.f, d[bf][2][jj]));     // B^T rows 0 / 5: [4 0 -5 0 1 0]
        }
    };
    auto columns = [&](int s) __attribute__((always_inline)) {                                      // (w B)[nu] -> V[nu][tile][c]
        float* const dst = lds + (s * TC + tj) * LDT + 4 * cg;
        const f32x4 t1 = fma4(w[2], -4.f, w[4]), t2 = fma4(w[1], -4.f, w[3]);
        const f32x4 t3 = w[4] - w[2], t4 = w[3] - w[1];
        *reinterpret_cast<f32x4*>(dst + 0 * TILES * LDT) = fma4(w[0], 4.f, fma4(w[2], -5.f, w[4]));
        *reinterpret_cast<f32x4*>(dst + 1 * TILES * LDT) = t1 + t2;
        *reinterpret_cast<f32x4*>(dst + 2 * TILES * LDT) = t1 - t2;
        *reinterpret_cast<f32x4*>(dst + 3 * TILES * LDT) = fma4(t4, 2.f, t3);
        *reinterpret_cast<f32x4*>(dst + 4 * TILES * LDT) = fma4(t4, -2.f, t3);
        *reinterpret_cast<f32x4*>(dst + 5 * TILES * LDT) = fma4(w[1], 4.f, fma4(w[3], -5.f, w[5]));
    };

    // weights: this wave's 16 KiB of a position are contiguous ([k-group][block][lane][4]); positions C * C floats apart
    const float* const ub = p.U + ns * (KG * 512);                   // uniform; + 16 lane bytes per lane
    const unsigned wl = 16u * lane;
    const float* const bfrag = lds + (16 * tg + li) * LDT + 4 * q;  // V fragment base, + nu * TILES * LDT + 16 kg

    f32x4 Y[4][4][2];                                                // [a][b][block]: conv2 output (4 a' + a, 4 b' + b) of tile li, 4 channels
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { Y[a][b][0] = f32x4{0.f, 0.f, 0.f, 0.f}; Y[a][b][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // ---- one pass: V_xi -> LDS, 6 GEMMs, fold.  The first two input batches of the pass are already in flight. -----------------
    auto pass = [&](auto kind_c, auto next_c, int xi, float c1, float c2, float c3, float a1, float a2, float a3) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr int NEXT = decltype(next_c)::value;                // kind of the following pass (-1: none)
        __builtin_amdgcn_sched_barrier(0);
#if defined(W43_PRIO) && W43_PRIO == 1
        __builtin_amdgcn_s_setprio(3);                               // experiment: the short load / vector phases ahead of the partner's MFMA stream
#elif defined(W43_PRIO) && W43_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int j = 0; j < 6; ++j) asm volatile("" : "+v"(coff[j]));    // opaque per pass (as uoff below): no hoisted 64-bit address per (row, column)
        if (EARLY < 1) issue(kind_c, 0);                             // (EARLY batches of this pass went out before the previous pass's fold)
#ifdef W43_SERIAL
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (k > 0) issue(kind_c, k);
            combine(kind_c, k, c1, c2, c3);
            if (k % 3 == 2) columns(k / 3);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
        if (EARLY < 2) issue(kind_c, 1);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            combine(kind_c, k, c1, c2, c3);
            if (k % 3 == 2) columns(k / 3);                          // before the next request: w, the column temporaries and TWO batches in flight do not fit
            if (k + 2 < 6) issue(kind_c, k + 2);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);                            // (the weight loads below must not rise into the transform: registers)
        int64_t uoff = (int64_t)xi * 6 * (C * C);
        asm volatile("" : "+s"(uoff));                                // opaque per pass: keeps hipcc from hoisting one address pair per load out of the xi loop
        const float* const ux = ub + uoff;
#define WB(s) (((s) / KG) * (C * C) + 512 * ((s) % KG))
        f32x4 wq[RD][2];
#pragma unroll
        for (int s = 0; s < RD; ++s) {                               // first weight fragments: in flight across the barrier
            wq[s][0] = ldg(ux + WB(s), wl);
            wq[s][1] = ldg(ux + WB(s) + 256, wl);
        }
        STAMP(1 + 5 * xi);
#if defined(W43_PRIO) && W43_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#elif defined(W43_PRIO) && W43_PRIO == 2
        __builtin_amdgcn_s_setprio(3);                               // experiment: the MFMA phase first
#endif
        lds_barrier();                                               // V complete
        STAMP(2 + 5 * xi);
        f32x4 Z[4][2];
#pragma unroll
        for (int b = 0; b < 4; ++b) { Z[b][0] = f32x4{0.f, 0.f, 0.f, 0.f}; Z[b][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        f32x4 acc[2][2];
        f32x4 bq[2];
        bq[0] = *reinterpret_cast<const f32x4*>(bfrag);
#pragma unroll
        for (int nu = 0; nu < 6; ++nu) {
            const int st = nu & 1;
            acc[st][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[st][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kg = 0; kg < KG; ++kg) {
                const int s = KG * nu + kg;
                if (s + 1 < 6 * KG)
                    bq[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(bfrag + ((s + 1) / KG) * TILES * LDT + 16 * ((s + 1) % KG));
                const f32x4 wa = wq[s % RD][0], wb = wq[s % RD][1];
                if (s + RD < 6 * KG) {
                    wq[s % RD][0] = ldg(ux + WB(s + RD), wl);
                    wq[s % RD][1] = ldg(ux + WB(s + RD) + 256, wl);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[st][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], bq[s & 1][j], acc[st][0], 0, 0, 0);   // D[channel][tile]
                    acc[st][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j], bq[s & 1][j], acc[st][1], 0, 0, 0);
#ifdef W43_ALT
                    __builtin_amdgcn_sched_barrier(0);                // keep the two accumulators alternating: hipcc otherwise queues 4 dependent MFMAs
#endif
                }
                __builtin_amdgcn_sched_barrier(0);                    // keep the prefetch distances as written
            }
            // fold over nu: Z[b] += A^T[b][nu] acc,  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
#pragma unroll
            for (int bl = 0; bl < 2; ++bl) {
                if (nu == 0) Z[0][bl] = Z[0][bl] + acc[0][bl];
                if (nu == 2) {                                        // nu = 1 (set 1) and nu = 2 (set 0)
                    const f32x4 sm = acc[1][bl] + acc[0][bl], df = acc[1][bl] - acc[0][bl];
                    Z[0][bl] = Z[0][bl] + sm; Z[2][bl] = Z[2][bl] + sm;
                    Z[1][bl] = Z[1][bl] + df; Z[3][bl] = Z[3][bl] + df;
                }
                if (nu == 4) {                                        // nu = 3 (set 1) and nu = 4 (set 0)
                    const f32x4 sm = acc[1][bl] + acc[0][bl], df = acc[1][bl] - acc[0][bl];
                    Z[0][bl] = Z[0][bl] + sm; Z[2][bl] = fma4(sm, 4.f, Z[2][bl]);
                    Z[1][bl] = fma4(df, 2.f, Z[1][bl]); Z[3][bl] = fma4(df, 8.f, Z[3][bl]);
                }
                if (nu == 5) Z[3][bl] = Z[3][bl] + acc[1][bl];
            }
            if (nu == 0 || nu == 2 || nu == 4)
                asm volatile("" : "+v"(Z[0][0]), "+v"(Z[0][1]), "+v"(Z[1][0]), "+v"(Z[1][1]), "+v"(Z[2][0]), "+v"(Z[2][1]), "+v"(Z[3][0]), "+v"(Z[3][1]));
        }
#undef WB
        STAMP(3 + 5 * xi);
        // the next pass's first input batches go out before the fold over xi and the barrier
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT == 9) {                                   // inside the xi = 1..4 loop: rows (1, 2, 3, 4) of pass xi + 1 <= 4, rows (1, 3, 5, -) of pass 5
#pragma unroll
            for (int k = 0; k < EARLY; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ro = xi < 4 ? roff[0][i + 1] : roff[0][i < 3 ? 2 * i + 1 : 5];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) d[k & 1][i][jj] = ldg(xim + ro, coff[2 * k + jj]);
                }
        } else {
            if constexpr (NEXT >= 0 && EARLY >= 1) issue(next_c, 0);
            if constexpr (NEXT >= 0 && EARLY >= 2) issue(next_c, 1);
        }
        // fold over xi: Y[a][b] += A^T[a][xi] Z[b]
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int bl = 0; bl < 2; ++bl) {
                if (KIND == 0) Y[0][b][bl] = Y[0][b][bl] + Z[b][bl];
                else if (KIND == 5) Y[3][b][bl] = Y[3][b][bl] + Z[b][bl];
                else {
                    Y[0][b][bl] = Y[0][b][bl] + Z[b][bl];
                    Y[1][b][bl] = fma4(Z[b][bl], a1, Y[1][b][bl]);
                    Y[2][b][bl] = fma4(Z[b][bl], a2, Y[2][b][bl]);
                    Y[3][b][bl] = fma4(Z[b][bl], a3, Y[3][b][bl]);
                }
            }
#ifdef W43_PIN_Y
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) asm volatile("" : "+v"(Y[a][b][0]), "+v"(Y[a][b][1]));
#endif
        STAMP(4 + 5 * xi);
        lds_barrier();                                               // every wave is done reading V of this pass
        STAMP(5 + 5 * xi);
    };

    if (EARLY >= 1) issue(IC<0>{}, 0);
    if (EARLY >= 2) issue(IC<0>{}, 1);
    pass(IC<0>{}, IC<1>{}, 0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int xi = 1; xi <= 4; ++xi) {
        // B^T rows 1..4: [0 -4 -4 1 1 0], [0 4 -4 -1 1 0], [0 -2 -1 2 1 0], [0 2 -1 -2 1 0];  A^T columns 1..4: (1,1,1,1), (1,-1,1,-1), (1,2,4,8), (1,-2,4,-8)
        const float sg = (xi & 1) ? 1.f : -1.f;                      // xi = 1, 3: +; 2, 4: -
        const bool lo = xi <= 2;
        const float c1 = lo ? -4.f * sg : -2.f * sg, c2 = lo ? -4.f : -1.f, c3 = lo ? sg : 2.f * sg;
        const float a1 = lo ? sg : 2.f * sg, a2 = lo ? 1.f : 4.f, a3 = lo ? sg : 8.f * sg;
        pass(IC<1>{}, IC<9>{}, xi, c1, c2, c3, a1, a2, a3);
    }
    pass(IC<5>{}, IC<-1>{}, 5, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);

    // ---- tails on two 128-pixel halves: half hf = output rows 2 hf, 2 hf + 1 of every tile = image rows row0 + {0, 1, 4, 5} + 2 hf ----
    float* const T = lds;
    const int li32 = lane & 31, hh = lane >> 5;                      // 32x32x2 layout of the tails
    const int wm = wave / WN, wn = wave % WN;                        // MI * 32 pixels x NI * 32 channels per wave
    const int tj0 = tj;                                              // row-coalesced view: thread (cg, tj0) owns pixels tj0 + RP i of the half
    float* const trow = T + tj0 * LDT + 4 * cg;
    const int ty = (16 * tg + li) / TC, tx = (16 * tg + li) % TC;
    f32x4 bt[RT][NI];
    auto tail_prefetch = [&](const float* __restrict__ wsrc) __attribute__((always_inline)) {
        const float* b0 = wsrc + (wn * NI) * (KS * 256);
#pragma unroll
        for (int u = 0; u < RT; ++u)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bt[u][ni] = ldg(b0 + ni * (KS * 256) + 256 * u, wl);
    };
    f32x16 acc[MI][NI];
    auto gemm_tail = [&](const float* __restrict__ wsrc) __attribute__((always_inline)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        const float* a0 = T + (wm * MI * 32 + li32) * LDT + 4 * hh;
        const float* b0 = wsrc + (wn * NI) * (KS * 256);
        f32x4 a[2][MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * LDT);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[(ks + 1) & 1][mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * LDT + 8 * (ks + 1));
            }
            f32x4 bc[NI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bc[ni] = bt[ks % RT][ni];
            if (ks + RT < KS) {
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bt[ks % RT][ni] = ldg(b0 + ni * (KS * 256) + 256 * (ks + RT), wl);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[ni][r], a[ks & 1][mi][r], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto acc_to_lds = [&]() __attribute__((always_inline)) {                                         // D[channel][pixel] -> T[pixel][channel], 128-bit writes
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[mi][ni][4 * g + e];
                    *reinterpret_cast<f32x4*>(T + (wm * MI * 32 + mi * 32 + li32) * LDT + (wn * NI + ni) * 32 + 8 * g + 4 * hh) = o;
                }
    };

#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        // t2 = ELU(conv2 + b3a) + b3b -> T[pixel][channel]; T row r' <-> image row row0 + 4 (r' >> 1) + 2 hf + (r' & 1)
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int bl = 0; bl < 2; ++bl) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = elu_act(Y[2 * hf + a2][b][bl][e] + p.act_a) + p.act_b;
                    *reinterpret_cast<f32x4*>(T + ((2 * ty + a2) * W + 4 * tx + b) * LDT + 32 * ns + 16 * bl + 4 * q) = o;
                }
        tail_prefetch(p.w3);
        const int64_t pixb = ((int64_t)img * p.H + row0 + 2 * hf) * W + tj0;
        float* xr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[r] = p.xio + (pixb + (int64_t)(4 * (r >> 1) + (r & 1)) * W) * C + 4 * cg;
        // residual rows: requested ahead of conv3 -- in the first half only half of them (the second half's Y is still live: registers)
        f32x4 res[16];
        constexpr int NRES0 = W43_NRES0;
#pragma unroll
        for (int i = 0; i < (hf == 0 ? NRES0 : 16); ++i) res[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr[(RP * i) / W] + ((RP * i) % W) * C));
        STAMP(31 + 8 * hf);
        lds_barrier();                                               // t2 complete
        STAMP(32 + 8 * hf);
        gemm_tail(p.w3);                                             // conv3
        STAMP(33 + 8 * hf);
        if (hf == 0) {
#pragma unroll
            for (int i = NRES0; i < 16; ++i) res[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr[(RP * i) / W] + ((RP * i) % W) * C));
        }
        if (TAIL == 2) tail_prefetch(p.w1n);
        lds_barrier();                                               // every wave is done reading t2
        acc_to_lds();
        lds_barrier();
        STAMP(34 + 8 * hf);
        // out = conv3 * scale + bias4 + x, in place over the residual stream, whole pixel rows per 8th of a workgroup
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4 t = *reinterpret_cast<const f32x4*>(trow + RP * i * LDT);
            t = t * p.t_scale;
            t = t + p.t_b4;
            t = t + res[i];
            __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(xr[(RP * i) / W] + ((RP * i) % W) * C));
            if (TAIL == 2) {                                          // next block's conv1 pre-op, back into T in place
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = elu_act(t[e] + p.n_b1a) + p.n_b1b;
                *reinterpret_cast<f32x4*>(trow + RP * i * LDT) = t;
            }
        }
        STAMP(35 + 8 * hf);
        if constexpr (TAIL == 2) {
            lds_barrier();
            gemm_tail(p.w1n);                                        // next block's conv1
            STAMP(36 + 8 * hf);
            lds_barrier();                                           // every wave is done reading T
            acc_to_lds();
            lds_barrier();
            float* const y0 = p.y2 + (xr[0] - p.xio);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                f32x4 t = *reinterpret_cast<const f32x4*>(trow + RP * i * LDT);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = elu_act(t[e] + p.n_b2a) + p.n_b2b;
                const int r = (RP * i) / W;
                __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(y0 + ((4 * (r >> 1) + (r & 1)) * W + (RP * i) % W) * C));
            }
        }
        STAMP(37 + 8 * hf);
        if (hf == 0) lds_barrier();                                  // T is rewritten by the second half's t2
        STAMP(38 + 8 * hf);
    }
}

// U[pos = 6 xi + nu] = (G g G^T)[xi][nu] for g = w[n][k][3][3], evaluated in fp64 and rounded once;
// G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
// -> [pos][n >> 5][k >> 4][(n >> 4) & 1][lane = ((k >> 2) & 3) * 16 + (n & 15)][k & 3]: what lane (li, q) feeds to MFMA k & 3 of the k-group.
__global__ void wino43_weight_kernel(const float* __restrict__ w, int c, float* __restrict__ U) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;              // over n * c + k
    if (i >= c * c) return;
    const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double g[3][3], t[6][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) g[a][b] = (double)w[(int64_t)i * 9 + a * 3 + b];
    for (int x = 0; x < 6; ++x)
        for (int b = 0; b < 3; ++b) t[x][b] = G[x][0] * g[0][b] + G[x][1] * g[1][b] + G[x][2] * g[2][b];
    const int n = i / c, k = i % c;
    const int64_t fo = ((((int64_t)(n >> 5) * (c / 16) + (k >> 4)) * 2 + ((n >> 4) & 1)) * 64 + ((k >> 2) & 3) * 16 + (n & 15)) * 4 + (k & 3);
    const int64_t cc = (int64_t)c * c;
    for (int x = 0; x < 6; ++x)
        for (int y = 0; y < 6; ++y)
            U[(x * 6 + y) * cc + fo] = (float)(t[x][0] * G[y][0] + t[x][1] * G[y][1] + t[x][2] * G[y][2]);
}

}  // namespace

namespace vqae {

// geometry only; whether a handle uses this form at all is decided when it is created (VQAE_WINO43=0: F(2x2, 3x3) everywhere)
bool wino43_supported(int c, int h, int w, int dtype) {
    // C = 32 on the 128-wide grid compiles and is correct (VQAE_WINO43_C32=1), but that level is bound by vector issue: the 30 % fewer
    // MFMAs buy nothing there (675 us either way), so it keeps F(2x2, 3x3) and its smaller rounding error
    static const bool c32 = getenv("VQAE_WINO43_C32") && atoi(getenv("VQAE_WINO43_C32"));
    const bool cw = (c == 256 && w == 32) || (c == 128 && w == 32) || (c == 64 && w == 64) || (c32 && c == 32 && w == 128);
    return dtype == VQAE_DT_F32 && cw && h >= 8 && h % 8 == 0;
}
bool wino43_enabled() {
    const char* e = getenv("VQAE_WINO43");
    return !(e && !atoi(e));
}

size_t wino43_weight_floats(int c) { return (size_t)36 * c * c; }

// w_oihw_dev [c][c][3][3] (PyTorch layout, device) -> U_dev [36][c][c] (fragment order above)
int wino43_transform_weight(const float* w_oihw_dev, int c, float* U_dev, hipStream_t stream) {
    VQAE_REQUIRE(c == 256 || c == 128 || c == 64 || c == 32, VQAE_ERR_UNSUPPORTED, "wino43_transform_weight: C = %d", c);
    wino43_weight_kernel<<<(unsigned)ceil_div(c * c, 256), 256, 0, stream>>>(w_oihw_dev, c, U_dev);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

template <int C>
static int launch_w43(W43K& k, int64_t M, bool chain, hipStream_t stream) {
    using K = W43Cfg<C>;
    static bool attr_set = false;
    static int pad = 0;
    if (!attr_set) {
#ifdef W43_TRACE
        pad = getenv("VQAE_W43_LDS_PAD") ? atoi(getenv("VQAE_W43_LDS_PAD")) : 0;      // experiment: one workgroup per CU
#endif
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)wino43_trunk_kernel<C, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES + pad));
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)wino43_trunk_kernel<C, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES + pad));
        attr_set = true;
    }
    const unsigned grid = (unsigned)(M / (8 * K::W));
    // executed matrix work: 36 GEMMs of K = C per 16 output pixels (K_eff = 2.25 C per pixel) + the 1x1 tails
    const double flops = 2.0 * (double)M * C * (2.25 * C + C + (chain ? C : 0));
    ProfScope prof(C >= 128 ? PROF_CONV3X3_TRUNK : PROF_NONE, stream, flops);
    if (chain) wino43_trunk_kernel<C, 2><<<grid, K::NT, K::LDS_BYTES + pad, stream>>>(k);
    else wino43_trunk_kernel<C, 1><<<grid, K::NT, K::LDS_BYTES + pad, stream>>>(k);
    prof.done();
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// Same contract as wino_trunk_tail (conv_wino.hip): t1 -> xio in place (+ t1_next); w3 / w1n in that file's fragment order (k-slice 8).
int wino43_trunk_tail(const float* t1, const float* U, const float* w3, float act_a, float act_b, float t_scale, float t_b4,
                      float* xio, const float* w1n, float n_b1a, float n_b1b, float n_b2a, float n_b2b, float* t1_next,
                      int batch, int h, int w, int c, hipStream_t stream) {
    if (batch == 0) return VQAE_OK;
    VQAE_REQUIRE(t1 && U && w3 && xio && (!w1n || t1_next), VQAE_ERR_INVALID, "wino43_trunk_tail: null pointer");
    VQAE_REQUIRE(wino43_supported(c, h, w, VQAE_DT_F32), VQAE_ERR_UNSUPPORTED, "wino43_trunk_tail: C = %d, H = %d, W = %d", c, h, w);
    const int64_t M = (int64_t)batch * h * w;
    VQAE_REQUIRE(M < (1ll << 31) - 1024, VQAE_ERR_UNSUPPORTED, "wino43_trunk_tail: too many pixels");
    W43K k;
    memset(&k, 0, sizeof(k));
    k.t1 = t1; k.U = U; k.w3 = w3; k.w1n = w1n; k.xio = xio; k.y2 = t1_next;
    k.H = h; k.M = (int)M;
    k.act_a = act_a; k.act_b = act_b; k.t_scale = t_scale; k.t_b4 = t_b4;
    k.n_b1a = n_b1a; k.n_b1b = n_b1b; k.n_b2a = n_b2a; k.n_b2b = n_b2b;
    static const int stag = getenv("VQAE_W43_STAG") ? atoi(getenv("VQAE_W43_STAG")) : 0;
    k.stag = stag; k.first_gen = 512;
    switch (c) {
        case 256: return launch_w43<256>(k, M, w1n != nullptr, stream);
        case 128: return launch_w43<128>(k, M, w1n != nullptr, stream);
        case 64: return launch_w43<64>(k, M, w1n != nullptr, stream);
        default: return launch_w43<32>(k, M, w1n != nullptr, stream);
    }
}

}  // namespace vqae

#ifdef W43_TRACE
extern "C" int vqae_debug_w43_trace(void* dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_w43_trace), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : -3;
}
#endif
