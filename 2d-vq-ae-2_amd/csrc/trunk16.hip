// 16-bit-native trunk Fixup block (torch.autocast semantics, BASELINE configs #3 bf16 / #4 fp16): one launch runs
//   conv2 (3x3 circular, conv_block.py:203) -> ELU -> conv3 (1x1) -> * scale + bias4 + residual (conv_block.py:204-214)
//   [-> the NEXT block's bias / ELU / conv1 (1x1) / ELU (conv_block.py:199-202)]
// on v_mfma_f32_32x32x16_{bf16,f16}, for 'same' blocks with C in {64, 128, 256} channels on grids 32 / 64 / 128 wide.
//
// What differs from conv_mfma.hip's TAIL kernel (which this replaces in the 16-bit modes; 277 us -> see DESIGN.md):
//   * t1 / t1_next -- the conv2 operand, i.e. values that autocast has ALREADY rounded to the 16-bit type -- live in HBM
//     as 16-bit (half the bytes, exact); only the residual stream x stays fp32, as it does in the reference (the
//     shape-(1,) fp32 Fixup scalars promote every elementwise op to fp32).
//   * the block input (4 image rows + one wrap-around halo row above and below = 128 output pixels) is staged ONCE in
//     LDS as 16-bit, [pixel][C + 8]; all nine taps read their MFMA operand from it with shifted addresses -- no per-tap
//     re-gather from global, no conversion or ELU inside the K loop, no barrier inside the K loop.
//   * weights are the MFMA *row* operand, streamed from L2 in fragment order ([n-tile][k-step][lane][8]: one wave-wide
//     load = 1 KiB contiguous) through a register ring; a wave owns 32 output channels x all 128 pixels, so every
//     weight fragment is loaded once per workgroup and feeds 4 MFMAs, and the result has the pixel on the lane and four
//     consecutive channels per register quad: every epilogue access is 8 / 16 bytes wide.
//
// Rounding points are those of conv_mfma.hip (= torch.autocast): conv operands and conv outputs RNE to the 16-bit
// type, fp32 accumulation, fp32 scalar bias / ELU / scale / residual arithmetic, no FMA contraction.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
using vqae::elu_act;

template <int DT> struct E16;
template <> struct E16<VQAE_DT_BF16> {
    using elem = __bf16; using x8 = bf16x8; using x4 = bf16x4;
    static __device__ __forceinline__ f32x16 mma(const x8& a, const x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(__bf16)v; }
};
template <> struct E16<VQAE_DT_F16> {
    using elem = _Float16; using x8 = f16x8; using x4 = f16x4;
    static __device__ __forceinline__ f32x16 mma(const x8& a, const x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(_Float16)v; }
};

using vqae::lds_barrier;                 // LDS-only workgroup barrier (common.h): global loads stay in flight across it

struct T16K {
    const void* __restrict__ t1;     // [M][C] 16-bit: round16(ELU(round16(conv1) + b2a) + b2b), the conv2 operand
    const void* __restrict__ w2f;    // conv2 weights, fragment order [C/32][9 * C/16][64][8] 16-bit
    const void* __restrict__ w3f;    // conv3 weights, fragment order [C/32][C/16][64][8]
    const void* __restrict__ w1nf;   // next block's conv1, same order (NEXT)
    float* xio;                      // [M][C] fp32 residual stream, updated in place
    void* t1n;                       // [M][C] 16-bit: the next block's t1 (NEXT)
    int H;                           // image rows (a multiple of the tile's row count)
    float act_a, act_b, t_scale, t_b4, n_b1a, n_b1b, n_b2a, n_b2b;
};

template <int C, int W, int MT> struct T16Cfg {
    static_assert(MT == 2 || MT == 4 || MT == 8 || MT == 16, "m-tiles per workgroup");
    static_assert(W == 32 || W == 64 || W == 128 || W == 256, "grid width");
    static_assert(C == 16 || C == 32 || C == 64 || C == 128 || C == 256, "channels");
    static constexpr int NW = C >= 32 ? C / 32 : 1;   // waves = 32-channel output slices (C = 16: one slice, half of it zero weights)
    static constexpr int NQ = C >= 32 ? 4 : C / 8;    // channel quads of a lane that exist (register quad q <-> channels 8 q + 4 h ..)
    static constexpr int WM = C <= 32 ? (MT < 4 ? MT : 4) : 1;   // wave groups along the pixels: with one channel slice (C <= 32) every m-tile
    static constexpr int MTW = MT / WM;               //   gets its own wave (4x the waves, each 4x shorter); MTW = m-tiles per wave
    static constexpr int NT = NW * WM * 64;           // threads
    // columns a workgroup spans (W / TW column blocks per row).  C = 64 / 128 on grids >= 64 wide: 4 x 32 tiles (a 6 x 34 halo is
    // 1.6x the tile; single-row 1 x 128 tiles re-read 3x) -- with the whole-line epilogues below: -22 % at C = 64, W = 64
    static constexpr int TW = (C >= 64 && C <= 128 && W >= 64) ? 32 : (W < 128 ? W : 128);
    static constexpr int CB = W / TW;
    static constexpr int SEG = TW / 32;               // 32-pixel segments per tile row
    static constexpr int R = MT / SEG;                // image rows per workgroup (MT m-tiles of 32 pixels)
    static_assert(MT % SEG == 0, "whole tile rows per workgroup");
    static constexpr int LW = TW + 2;                 // LDS row: the tile's columns + one (wrap-around) halo column each side
    static constexpr int PS = 2 * C + 16;             // LDS bytes per pixel: odd 16-B slot stride -> conflict-free b128 reads
    static constexpr int KS = C / 16;                 // k-slices per tap
    static constexpr int IMG_BYTES = (R + 2) * LW * PS;
    // C >= 64: the epilogues' global accesses go through a wave-private LDS transpose (32 pixels x this wave's 32 channels)
    // so that every wave-wide access is 1 KiB of whole 128-byte lines instead of 64 row pieces of 16 bytes (the L1 spends
    // 62 % of its cycles waiting on outstanding misses in these kernels, and three quarters of its accesses were such pieces)
    static constexpr bool EPI = C >= 64;              // measured on one box: C = 128 123 -> 114 us, C = 64 245 -> 190 us, C = 256 -2.5 %
    static constexpr int EPI_RS = 144;                // scratch row stride (128 B of fp32 + one 16-B slot: conflict-free both ways)
    static constexpr int EPI_BYTES = EPI ? 32 * EPI_RS : 0;
    static constexpr int LDS_BYTES = IMG_BYTES + NW * WM * EPI_BYTES;
};

// Developer aid (off by default; tools/t16_trace.py): per-phase s_memtime stamps of every wave.
#ifdef VQAE_T16_TRACE
__device__ unsigned long long* g_t16_trace = nullptr;
__device__ int g_t16_dbg = 0;         // experiments: 1 skip the staging loads, 2 skip the residual loads, 4 skip the stores
#define DBG(bit) (g_t16_dbg & (bit))
#define STAMP(i) do { if (lane == 0 && g_t16_trace) g_t16_trace[((int64_t)blockIdx.x * 8 + wv) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#define DBG(bit) false
#endif

constexpr int NB_MAX = 8;                             // weight-fragment ring depth (k-steps in flight per wave): 8, or 6 at C = 64

template <int C, int W, int MT, int DT, bool NEXT>
__global__ __launch_bounds__((T16Cfg<C, W, MT>::NT), 2)
void trunk16_kernel(const T16K p) {
    using K = T16Cfg<C, W, MT>;
    using E = E16<DT>;
    using x8 = typename E::x8;
    using x4 = typename E::x4;
    constexpr int NT = K::NT, SEG = K::SEG, R = K::R, PS = K::PS, KS = K::KS, TW = K::TW, LW = K::LW, CB = K::CB, NQ = K::NQ;
    constexpr int NW = K::NW, MTW = K::MTW;
    constexpr int NS2 = 9 * KS;
    constexpr int NB = (3 * KS) % NB_MAX == 0 ? NB_MAX : ((3 * KS) % 6 == 0 ? 6 : 3);
    extern __shared__ __attribute__((aligned(16))) char lds[];      // A: [(R + 2) x (TW + 2) pixels][PS];  T: [32 MT pixels][PS] over it

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wv % NW;                                       // n-tile: output channels [32 wave, 32 wave + 32)
    const int m0 = (wv / NW) * MTW;                                 // first of this wave's MTW m-tiles
    const int x = lane & 31, h = lane >> 5;

#ifdef VQAE_T16_TRACE
    if (!NEXT && lane == 0 && g_t16_trace) {                        // slot 6 (unused without NEXT): where this workgroup ran
        const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);           // HW_ID[15:0] (cu_id, sh_id, se_id in 15:8)
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);          // XCC_ID[3:0]
        g_t16_trace[((int64_t)blockIdx.x * 8 + wv) * 8 + 6] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    // XCD-contiguous tile order: neighbouring row groups of an image (shared halo rows) meet in one L2 (speed only)
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tiles_per_img = (p.H / R) * CB;
    const int img = tile / tiles_per_img;
    const int t_in = tile - img * tiles_per_img;
    const int y0 = (t_in / CB) * R, x0 = (t_in % CB) * TW;
    const int64_t pix0 = ((int64_t)img * p.H + y0) * W + x0;        // first output pixel of this tile (NHWC pixel index)
    auto moff = [&](int mi) { const int g = m0 + mi; return (g / SEG) * W + (g % SEG) * 32; };   // pixel offset of the wave's m-tile mi from pix0
    STAMP(0);

    // first ring of conv2 weight fragments (L2), requested ahead of the input rows
    // ---- conv2: acc[mi] (32 channels x 32 pixels) += W2[tap] (row operand, from L2) x A[tap-shifted pixels] (LDS) --------
    const char* const w2p = (const char*)p.w2f + ((int64_t)wave * NS2 * 64 + lane) * 16;
    x8 wq[NB];
#pragma unroll
    for (int s = 0; s < NB; ++s) wq[s] = *reinterpret_cast<const x8*>(w2p + s * 1024);
    // ---- stage the (R + 2) x (TW + 2) input pixels (wrap-around halo rows and columns) in LDS, 16-bit, once --------------
    {
        constexpr int CPP = C * 2 / 16;                             // 16-byte chunks per pixel
        constexpr int NCH = (R + 2) * LW * CPP;
        constexpr int PER = (NCH + NT - 1) / NT;
        constexpr int GRP = 13;                                     // loads in flight per thread (bounds the staging registers)
        const char* const src = (const char*)p.t1 + (int64_t)img * p.H * W * C * 2;
#pragma unroll
        for (int i0 = 0; i0 < PER; i0 += GRP) {
            u32x4 v[GRP];
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                if (i0 + i < PER) {
                    int c = tid + (i0 + i) * NT;
                    c = c < NCH ? c : NCH - 1;                      // ragged last sweep: re-read the last chunk
                    const int px = c / CPP, part = c % CPP;
                    const int br = px / LW, lx = px % LW;
                    int iy = y0 - 1 + br;
                    iy = iy < 0 ? iy + p.H : (iy >= p.H ? iy - p.H : iy);
                    const int gx = (x0 + lx - 1) & (W - 1);
                    if (!DBG(1)) v[i] = *reinterpret_cast<const u32x4*>(src + ((int64_t)(iy * W + gx) * C * 2 + part * 16));
                    else v[i] = u32x4{0x3c003c00u + (unsigned)c, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
                }
            }
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                if (i0 + i < PER) {
                    const int c = tid + (i0 + i) * NT;
                    if (NCH % NT == 0 || c < NCH) *reinterpret_cast<u32x4*>(lds + (c / CPP) * PS + (c % CPP) * 16) = v[i];
                }
            }
        }
    }

    int abase[MTW][3];                                              // byte offset of (own m-tile, dx): its tile row + pixel column + lane's k half
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int g = m0 + mi;
            abase[mi][dx] = ((g / SEG) * LW + (g % SEG) * 32 + x + dx) * PS + 16 * h;     // LDS column 0 is image column x0 - 1
        }
    f32x16 acc[MTW];
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
    lds_barrier();
    STAMP(1);
    static_assert((3 * KS) % NB == 0, "ring depth must divide a tap row's k-steps");
    // one tap row (dy) per trip of a real loop, its 3 * KS k-steps unrolled: keeps the scheduling window (and the
    // registers the compiler spends on hoisted LDS reads) bounded.  The ring runs NB steps ahead across trips; the
    // last trip's look-ahead reads NB KiB past this wave's fragments (the next n-tile's, or the buffer's tail pad).
    const char* wrow = w2p;
    const char* arow = lds;
    // Software pipeline, pinned with sched_group_barrier (left alone, hipcc batches the ring's refills and waits for the
    // first of them right after issuing it: one exposed L2 round trip per 8 k-steps): per k-step ONE weight-fragment
    // load (for step s + NB), then 4 x { MFMA of step s, LDS fragment read of step s + 1 }.
    auto load_a = [&](x8 (&dst)[MTW], const char* base, int s) {
        const int dx = s / KS, ks = s % KS;
#pragma unroll
        for (int mi = 0; mi < MTW; ++mi) dst[mi] = *reinterpret_cast<const x8*>(base + abase[mi][dx] + ks * 32);
    };
    x8 af[2][MTW];
    load_a(af[0], arow, 0);
    if constexpr ((3 * KS) % 2 == 0) {
#pragma unroll 1
        for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
            for (int s = 0; s < 3 * KS; ++s) {
                const x8 wc = wq[s % NB];
                wq[s % NB] = *reinterpret_cast<const x8*>(wrow + (s + NB) * 1024);
                if (s + 1 < 3 * KS) load_a(af[(s + 1) & 1], arow, s + 1);
                else if (dy < 2) load_a(af[0], arow + LW * PS, 0);   // 3 * KS is even: the next tap row starts in af[0]
#pragma unroll
                for (int mi = 0; mi < MTW; ++mi) acc[mi] = E::mma(wc, af[s & 1][mi], acc[mi]);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
#pragma unroll
                for (int mi = 0; mi < MTW; ++mi) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
            }
            wrow += 3 * KS * 1024;
            arow += LW * PS;
        }
    } else {                                                        // C = 16: 9 k-steps in all, fully unrolled (static buffer parity)
#pragma unroll
        for (int g = 0; g < 9 * KS; ++g) {
            const x8 wc = wq[g % NB];
            wq[g % NB] = *reinterpret_cast<const x8*>(wrow + (g + NB) * 1024);
            if (g + 1 < 9 * KS) load_a(af[(g + 1) & 1], arow + ((g + 1) / (3 * KS)) * LW * PS, (g + 1) % (3 * KS));
#pragma unroll
            for (int mi = 0; mi < MTW; ++mi) acc[mi] = E::mma(wc, af[g & 1][mi], acc[mi]);
        }
    }
    STAMP(2);

    // result layout: lane = pixel x of m-tile mi, register r = channel 32 wave + (r & 3) + 8 (r >> 2) + 4 h
    const int cbase = wave * 32 + 4 * h;                            // + 8 q + {0..3}
    auto to_T = [&](const f32x4& v, int mi, int q) {                // 4 consecutive channels of one pixel -> T, 16-bit
        *reinterpret_cast<x4*>(lds + ((m0 + mi) * 32 + x) * PS + (cbase + 8 * q) * 2) = __builtin_convertvector(v, x4);
    };
    // 1x1 tails: acc = Wf (C x C, row operand, fragments through the ring w1) x T (K = C).  The ring is filled by
    // w_prefetch well ahead of its gemm (before the barrier / activation work in front of it).
    constexpr int NR = KS < NB ? KS : NB;
    x8 w1[NR];
    auto w_prefetch = [&](const void* wf) -> const char* {
        const char* const wp = (const char*)wf + ((int64_t)wave * KS * 64 + lane) * 16;
#pragma unroll
        for (int s = 0; s < NR; ++s) w1[s] = *reinterpret_cast<const x8*>(wp + s * 1024);
        return wp;
    };
    auto gemm1x1 = [&](const char* wp) {
#pragma unroll
        for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
        lds_barrier();                                            // T complete
        auto load_b = [&](x8 (&dst)[MTW], int s) {
#pragma unroll
            for (int mi = 0; mi < MTW; ++mi) dst[mi] = *reinterpret_cast<const x8*>(lds + ((m0 + mi) * 32 + x) * PS + 16 * h + s * 32);
        };
        x8 bf[2][MTW];
        load_b(bf[0], 0);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const x8 wc = w1[s % NR];
            if (s + NR < KS) w1[s % NR] = *reinterpret_cast<const x8*>(wp + (s + NR) * 1024);
            if (s + 1 < KS) load_b(bf[(s + 1) & 1], s + 1);
#pragma unroll
            for (int mi = 0; mi < MTW; ++mi) acc[mi] = E::mma(wc, bf[s & 1][mi], acc[mi]);
            if (s + NR < KS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
#pragma unroll
            for (int mi = 0; mi < MTW; ++mi) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (s + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
    };

    // ---- t2 = round(ELU(round(conv2) + b3a) + b3b) -> T (over the dead input rows) ------------------------------------
    const char* const wp3 = w_prefetch(p.w3f);                      // in flight across the barrier and the activation work
    lds_barrier();                                                // every wave is done reading A
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = elu_act(E::rnd(acc[mi][4 * q + e]) + p.act_a) + p.act_b;
            to_T(v, mi, q);                                         // the cast is the conv3 input rounding
        }

    // residual rows: requested now, consumed after conv3.
    //   direct form (C < 64): 16 B per lane = 4 consecutive channels of the lane's pixel (a 512-byte-strided row piece);
    //   EPI form: whole lines -- instruction i of an m-tile covers pixels 8 i .. 8 i + 7 x this wave's 128 bytes (lane L:
    //   pixel 8 i + L / 8, 16-byte chunk L % 8) -- and a wave-private LDS transpose turns them into the MFMA layout.
    constexpr bool EPI = K::EPI;
    constexpr int RS = K::EPI_RS;
    char* const S = lds + K::IMG_BYTES + wv * K::EPI_BYTES;         // this wave's transpose scratch (EPI)
    float* const xrow = p.xio + (pix0 + x) * C + cbase;             // direct form
    char* const xlin = (char*)(p.xio + pix0 * C) + ((lane >> 3) * C + wave * 32) * 4 + (lane & 7) * 16;   // EPI form, + 8 i pixels
    f32x4 xr[MTW][4];
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if constexpr (EPI) xr[mi][q] = *reinterpret_cast<const f32x4*>(xlin + ((int64_t)moff(mi) + 8 * q) * (C * 4));
            else if (!DBG(2)) xr[mi][q] = *reinterpret_cast<const f32x4*>(xrow + (int64_t)moff(mi) * C + 8 * q);
            else xr[mi][q] = f32x4{1.f, 2.f, 3.f, (float)lane};
        }

    STAMP(3);
    gemm1x1(wp3);                                                   // conv3
    STAMP(4);
    const char* wp1 = nullptr;
    if (NEXT) wp1 = w_prefetch(p.w1nf);

    // ---- out = round(conv3) * scale + bias4 + x, in place; u = round(ELU(out + b1a') + b1b') for the next conv1 --------
    if (NEXT) lds_barrier();                                      // conv3 finished reading T
    char* const s_lin = S + (lane >> 3) * RS + (lane & 7) * 16;     // scratch address of this lane's line piece (+ 8 i rows)
    char* const s_mma = S + x * RS + h * 16;                        // ... of its MFMA-layout piece (+ 32 q bytes)
    auto finish = [&](int mi, f32x4 (&xv)[4]) {
        if constexpr (EPI) {                                        // line pieces -> scratch -> MFMA layout (LDS runs a wave's accesses in order)
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(s_lin + 8 * q * RS) = xv[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const f32x4*>(s_mma + 32 * q);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4 t, u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = E::rnd(acc[mi][4 * q + e]) * p.t_scale;
                v = v + p.t_b4;
                v = v + xv[q][e];
                t[e] = v;
                u[e] = elu_act(v + p.n_b1a) + p.n_b1b;
            }
            if constexpr (EPI) *reinterpret_cast<f32x4*>(s_mma + 32 * q) = t;
            else if (!DBG(4) || t[0] == 12345.f) *reinterpret_cast<f32x4*>(xrow + (int64_t)moff(mi) * C + 8 * q) = t;
            if (NEXT) to_T(u, mi, q);
        }
        if constexpr (EPI) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<f32x4*>(xlin + ((int64_t)moff(mi) + 8 * q) * (C * 4)) = *reinterpret_cast<const f32x4*>(s_lin + 8 * q * RS);
        }
    };
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi) finish(mi, xr[mi]);
    STAMP(5);
    if constexpr (NEXT) {
        gemm1x1(wp1);                                               // the next block's conv1
        STAMP(6);
        typename E::elem* const trow = (typename E::elem*)p.t1n + (pix0 + x) * C + cbase;
        // EPI form: 16-bit rows of this wave are 64 bytes: instruction i covers pixels 16 i .. 16 i + 15 (lane L: pixel 16 i + L / 4, chunk L % 4)
        char* const tlin = (char*)((typename E::elem*)p.t1n + pix0 * C) + ((lane >> 2) * C + wave * 32) * 2 + (lane & 3) * 16;
        char* const s_tl = S + (lane >> 2) * 80 + (lane & 3) * 16;  // 80-byte scratch rows for the 16-bit tile
        char* const s_tm = S + x * 80 + h * 8;
#pragma unroll
        for (int mi = 0; mi < MTW; ++mi) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = elu_act(E::rnd(acc[mi][4 * q + e]) + p.n_b2a) + p.n_b2b;
                if constexpr (EPI) *reinterpret_cast<x4*>(s_tm + 16 * q) = __builtin_convertvector(v, x4);
                else if (!DBG(4) || v[0] == 12345.f) *reinterpret_cast<x4*>(trow + (int64_t)moff(mi) * C + 8 * q) = __builtin_convertvector(v, x4);
            }
            if constexpr (EPI) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    *reinterpret_cast<u32x4*>(tlin + ((int64_t)moff(mi) + 16 * i) * (C * 2)) = *reinterpret_cast<const u32x4*>(s_tl + 16 * i * 80);
            }
        }
    }
    STAMP(7);
}

// ------------------------------------------------------------------------------------------------------------------
// packed fp32 [C n][taps * C] (tap-major K, vqae_conv_pack_weight_f32; already rounded to the 16-bit type) ->
// fragment order [C/32 n-tiles][taps * C/16 k-steps][64 lanes][8]: lane (r, h) of k-step (tap, ks) holds
// w[n = 32 nt + r][tap][k = 16 ks + 8 h + j], j = 0..7 -- the MFMA row-operand fragment of that step.
template <typename EL>
__global__ void pack16_kernel(const float* __restrict__ w, int c, int taps, EL* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int ks_n = c / 16;
    if (i >= (int64_t)(c < 32 ? 32 : c) * taps * c) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const int64_t st = i >> 9;
    const int s = (int)(st % (taps * ks_n)), nt = (int)(st / (taps * ks_n));
    const int tap = s / ks_n, ks = s % ks_n;
    const int n = nt * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + j;
    out[i] = (EL)w[(int64_t)n * taps * c + tap * c + k];              // rows >= c (c = 16) are the packed layout's zero padding
}

// conv1 of the block at the HEAD of a chain (the others get theirs from the previous block's launch), 16-bit modes:
//   t1 = round16(ELU(round16(conv1x1(round16(ELU(x + b1a) + b1b))) + b2a) + b2b),   x fp32 [M][C] -> t1 16-bit [M][C]
// A streaming launch (4 + 2 bytes per channel and pixel): no LDS tile, a wave takes G groups of 32 consecutive pixels per
// trip of a grid-stride loop -- lane (pixel, k half) loads its 8 channels per k-step straight from global, the weights
// (fragment order, <= 32 KiB; C <= 128) are the MFMA row operand from LDS, and the lane stores the 4 consecutive
// channels of each register quad as 8 bytes.  The next trip's rows are requested before this trip's MFMAs.
// (Before: the generic fp32 conv launch + a separate rounding pass -- 0.62 + 0.27 ms at C = 16 on 256 x 256, batch 256.)
// OUT32: the fp32 value ELU(round16(conv1) + b2a) + b2b is stored as it is ('up' blocks: a bicubic resize, not a conv, reads it)
template <int C, int DT, bool OUT32>
__global__ __launch_bounds__(256, 2)
void head16_kernel(const float* __restrict__ x, const void* __restrict__ w1f, float b1a, float b1b, float b2a, float b2b,
                   void* __restrict__ t1, int n_groups) {
    using E = E16<DT>;
    using x8 = typename E::x8;
    using x4 = typename E::x4;
    constexpr int KU = C / 16, NT = C < 32 ? 1 : C / 32, NQ = C >= 32 ? 4 : C / 8;
    constexpr int G = C <= 16 ? 4 : (C <= 32 ? 2 : 1);             // 32-pixel groups per wave and trip (8+ x 16 B per lane in flight)
    constexpr bool WLDS = true;                                     // weights staged in LDS (<= 32 KiB)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int li = lane & 31, hh = lane >> 5;
    if constexpr (WLDS) {
        for (int i = tid; i < NT * KU * 64; i += 256) reinterpret_cast<u32x4*>(lds)[i] = reinterpret_cast<const u32x4*>(w1f)[i];
        __syncthreads();
    }
    auto wfrag = [&](int ct, int u) -> x8 {
        if constexpr (WLDS) return *reinterpret_cast<const x8*>(lds + ((ct * KU + u) * 64 + lane) * 16);
        else return *reinterpret_cast<const x8*>((const char*)w1f + ((int64_t)(ct * KU + u) * 64 + lane) * 16);
    };
    const int wave_g = (blockIdx.x * 4 + (tid >> 6)) * G;           // first group of this wave's first trip
    const int stride = gridDim.x * 4 * G;
    f32x4 xin[G][KU][2];
    auto load = [&](int g0) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int g = g0 + i < n_groups ? g0 + i : n_groups - 1;
            const float* src = x + ((int64_t)g * 32 + li) * C + 8 * hh;
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                xin[i][u][0] = *reinterpret_cast<const f32x4*>(src + 16 * u);
                xin[i][u][1] = *reinterpret_cast<const f32x4*>(src + 16 * u + 4);
            }
        }
    };
    constexpr bool AHEAD = C <= 64;                                 // wider rows: the register budget goes to one trip's operands
    if (AHEAD && wave_g < n_groups) load(wave_g);
    for (int g0 = wave_g; g0 < n_groups; g0 += stride) {
        if (!AHEAD) load(g0);
        x8 xa[G][KU];
#pragma unroll
        for (int i = 0; i < G; ++i)
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                f32x8 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = elu_act(xin[i][u][0][e] + b1a) + b1b;
                    v[4 + e] = elu_act(xin[i][u][1][e] + b1a) + b1b;
                }
                xa[i][u] = __builtin_convertvector(v, x8);          // conv1 input cast
            }
        if (AHEAD && g0 + stride < n_groups) load(g0 + stride);     // next trip's rows, under this trip's MFMAs and stores
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (g0 + i >= n_groups) break;
            typename E::elem* const dst = (typename E::elem*)t1 + ((int64_t)(g0 + i) * 32 + li) * C + 4 * hh;
            float* const dst32 = (float*)t1 + ((int64_t)(g0 + i) * 32 + li) * C + 4 * hh;
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int u = 0; u < KU; ++u) acc = E::mma(wfrag(ct, u), xa[i][u], acc);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = elu_act(E::rnd(acc[4 * q + e]) + b2a) + b2b;   // conv1 output cast
                    if constexpr (OUT32) *reinterpret_cast<f32x4*>(dst32 + 32 * ct + 8 * q) = o;
                    else *reinterpret_cast<x4*>(dst + 32 * ct + 8 * q) = __builtin_convertvector(o, x4);   // conv2 input cast
                }
            }
        }
    }
}

template <typename EL>
__global__ void round_pack16_kernel(const float* __restrict__ src, EL* __restrict__ dst, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    typedef EL el4 __attribute__((ext_vector_type(4)));
    reinterpret_cast<el4*>(dst)[i] = __builtin_convertvector(v, el4);
}

template <int C, int W, int MT, int DT>
int launch_t16(const T16K& k, bool next, int64_t n_px, hipStream_t stream) {
    using K = T16Cfg<C, W, MT>;
    static bool attr_set = false;
    static int pad = 0;
    if (!attr_set) {
#ifdef VQAE_T16_TRACE
        pad = getenv("VQAE_T16_LDS_PAD") ? atoi(getenv("VQAE_T16_LDS_PAD")) : 0;      // experiment: fewer workgroups per CU
#endif
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)trunk16_kernel<C, W, MT, DT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES + pad));
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)trunk16_kernel<C, W, MT, DT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES + pad));
        attr_set = true;
    }
    const int n_tiles = (int)(n_px / (32 * MT));
    vqae::ProfScope prof(C >= 128 && W == 32 ? vqae::PROF_CONV3X3_TRUNK : 0, stream, 2.0 * (double)n_px * C * (9.0 * C + C + (next ? C : 0)));
    if (next) trunk16_kernel<C, W, MT, DT, true><<<n_tiles, K::NT, K::LDS_BYTES + pad, stream>>>(k);
    else trunk16_kernel<C, W, MT, DT, false><<<n_tiles, K::NT, K::LDS_BYTES + pad, stream>>>(k);
    prof.done();
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

template <int DT>
int launch_t16_cw(const T16K& k, bool next, int c, int w, int64_t n_px, hipStream_t stream) {
    const int h_rows = k.H;
    // 4 m-tiles (128 pixels) per workgroup everywhere.  2 m-tiles (twice the resident workgroups) measured slower at
    // C = 128: every weight fragment then feeds 2 MFMAs instead of 4 and the kernel becomes bound by the L1 address path.
    if (c == 128 && w == 32) return launch_t16<128, 32, 4, DT>(k, next, n_px, stream);
    if (c == 256 && w == 32) return launch_t16<256, 32, 4, DT>(k, next, n_px, stream);
    if (c == 64 && w == 64) return launch_t16<64, 64, 4, DT>(k, next, n_px, stream);
    if (c == 128 && w == 64) return launch_t16<128, 64, 4, DT>(k, next, n_px, stream);
    if (c == 64 && w == 128) return launch_t16<64, 128, 4, DT>(k, next, n_px, stream);
    // C <= 32: 16 m-tiles (512 pixels, 4 per wave) per workgroup where the rows allow it -- these levels are bound by the
    // latency chain of a tile (stage -> conv2 -> conv3 -> conv1'), so more pixels per wave in flight is what pays
    // (cfg A bf16 +3.3 %, cfg B f16 encode +2.9 % over 4 m-tiles; 8: +2.2 %).  VQAE_T16_MT = 4 / 8 / 16 overrides.
    static const int mt_small = getenv("VQAE_T16_MT") ? atoi(getenv("VQAE_T16_MT")) : 16;
    if (mt_small == 16 && h_rows % 4 == 0) {
        if (c == 32 && w == 128) return launch_t16<32, 128, 16, DT>(k, next, n_px, stream);
        if (c == 32 && w == 256) return launch_t16<32, 256, 16, DT>(k, next, n_px, stream);
        if (c == 16 && w == 128) return launch_t16<16, 128, 16, DT>(k, next, n_px, stream);
        if (c == 16 && w == 256) return launch_t16<16, 256, 16, DT>(k, next, n_px, stream);
    }
    if (mt_small == 8 && h_rows % 2 == 0) {
        if (c == 32 && w == 128) return launch_t16<32, 128, 8, DT>(k, next, n_px, stream);
        if (c == 32 && w == 256) return launch_t16<32, 256, 8, DT>(k, next, n_px, stream);
        if (c == 16 && w == 128) return launch_t16<16, 128, 8, DT>(k, next, n_px, stream);
        if (c == 16 && w == 256) return launch_t16<16, 256, 8, DT>(k, next, n_px, stream);
    }
    if (c == 32 && w == 128) return launch_t16<32, 128, 4, DT>(k, next, n_px, stream);
    if (c == 32 && w == 256) return launch_t16<32, 256, 4, DT>(k, next, n_px, stream);
    if (c == 16 && w == 128) return launch_t16<16, 128, 4, DT>(k, next, n_px, stream);
    if (c == 16 && w == 256) return launch_t16<16, 256, 4, DT>(k, next, n_px, stream);
    return vqae::fail(VQAE_ERR_UNSUPPORTED, "trunk16: C = %d on a %d-wide grid", c, w);
}

}  // namespace

namespace vqae {

// (C, grid width) pairs with a kernel: the trunk of cfg A / B (128 @ 32), cfg C (256 @ 32) and the levels above them
bool trunk16_supported(int c, int h, int w, int dtype) {
    static const bool off = getenv("VQAE_NO_TRUNK16") && atoi(getenv("VQAE_NO_TRUNK16"));
    if (off || (dtype != VQAE_DT_BF16 && dtype != VQAE_DT_F16)) return false;
    const bool cw = (c == 128 && w == 32) || (c == 256 && w == 32) || (c == 64 && w == 64) || (c == 128 && w == 64) ||
                    (c == 64 && w == 128) || (c == 32 && w == 128) || (c == 32 && w == 256) || (c == 16 && w == 128) || (c == 16 && w == 256);
    const int tw = (c >= 64 && c <= 128 && w >= 64) ? 32 : (w < 128 ? w : 128);       // T16Cfg::TW
    return cw && h >= 1 && h % (128 / tw) == 0;
}

// + the ring's look-ahead past the last n-tile's fragments (trunk16_kernel reads, never uses, NB KiB beyond them)
size_t trunk16_weight_bytes(int c, int taps) { return (size_t)(c < 32 ? 32 : c) * c * taps * 2 + (size_t)(NB_MAX + 1) * 1024; }

// packed (vqae_conv_pack_weight_f32, rounded) fp32 weights [c][taps * c] on the device -> 16-bit fragment order
int trunk16_pack_weight(const float* w_packed_dev, int c, int taps, int dtype, void* out_dev, hipStream_t stream) {
    VQAE_REQUIRE(c % 16 == 0 && (taps == 1 || taps == 9), VQAE_ERR_INVALID, "trunk16_pack_weight: c %d taps %d", c, taps);
    const int64_t n = (int64_t)(c < 32 ? 32 : c) * c * taps;
    if (dtype == VQAE_DT_BF16) pack16_kernel<__bf16><<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(w_packed_dev, c, taps, (__bf16*)out_dev);
    else pack16_kernel<_Float16><<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(w_packed_dev, c, taps, (_Float16*)out_dev);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// chain-head conv1 (head16_kernel): x fp32 [M][c] -> t1 16-bit [M][c] (out32: fp32, not rounded after the activation); w1f from trunk16_pack_weight(.., taps = 1); M % 32 == 0
bool trunk16_head_supported(int c, int64_t m, int dtype) {
    static const bool off = getenv("VQAE_NO_T16_HEAD") && atoi(getenv("VQAE_NO_T16_HEAD"));
    if (off || (dtype != VQAE_DT_BF16 && dtype != VQAE_DT_F16)) return false;
    return (c == 16 || c == 32 || c == 64 || c == 128) && m > 0 && m % 32 == 0 && m / 32 < (1ll << 31);   // C = 256: generic conv + cast
}

template <int C, int DT>
static int launch_head16(const float* x, const void* w1f, float b1a, float b1b, float b2a, float b2b, void* t1, int64_t m,
                         bool out32, hipStream_t stream) {
    constexpr int KU = C / 16, NT = C < 32 ? 1 : C / 32, G = C <= 16 ? 4 : (C <= 32 ? 2 : 1);
    constexpr int lds_bytes = C <= 128 ? NT * KU * 1024 : 0;
    const int n_groups = (int)(m / 32);
    const int64_t wgs = ceil_div(n_groups, 4 * G);
    const unsigned grid = (unsigned)(wgs < 2048 ? wgs : 2048);      // 8 workgroups per CU; the rest by grid stride
    if (out32) head16_kernel<C, DT, true><<<grid, 256, lds_bytes, stream>>>(x, w1f, b1a, b1b, b2a, b2b, t1, n_groups);
    else head16_kernel<C, DT, false><<<grid, 256, lds_bytes, stream>>>(x, w1f, b1a, b1b, b2a, b2b, t1, n_groups);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

int trunk16_head(const float* x, const void* w1f, float b1a, float b1b, float b2a, float b2b, void* t1, int64_t m, int c,
                 int dtype, bool out32, hipStream_t stream) {
    VQAE_REQUIRE(x && w1f && t1, VQAE_ERR_INVALID, "trunk16_head: null pointer");
    VQAE_REQUIRE(trunk16_head_supported(c, m, dtype), VQAE_ERR_UNSUPPORTED, "trunk16_head: C = %d, M = %lld, dtype %d", c, (long long)m, dtype);
#define VQAE_H16(C_) (dtype == VQAE_DT_BF16 ? launch_head16<C_, VQAE_DT_BF16>(x, w1f, b1a, b1b, b2a, b2b, t1, m, out32, stream) \
                                            : launch_head16<C_, VQAE_DT_F16>(x, w1f, b1a, b1b, b2a, b2b, t1, m, out32, stream))
    switch (c) {
        case 16: return VQAE_H16(16);
        case 32: return VQAE_H16(32);
        case 64: return VQAE_H16(64);
        default: return VQAE_H16(128);
    }
#undef VQAE_H16
}

// fp32 [n] (n % 4 == 0) -> 16-bit, RNE: t1 of a chain head produced by the generic conv1 launch
int trunk16_round_pack(const float* src, void* dst, int64_t n, int dtype, hipStream_t stream) {
    VQAE_REQUIRE(n % 4 == 0, VQAE_ERR_INVALID, "trunk16_round_pack: n %% 4");
    const int64_t n4 = n / 4;
    if (dtype == VQAE_DT_BF16) round_pack16_kernel<__bf16><<<(unsigned)ceil_div(n4, 256), 256, 0, stream>>>(src, (__bf16*)dst, n4);
    else round_pack16_kernel<_Float16><<<(unsigned)ceil_div(n4, 256), 256, 0, stream>>>(src, (_Float16*)dst, n4);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// One trunk Fixup block: t1 (16-bit) -> xio updated in place (+ t1_next, 16-bit, when w1nf != null).
int trunk16_block(const void* t1, const void* w2f, const void* w3f, float act_a, float act_b, float t_scale, float t_b4,
                  float* xio, const void* w1nf, float n_b1a, float n_b1b, float n_b2a, float n_b2b, void* t1_next,
                  int batch, int h, int w, int c, int dtype, hipStream_t stream) {
    if (batch == 0) return VQAE_OK;
    VQAE_REQUIRE(t1 && w2f && w3f && xio && (!w1nf || t1_next), VQAE_ERR_INVALID, "trunk16_block: null pointer");
    VQAE_REQUIRE(trunk16_supported(c, h, w, dtype), VQAE_ERR_UNSUPPORTED, "trunk16_block: C = %d, H = %d, W = %d, dtype %d", c, h, w, dtype);
    const int64_t M = (int64_t)batch * h * w;
    VQAE_REQUIRE(M / 128 < (1ll << 31) - 8, VQAE_ERR_UNSUPPORTED, "trunk16_block: too many pixels");
    T16K k;
    k.t1 = t1; k.w2f = w2f; k.w3f = w3f; k.w1nf = w1nf; k.xio = xio; k.t1n = t1_next; k.H = h;
    k.act_a = act_a; k.act_b = act_b; k.t_scale = t_scale; k.t_b4 = t_b4;
    k.n_b1a = n_b1a; k.n_b1b = n_b1b; k.n_b2a = n_b2a; k.n_b2b = n_b2b;
    if (dtype == VQAE_DT_BF16) return launch_t16_cw<VQAE_DT_BF16>(k, w1nf != nullptr, c, w, M, stream);
    return launch_t16_cw<VQAE_DT_F16>(k, w1nf != nullptr, c, w, M, stream);
}

}  // namespace vqae

#ifdef VQAE_T16_TRACE
extern "C" int vqae_debug_t16_dbg(int bits) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_t16_dbg), &bits, sizeof(bits)) == hipSuccess ? 0 : -3;
}
extern "C" int vqae_debug_t16_trace(void* dev_buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_t16_trace), &dev_buf, sizeof(dev_buf)) == hipSuccess ? 0 : -3;
}
#endif
