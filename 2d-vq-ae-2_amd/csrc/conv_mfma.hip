// fp32 implicit-GEMM convolution on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces the torch.nn.Conv2d call sites of PreActFixupResBlock.forward
// (reference vq_ae/layers/conv_block.py:196-216): 1x1 "proj2d", 3x3 circular "same2d",
// 2x2/stride-2 "down2d", with the Fixup scalar-bias / ELU pre-op fused into the operand load and the
// scale / bias / residual / next-conv-pre-activation fused into the epilogue.
//
// GEMM view: M = B*Ho*Wo output pixels (NHWC, so the GEMM row index IS the output pixel index),
// N = Cout, K = taps*Cin with k = tap*Cin + ci.  One 256-thread workgroup (4 waves, one per SIMD,
// two workgroups resident per CU) owns a 128 x NT output tile; every K-step stages a 128 x KC
// activation tile (gathered per tap: circular wrap / stride / zero pad resolved per pixel) and an
// NT x KC weight tile through LDS (register prefetch of step s+1 under the MFMAs of step s, two LDS
// buffers, one barrier per step).  fp32 MFMA issues one 32x32x2 every 64 cycles per SIMD, so operand
// staging is almost free; the kernel is bound by the matrix pipe (157 TFLOP/s dense fp32 peak).
//
// LDS tiles are [row][KC + 4] floats: the +4 (one 16-B slot, odd slot stride) makes the
// ds_read_b128 fragment reads conflict-free (MI355X_MICROARCH.md §LDS).  Lane (i = l&31, h = l>>5)
// reads 4 consecutive k at 8u + 4h and feeds them to 4 consecutive MFMAs; A and B use the same k
// permutation, which the sum over k does not see.
#include "common.h"
#include <cstdlib>

namespace vqae {
bool conv_small_k_supported(const vqae_conv_args* a);
int conv_small_k(const vqae_conv_args* a, const float* x, const float* w, const float* bias_vec, const float* residual,
                 float* y, hipStream_t stream);
}  // namespace vqae

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvK {
    const float* __restrict__ x;
    const float* __restrict__ w;         // packed [Npad][Ktot]
    const float* __restrict__ bias_vec;  // [Cout] or null
    const float* residual;               // [M][Cout] or null; may alias y (in-place residual add)
    float* y;                            // [M][Cout]
    int B, H, W, Cin, Cout, Ho, Wo, ks, stride, pad, pad_mode;
    int M, Ktot, n_chunks, n_steps;
    int pre_mode;
    float pre_a, pre_b;
    int has_scale, has_bias_s, has_act;
    float scale, bias_s, act_a, act_b;
    // fused tail (trunk, C = 128): conv3 of this block [+ conv1 of the next block], see TAIL below
    const float* __restrict__ w3;        // [C][C] in MFMA fragment order for this engine (vqae::wino_frag_weight, k-slice SK)
    const float* __restrict__ w1n;       // same, the NEXT block's conv1 (TAIL == 2)
    float* y2;                           // [M][128]: next block's t1 (TAIL == 2)
    float t_scale, t_b4, n_b1a, n_b1b, n_b2a, n_b2b;
    const float* __restrict__ gate;      // [B][Cin] per-image channel gate (PRE == VQAE_PRE_CHANNEL_GATE; MBConv SE)
    int dt;                              // VQAE_DT_*: autocast rounding points
    int m16;                             // 16-bit MFMA engine (dt != 0, cin % 32 == 0, no zero padding)
};

using vqae::elu_act;                     // ELU(alpha = 1), branch-free (common.h)

// ------------------------------------------------------------------------------------------------
// fp32 MFMA and VALU instructions do not co-execute on gfx950 (SQ_VALU_MFMA_COEXEC_CYCLES = 0 for this
// kernel; the f32 MFMA runs on the fp32 vector datapath), so every VALU instruction in the K loop is
// paid in matrix-pipe time.  The loop is therefore written to issue almost none:
//   * all per-pixel gather geometry (wrapped / clamped row and column offsets of every tap, zero-pad
//     bits) is computed once per workgroup into registers; a K-step costs no address arithmetic --
//     loads use a wave-uniform base (SGPR: image base + channel chunk) plus a 32-bit lane offset that
//     only changes when the tap changes (one add per pixel per tap);
//   * the loop is unrolled over the two LDS buffers so every ds_read / ds_write address is
//     base VGPR + immediate;
//   * the epilogue addresses through buffer descriptors (hardware range check drops the M tail).
// ------------------------------------------------------------------------------------------------

// ---- matrix-engine policy: exact-fp32 MFMA (32x32x2, 4 per 8-deep k-slice) or 16-bit MFMA (32x32x16, one per
// 16-deep k-slice; autocast modes only).  Both use the same 32x32 C/D layout and the same LDS image idea:
// [row][KC + PAD] elements, lane (i = l&31, h = l>>5) reads SK/2 consecutive k at SK*u + (SK/2)*h.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int DT, bool M16> struct MP {               // fp32 engine (any DT: DT only adds rounding points)
    using elem = float;
    using frag = f32x4;
    static constexpr int SK = 8, PAD = 4;
    static __device__ __forceinline__ void mma(f32x16& acc, const frag& a, const frag& b) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b[r], acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void store4(elem* d, f32x4 v) { *reinterpret_cast<f32x4*>(d) = v; }
    static __device__ __forceinline__ frag load_b(const float* q) { return *reinterpret_cast<const f32x4*>(q); }
    static __device__ __forceinline__ elem cvt(float v) { return v; }
};
template <> struct MP<VQAE_DT_BF16, true> {
    using elem = __bf16;
    using frag = bf16x8;
    static constexpr int SK = 16, PAD = 8;
    static __device__ __forceinline__ void mma(f32x16& acc, const frag& a, const frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void store4(elem* d, f32x4 v) { *reinterpret_cast<bf16x4*>(d) = __builtin_convertvector(v, bf16x4); }
    static __device__ __forceinline__ frag load_b(const float* q) {
        const bf16x4 lo = __builtin_convertvector(*reinterpret_cast<const f32x4*>(q), bf16x4);
        const bf16x4 hi = __builtin_convertvector(*reinterpret_cast<const f32x4*>(q + 4), bf16x4);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    static __device__ __forceinline__ elem cvt(float v) { return (__bf16)v; }
};
template <> struct MP<VQAE_DT_F16, true> {
    using elem = _Float16;
    using frag = f16x8;
    static constexpr int SK = 16, PAD = 8;
    static __device__ __forceinline__ void mma(f32x16& acc, const frag& a, const frag& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    static __device__ __forceinline__ void store4(elem* d, f32x4 v) { *reinterpret_cast<f16x4*>(d) = __builtin_convertvector(v, f16x4); }
    static __device__ __forceinline__ frag load_b(const float* q) {
        const f16x4 lo = __builtin_convertvector(*reinterpret_cast<const f32x4*>(q), f16x4);
        const f16x4 hi = __builtin_convertvector(*reinterpret_cast<const f32x4*>(q + 4), f16x4);
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    static __device__ __forceinline__ elem cvt(float v) { return (_Float16)v; }
};
template <int DT> __device__ __forceinline__ float round_ct(float v) {
    if (DT == VQAE_DT_BF16) return (float)(__bf16)v;
    if (DT == VQAE_DT_F16) return (float)(_Float16)v;
    return v;
}

// TAIL (KC = 32, Cin = Cout = NT in {64, 128}): after the 3x3 conv2 of a trunk Fixup block the same
// workgroup also runs   TAIL >= 1: conv3 (1x1) + scale/bias4 + residual  -> block output (in place over x)
//                       TAIL == 2: conv1 (1x1) of the NEXT block on that output tile -> its t1.
// The 128 x C activation tile goes accumulator -> LDS ([128][C + 4], over the dead staging buffers) ->
// A fragments; the C x C weight matrices (<= 64 KB) are read as B fragments straight from L2.  This removes the
// two HBM-bound 1x1 launches per trunk block (4 of the 7 activation passes) with no halo recompute.
template <int NT, int KC, int PRE, bool PADZ, int TAIL, int DT, bool M16>
__global__ __launch_bounds__(256, 2)
void conv_mfma_kernel(const ConvK p) {
    // DT: autocast rounding points compiled in (the fp32 build has none); M16: 16-bit MFMA engine
    using P = MP<DT, M16>;
    using elem = typename P::elem;
    using frag = typename P::frag;
    constexpr bool R16 = DT != VQAE_DT_F32;
    auto rnd = [&](float v) { return round_ct<DT>(v); };
    constexpr int SK = P::SK;                        // k-slice per fragment read
    constexpr int LDR = KC + P::PAD;                 // LDS row stride (elements)
    constexpr int WN = (NT == 128) ? 2 : 1;          // waves along N
    constexpr int WM = 4 / WN;                       // waves along M
    constexpr int MI = 128 / (WM * 32);              // 32-row MFMA tiles per wave along M
    constexpr int NI = NT / (WN * 32);               // along N
    constexpr int A_PT = KC / 8;                     // float4 per thread for the A tile
    constexpr int B_F4 = NT * KC / 4;                // float4 in the B tile
    constexpr int B_PT = (B_F4 + 255) / 256;
    constexpr int C4 = KC / 4;                       // float4 per tile row
    constexpr int A_BUF = 128 * LDR;                 // elements per A buffer
    constexpr int B_BUF = NT * LDR;

    extern __shared__ __attribute__((aligned(16))) float lds[];   // A0 A1 B0 B1
    elem* const Abuf0 = reinterpret_cast<elem*>(lds);
    elem* const Bbuf0 = Abuf0 + 2 * A_BUF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs; give each XCD a
    // contiguous range of M tiles so 3x3 halo rows are shared in one L2 (speed only).
    int tile_m;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile_m = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int m0 = tile_m * 128;
    const int n0 = blockIdx.y * NT;

    // ---- one-time gather geometry ----------------------------------------------------------------
    const int hw_o = p.Ho * p.Wo;
    const int img0 = m0 / hw_o;                                              // wave-uniform
    const float* const xb = p.x + (int64_t)img0 * p.H * p.W * p.Cin;         // uniform base of this tile's images
    const bool circ = p.pad_mode == VQAE_PAD_CIRCULAR;
    const int a_c4 = tid % C4;
    int yo0[A_PT], yo1[A_PT], yo2[A_PT], xo0[A_PT], xo1[A_PT], xo2[A_PT];   // element offsets (relative to xb) per tap row / column
    unsigned zbits[A_PT];                            // bit (3*dy + dx): tap lies in the zero padding
    int a_wr[A_PT];                                  // LDS write offset (floats) of this thread's A pieces
    int g_off[A_PT];                                 // PRE == CHANNEL_GATE: offset of this pixel's image row in p.gate
#pragma unroll
    for (int i = 0; i < A_PT; ++i) {
        const int row = tid / C4 + i * (256 / C4);
        a_wr[i] = row * LDR + a_c4 * 4;
        int m = m0 + row;
        m = m < p.M ? m : p.M - 1;                   // tail rows re-gather the last pixel; their stores are dropped
        const int b = m / hw_o, rem = m - b * hw_o;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int imgoff = (b - img0) * p.H * p.W * p.Cin;
        g_off[i] = b * p.Cin + a_c4 * 4;
        unsigned zy = 0, zx = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int iy0 = oy * p.stride + d - p.pad, ix0 = ox * p.stride + d - p.pad;
            const int wy = iy0 < 0 ? iy0 + p.H : (iy0 >= p.H ? iy0 - p.H : iy0);
            const int wx = ix0 < 0 ? ix0 + p.W : (ix0 >= p.W ? ix0 - p.W : ix0);
            const int cy = iy0 < 0 ? 0 : (iy0 >= p.H ? p.H - 1 : iy0);
            const int cx = ix0 < 0 ? 0 : (ix0 >= p.W ? p.W - 1 : ix0);
            zy |= (iy0 != cy) ? (1u << d) : 0u;
            zx |= (ix0 != cx) ? (1u << d) : 0u;
            const int iy = circ ? wy : cy, ix = circ ? wx : cx;
            const int yv = imgoff + iy * p.W * p.Cin, xv = ix * p.Cin + a_c4 * 4;
            if (d == 0) { yo0[i] = yv; xo0[i] = xv; } else if (d == 1) { yo1[i] = yv; xo1[i] = xv; } else { yo2[i] = yv; xo2[i] = xv; }
        }
        unsigned z = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) z |= (((zy >> (t / 3)) | (zx >> (t % 3))) & 1u) << t;
        zbits[i] = z;
    }
    int b_off[B_PT], b_wr[B_PT];
#pragma unroll
    for (int i = 0; i < B_PT; ++i) {
        const int f = tid + i * 256;
        const int fc = (B_F4 % 256 == 0 || f < B_F4) ? f : 0;
        const int n = fc / C4, c4 = fc % C4;
        b_off[i] = n * p.Ktot + c4 * 4;
        b_wr[i] = n * LDR + c4 * 4;
    }
    const float* const wbase = p.w + (int64_t)n0 * p.Ktot;

    // ---- K-step state: (tap, chunk) of the NEXT step to gather, current lane offsets --------------
    int nx_tap = 0, nx_chunk = 0;                    // wave-uniform
    int cur[A_PT];
    unsigned curz = 0;
    auto set_tap = [&](int tap) {
        const int dy = tap / p.ks, dx = tap - dy * p.ks;
        curz = 0;
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            const int yo = dy == 0 ? yo0[i] : (dy == 1 ? yo1[i] : yo2[i]);
            const int xo = dx == 0 ? xo0[i] : (dx == 1 ? xo1[i] : xo2[i]);
            cur[i] = yo + xo;
            if (PADZ) curz |= ((zbits[i] >> (dy * 3 + dx)) & 1u) << i;
        }
    };
    set_tap(0);

    f32x4 ra[A_PT], rb[B_PT], rg[A_PT];
    unsigned raz = 0;
    auto gather = [&]() {                            // issue the loads of step (nx_tap, nx_chunk), then advance
        const float* xs = xb + nx_chunk * KC;                                          // uniform
        const float* ws = wbase + (nx_tap * p.n_chunks + nx_chunk) * KC;               // uniform
#pragma unroll
        for (int i = 0; i < A_PT; ++i) ra[i] = *reinterpret_cast<const f32x4*>(xs + cur[i]);
        if (PRE == VQAE_PRE_CHANNEL_GATE) {
#pragma unroll
            for (int i = 0; i < A_PT; ++i) rg[i] = *reinterpret_cast<const f32x4*>(p.gate + g_off[i] + nx_chunk * KC);
        }
#pragma unroll
        for (int i = 0; i < B_PT; ++i) rb[i] = *reinterpret_cast<const f32x4*>(ws + b_off[i]);
        raz = curz;
    };
    auto advance = [&]() {                           // move (nx_tap, nx_chunk) to the following step
        if (++nx_chunk == p.n_chunks) {
            nx_chunk = 0;
            ++nx_tap;
            if (nx_tap < p.ks * p.ks) set_tap(nx_tap);
        }
    };
    auto stage = [&](elem* As, elem* Bs) {           // registers -> LDS (+ Fixup pre-op, zero padding)
#pragma unroll
        for (int i = 0; i < A_PT; ++i) {
            f32x4 v = ra[i];
            if (PRE == VQAE_PRE_CHANNEL_GATE) {
                v = v * rg[i];
            } else if (PRE != VQAE_PRE_NONE) {
                v = v + p.pre_a;
                if (PRE == VQAE_PRE_BIAS_ELU_BIAS) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = elu_act(v[e]) + p.pre_b;
                }
            }
            if (PADZ) { if ((raz >> i) & 1u) v = (f32x4)(0.f); }
            if (R16) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rnd(v[e]);                        // cast at the conv input
            }
            P::store4(As + a_wr[i], v);
        }
#pragma unroll
        for (int i = 0; i < B_PT; ++i)
            if (B_F4 % 256 == 0 || tid + i * 256 < B_F4) P::store4(Bs + b_wr[i], rb[i]);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const elem* const a_frag = Abuf0 + (wm * MI * 32 + (lane & 31)) * LDR + (SK / 2) * (lane >> 5);
    const elem* const b_frag = Bbuf0 + (wn * NI * 32 + (lane & 31)) * LDR + (SK / 2) * (lane >> 5);
    auto compute = [&](const elem* As, const elem* Bs) {
        // fragment reads are software-pipelined one k-slice ahead of the MFMAs that use them
        frag a[2][MI], b[2][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const frag*>(As + mi * 32 * LDR);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b[0][ni] = *reinterpret_cast<const frag*>(Bs + ni * 32 * LDR);
#pragma unroll
        for (int u = 0; u < KC / SK; ++u) {
            if (u + 1 < KC / SK) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    a[(u + 1) & 1][mi] = *reinterpret_cast<const frag*>(As + mi * 32 * LDR + SK * (u + 1));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    b[(u + 1) & 1][ni] = *reinterpret_cast<const frag*>(Bs + ni * 32 * LDR + SK * (u + 1));
            }
            if constexpr (!M16) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u & 1][mi][r], b[u & 1][ni][r], acc[mi][ni], 0, 0, 0);
            } else {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni) P::mma(acc[mi][ni], a[u & 1][mi], b[u & 1][ni]);
            }
        }
    };

    // One K-step as a single scheduling region: the gather's loads are issued between the first MFMAs,
    // the LDS stores of the gathered tile between the last ones, so the matrix pipe (64 cycles per
    // 32x32x2 MFMA) never drains while this wave issues memory instructions.
    constexpr int N_MFMA = M16 ? MI * NI * (KC / 16) : MI * NI * (KC / 2);   // MFMAs per wave per step
    constexpr int N_LD = A_PT + B_PT;                // LDS stores (and, without the gate, global loads) per thread per step
    auto step = [&](const elem* As, const elem* Bs, elem* Asn, elem* Bsn) {
        gather();
        compute(As, Bs);
        stage(Asn, Bsn);
        if (!M16 && N_MFMA >= 4 * N_LD && PRE != VQAE_PRE_CHANNEL_GATE) {
#pragma unroll
            for (int i = 0; i < N_LD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // 1 VMEM read
            }
            __builtin_amdgcn_sched_group_barrier(0x008, N_MFMA - 3 * N_LD, 0);
#pragma unroll
            for (int i = 0; i < N_LD; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // 1 DS write
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // 2 MFMA
            }
        }
    };

    // prologue: step 0 -> buffer 0
    gather();
    advance();
    stage(Abuf0, Bbuf0);
    __syncthreads();

    // main loop, unrolled over the two LDS buffers
    int remaining = p.n_steps - 1;                   // steps still to be gathered
    while (remaining >= 2) {
        step(a_frag, b_frag, Abuf0 + A_BUF, Bbuf0 + B_BUF);
        __syncthreads();
        advance();
        step(a_frag + A_BUF, b_frag + B_BUF, Abuf0, Bbuf0);
        __syncthreads();
        advance();
        remaining -= 2;
    }
    if (remaining == 1) {
        step(a_frag, b_frag, Abuf0 + A_BUF, Bbuf0 + B_BUF);
        __syncthreads();
        compute(a_frag + A_BUF, b_frag + B_BUF);
    } else {
        compute(a_frag, b_frag);
    }

    if constexpr (TAIL > 0) {
        static_assert(NT == 128 || NT == 64, "fused tail: the N tile must cover all C = NT channels");
        constexpr int CC = NT;                              // channels of the block
        constexpr int LDT = CC + P::PAD;
        elem* const T = reinterpret_cast<elem*>(lds);       // [128][CC + PAD], aliases the staging buffers
        const int li = lane & 31, hh = lane >> 5;
        const int rows_valid_t = (p.M - m0 < 128) ? (p.M - m0) : 128;
        const unsigned range_t = (unsigned)rows_valid_t * (unsigned)CC * 4u;

        auto acc_to_lds = [&]() {                           // accumulator (C/D layout) -> T[row][n]
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        T[(wm * MI * 32 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * LDT + wn * NI * 32 + ni * 32 + li] = P::cvt(acc[mi][ni][r]);
        };
        auto gemm_tail = [&](const float* __restrict__ wsrc) {   // acc = T (128 x CC) x wsrc^T, K = CC
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            const elem* af = T + (wm * MI * 32 + li) * LDT + (SK / 2) * hh;
            // fragment order: one wave-wide load = 64 x SK/2 consecutive floats (row-major weights made it touch 32
            // cache lines; with 16-bit MFMAs the tails were bound by these loads)
            const float* bf = wsrc + (wn * NI) * (32 * CC) + (SK / 2) * lane;
            constexpr int NG = CC / (4 * SK);               // groups of 4 k-slices
            frag bq[2][4][NI];                              // B fragments, 4 k-slices per group, 2 groups in flight
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bq[0][u][ni] = P::load_b(bf + ni * 32 * CC + 32 * SK * u);
#pragma unroll
            for (int ug = 0; ug < NG; ++ug) {
                if (ug + 1 < NG) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            bq[(ug + 1) & 1][u][ni] = P::load_b(bf + ni * 32 * CC + 32 * SK * (4 * (ug + 1) + u));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    frag a[MI];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) a[mi] = *reinterpret_cast<const frag*>(af + mi * 32 * LDT + SK * (4 * ug + u));
                    if constexpr (!M16) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                                for (int ni = 0; ni < NI; ++ni)
                                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][r], bq[ug & 1][u][ni][r], acc[mi][ni], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni) P::mma(acc[mi][ni], a[mi], bq[ug & 1][u][ni]);
                    }
                }
            }
        };

        // t2 = ELU(conv2 + b3a) + b3b
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r)           // conv2 output cast, fp32 activation, conv3 input cast
                    acc[mi][ni][r] = rnd(elu_act(rnd(acc[mi][ni][r]) + p.act_a) + p.act_b);
        __syncthreads();                                    // every wave is done with the last K-step's tiles
        acc_to_lds();
        __syncthreads();
        gemm_tail(p.w3);                                    // conv3

        // out = conv3 * scale + bias4 + x, in place over the residual stream
        const __amdgpu_buffer_rsrc_t o_rsrc =
            __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)m0 * CC, 0, (int)range_t, 0x00020000);
        const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.residual + (int64_t)m0 * CC), 0, (int)range_t, 0x00020000);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const unsigned base = (unsigned)((wm * MI * 32 + mi * 32 + 4 * hh) * (CC * 4) + (wn * NI * 32 + ni * 32 + li) * 4);
                float res[16];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(x_rsrc, base + ((r & 3) + 8 * (r >> 2)) * (CC * 4), 0, 0));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float t = rnd(acc[mi][ni][r]) * p.t_scale;
                    t = t + p.t_b4;
                    t = t + res[r];
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), o_rsrc, base + ((r & 3) + 8 * (r >> 2)) * (CC * 4), 0, 0);
                    if (TAIL == 2) acc[mi][ni][r] = rnd(elu_act(t + p.n_b1a) + p.n_b1b);     // next block's conv1 pre-op
                }
            }
        if constexpr (TAIL == 2) {
            __syncthreads();                                // conv3 finished reading T
            acc_to_lds();
            __syncthreads();
            gemm_tail(p.w1n);                               // next block's conv1
            const __amdgpu_buffer_rsrc_t t_rsrc =
                __builtin_amdgcn_make_buffer_rsrc(p.y2 + (int64_t)m0 * CC, 0, (int)range_t, 0x00020000);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const unsigned base = (unsigned)((wm * MI * 32 + mi * 32 + 4 * hh) * (CC * 4) + (wn * NI * 32 + ni * 32 + li) * 4);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float t = elu_act(rnd(acc[mi][ni][r]) + p.n_b2a) + p.n_b2b;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), t_rsrc, base + ((r & 3) + 8 * (r >> 2)) * (CC * 4), 0, 0);
                    }
                }
        }
        return;
    }

    // ---- epilogue ---------------------------------------------------------------------------------
    // C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    // Output / residual rows of this tile are addressed through buffer descriptors based at row m0 whose
    // range ends at row M: the hardware drops the stores (and zero-fills the loads) of the M tail.
    const int rows_valid = (p.M - m0 < 128) ? (p.M - m0) : 128;
    const unsigned range = (unsigned)rows_valid * (unsigned)p.Cout * 4u;
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(p.y + (int64_t)m0 * p.Cout, 0, (int)range, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.residual ? p.residual + (int64_t)m0 * p.Cout : p.y), 0, p.residual ? (int)range : 0, 0x00020000);
    const int col = lane & 31;
    const int rhalf = 4 * (lane >> 5);
    const int row_bytes = p.Cout * 4;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int nl = wn * NI * 32 + ni * 32 + col;
        const bool n_ok = n0 + nl < p.Cout;
        const float bv = (p.bias_vec && n_ok) ? p.bias_vec[n0 + nl] : 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            // a column past Cout gets an offset beyond the range -> dropped by the range check
            const unsigned base = n_ok ? (unsigned)((wm * MI * 32 + mi * 32 + rhalf) * row_bytes + (n0 + nl) * 4) : 0xF0000000u;
            float res[16];
            if (p.residual) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rsrc, base + ((r & 3) + 8 * (r >> 2)) * row_bytes, 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float t = acc[mi][ni][r];
                if (p.bias_vec) t = t + bv;                 // the conv's own bias is part of the (16-bit) conv output
                t = rnd(t);
                if (p.has_scale) { t = t * p.scale; t = t + p.bias_s; }
                else if (p.has_bias_s) { t = t + p.bias_s; }
                if (p.residual) t = t + res[r];
                if (p.has_act == VQAE_ACT_SILU) t = t / (1.0f + expf(-t));
                else if (p.has_act) t = elu_act(t + p.act_a) + p.act_b;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, t), y_rsrc, base + ((r & 3) + 8 * (r >> 2)) * row_bytes, 0, 0);
            }
        }
    }
}

__global__ void pack_weight_kernel(const float* __restrict__ w, int cout, int cin, int ks, int npad,
                                   float* __restrict__ out) {
    const int ktot = ks * ks * cin;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)npad * ktot) return;
    const int n = (int)(i / ktot), k = (int)(i % ktot);
    const int tap = k / cin, ci = k % cin;
    out[i] = (n < cout) ? w[((int64_t)n * cin + ci) * ks * ks + tap] : 0.f;
}

template <int NT, int KC, int PRE, bool PADZ, int TAIL, int DT, bool M16>
int launch_r(const ConvK& k, hipStream_t stream) {
    using P = MP<DT, M16>;
    const int npad = (int)vqae::round_up(k.Cout, 32);
    dim3 grid((unsigned)vqae::ceil_div(k.M, 128), (unsigned)vqae::ceil_div(npad, NT));
    constexpr int lds_bytes = 2 * (128 + NT) * (KC + P::PAD) * (int)sizeof(typename P::elem);
    static bool attr_set = false;
    if (!attr_set) {
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)conv_mfma_kernel<NT, KC, PRE, PADZ, TAIL, DT, M16>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    const int cls = (NT == 128 && KC == 32 && k.Cin >= 128) ? (k.ks == 3 ? vqae::PROF_CONV3X3_TRUNK : (k.ks == 1 ? vqae::PROF_CONV1X1_TRUNK : 0)) : 0;
    const double flops = 2.0 * k.M * (double)k.Cout * ((double)k.Ktot + (TAIL >= 1 ? (double)k.Cout : 0.0) + (TAIL == 2 ? (double)k.Cout : 0.0));
    vqae::ProfScope prof(cls, stream, flops);
    conv_mfma_kernel<NT, KC, PRE, PADZ, TAIL, DT, M16><<<grid, 256, lds_bytes, stream>>>(k);
    prof.done();
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

template <int NT, int KC, int PRE, bool PADZ, int TAIL = 0>
int launch(const ConvK& k, hipStream_t stream) {
    if (k.m16) {          // 16-bit MFMA engine: KC in {32, 64}, circular / no padding only
        if constexpr (!PADZ && KC >= 32) {
            if (k.dt == VQAE_DT_BF16) return launch_r<NT, KC, PRE, PADZ, TAIL, VQAE_DT_BF16, true>(k, stream);
            return launch_r<NT, KC, PRE, PADZ, TAIL, VQAE_DT_F16, true>(k, stream);
        } else {
            return vqae::fail(VQAE_ERR_UNSUPPORTED, "conv2d: 16-bit engine selected for an unsupported variant");
        }
    }
    if constexpr (KC <= 32) {
        if (k.dt == VQAE_DT_BF16) return launch_r<NT, KC, PRE, PADZ, TAIL, VQAE_DT_BF16, false>(k, stream);
        if (k.dt == VQAE_DT_F16) return launch_r<NT, KC, PRE, PADZ, TAIL, VQAE_DT_F16, false>(k, stream);
        return launch_r<NT, KC, PRE, PADZ, TAIL, VQAE_DT_F32, false>(k, stream);
    } else {
        return vqae::fail(VQAE_ERR_UNSUPPORTED, "conv2d: KC 64 needs the 16-bit engine");
    }
}

template <int NT, int KC>
int launch_pre(const ConvK& k, hipStream_t stream) {
    const bool padz = k.pad > 0 && k.pad_mode == VQAE_PAD_ZEROS;
    if (padz && k.pre_mode == VQAE_PRE_CHANNEL_GATE) return vqae::fail(VQAE_ERR_UNSUPPORTED, "conv2d: gated load with zero padding");
    if (padz) {       // zero padding is only used by tests / generic callers: one (slower) variant
        switch (k.pre_mode) {
            case VQAE_PRE_NONE: return launch<NT, KC, VQAE_PRE_NONE, true>(k, stream);
            case VQAE_PRE_BIAS: return launch<NT, KC, VQAE_PRE_BIAS, true>(k, stream);
            default: return launch<NT, KC, VQAE_PRE_BIAS_ELU_BIAS, true>(k, stream);
        }
    }
    switch (k.pre_mode) {
        case VQAE_PRE_NONE: return launch<NT, KC, VQAE_PRE_NONE, false>(k, stream);
        case VQAE_PRE_BIAS: return launch<NT, KC, VQAE_PRE_BIAS, false>(k, stream);
        case VQAE_PRE_CHANNEL_GATE:
            if constexpr (KC == 32) {
                if (k.dt == VQAE_DT_F32 && k.ks == 1) return launch_r<NT, KC, VQAE_PRE_CHANNEL_GATE, false, 0, VQAE_DT_F32, false>(k, stream);
            }
            return vqae::fail(VQAE_ERR_UNSUPPORTED, "conv2d: the gated operand load needs a fp32 1x1 conv with cin %% 32 == 0");
        default: return launch<NT, KC, VQAE_PRE_BIAS_ELU_BIAS, false>(k, stream);
    }
}

template <int NT>
int launch_kc(const ConvK& k, int kc, hipStream_t stream) {
    switch (kc) {
        case 64: return launch_pre<NT, 64>(k, stream);
        case 32: return launch_pre<NT, 32>(k, stream);
        case 16: return launch_pre<NT, 16>(k, stream);
        default: return launch_pre<NT, 8>(k, stream);
    }
}

}  // namespace

extern "C" size_t vqae_conv_packed_floats(int cout, int cin, int ksize) {
    // rows padded to a multiple of 128 so every N tile (32/64/128 wide) reads in-bounds rows
    return (size_t)vqae::round_up(cout, 128) * ksize * ksize * cin;
}

extern "C" int vqae_conv_pack_weight_f32(const float* w, int cout, int cin, int ks, float* packed, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VQAE_REQUIRE(w && packed, VQAE_ERR_INVALID, "pack_weight: null pointer");
    VQAE_REQUIRE(ks >= 1 && ks <= 3 && cin >= 1 && cout >= 1, VQAE_ERR_INVALID, "pack_weight: bad shape");
    const int npad = (int)vqae::round_up(cout, 128);
    const int64_t tot = (int64_t)npad * ks * ks * cin;
    pack_weight_kernel<<<(unsigned)vqae::ceil_div(tot, 256), 256, 0, stream>>>(w, cout, cin, ks, npad, packed);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

namespace {
int fill_conv(const vqae_conv_args* a, const float* x, const float* w, const float* bias_vec, const float* residual,
              float* y, ConvK* out, int* kc_out) {
    VQAE_REQUIRE(a, VQAE_ERR_INVALID, "conv2d: null args");
    VQAE_REQUIRE(x && w && y, VQAE_ERR_INVALID, "conv2d: null pointer");
    VQAE_REQUIRE(a->cin % 8 == 0 && a->cin >= 8, VQAE_ERR_UNSUPPORTED,
                 "conv2d: cin %d must be a multiple of 8 (use vqae_conv3x3_direct_f32 for stems)", a->cin);
    VQAE_REQUIRE(a->ksize >= 1 && a->ksize <= 3 && (a->stride == 1 || a->stride == 2) && a->pad >= 0 && a->pad <= 1,
                 VQAE_ERR_UNSUPPORTED, "conv2d: unsupported geometry k%d s%d p%d", a->ksize, a->stride, a->pad);
    VQAE_REQUIRE(a->batch >= 0 && a->in_h >= 1 && a->in_w >= 1 && a->cout >= 1, VQAE_ERR_INVALID, "conv2d: bad shape");
    const int Ho = (a->in_h + 2 * a->pad - a->ksize) / a->stride + 1;
    const int Wo = (a->in_w + 2 * a->pad - a->ksize) / a->stride + 1;
    VQAE_REQUIRE(Ho >= 1 && Wo >= 1, VQAE_ERR_INVALID, "conv2d: empty output");
    if (a->pad_mode == VQAE_PAD_CIRCULAR)
        VQAE_REQUIRE(a->pad <= a->in_h && a->pad <= a->in_w, VQAE_ERR_INVALID, "conv2d: circular pad larger than input");
    const int64_t M = (int64_t)a->batch * Ho * Wo;
    VQAE_REQUIRE(M < (1ll << 31) - 256, VQAE_ERR_UNSUPPORTED, "conv2d: too many output pixels (%lld)", (long long)M);
    VQAE_REQUIRE(a->pre_mode >= VQAE_PRE_NONE && a->pre_mode <= VQAE_PRE_CHANNEL_GATE, VQAE_ERR_INVALID, "conv2d: pre_mode %d", a->pre_mode);
    VQAE_REQUIRE(a->has_act >= 0 && a->has_act <= VQAE_ACT_SILU, VQAE_ERR_INVALID, "conv2d: has_act %d", a->has_act);
    // lane offsets are 32-bit and relative to the first image of a 128-pixel tile (which spans at most
    // 128 / (Ho*Wo) + 2 images)
    VQAE_REQUIRE(((int64_t)128 / (Ho * Wo) + 3) * a->in_h * a->in_w * a->cin < (1ll << 30), VQAE_ERR_UNSUPPORTED,
                 "conv2d: image too large for 32-bit tile-relative offsets");
    ConvK k;
    memset(&k, 0, sizeof(k));
    k.x = x; k.w = w; k.bias_vec = bias_vec; k.residual = residual; k.y = y;
    k.B = a->batch; k.H = a->in_h; k.W = a->in_w; k.Cin = a->cin; k.Cout = a->cout; k.Ho = Ho; k.Wo = Wo;
    k.ks = a->ksize; k.stride = a->stride; k.pad = a->pad;
    k.pad_mode = (a->pad == 0) ? VQAE_PAD_ZEROS : a->pad_mode;    // pad 0: the bounds test always passes
    VQAE_REQUIRE(k.pad_mode == VQAE_PAD_ZEROS || k.pad_mode == VQAE_PAD_CIRCULAR, VQAE_ERR_INVALID,
                 "conv2d: pad_mode %d", a->pad_mode);
    k.M = (int)M; k.Ktot = a->ksize * a->ksize * a->cin;
    static const bool no_m16 = getenv("VQAE_NO_MFMA16") && atoi(getenv("VQAE_NO_MFMA16"));
    const bool padz = a->pad > 0 && a->pad_mode == VQAE_PAD_ZEROS;
    k.m16 = (a->dtype != VQAE_DT_F32 && a->cin % 32 == 0 && !padz && !no_m16) ? 1 : 0;
    const int kc = k.m16 ? ((a->cin % 64 == 0) ? 64 : 32) : ((a->cin % 32 == 0) ? 32 : (a->cin % 16 == 0) ? 16 : 8);
    k.n_chunks = a->cin / kc;
    k.n_steps = a->ksize * a->ksize * k.n_chunks;
    k.pre_mode = a->pre_mode; k.pre_a = a->pre_a; k.pre_b = a->pre_b;
    k.has_scale = a->has_scale; k.has_bias_s = a->has_bias_s; k.has_act = a->has_act;
    k.scale = a->scale; k.bias_s = a->bias_s; k.act_a = a->act_a; k.act_b = a->act_b;
    VQAE_REQUIRE(a->dtype >= VQAE_DT_F32 && a->dtype <= VQAE_DT_F16, VQAE_ERR_INVALID, "conv2d: dtype %d", a->dtype);
    k.dt = a->dtype;
    *out = k;
    *kc_out = kc;
    return VQAE_OK;
}
}  // namespace

static int conv2d_impl(const vqae_conv_args* a, const float* x, const float* gate, const float* w, const float* bias_vec,
                       const float* residual, float* y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VQAE_REQUIRE(a, VQAE_ERR_INVALID, "conv2d: null args");
    if (a->batch == 0) return VQAE_OK;
    ConvK k;
    int kc;
    const int rc = fill_conv(a, x, w, bias_vec, residual, y, &k, &kc);
    if (rc) return rc;
    static const bool no_small = getenv("VQAE_NO_SMALL_K") && atoi(getenv("VQAE_NO_SMALL_K"));
    if (!gate && !no_small && vqae::conv_small_k_supported(a))       // 8-channel level: VALU kernel at the HBM rate
        return vqae::conv_small_k(a, x, w, bias_vec, residual, y, stream);
    VQAE_REQUIRE((a->pre_mode == VQAE_PRE_CHANNEL_GATE) == (gate != nullptr), VQAE_ERR_INVALID,
                 "conv2d: VQAE_PRE_CHANNEL_GATE and the gate pointer go together (vqae_conv2d_gated_f32)");
    k.gate = gate;
    if (a->cout <= 32) return launch_kc<32>(k, kc, stream);
    if (a->cout <= 64) return launch_kc<64>(k, kc, stream);
    return launch_kc<128>(k, kc, stream);
}

extern "C" int vqae_conv2d_f32(const vqae_conv_args* a, const float* x, const float* w, const float* bias_vec,
                               const float* residual, float* y, void* stream) {
    return conv2d_impl(a, x, nullptr, w, bias_vec, residual, y, stream);
}

extern "C" int vqae_conv2d_gated_f32(const vqae_conv_args* a, const float* x, const float* gate, const float* w,
                                     const float* bias_vec, const float* residual, float* y, void* stream) {
    VQAE_REQUIRE(gate, VQAE_ERR_INVALID, "conv2d_gated: null gate");
    return conv2d_impl(a, x, gate, w, bias_vec, residual, y, stream);
}

namespace vqae {
// k-slice width of the engine conv_trunk_tail will use for this dtype / channel count (8: fp32 MFMA, 16: 16-bit MFMA);
// w3 / w1n must be in fragment order for that width
int conv_tail_kslice(int dtype, int cin) {
    static const bool no_m16 = getenv("VQAE_NO_MFMA16") && atoi(getenv("VQAE_NO_MFMA16"));
    return (dtype != VQAE_DT_F32 && cin % 32 == 0 && !no_m16) ? 16 : 8;
}

// Trunk Fixup block tail fusion (internal to the handle): conv2 (3x3 circular, C = 128, `a` carries its
// geometry and its ELU epilogue) + conv3 (+ the next block's conv1 when w1n != null).
//   t1 [M][128] -> xio [M][128] updated in place (block output) and, if w1n, t1_next [M][128].
int conv_trunk_tail(const vqae_conv_args* a, const float* t1, const float* w2, const float* w3, float t_scale,
                    float t_b4, float* xio, const float* w1n, float n_b1a, float n_b1b, float n_b2a, float n_b2b,
                    float* t1_next, hipStream_t stream) {
    if (a->batch == 0) return VQAE_OK;
    VQAE_REQUIRE((a->cin == 128 || a->cin == 64) && a->cout == a->cin && a->ksize == 3 && a->has_act && a->pre_mode == VQAE_PRE_NONE &&
                 a->pad == 1 && a->pad_mode == VQAE_PAD_CIRCULAR, VQAE_ERR_UNSUPPORTED, "conv_trunk_tail: not a trunk conv2");
    VQAE_REQUIRE(w3 && xio && (!w1n || t1_next), VQAE_ERR_INVALID, "conv_trunk_tail: null pointer");
    ConvK k;
    int kc;
    const int rc = fill_conv(a, t1, w2, nullptr, xio, xio, &k, &kc);
    if (rc) return rc;
    k.w3 = w3; k.w1n = w1n; k.y2 = t1_next;
    k.t_scale = t_scale; k.t_b4 = t_b4; k.n_b1a = n_b1a; k.n_b1b = n_b1b; k.n_b2a = n_b2a; k.n_b2b = n_b2b;
    if (a->cin == 64) {
        if (kc == 64) return w1n ? launch<64, 64, VQAE_PRE_NONE, false, 2>(k, stream) : launch<64, 64, VQAE_PRE_NONE, false, 1>(k, stream);
        return w1n ? launch<64, 32, VQAE_PRE_NONE, false, 2>(k, stream) : launch<64, 32, VQAE_PRE_NONE, false, 1>(k, stream);
    }
    if (kc == 64) return w1n ? launch<128, 64, VQAE_PRE_NONE, false, 2>(k, stream) : launch<128, 64, VQAE_PRE_NONE, false, 1>(k, stream);
    return w1n ? launch<128, 32, VQAE_PRE_NONE, false, 2>(k, stream) : launch<128, 32, VQAE_PRE_NONE, false, 1>(k, stream);
}
}  // namespace vqae
