// Winograd F(2x2, 3x3) form of the trunk Fixup blocks, fp32: C = 128 (and 256) channels on a 32-wide grid (the code-grid
// resolution), C = 64 on a 64-wide grid and C = 32 on a 128-wide grid (the levels above it):
//   conv2 (3x3 circular, conv_block.py:208)  as  Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A
// followed, in the same workgroup, by the block's conv3 (+ scale / bias4 / residual) and the NEXT block's conv1 --
// the same fusion as conv_mfma.hip's TAIL, whose tail this kernel repeats.
//
// Why: fp32 MFMA is the roofline of this path (157.3 TFLOP/s) and the 3x3 conv is 9/11 of a trunk block's matrix
// work.  F(2x2, 3x3) needs 16 multiplies per 2x2 output tile and channel pair instead of 36: the 3x3 becomes 16
// independent [tiles x 128] x [128 x 128] GEMMs, 2.25x fewer MFMAs (K_eff = 512 instead of 1152 per output pixel).
// The transforms are additions only (B^T, A^T have entries 0, +-1; G's halves are folded into the pre-transformed
// weights on the host), a few VALU instructions per element.  Result differs from the direct form by fp32 rounding
// only (measured in tests/test_model_gpu.py::test_winograd_trunk_equals_direct).
//
// Work split (written for C = 128; C = 64 and C = 32 in brackets; C = 256: same tile, 512 threads = 8 channel slices).  A 256-thread workgroup owns 4 image rows =
// 128 [256, 512] output pixels = 32 [64, 128] Winograd tiles (2 tile rows x 16 [32, 64] tile columns).  A wave owns 32
// output channels and 32 tiles: 4 channel slices x 1 tile group [2 x 2, 1 x 4].  The 4x4 transformed domain is walked
// one row xi at a time (4 passes):
//   transform  V_xi[nu][tile][c] = (B^T d B)[xi][nu], nu = 0..3, for all 128 input channels -> LDS (4 x 32 x 132 floats);
//              the input rows come straight from global/L2 (t1 of the block, written by the previous launch)
//   GEMM       acc[nu] (32 tiles x 32 channels, one 32x32 MFMA tile) += V_xi[nu] x U[xi, nu]^T over K = 128:
//              A fragments from LDS, B fragments (pre-transformed weights, 1 MiB per layer, L2-resident) from global
//   fold       Y[a][b] += A^T[a][xi] * A^T[b][nu] * acc   in registers, right after each nu's K loop: the lane that holds
//              (tile, channel) of one (xi, nu) product holds it for all of them
// so only one 16-register accumulator + 4 x 16 output registers are live and two workgroups fit a CU; while one
// transforms, the other one's MFMAs run.  Then t2 = ELU(Y + b3a) + b3b goes to LDS as the [128 px][132] A operand of
// the conv3 / next-conv1 tails.
//
// Measured (cfg B, batch 256, `-DVQAE_WINO_TRACE` stamps): a tile costs a wave ~98 k cycles of MFMA issue, ~10 k of VALU
// (fold 576, ELU 1150, transforms 512 instructions ...; fp32 MFMA and VALU do not co-execute) and ~20 k of exposed
// waits across its 14 barrier-separated phases; a lone workgroup per CU takes 154 k cycles per tile, two take 260 k for
// two.  DESIGN.md section 4 has the roofline numbers, section 8 what would move them.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
using vqae::elu_act;

struct WinoK {
    const float* __restrict__ t1;        // [M][C] conv2 input (= ELU(conv1(.) + b2a) + b2b)
    const float* __restrict__ U;         // G g G^T in fragment order [16 pos][C/32 n-tiles][C/8 k-slices][64 lanes][4]
    const float* __restrict__ w3;        // [C][C] in fragment order [C/32 n-tiles][C/8 k-slices][64 lanes][4]
    const float* __restrict__ w1n;       // same, the next block's conv1 (TAIL == 2)
    float* xio;                          // [M][C] residual stream, updated in place
    float* y2;                           // [M][C] next block's t1 (TAIL == 2)
    int H, Wimg, M;                      // image rows / columns (Wimg a multiple of the workgroup's column span); M = B * H * Wimg
    float act_a, act_b, t_scale, t_b4, n_b1a, n_b1b, n_b2a, n_b2b;
#ifdef VQAE_WINO_TRACE
    unsigned long long* trace;           // [wg][4 waves][32] s_memtime stamps (developer build only)
#endif
};

// Developer aid (off by default): per-phase s_memtime stamps of every wave -> gpurun_out/wino_trace.bin.  It showed that
// the cost of this kernel's first version sat in the L1 tag pipe (row-major weight fragments), not in HBM or the MFMAs.
#ifdef VQAE_WINO_TRACE
#define STAMP(i) do { if (lane == 0 && p.trace && wave < 4) p.trace[((int64_t)blockIdx.x * 4 + wave) * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

constexpr int BPF = 6;                   // weight-fragment prefetch distance (k-slices)

template <int C> struct WinoCfg {
#ifdef VQAE_WINO_SMALL_WG                               // experiment: half-size workgroups (2 waves), 4 per CU, at C = 32 / 64
    static constexpr int W = C >= 128 ? 32 : (C == 64 ? 32 : 64);
    static constexpr int NT = C == 256 ? 512 : (C >= 128 ? 256 : 128);
#else
    static constexpr int W = C >= 128 ? 32 : (C == 64 ? 64 : 128);   // columns a workgroup spans (the image may be k times wider)
    static constexpr int NT = C == 256 ? 512 : 256;    // threads: at C = 256 eight waves (one per 32-channel slice), one workgroup per CU
#endif
    static constexpr int NW = NT / 64;
    static constexpr int PX = 4 * W;                   // output pixels per workgroup (4 image rows)
    static constexpr int TILES = PX / 4;               // 2x2 output tiles per workgroup
    static constexpr int TC = W / 2;                   // tile columns
    static constexpr int NS = C / 32;                  // 32-channel slices
    static constexpr int KS = C / 8;                   // k-slices (8 channels) per GEMM
    static constexpr int C4 = C / 4;                   // float4 per pixel
    static constexpr int RP = NT / C4;                 // pixels (or tile columns) covered by one sweep of the workgroup's threads
    static constexpr int LDT = C + 4;                  // LDS row stride (floats): conflict-free ds_read_b128 fragments
    static constexpr int NI = C >= 64 ? 2 : 1;         // tails: 32-channel tiles per wave
    static constexpr int WN = C / (32 * NI);           //        waves along the channels, NW / WN along the pixels
    static constexpr int MI = PX / ((NW / WN) * 32);   //        32-pixel tiles per wave (MI * NI = 4 accumulators)
    static constexpr int LDS_BYTES = PX * LDT * 4;     // V[4][TILES][LDT] and T[PX][LDT] overlay each other
#ifdef VQAE_WINO_PREF_ALL
    static constexpr int PREF = 1;
#else
    // C <= 64: the first half of the NEXT pass's input rows is requested before this pass's MFMAs (64 registers in flight).  These
    // levels move as many bytes per tile as C = 128 with 1/4 (C = 32) or 1/2 (C = 64) of the matrix work, and without it a tile
    // takes HBM time PLUS matrix time (round 3).  At C = 128 the same prefetch measured slower (section 4, negative results).
    static constexpr int PREF = C <= 64 ? 1 : 0;
#endif
};

// Fragment order of a [C n][C k] matrix: element (n, k) of the 32-row tile n >> 5 and 8-wide k-slice k >> 3 goes to
// lane (k >> 2 & 1) * 32 + (n & 31), component k & 3 -- what lane (li = n & 31, hh) feeds to MFMA number k & 3 of the slice.
__device__ __forceinline__ int frag_offset(int n, int k, int c, int sk = 8) {   // sk: k-slice width (16 for 16-bit MFMA)
    const int h = sk / 2;
    return (((n >> 5) * (c / sk) + k / sk) * 64 + ((k / h) & 1) * 32 + (n & 31)) * h + k % h;
}

// DT: autocast rounding points compiled in (16-bit modes, C = 32 only: every conv operand and conv output is rounded
// to bf16 / f16, the arithmetic stays fp32 -- conv(x16, w16) accumulated in fp32 is what torch.autocast computes, and its
// Winograd form differs from the direct one by fp32 rounding only).
// WIDE: the grid is k > 1 workgroup spans wide (column blocks); the exact-width case keeps all geometry compile-time.
template <int C, int TAIL, int DT, bool WIDE>
__global__ __launch_bounds__((WinoCfg<C>::NT), 2)
void wino_trunk_kernel(const WinoK p) {
    using K = WinoCfg<C>;
    auto rnd = [](float v) -> float {
        if (DT == VQAE_DT_BF16) return (float)(__bf16)v;
        if (DT == VQAE_DT_F16) return (float)(_Float16)v;
        return v;
    };
    auto rnd4 = [&](f32x4 v) -> f32x4 {
        if (DT != VQAE_DT_F32) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rnd(v[e]);
        }
        return v;
    };
    constexpr int W = K::W, PX = K::PX, TC = K::TC, NS = K::NS, KS = K::KS, C4 = K::C4, RP = K::RP, LDT = K::LDT, WN = K::WN;
    constexpr int MI = K::MI, NI = K::NI;
    constexpr bool PREF = K::PREF && !WIDE;                           // column-blocked grids carry more geometry: no registers left for it
    constexpr int STEPS = 4 * KS;                                    // k-slices per pass (4 nu)
    extern __shared__ __attribute__((aligned(16))) float lds[];      // V[4][TILES][LDT]  /  T[PX][LDT]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int ns = wave % NS, mt = wave / NS;                        // main phase: channel slice, 32-tile group

    // XCD-contiguous tile order (as conv_mfma.hip): neighbouring row groups of an image share an L2
    int tile_m;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        tile_m = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // tile -> (image, group of 4 rows, block of W columns); column blocks of a row group are neighbours in the order
    const int Wimg = WIDE ? p.Wimg : W;
    const int cblocks = WIDE ? p.Wimg / W : 1;
    const int tpi = (p.H / 4) * cblocks;
    const int img = tile_m / tpi;
    const int rem = tile_m - img * tpi;
    const int row0 = 4 * (rem / cblocks);                             // first image row of this workgroup
    const int col0 = W * (rem % cblocks);                             // first image column
    const int hw = p.H * Wimg;
    const float* const xim = p.t1 + (int64_t)img * hw * C;

    // ---- transform geometry: thread -> channel group cg, tile column tj0 (+ RP for odd items), tile rows 0 / 1 -------
    const int cg = tid % C4;
    const int tj0 = tid / C4;                                        // 0 .. RP - 1
    int coff[2][4], roff[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = col0 + 2 * (tj0 + RP * s) - 1 + j;
            c = WIDE ? (c < 0 ? c + Wimg : (c >= Wimg ? c - Wimg : c)) : (c & (W - 1));   // circular in x
            coff[s][j] = c * C + 4 * cg;
            int r = row0 + 2 * s - 1 + j;                            // tile row s: input rows row0 + 2s - 1 .. + 2
            r = r < 0 ? r + p.H : (r >= p.H ? r - p.H : r);          // circular in y
            roff[s][j] = r * Wimg * C;
        }
    }

    // Weight fragments are stored in the order the MFMA consumes them: one wave-wide 128-bit load = 1 KiB of
    // consecutive memory (8 cache lines).  Row-major weights would make the same load touch 32 lines, 32 B of each,
    // and the L1 tag pipe -- one line per cycle -- would cost as much as the MFMAs it feeds.
    const float* const ub = p.U + (int64_t)ns * (KS * 256) + 4 * lane;            // + pos * C * C + u * 256
    const float* const af = lds + (mt * 32 + li) * LDT + 4 * hh;                   // A fragment base, + nu * TILES * LDT

    // raw input rows of transform pass xi, in two halves (tile row 0 / 1) to bound the registers in flight:
    // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
    f32x4 v[4][4], la[2][4], lb[2][4];
    auto tr_load = [&](int xi, int s_r) {                            // tile row s_r: items (it = 2 * s_r + s_c)
        const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1);              // row combination: d[ra] (+|-) d[rb]
        const int rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
#pragma unroll
        for (int s_c = 0; s_c < 2; ++s_c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                la[s_c][j] = rnd4(*reinterpret_cast<const f32x4*>(xim + roff[s_r][ra] + coff[s_c][j]));   // conv2 input cast
                lb[s_c][j] = rnd4(*reinterpret_cast<const f32x4*>(xim + roff[s_r][rb] + coff[s_c][j]));
            }
    };
    auto tr_combine = [&](int xi, int s_r) {
#pragma unroll
        for (int s_c = 0; s_c < 2; ++s_c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[2 * s_r + s_c][j] = xi == 1 ? la[s_c][j] + lb[s_c][j] : la[s_c][j] - lb[s_c][j];
    };
    tr_load(0, 0);
    tr_combine(0, 0);
    tr_load(0, 1);
    tr_combine(0, 1);

    // Tail operands requested while the last pass's MFMAs run.  vmcnt retires in order (stores included), so a load
    // issued behind a slow one waits for it: the residual rows (HBM) go out after the last weight-fragment load of
    // the main phase, the first weight fragments of each tail GEMM before the stores of the epilogue in front of it.
    const int wm = wave / WN, wn = wave % WN;                        // tails: MI*32 pixels x NI*32 channels per wave
    // row-coalesced view of the tile: thread (cg, tj0) owns pixels tj0 + RP * i, i.e. row (RP * i) / W, column
    // tj0 + (RP * i) % W of the workgroup's 4 x W block
    // (four row pointers + immediate column offsets: the addresses cost no registers beyond these)
    const int64_t pix0 = ((int64_t)img * p.H + row0) * Wimg + col0 + tj0;
    float* xr[4];                                                     // set where first needed (register budget of the main phase)
    f32x4 res[16];
    f32x4 bt[2][4][NI];
    auto tail_prefetch = [&](const float* __restrict__ wsrc) {
        const float* b0 = wsrc + (wn * NI) * (KS * 256) + 4 * lane;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bt[0][u][ni] = *reinterpret_cast<const f32x4*>(b0 + ni * (KS * 256) + 256 * u);
    };

    STAMP(0);
    f32x16 y00, y01, y10, y11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { y00[r] = 0.f; y01[r] = 0.f; y10[r] = 0.f; y11[r] = 0.f; }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        // ---- row xi of B^T d B for all tiles and channels -> V[nu][tile][c] (v holds the row combinations) ---------
        __builtin_amdgcn_sched_barrier(0);
        if (xi > 0) __syncthreads();                                  // every wave is done with V of pass xi - 1
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int tile = (it >> 1) * TC + tj0 + RP * (it & 1);
            float* dst = lds + tile * LDT + 4 * cg;
            *reinterpret_cast<f32x4*>(dst + 0 * K::TILES * LDT) = v[it][0] - v[it][2];
            *reinterpret_cast<f32x4*>(dst + 1 * K::TILES * LDT) = v[it][1] + v[it][2];
            *reinterpret_cast<f32x4*>(dst + 2 * K::TILES * LDT) = v[it][2] - v[it][1];
            *reinterpret_cast<f32x4*>(dst + 3 * K::TILES * LDT) = v[it][1] - v[it][3];
        }

        // ---- GEMM: acc = V[nu] x U[xi, nu]^T for nu = 0..3, STEPS steps of one k-slice (8 channels) each -------------
        const float* const ux = ub + (int64_t)(4 * xi) * (C * C);
#define WB(s) (((s) / KS) * (C * C) + 256 * ((s) % KS))
        f32x4 bq[BPF];
#pragma unroll
        for (int s = 0; s < BPF; ++s)                                  // first B fragments: in flight across the barrier
            bq[s] = *reinterpret_cast<const f32x4*>(ux + WB(s));
        __syncthreads();
        STAMP(14 + 4 * xi);
        if (PREF && xi < 3) tr_load(xi + 1, 0);                     // in flight under this pass's MFMAs (v is dead until then)
        f32x4 aq[2];
        aq[0] = *reinterpret_cast<const f32x4*>(af);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            // One accumulator live at a time (register budget).  A lone dependent chain issues one 32x32x2 MFMA per
            // ~72 cycles instead of 64; a second (even / odd k-slice) accumulator measured no gain with two waves per SIMD.
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                const int s = KS * nu + u;
                if (s + 1 < STEPS)
                    aq[(s + 1) & 1] = *reinterpret_cast<const f32x4*>(af + ((s + 1) / KS) * K::TILES * LDT + 8 * ((s + 1) % KS));
                const f32x4 b = bq[s % BPF];
                if (s + BPF < STEPS)
                    bq[s % BPF] = *reinterpret_cast<const f32x4*>(ux + WB(s + BPF));
                if (xi == 3 && s == STEPS - BPF) {                     // main phase has no more loads to issue
                    tail_prefetch(p.w3);
#pragma unroll
                    for (int r = 0; r < 4; ++r) xr[r] = p.xio + (pix0 + (int64_t)r * Wimg) * C + 4 * cg;
#pragma unroll
                    for (int i = 0; i < 16; ++i) res[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr[(RP * i) / W] + ((RP * i) % W) * C));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b[r], aq[s & 1][r], acc, 0, 0, 0);   // D[channel][tile]
                __builtin_amdgcn_sched_barrier(0);                    // keep the prefetch distances as written (register budget)
            }
            // fold: Y[a][b] += A^T[a][xi] * A^T[b][nu] * acc,  A^T = [1 1 1 0; 0 1 -1 -1]
            const int ca0 = xi <= 2 ? 1 : 0, ca1 = xi == 0 ? 0 : (xi == 1 ? 1 : -1);
            const int cb0 = nu <= 2 ? 1 : 0, cb1 = nu == 0 ? 0 : (nu == 1 ? 1 : -1);
            if (ca0 * cb0 != 0) y00 = y00 + acc;
            if (ca0 * cb1 == 1) y01 = y01 + acc; else if (ca0 * cb1 == -1) y01 = y01 - acc;
            if (ca1 * cb0 == 1) y10 = y10 + acc; else if (ca1 * cb0 == -1) y10 = y10 - acc;
            if (ca1 * cb1 == 1) y11 = y11 + acc; else if (ca1 * cb1 == -1) y11 = y11 - acc;
            // pin the fold here: left to itself the compiler defers these adds to the end of the kernel and spills
            // every (xi, nu) accumulator to scratch meanwhile
            asm volatile("" : "+v"(y00), "+v"(y01), "+v"(y10), "+v"(y11));
            if (nu < 3) STAMP(15 + 4 * xi + nu);
        }
#undef WB
        // Next pass's input rows are requested only now.  Requesting them under this pass's MFMAs measured SLOWER
        // (0.499 vs 0.478 ms per launch), as did a longer weight-fragment prefetch: more loads in flight per CU back
        // up the vector-memory queue and the in-order wave stalls at issue, MFMAs included.  The partner workgroup of
        // the CU covers this wait.
        if (xi < 3) {
            if (!PREF) tr_load(xi + 1, 0);
            tr_combine(xi + 1, 0); tr_load(xi + 1, 1); tr_combine(xi + 1, 1);
        }
        STAMP(1 + xi);
    }

    // ---- t2 = ELU(conv2 + b3a) + b3b -> T[pixel][channel].  The MFMAs ran with the weights as the row operand, so a
    // lane holds tile (mt * 32 + lane & 31) and, per register group g = r >> 2, four consecutive channels
    // 32 ns + 8g + 4*hh + (r & 3): every LDS write below is 128 bits.
    __syncthreads();                                                  // every wave is done with V of pass 3
    float* const T = lds;
    {
        const int tile = mt * 32 + li;
        const int px = (2 * (tile / TC)) * W + 2 * (tile % TC);       // top-left pixel of this lane's 2x2 output tile
        float* const d = T + px * LDT + 32 * ns + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o00, o01, o10, o11;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                o00[e] = rnd(elu_act(rnd(y00[4 * g + e]) + p.act_a) + p.act_b);   // conv2 output cast, conv3 input cast
                o01[e] = rnd(elu_act(rnd(y01[4 * g + e]) + p.act_a) + p.act_b);
                o10[e] = rnd(elu_act(rnd(y10[4 * g + e]) + p.act_a) + p.act_b);
                o11[e] = rnd(elu_act(rnd(y11[4 * g + e]) + p.act_a) + p.act_b);
            }
            *reinterpret_cast<f32x4*>(d + 8 * g) = o00;
            *reinterpret_cast<f32x4*>(d + 8 * g + LDT) = o01;
            *reinterpret_cast<f32x4*>(d + 8 * g + W * LDT) = o10;
            *reinterpret_cast<f32x4*>(d + 8 * g + (W + 1) * LDT) = o11;
        }
    }
    STAMP(5);
    __syncthreads();
    STAMP(6);

    // ---- tails: 64 px x 64 ch per wave over the PX x C tile, weight fragments straight from L2, D[channel][pixel] ------
    f32x16 acc[MI][NI];
    auto gemm_tail = [&](const float* __restrict__ wsrc) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        const float* a0 = T + (wm * MI * 32 + li) * LDT + 4 * hh;
        const float* b0 = wsrc + (wn * NI) * (KS * 256) + 4 * lane;
        f32x4 a[2][MI];                                               // A fragments, read one k-slice ahead of their MFMAs
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * LDT);
#pragma unroll
        for (int ug = 0; ug < KS / 4; ++ug) {                         // bt[0] = first 4 k-slices: tail_prefetch()
            if (ug + 1 < KS / 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        bt[(ug + 1) & 1][u][ni] = *reinterpret_cast<const f32x4*>(b0 + ni * (KS * 256) + 256 * (4 * (ug + 1) + u));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ks = 4 * ug + u;
                if (ks + 1 < KS) {
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        a[(ks + 1) & 1][mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * LDT + 8 * (ks + 1));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bt[ug & 1][u][ni][r], a[ks & 1][mi][r], acc[mi][ni], 0, 0, 0);
            }
        }
    };
    auto acc_to_lds = [&]() {                                         // D[channel][pixel] -> T[pixel][channel], 128-bit writes
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = acc[mi][ni][4 * g + e];
                    *reinterpret_cast<f32x4*>(T + (wm * MI * 32 + mi * 32 + li) * LDT + (wn * NI + ni) * 32 + 8 * g + 4 * hh) = o;
                }
    };
    float* const trow = T + tj0 * LDT + 4 * cg;                       // row-coalesced view: + (RP i) rows

    gemm_tail(p.w3);                                                  // conv3
    STAMP(7);
    if (TAIL == 2) tail_prefetch(p.w1n);                              // ahead of the epilogue's stores
    __syncthreads();                                                  // every wave is done reading t2
    acc_to_lds();
    __syncthreads();
    STAMP(8);
    // out = conv3 * scale + bias4 + x, in place over the residual stream, whole pixel rows per RP-th of a workgroup
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        f32x4 t = rnd4(*reinterpret_cast<const f32x4*>(trow + RP * i * LDT));       // conv3 output cast
        t = t * p.t_scale;
        t = t + p.t_b4;
        t = t + res[i];
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(xr[(RP * i) / W] + ((RP * i) % W) * C));
        if (TAIL == 2) {                                              // next block's conv1 pre-op, back into T in place
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = rnd(elu_act(t[e] + p.n_b1a) + p.n_b1b);   // next conv1 input cast
            *reinterpret_cast<f32x4*>(trow + RP * i * LDT) = t;
        }
    }
    STAMP(9);
    if constexpr (TAIL == 2) {
        __syncthreads();
        STAMP(10);
        gemm_tail(p.w1n);                                             // next block's conv1
        STAMP(11);
        __syncthreads();                                              // every wave is done reading T
        acc_to_lds();
        __syncthreads();
        STAMP(12);
        float* yr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) yr[r] = p.y2 + (pix0 + (int64_t)r * Wimg) * C + 4 * cg;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4 t = *reinterpret_cast<const f32x4*>(trow + RP * i * LDT);
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = elu_act(rnd(t[e]) + p.n_b2a) + p.n_b2b;   // next conv1 output cast
            __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(yr[(RP * i) / W] + ((RP * i) % W) * C));
        }
        STAMP(13);
    }
}

// conv1 of the block at the head of a 'same'-block chain (the others get theirs from the previous block's tail):
//   y = ELU(conv1x1(ELU(x + pa) + pb) + aa) + ab,   C -> C channels, fp32          (conv_block.py:199-206)
// Same machinery as the tails above: the activated input tile is staged once in LDS ([PX][C + 4], row-coalesced
// 128-bit loads), the weights (fragment order) are the MFMA row operand, the result goes back through LDS to
// row-coalesced 128-bit stores.  An HBM-bound launch (2 C floats per pixel); the implicit-GEMM engine ran it at
// 2.3 TB/s because its per-workgroup setup is paid on four K steps.
template <int C>
__global__ __launch_bounds__((WinoCfg<C>::NT), 2)
void fixup_conv1_kernel(const float* __restrict__ x, const float* __restrict__ w1f, float pa, float pb, float aa, float ab,
                        float* __restrict__ y) {
    using K = WinoCfg<C>;
    constexpr int PX = K::PX, KS = K::KS, C4 = K::C4, RP = K::RP, LDT = K::LDT, WN = K::WN, MI = K::MI, NI = K::NI;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // T[PX][LDT]
    float* const T = lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int cg = tid % C4, p0 = tid / C4;
    const int64_t m0 = (int64_t)blockIdx.x * PX;
    const float* const xrow = x + (m0 + p0) * C + 4 * cg;
    float* const trow = T + p0 * LDT + 4 * cg;
    f32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const f32x4*>(xrow + (int64_t)(RP * i) * C);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = elu_act(v[i][e] + pa) + pb;
        *reinterpret_cast<f32x4*>(trow + RP * i * LDT) = v[i];
    }
    __syncthreads();
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    {
        const float* a0 = T + (wm * MI * 32 + li) * LDT + 4 * hh;
        const float* b0 = w1f + (wn * NI) * (KS * 256) + 4 * lane;
        f32x4 a[2][MI], bq[2][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * LDT);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bq[0][ni] = *reinterpret_cast<const f32x4*>(b0 + ni * (KS * 256));
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            if (u + 1 < KS) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[(u + 1) & 1][mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * LDT + 8 * (u + 1));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) bq[(u + 1) & 1][ni] = *reinterpret_cast<const f32x4*>(b0 + ni * (KS * 256) + 256 * (u + 1));
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[u & 1][ni][r], a[u & 1][mi][r], acc[mi][ni], 0, 0, 0);
        }
    }
    __syncthreads();                                                  // every wave is done reading T
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = elu_act(acc[mi][ni][4 * g + e] + aa) + ab;
                *reinterpret_cast<f32x4*>(T + (wm * MI * 32 + mi * 32 + li) * LDT + (wn * NI + ni) * 32 + 8 * g + 4 * hh) = o;
            }
    __syncthreads();
    float* const yrow = y + (m0 + p0) * C + 4 * cg;
#pragma unroll
    for (int i = 0; i < 16; ++i)
        *reinterpret_cast<f32x4*>(yrow + (int64_t)(RP * i) * C) = *reinterpret_cast<const f32x4*>(trow + RP * i * LDT);
}

// Persistent form of fixup_conv1_kernel (round 3).  One launch = 2 workgroups per CU that loop over the 128-pixel tiles; the
// NEXT tile's input rows are requested (16 x 16 B per thread, kept in registers) before this tile's MFMAs start and the
// phase boundaries are LDS-only barriers (a __syncthreads() would drain those loads), so a workgroup's HBM reads run under
// its own matrix work instead of only under its CU partner's: the one-shot form spends load + MFMA + store back to back
// (33 k cycles per tile against 16.4 k of MFMA issue and 12.8 k of HBM time per tile and CU).
template <int C>
__global__ __launch_bounds__((WinoCfg<C>::NT), 2)
void fixup_conv1p_kernel(const float* __restrict__ x, const float* __restrict__ w1f, float pa, float pb, float aa, float ab,
                         float* __restrict__ y, const int n_tiles) {
    using K = WinoCfg<C>;
    constexpr int PX = K::PX, KS = K::KS, C4 = K::C4, RP = K::RP, LDT = K::LDT, WN = K::WN, MI = K::MI, NI = K::NI;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // T[PX][LDT]
    float* const T = lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int cg = tid % C4, p0 = tid / C4;
    const int wm = wave / WN, wn = wave % WN;
    float* const trow = T + p0 * LDT + 4 * cg;
    const float* const a0 = T + (wm * MI * 32 + li) * LDT + 4 * hh;
    const float* const b0 = w1f + (wn * NI) * (KS * 256) + 4 * lane;
    int tile = blockIdx.x;
    f32x4 v[16];
    if (tile < n_tiles) {
        const float* const xrow = x + ((int64_t)tile * PX + p0) * C + 4 * cg;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xrow + (int64_t)(RP * i) * C));
    }
    for (; tile < n_tiles; tile += gridDim.x) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = elu_act(v[i][e] + pa) + pb;
            *reinterpret_cast<f32x4*>(trow + RP * i * LDT) = v[i];
        }
        vqae::lds_barrier();
        const int nt = tile + gridDim.x;
        if (nt < n_tiles) {                                            // next tile's rows: in flight under this tile's MFMAs
            const float* const xrow = x + ((int64_t)nt * PX + p0) * C + 4 * cg;
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xrow + (int64_t)(RP * i) * C));
        }
        f32x16 acc[MI][NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        {
            f32x4 a[2][MI], bq[2][NI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * LDT);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bq[0][ni] = *reinterpret_cast<const f32x4*>(b0 + ni * (KS * 256));
#pragma unroll 1
            for (int u2 = 0; u2 < KS; u2 += 2) {                       // a real loop (two k-slices per trip): bounds what hipcc hoists
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    const int u = u2 + uu;
                    if (u + 1 < KS) {
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) a[(uu + 1) & 1][mi] = *reinterpret_cast<const f32x4*>(a0 + mi * 32 * LDT + 8 * (u + 1));
#pragma unroll
                        for (int ni = 0; ni < NI; ++ni) bq[(uu + 1) & 1][ni] = *reinterpret_cast<const f32x4*>(b0 + ni * (KS * 256) + 256 * (u + 1));
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NI; ++ni)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[uu & 1][ni][r], a[uu & 1][mi][r], acc[mi][ni], 0, 0, 0);
                }
            }
        }
        vqae::lds_barrier();                                           // every wave is done reading T
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = elu_act(acc[mi][ni][4 * g + e] + aa) + ab;
                    *reinterpret_cast<f32x4*>(T + (wm * MI * 32 + mi * 32 + li) * LDT + (wn * NI + ni) * 32 + 8 * g + 4 * hh) = o;
                }
        vqae::lds_barrier();
        float* const yrow = y + ((int64_t)tile * PX + p0) * C + 4 * cg;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            *reinterpret_cast<f32x4*>(yrow + (int64_t)(RP * i) * C) = *reinterpret_cast<const f32x4*>(trow + RP * i * LDT);
        vqae::lds_barrier();                                           // T is rewritten by the next tile's staging
    }
}

// U[xi*4 + nu] = (G g G^T)[xi][nu] for g = w[n][k][3][3];  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
__global__ void wino_weight_kernel(const float* __restrict__ w, int c, int dt, float* __restrict__ U) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;              // over n * c + k
    if (i >= c * c) return;
    float g[3][3], t[4][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = vqae::round_dt(w[(int64_t)i * 9 + a * 3 + b], dt);     // autocast weight cast
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
        t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
        t[3][b] = g[2][b];
    }
    const int fo = frag_offset(i / c, i % c, c);
    const int64_t cc = (int64_t)c * c;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        U[(a * 4 + 0) * cc + fo] = t[a][0];
        U[(a * 4 + 1) * cc + fo] = 0.5f * (t[a][0] + t[a][1] + t[a][2]);
        U[(a * 4 + 2) * cc + fo] = 0.5f * (t[a][0] - t[a][1] + t[a][2]);
        U[(a * 4 + 3) * cc + fo] = t[a][2];
    }
}

// packed [c n][c k] (vqae_conv_pack_weight_f32) -> fragment order
__global__ void frag_weight_kernel(const float* __restrict__ w, int c, int sk, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c * c) return;
    out[frag_offset(i / c, i % c, c, sk)] = w[i];
}

template <int C, int DT, bool WIDE>
int launch_wino_w(const WinoK& k, bool chain, hipStream_t stream) {
    using K = WinoCfg<C>;
    static bool attr_set = false;
    if (!attr_set) {
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)wino_trunk_kernel<C, 1, DT, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)wino_trunk_kernel<C, 2, DT, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
        attr_set = true;
    }
    const unsigned grid = (unsigned)(k.M / K::PX);
    // executed matrix work: 16 GEMMs of K = C per 4 output pixels (K_eff = 4 C per pixel) + the 1x1 tails
    const double flops = 2.0 * (double)k.M * C * (4.0 * C + C + (chain ? C : 0));
    vqae::ProfScope prof(C >= 128 ? vqae::PROF_CONV3X3_TRUNK : vqae::PROF_NONE, stream, flops);
    if (chain) wino_trunk_kernel<C, 2, DT, WIDE><<<grid, K::NT, K::LDS_BYTES, stream>>>(k);
    else wino_trunk_kernel<C, 1, DT, WIDE><<<grid, K::NT, K::LDS_BYTES, stream>>>(k);
    prof.done();
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

template <int C, int DT = VQAE_DT_F32>
int launch_wino(const WinoK& k, bool chain, hipStream_t stream) {
    return k.Wimg == WinoCfg<C>::W ? launch_wino_w<C, DT, false>(k, chain, stream) : launch_wino_w<C, DT, true>(k, chain, stream);
}

}  // namespace

namespace vqae {

// fp32: C in {256, 128, 64, 32} on a grid whose width is a multiple of the workgroup's column span (32, 64, 128); 16-bit autocast
// modes: C = 32 only (the wider levels have a 16-bit MFMA kernel with fused tails, conv_mfma.hip)
bool wino_trunk_supported(int c, int h, int w, int dtype) {
    const int span = (c == 256 || c == 128) ? 32 : (c == 64 ? WinoCfg<64>::W : (c == 32 ? WinoCfg<32>::W : 0));
    if (span == 0 || (dtype != VQAE_DT_F32 && c != 32)) return false;
    return w >= span && w % span == 0 && h >= 4 && h % 4 == 0;
}

size_t wino_weight_floats(int c) { return (size_t)16 * c * c; }

// w_oihw_dev [c][c][3][3] (PyTorch layout, device) -> U_dev [16][c][c] (fragment order)
int wino_transform_weight(const float* w_oihw_dev, int c, int dtype, float* U_dev, hipStream_t stream) {
    wino_weight_kernel<<<(unsigned)ceil_div(c * c, 256), 256, 0, stream>>>(w_oihw_dev, c, dtype, U_dev);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// chain-head conv1 (fixup_conv1_kernel): fp32, C in {128, 64, 32}, M a multiple of the kernel's pixel tile
bool fixup_conv1_supported(int c, int64_t m) {
    if (c != 256 && c != 128 && c != 64 && c != 32) return false;
    const int px = c >= 128 ? 128 : (c == 64 ? WinoCfg<64>::PX : WinoCfg<32>::PX);
    return m > 0 && m % px == 0;
}

template <int C>
static int launch_conv1(const float* x, const float* w1f, float pa, float pb, float aa, float ab, float* y, int64_t m,
                        hipStream_t stream) {
    using K = WinoCfg<C>;
    static bool attr_set = false;
    if (!attr_set) {
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)fixup_conv1_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
        attr_set = true;
    }
    static bool attr_p = false;
    static int n_cu = 256;
    if (!attr_p) {
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)fixup_conv1p_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES));
        int dev = 0;
        hipDeviceProp_t prop;
        VQAE_HIP_CHECK(hipGetDevice(&dev));
        VQAE_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        if (prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
        attr_p = true;
    }
    // read per call (a chain head: a handful of launches per forward pass) so that tests can switch forms inside one process
    const char* e1 = getenv("VQAE_CONV1_ONESHOT");
    const char* e2 = getenv("VQAE_CONV1_PERSIST_MIN_TILES");
    const bool oneshot = e1 && atoi(e1);
    const int64_t n_tiles = m / K::PX;
    const int wgs = (int)std::min<int64_t>((K::LDS_BYTES * 2 <= 160 * 1024 ? 2 : 1) * n_cu, n_tiles);   // resident workgroups
    const int64_t min_tiles = e2 ? atoll(e2) : 4ll * n_cu;          // below ~2 tiles per workgroup the loop has nothing to overlap
    ProfScope prof(C >= 128 ? PROF_CONV1X1_TRUNK : PROF_NONE, stream, 2.0 * (double)m * C * C);
    if (oneshot || C > 128 || n_tiles < min_tiles)
        fixup_conv1_kernel<C><<<(unsigned)n_tiles, K::NT, K::LDS_BYTES, stream>>>(x, w1f, pa, pb, aa, ab, y);
    else
        fixup_conv1p_kernel<C><<<(unsigned)wgs, K::NT, K::LDS_BYTES, stream>>>(x, w1f, pa, pb, aa, ab, y, (int)n_tiles);
    prof.done();
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

int fixup_conv1(const float* x, const float* w1f, float pa, float pb, float aa, float ab, float* y, int64_t m, int c,
                hipStream_t stream) {
    VQAE_REQUIRE(x && w1f && y, VQAE_ERR_INVALID, "fixup_conv1: null pointer");
    VQAE_REQUIRE(fixup_conv1_supported(c, m), VQAE_ERR_UNSUPPORTED, "fixup_conv1: C = %d, M = %lld", c, (long long)m);
    VQAE_REQUIRE(m / 128 < (1ll << 31), VQAE_ERR_UNSUPPORTED, "fixup_conv1: too many pixels");
    if (c == 256) return launch_conv1<256>(x, w1f, pa, pb, aa, ab, y, m, stream);
    if (c == 128) return launch_conv1<128>(x, w1f, pa, pb, aa, ab, y, m, stream);
    if (c == 64) return launch_conv1<64>(x, w1f, pa, pb, aa, ab, y, m, stream);
    return launch_conv1<32>(x, w1f, pa, pb, aa, ab, y, m, stream);
}

// packed [c][c] 1x1 weights (device) -> fragment order (device); sk = 8 (fp32 MFMA k-slice) or 16 (16-bit MFMA)
int wino_frag_weight(const float* w_packed_dev, int c, int sk, float* out_dev, hipStream_t stream) {
    frag_weight_kernel<<<(unsigned)ceil_div(c * c, 256), 256, 0, stream>>>(w_packed_dev, c, sk, out_dev);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// Same contract as conv_trunk_tail (conv_mfma.hip), fp32, (C, W) in {(128, 32), (64, 64), (32, 128)}: t1 -> xio in place (+ t1_next).
// U, w3, w1n in fragment order.
int wino_trunk_tail(const float* t1, const float* U, const float* w3, float act_a, float act_b, float t_scale, float t_b4,
                    float* xio, const float* w1n, float n_b1a, float n_b1b, float n_b2a, float n_b2b, float* t1_next,
                    int batch, int h, int w, int c, int dtype, hipStream_t stream) {
    if (batch == 0) return VQAE_OK;
    VQAE_REQUIRE(t1 && U && w3 && xio && (!w1n || t1_next), VQAE_ERR_INVALID, "wino_trunk_tail: null pointer");
    VQAE_REQUIRE(wino_trunk_supported(c, h, w, dtype), VQAE_ERR_UNSUPPORTED, "wino_trunk_tail: C = %d, H = %d, W = %d, dtype %d", c, h, w, dtype);
    const int64_t M = (int64_t)batch * h * w;
    VQAE_REQUIRE(M < (1ll << 31) - 256, VQAE_ERR_UNSUPPORTED, "wino_trunk_tail: too many pixels");
    WinoK k;
    memset(&k, 0, sizeof(k));
    k.t1 = t1; k.U = U; k.w3 = w3; k.w1n = w1n; k.xio = xio; k.y2 = t1_next;
    k.H = h; k.Wimg = w; k.M = (int)M;
    k.act_a = act_a; k.act_b = act_b; k.t_scale = t_scale; k.t_b4 = t_b4;
    k.n_b1a = n_b1a; k.n_b1b = n_b1b; k.n_b2a = n_b2a; k.n_b2b = n_b2b;
#ifdef VQAE_WINO_TRACE
    static unsigned long long* trace = nullptr;
    if (!trace) (void)hipMalloc((void**)&trace, (size_t)4096 * 128 * 8);
    k.trace = (c == 128 && M / 128 <= 4096) ? trace : nullptr;
#endif
    const int rc = dtype == VQAE_DT_BF16 ? launch_wino<32, VQAE_DT_BF16>(k, w1n != nullptr, stream)
                 : dtype == VQAE_DT_F16 ? launch_wino<32, VQAE_DT_F16>(k, w1n != nullptr, stream)
                 : c == 256 ? launch_wino<256>(k, w1n != nullptr, stream)
                 : c == 128 ? launch_wino<128>(k, w1n != nullptr, stream)
                 : (c == 64 ? launch_wino<64>(k, w1n != nullptr, stream) : launch_wino<32>(k, w1n != nullptr, stream));
#ifdef VQAE_WINO_TRACE
    static int launches = 0;
    if (rc == VQAE_OK && c == 128 && w1n && M / 128 == 2048 && ++launches == 200) {   // one steady-state launch of the B = 256 bench
        (void)hipStreamSynchronize(stream);
        unsigned long long* host = new unsigned long long[(size_t)2048 * 128];
        (void)hipMemcpy(host, trace, (size_t)2048 * 128 * 8, hipMemcpyDeviceToHost);
        FILE* f = fopen("gpurun_out/wino_trace.bin", "wb");
        if (f) { fwrite(host, 8, (size_t)2048 * 128, f); fclose(f); }
        delete[] host;
    }
#endif
    return rc;
}

}  // namespace vqae
