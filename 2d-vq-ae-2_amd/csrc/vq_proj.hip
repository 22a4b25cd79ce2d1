// Fused projected vector quantiser for the reference's default codebook (projection_dim = 8):
//   ProjectedEMAVectorQuantizer2d.forward = proj_out(VQ(proj_in(x)))            (vq_ae/layers/vq.py:190-192)
//     z   = proj_in(x)                       1x1 conv C -> 8 with bias          (vq.py:178-182)
//     idx = argmin_k (sum_j |z_j - e_kj|^4)^(1/4), lowest k on ties             (vq.py:121-129, p = inputs.dim() = 4)
//     q   = z + (e[idx] - z)                 straight-through forward value     (vq.py:130,146)
//     out = proj_out(q)                      1x1 conv 8 -> C with bias          (vq.py:183-187)
//     loss = beta * mean((z - e[idx])^2)     in the 8-D space                   (vq.py:143)
// in ONE pass over the activation: a row of x (C fp32 channels) is read once, its 8-D projection never leaves registers
// for the distance loop, and the C-channel output row is written once -- this is the one configuration where the
// lookup's algorithmic HBM traffic (2 C * 4 B per row) is comparable to its arithmetic.
//
// Work split: ONE ROW PER LANE.  Everything a lane needs besides its own row is wave-uniform -- proj_in / proj_out
// weights, biases, the 8-float codebook rows (17 KB in all at K = 256, C = 128) -- and is staged once per workgroup in
// LDS, from where every lane reads the SAME address (a broadcast read: no bank conflicts, no cross-lane traffic):
// z_j += x_c * Wt_in[c][j];  d = z_j - e_kj;  out_c += q_j * W_out[c][j].  (Through the scalar cache instead -- SGPR
// operands -- the kernel ran at 0.9 TB/s: s_load results return out of order, so every group of scalar loads cost a
// full lgkmcnt(0) round trip that 4 waves per SIMD could not cover.)
// Per (row, code): 8 sub + 8 mul + 8 fma in channel order (the tier-1 recipe of vq_kernels.hip) + 5 to keep
// (best, second best, argmin).  Rows whose best/second gap lies inside the evaluation noise are flagged and re-evaluated
// bit-faithfully by vq_tier2_kernel on the stored 8-D z (32 B per row); their output rows are then patched.
//
// 16-bit autocast modes (DT): proj_in / proj_out are convolutions, so their operands and results are rounded to the
// 16-bit type exactly as vqae_conv2d_f32 does (weights / biases arrive pre-rounded); the distance, q and the loss stay fp32.
#include "common.h"

namespace vqae {
int vq_tier2_run(const float* z, const float* embed, int K, int D, int* idx32, const int* flag_count, const int* flag_list,
                 hipStream_t stream);
int vq_loss_from_idx(const float* z, const float* embed, const int* idx32, int64_t N, int D, float commitment, double* partials,
                     float* loss, hipStream_t stream);
int vq_write_idx(const int* idx32, int64_t N, void* out, int idx_dtype, hipStream_t stream);
}  // namespace vqae

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int PD = 8;                                    // projection_dim

struct VqProjK {
    const float* __restrict__ x;         // [N][C]
    const float* __restrict__ wt_in;     // [C][8]   proj_in.weight transposed
    const float* __restrict__ b_in;      // [8]
    const float* __restrict__ embed;     // [K][8]
    const float* __restrict__ w_out;     // [C][8]   proj_out.weight (PyTorch [C][8][1][1])
    const float* __restrict__ b_out;     // [C]
    float* __restrict__ out;             // [N][C]
    float* __restrict__ z;               // [N][8]
    int* __restrict__ idx32;             // [N]
    float* __restrict__ margin;          // [N] or null
    int* __restrict__ flag_count;
    int* __restrict__ flag_list;
    int64_t N;
    int C, K;
    float thr;
};

template <int DT> __device__ __forceinline__ float rnd16(float v) {
    if (DT == VQAE_DT_BF16) return (float)(__bf16)v;
    if (DT == VQAE_DT_F16) return (float)(_Float16)v;
    return v;
}

constexpr int VP_THREADS = 128;

template <int DT>
__global__ __launch_bounds__(VP_THREADS)
void vq_proj_fused_kernel(const VqProjK p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const s_win = lds;                                                // [C][8]
    float* const s_wout = s_win + p.C * PD;                                  // [C][8]
    float* const s_bout = s_wout + p.C * PD;                                 // [C]
    float* const s_bin = s_bout + p.C;                                       // [8]
    float* const s_emb = s_bin + PD;                                         // [Kpad4][8]
    {
        const int tid = threadIdx.x;
        for (int i = tid; i < p.C * PD / 4; i += VP_THREADS) {
            reinterpret_cast<f32x4*>(s_win)[i] = reinterpret_cast<const f32x4*>(p.wt_in)[i];
            reinterpret_cast<f32x4*>(s_wout)[i] = reinterpret_cast<const f32x4*>(p.w_out)[i];
        }
        for (int i = tid; i < p.C / 4; i += VP_THREADS) reinterpret_cast<f32x4*>(s_bout)[i] = reinterpret_cast<const f32x4*>(p.b_out)[i];
        if (tid < PD) s_bin[tid] = p.b_in[tid];
        const int kpad = (p.K + 3) & ~3;
        for (int i = tid; i < kpad * PD / 4; i += VP_THREADS)               // pad codes: copies of the last one (never win a strict '<')
            reinterpret_cast<f32x4*>(s_emb)[i] = reinterpret_cast<const f32x4*>(p.embed)[i < p.K * PD / 4 ? i : (p.K - 1) * PD / 4 + (i & 1)];
    }
    __syncthreads();

    const int64_t row_raw = (int64_t)blockIdx.x * VP_THREADS + threadIdx.x;
    const bool live = row_raw < p.N;
    const int64_t row = live ? row_raw : p.N - 1;                            // tail lanes recompute the last row, store nothing
    const float* __restrict__ xr = p.x + row * p.C;

    // ---- z = proj_in(x): k-ordered fp32 fma chain per output, bias last (the order of vqae_conv2d_f32's epilogue) ----
    float z[PD];
#pragma unroll
    for (int j = 0; j < PD; ++j) z[j] = 0.f;
#pragma unroll 8
    for (int c = 0; c < p.C; c += 4) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xe = rnd16<DT>(xv[e]);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(s_win + (c + e) * PD);         // broadcast reads
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(s_win + (c + e) * PD + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { z[j] = __builtin_fmaf(xe, w0[j], z[j]); z[j + 4] = __builtin_fmaf(xe, w1[j], z[j + 4]); }
        }
    }
#pragma unroll
    for (int j = 0; j < PD; ++j) z[j] = rnd16<DT>(z[j] + s_bin[j]);
    if (live) {
        *reinterpret_cast<f32x4*>(p.z + row * PD) = (f32x4){z[0], z[1], z[2], z[3]};
        *reinterpret_cast<f32x4*>(p.z + row * PD + 4) = (f32x4){z[4], z[5], z[6], z[7]};
    }

    // ---- tier-1 argmin over the codebook, 4 codes in flight to cover the fma latency -----------------------------------
    float b1 = INFINITY, b2 = INFINITY;
    int i1 = 0;
    const int kpad = (p.K + 3) & ~3;
#pragma unroll 2
    for (int k0 = 0; k0 < kpad; k0 += 4) {
        float s4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x4 e0 = *reinterpret_cast<const f32x4*>(s_emb + (k0 + u) * PD);
            const f32x4 e1 = *reinterpret_cast<const f32x4*>(s_emb + (k0 + u) * PD + 4);
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                float d = z[j] - (j < 4 ? e0[j] : e1[j - 4]);
                d = d * d;
                a = __builtin_fmaf(d, d, a);
            }
            s4[u] = a;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {                                        // k ascends: strict '<' keeps the lowest index
            const bool better = s4[u] < b1;
            if (k0 + u < p.K) b2 = fminf(b2, fmaxf(b1, s4[u]));              // a pad code (copy of K - 1) must not become "second best"
            i1 = better ? k0 + u : i1;
            b1 = fminf(b1, s4[u]);
        }
    }
    if (live) {
        p.idx32[row] = i1;
        const float gap = b2 - b1;
        if (p.margin) p.margin[row] = (b2 > 0.f) ? gap / b2 : 0.f;
        if (!(gap > p.thr * b2)) {                                           // inside evaluation noise, exact tie, or NaN
            const int slot = atomicAdd(p.flag_count, 1);
            p.flag_list[slot] = (int)row;
        }
    }

    // ---- q = z + (e[idx] - z); out = round(sum_j round(q_j) * W_out[c][j] + b_out[c]) ------------------------------------
    float qr[PD];
    {
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(s_emb + i1 * PD), e1 = *reinterpret_cast<const f32x4*>(s_emb + i1 * PD + 4);
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            const float ev = j < 4 ? e0[j] : e1[j - 4];
            qr[j] = rnd16<DT>(z[j] + (ev - z[j]));
        }
    }
    // Output rows: computed 2 rows per wave instruction, lane L -> row 2 i + (L >> 5), channels 4 (L & 31) .. + 3 of a 128-channel
    // slab, so a store instruction writes two whole 512-byte row pieces.  (One row per lane -- 16 B per lane at a C * 4-byte
    // stride, 64 line pieces per instruction -- made the stores cost more than everything else in the kernel.)  q of the
    // row arrives by a cross-lane read; the fma chain per output is the one of the row-per-lane form, bit for bit.
    const int lane = threadIdx.x & 63;
    const int64_t row_w0 = (int64_t)blockIdx.x * VP_THREADS + (threadIdx.x & ~63);       // first row of this wave
    for (int c0 = 0; c0 < p.C; c0 += 128) {
        const int cl = c0 + 4 * (lane & 31);
        const bool cvalid = cl < p.C;
        const int cs = cvalid ? cl : 0;
        f32x4 wv0[4], wv1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            wv0[e] = *reinterpret_cast<const f32x4*>(s_wout + (cs + e) * PD);
            wv1[e] = *reinterpret_cast<const f32x4*>(s_wout + (cs + e) * PD + 4);
        }
        const f32x4 bo = *reinterpret_cast<const f32x4*>(s_bout + cs);
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int src = 2 * i + (lane >> 5);
            float qs[PD];
#pragma unroll
            for (int j = 0; j < PD; ++j) qs[j] = __shfl(qr[j], src);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) a = __builtin_fmaf(qs[j], wv0[e][j], a);
#pragma unroll
                for (int j = 0; j < 4; ++j) a = __builtin_fmaf(qs[j + 4], wv1[e][j], a);
                o[e] = rnd16<DT>(a + bo[e]);
            }
            if (cvalid && row_w0 + src < p.N) *reinterpret_cast<f32x4*>(p.out + (row_w0 + src) * p.C + cl) = o;
        }
    }
}

// ---- round 3: 16 rows per wave step, projections on the fp32 MFMA, search filtered on the f16 MFMA, persistent waves ----------
// The row-per-lane kernel above runs at 30-34 % of HBM: its x loads / out stores are 16 B per lane at a C * 4-byte stride (64
// line pieces per instruction), every wave of the chip is resident at once and walks load -> search -> store in lock-step
// (HBM idles during the search), and search + projections are ~2 200 vector instructions per 16 rows (PMC: the vector ALUs
// are the busiest unit; at 2 cycles per instruction they alone need ~40 us of the 95).  Here a wave step is 16 rows,
// lane = (g = lane >> 4, r = lane & 15):
//   proj_in   z^T[8 (of 16)][16 rows] = W_in[8 x C] . x^T[C x 16]  on v_mfma_f32_16x16x4_f32: lane (g, r) feeds x[row r][16 s +
//             4 g + e] (one 16-byte load per s: 64 contiguous bytes per row and instruction) and ends up with z_{4g..4g+3}(row r)
//   filter    sum_j (z_j - e_kj)^4 - sum_j z_j^4 = [z^3 | z^2 | z | 1] . [-4 e | 6 e^2 | -4 e^3 | sum e^4]: ONE
//             v_mfma_f32_16x16x32_f16 per 16 rows x 16 codes (K' = 25 of 32), and a second one on the operands' absolute values:
//             every operand is rounded to f16 once, so the exact score lies within eta * A of S (A = the sum of the products'
//             magnitudes, eta = 1.02 * 2^-10) -- a PER-CODE interval.  (The per-row Hoelder bound of vq_filter.hip keeps ~10 codes
//             per row in 8 dimensions: measured slower than the exact scan.)  Codes whose lower bound does not exceed the row's
//             smallest upper bound (+ the near-tie window) survive: 1-3 per row, into a per-wave pair list
//   exact     the survivors are evaluated with the tier-1 recipe (d = z - e, d^2, fma(d^2, d^2, .) in channel order: the same
//             instructions as above, bit for bit); two rounds of 64-bit LDS minima on (distance bits << 32 | code) leave the
//             best and second best of each row, ties to the lower index; near ties are flagged for tier 2 as before.  The
//             filter only decides which codes are LOOKED AT.  Rows outside the f16 range of z^3, a full pair list, K > 256 or
//             not a multiple of 16, or a codebook with NaN / Inf: the wave step scans every code exactly (the loop below).
//   proj_out  out^T[16 ch][16 rows] = W_out[16 x 8] . q^T[8 x 16] per 16-channel block, bias as the accumulator's start:
//             lane (g, r) holds 4 consecutive channels of row r -> one 16-byte store, 64 contiguous bytes per row
// and a wave loops over steps (persistent grid) with the NEXT step's x already requested (32 registers).
constexpr int VP16_CAP = 128;                                                 // (row, code) pairs per wave step

template <int C, int DT, int WPS, bool PF, int NWV>
__global__ __launch_bounds__(64 * NWV, WPS)
void vq_proj16_kernel(const VqProjK p, const int n_units, const int use_filter) {
    constexpr int NS = C / 16;                                                // 16-channel slices
    typedef unsigned long long u64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int kpad = (p.K + 3) & ~3;
    float* const s_emb = lds;                                                 // [Kpad4][8]
    float* const s_win = s_emb + kpad * PD;                                   // [C][16 + 1]: W_in^T, j padded to 16 with zeros
    float* const s_bout = s_win + C * 17;                                     // [C]
    float* const s_bin = s_bout + C;                                          // [16] (8 real)
    float* const s_red = s_bin + 16;                                          // [8] block reductions
    float* const s_wout = s_red + 16;                                          // [C][8 + 1]: W_out rows (proj_out's A operand)
    char* const s_wave = reinterpret_cast<char*>(s_wout + C * 9 + (C & 3 ? 0 : 0));   // per-wave scratch, 1 296 B each (16-byte aligned)
    constexpr int WSCR = 16 * PD * 4 + 16 * 8 * 2 + VP16_CAP * 4 + 16;
    f16x8* const s_tab = reinterpret_cast<f16x8*>(s_wave + NWV * WSCR);         // [K / 16][64 lanes]: filter's code operand (K <= 256)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    float* const w_z = reinterpret_cast<float*>(s_wave + wave * WSCR);        // [16 rows][8]
    u64* const w_best = reinterpret_cast<u64*>(w_z + 16 * PD);                // [16]
    u64* const w_second = w_best + 16;                                        // [16]
    unsigned* const w_list = reinterpret_cast<unsigned*>(w_second + 16);      // [CAP] row << 16 | code
    int* const w_cnt = reinterpret_cast<int*>(w_list + VP16_CAP);
    for (int i = tid; i < kpad * PD / 4; i += 64 * NWV)                            // pad codes: copies of the last one, masked in the search
        reinterpret_cast<f32x4*>(s_emb)[i] = reinterpret_cast<const f32x4*>(p.embed)[i < p.K * PD / 4 ? i : (p.K - 1) * PD / 4 + (i & 1)];
    for (int i = tid; i < C * 16; i += 64 * NWV) s_win[(i >> 4) * 17 + (i & 15)] = (i & 15) < PD ? p.wt_in[(i >> 4) * PD + (i & 15)] : 0.f;
    for (int i = tid; i < C; i += 64 * NWV) s_bout[i] = p.b_out[i];
    if (tid < 16) s_bin[tid] = tid < PD ? p.b_in[tid] : 0.f;
    for (int i = tid; i < C * PD; i += 64 * NWV) s_wout[(i >> 3) * 9 + (i & 7)] = p.w_out[i];
    const float* const wo_p = s_wout + r * 9 + g;                             // proj_out's A operand: W_out[16 b + r][g], [..][4 + g]
    __syncthreads();
    // ---- filter set-up: scale 8 / max |e| (the score is homogeneous: the argmin does not move), code operand table in fragment order
    bool filt = use_filter && p.K <= 256 && (p.K & 15) == 0;
    float sc = 1.f;
    if (filt) {
        float m = 0.f;
        for (int i = tid; i < p.K * PD; i += 64 * NWV) { const float a = __builtin_fabsf(s_emb[i]); m = a > m ? a : (a != a ? INFINITY : m); }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) s_red[wave] = m;
        __syncthreads();
        float emax = s_red[0];
#pragma unroll
        for (int w_ = 1; w_ < NWV; ++w_) emax = __builtin_fmaxf(emax, s_red[w_]);
        filt = emax > 0.f && emax < INFINITY;                                 // workgroup-uniform
        sc = filt ? 8.f / emax : 1.f;
        for (int i = tid; i < (p.K / 16) * 64; i += 64 * NWV) {                    // fragment of lane (gg, c) of code block blk
            const int blk = i >> 6, lp = i & 63, c = lp & 15, gg = lp >> 4;
            const float* e = s_emb + (16 * blk + c) * PD;
            f16x8 h;
            float e4 = 0.f;
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const float t = sc * e[j], t2 = t * t;
                e4 += t2 * t2;
                h[j] = (_Float16)(gg == 0 ? -4.f * t : (gg == 1 ? 6.f * t2 : (gg == 2 ? -4.f * t2 * t : 0.f)));
            }
            if (gg == 3) h[0] = (_Float16)e4;                                 // <= 8 * 8^4 = 32 768: inside the f16 range
            s_tab[i] = h;
        }
        __syncthreads();
    }

    const int stride = gridDim.x * NWV;
    int u = blockIdx.x * NWV + wave;
    auto load_unit = [&](int unit, f32x4 (&xv)[NS]) {
        int64_t row = (int64_t)unit * 16 + r;
        row = row < p.N ? row : p.N - 1;                                      // tail rows recompute the last row, store nothing
        const float* __restrict__ xr = p.x + row * C + 4 * g;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) xv[s_] = *reinterpret_cast<const f32x4*>(xr + 16 * s_);
    };
    auto proj_in = [&](const f32x4 (&xv)[NS]) -> f32x4 {                      // -> z_{4g+i}(row r), i = 0..3 (lanes g < 2)
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        const float* wa = s_win + (4 * g) * 17 + r;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float w = wa[(16 * s_ + e) * 17];
                const float xe = rnd16<DT>(xv[s_][e]);
                if ((s_ & 1) == 0) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, xe, a0, 0, 0, 0);
                else a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, xe, a1, 0, 0, 0);
            }
        f32x4 zc;
        const f32x4 bi = *reinterpret_cast<const f32x4*>(s_bin + 4 * g);
#pragma unroll
        for (int i = 0; i < 4; ++i) zc[i] = rnd16<DT>((a0[i] + a1[i]) + bi[i]);
        return zc;
    };
    auto wave_lds_sync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };   // a wave's LDS accesses complete in order
    auto tier1 = [&](const float (&z)[PD], const float* __restrict__ e) -> float {     // the exact evaluation, channel order
        const f32x4 e0 = *reinterpret_cast<const f32x4*>(e), e1 = *reinterpret_cast<const f32x4*>(e + 4);
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < PD; ++j) {
            float d = z[j] - (j < 4 ? e0[j] : e1[j - 4]);
            d = d * d;
            a = __builtin_fmaf(d, d, a);
        }
        return a;
    };

    f32x4 xv[NS];
    f32x4 zc = {0.f, 0.f, 0.f, 0.f};
    if (PF && u < n_units) {
        load_unit(u, xv);
        zc = proj_in(xv);
    }
    for (; u < n_units; u += stride) {
        const int un = u + stride;
        if (PF) {
            if (un < n_units) load_unit(un, xv);                              // in flight under this step's search (32 registers)
        } else {                                                              // no prefetch: fewer registers, more resident waves
            load_unit(u, xv);
            zc = proj_in(xv);
        }
        const int64_t row = (int64_t)u * 16 + r;
        const bool live = row < p.N;
        if (live && g < 2) *reinterpret_cast<f32x4*>(p.z + row * PD + 4 * g) = zc;
        float z[PD];
#pragma unroll
        for (int i = 0; i < 4; ++i) { z[i] = __shfl(zc[i], r); z[4 + i] = __shfl(zc[i], 16 + r); }

        float b1 = INFINITY, b2 = INFINITY;
        int i1 = 0;
        bool full = !filt;                                                    // wave-uniform: scan every code exactly
        if (filt) {
            // ---- row operand [z^3 | z^2 | z | 1] (scaled), per-row error bound --------------------------------------------------------
            float zs[PD], x4 = 0.f, amax = 0.f;
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                zs[j] = z[j] * sc;
                const float t = zs[j] * zs[j];
                x4 = __builtin_fmaf(t, t, x4);
                amax = __builtin_fmaxf(amax, __builtin_fabsf(zs[j]));
            }
            f16x8 fa;
#pragma unroll
            for (int j = 0; j < PD; ++j) {
                const float t2 = zs[j] * zs[j];
                fa[j] = (_Float16)(g == 0 ? t2 * zs[j] : (g == 1 ? t2 : (g == 2 ? zs[j] : (j == 0 ? 1.f : 0.f))));
            }
            const bool out_of_range = !(amax < 30.f);                         // NaN / Inf rows too
            if (g == 0) {                                                     // z rows + per-row state for the exact stage
                *reinterpret_cast<f32x4*>(w_z + r * PD) = (f32x4){z[0], z[1], z[2], z[3]};
                *reinterpret_cast<f32x4*>(w_z + r * PD + 4) = (f32x4){z[4], z[5], z[6], z[7]};
                w_best[r] = ~0ull;
                w_second[r] = ~0ull;
            }
            if (lane == 0) *w_cnt = 0;
            // ---- interval filter.  S = [f] . [c] on the f16 MFMA and A = [|f|] . [|c|] (the same operands with the sign bits cleared):
            // every operand was rounded to f16 once (2^-11 relative), so the exact score lies in [S - eta A, S + eta A], eta = 1.02 * 2^-10
            // (A is itself computed from the rounded operands: the 2 % cover that and the fp32 accumulation of 25 terms).  T = the
            // smallest upper bound of the row; the true best has S_best <= T, so every code within the exact evaluation's near-tie
            // window w of the best satisfies  S_k - eta A_k <= T + w.  Two passes over the code blocks (an MFMA is 16 cycles; keeping
            // the scores would cost 128 registers).
            constexpr float ETA = 1.02f / 1024.f;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            auto absf16 = [](const f16x8& v) -> f16x8 {
                u32x4 b = __builtin_bit_cast(u32x4, v);
                b &= 0x7FFF7FFFu;
                return __builtin_bit_cast(f16x8, b);
            };
            const f16x8 fa_abs = absf16(fa);
            const int nblk = p.K >> 4;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            float mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
#pragma unroll 4
            for (int blk = 0; blk < nblk; ++blk) {
                const f16x8 cb = s_tab[blk * 64 + lane];
                const f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, cb, zero4, 0, 0, 0);
                const f32x4 av = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa_abs, absf16(cb), zero4, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) mn[i] = __builtin_fminf(mn[i], __builtin_fmaf(ETA, av[i], sv[i]));
            }
            float lim[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int off = 1; off <= 8; off <<= 1) mn[i] = __builtin_fminf(mn[i], __shfl_xor(mn[i], off));
                const float x4r = __shfl(x4, 4 * g + i);                      // rows 4 g + i: their sum z^4 sits in lane 4 g + i
                lim[i] = mn[i] + 1e-5f * (__builtin_fabsf(mn[i]) + x4r) + 1e-30f;   // w: 1e-5 of the distance scale >> thr = 2.9e-6
            }
            wave_lds_sync();
            // ---- survivors -> pair list ---------------------------------------------------------------------------------------------------
#pragma unroll 4
            for (int blk = 0; blk < nblk; ++blk) {
                const f16x8 cb = s_tab[blk * 64 + lane];
                const f32x4 sv = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, cb, zero4, 0, 0, 0);
                const f32x4 av = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa_abs, absf16(cb), zero4, 0, 0, 0);
                f32x4 lo;
#pragma unroll
                for (int i = 0; i < 4; ++i) lo[i] = __builtin_fmaf(-ETA, av[i], sv[i]);
                if (lo[0] <= lim[0] || lo[1] <= lim[1] || lo[2] <= lim[2] || lo[3] <= lim[3]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (lo[i] <= lim[i]) {
                            const int slot = atomicAdd(w_cnt, 1);
                            if (slot < VP16_CAP) w_list[slot] = ((unsigned)(4 * g + i) << 16) | (unsigned)(16 * blk + r);
                        }
                }
            }
            wave_lds_sync();
            const int np = *w_cnt;
            full = np > VP16_CAP || __any(out_of_range) || np < 1;            // wave-uniform
            if (!full) {
                // ---- exact distances of the survivors, one pair per lane and trip -------------------------------------------------------------
                auto key_of = [](float dv, unsigned code) -> u64 { return ((u64)__float_as_uint(dv) << 32) | code; };   // d >= 0: bit order = value order
                for (int base = 0; base < np; base += 64) {
                    const int t = base + lane;
                    const unsigned pr = w_list[t < np ? t : np - 1];
                    const int prow = (int)(pr >> 16), pcode = (int)(pr & 0xFFFFu);
                    float zr[PD];
                    const f32x4 z0 = *reinterpret_cast<const f32x4*>(w_z + prow * PD), z1 = *reinterpret_cast<const f32x4*>(w_z + prow * PD + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { zr[j] = z0[j]; zr[4 + j] = z1[j]; }
                    const u64 key = key_of(tier1(zr, s_emb + pcode * PD), (unsigned)pcode);
                    if (t < np) atomicMin(&w_best[prow], key);
                    wave_lds_sync();
                    if (t < np && key != w_best[prow]) atomicMin(&w_second[prow], key);
                    wave_lds_sync();
                }
                if (np > 64) {                                                // several trips: redo `second` against the final `best`
                    if (g == 0) w_second[r] = ~0ull;
                    wave_lds_sync();
                    for (int base = 0; base < np; base += 64) {
                        const int t = base + lane;
                        const unsigned pr = w_list[t < np ? t : np - 1];
                        const int prow = (int)(pr >> 16), pcode = (int)(pr & 0xFFFFu);
                        float zr[PD];
                        const f32x4 z0 = *reinterpret_cast<const f32x4*>(w_z + prow * PD), z1 = *reinterpret_cast<const f32x4*>(w_z + prow * PD + 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) { zr[j] = z0[j]; zr[4 + j] = z1[j]; }
                        const u64 key = key_of(tier1(zr, s_emb + pcode * PD), (unsigned)pcode);
                        if (t < np && key != w_best[prow]) atomicMin(&w_second[prow], key);
                    }
                    wave_lds_sync();
                }
                const u64 kb = w_best[r], ks = w_second[r];
                b1 = __uint_as_float((unsigned)(kb >> 32));
                i1 = (int)(kb & 0xFFFFu);
                b2 = ks == ~0ull ? 3.0e38f : __uint_as_float((unsigned)(ks >> 32));    // a lone survivor: no near tie
                full = __any(kb == ~0ull);                                    // a row without a survivor (NaN scores): scan exactly
            }
        }
        if (full) {
            // ---- exact scan of every code: lane (g, r) takes the codes k = 4 i + g, 4 codes in flight ----------------------------------------
            b1 = INFINITY; b2 = INFINITY; i1 = 0;
            const float* eb = s_emb + g * PD;
#pragma unroll 1
            for (int k0 = 0; k0 < kpad; k0 += 16) {
                float s4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) s4[t] = tier1(z, eb + (k0 + 4 * t) * PD);
#pragma unroll
                for (int t = 0; t < 4; ++t) {                                 // k ascends: strict '<' keeps the lowest index
                    const int k = k0 + 4 * t + g;
                    const float sv = k < p.K ? s4[t] : INFINITY;              // pad codes / the ragged last group never win
                    const bool better = sv < b1;
                    b2 = fminf(b2, fmaxf(b1, sv));
                    i1 = better ? k : i1;
                    b1 = fminf(b1, sv);
                }
            }
            // merge the four code subsets of a row (lanes r, 16 + r, 32 + r, 48 + r): smaller sum, lower index on equal sums
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {
                const float ob1 = __shfl_xor(b1, m), ob2 = __shfl_xor(b2, m);
                const int oi1 = __shfl_xor(i1, m);
                const bool take = ob1 < b1 || (ob1 == b1 && oi1 < i1);
                b2 = fminf(fminf(b2, ob2), fmaxf(b1, ob1));
                i1 = take ? oi1 : i1;
                b1 = fminf(b1, ob1);
            }
        }
        if (live && g == 0) {
            p.idx32[row] = i1;
            const float gap = b2 - b1;
            if (p.margin) p.margin[row] = (b2 > 0.f) ? gap / b2 : 0.f;
            if (!(gap > p.thr * b2)) {                                        // inside evaluation noise, exact tie, or NaN
                const int slot = atomicAdd(p.flag_count, 1);
                p.flag_list[slot] = (int)row;
            }
        }
        // ---- q = z + (e[idx] - z) for j = g and 4 + g; out^T = W_out . q^T + b_out ----------------------------------------------------
        const float zl = g == 0 ? z[0] : (g == 1 ? z[1] : (g == 2 ? z[2] : z[3]));
        const float zh = g == 0 ? z[4] : (g == 1 ? z[5] : (g == 2 ? z[6] : z[7]));
        const float el = s_emb[i1 * PD + g], eh = s_emb[i1 * PD + 4 + g];
        const float ql = rnd16<DT>(zl + (el - zl)), qh = rnd16<DT>(zh + (eh - zh));
        float* __restrict__ orow = p.out + row * C + 4 * g;
#pragma unroll
        for (int b = 0; b < NS; ++b) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(s_bout + 16 * b + 4 * g);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wo_p[16 * b * 9], ql, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wo_p[16 * b * 9 + 4], qh, acc, 0, 0, 0);
            if (DT != VQAE_DT_F32) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = rnd16<DT>(acc[i]);
            }
            if (live) *reinterpret_cast<f32x4*>(orow + 16 * b) = acc;
        }
        if (PF && un < n_units) zc = proj_in(xv);
    }
}

template <int C, int DT>
int launch_vq_proj16(const VqProjK& k, hipStream_t stream) {
    const int n_units = (int)vqae::ceil_div(k.N, 16);
    static const bool nofilter = getenv("VQAE_VQ16_NOFILTER") && atoi(getenv("VQAE_VQ16_NOFILTER"));
    const int use_filter = !nofilter && !k.margin && k.K <= 256 && (k.K & 15) == 0;   // the margin output wants the true second best
    // VQAE_VQ16_WPS: waves per SIMD the kernel is compiled and launched for.  3 keeps the register prefetch of the next step's
    // rows (256-thread workgroups, 3 per CU); 4 drops it for more resident waves (512-thread workgroups, 2 per CU)
    static const int wps = getenv("VQAE_VQ16_WPS") ? atoi(getenv("VQAE_VQ16_WPS")) : 4;
    const int nwv = wps >= 5 ? 10 : (wps == 4 ? 8 : 4);
    const size_t lds_bytes = ((size_t)((k.K + 3) & ~3) * PD + (size_t)C * 17 + C + 16 + 16 + (size_t)C * 9) * sizeof(float)
                             + (size_t)nwv * (16 * PD * 4 + 16 * 8 * 2 + VP16_CAP * 4 + 16) + (use_filter ? (size_t)(k.K / 16) * 1024 : 0);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        VQAE_HIP_CHECK(hipGetDevice(&dev));
        VQAE_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int wg_per_cu = std::max(1, std::min(wps >= 4 ? 2 : 3, (int)(160 * 1024 / (lds_bytes + 256))));
    const unsigned grid = (unsigned)std::min<int64_t>(vqae::ceil_div(n_units, nwv), (int64_t)n_cu * wg_per_cu);
    vqae::ProfScope prof(vqae::PROF_VQ_TIER1, stream, (double)k.N * (29.0 * k.K + 4.0 * PD * k.C));
    if (wps >= 5) vq_proj16_kernel<C, DT, 5, false, 10><<<grid, 640, lds_bytes, stream>>>(k, n_units, use_filter);
    else if (wps == 4) vq_proj16_kernel<C, DT, 4, false, 8><<<grid, 512, lds_bytes, stream>>>(k, n_units, use_filter);
    else vq_proj16_kernel<C, DT, 3, true, 4><<<grid, 256, lds_bytes, stream>>>(k, n_units, use_filter);
    prof.done();
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// rows re-assigned by tier 2: recompute their output rows from the stored z and the corrected index
template <int DT>
__global__ __launch_bounds__(64)
void vq_proj_patch_kernel(const VqProjK p) {
    const int nflag = *p.flag_count;
    for (int i = blockIdx.x * 64 + threadIdx.x; i < nflag; i += gridDim.x * 64) {
        const int row = p.flag_list[i];
        const float* __restrict__ zr = p.z + (int64_t)row * PD;
        const float* __restrict__ eb = p.embed + p.idx32[row] * PD;
        float q[PD];
#pragma unroll
        for (int j = 0; j < PD; ++j) q[j] = zr[j] + (eb[j] - zr[j]);
        float qr[PD];
#pragma unroll
        for (int j = 0; j < PD; ++j) qr[j] = rnd16<DT>(q[j]);
        float* __restrict__ orow = p.out + (int64_t)row * p.C;
        for (int c = 0; c < p.C; ++c) {                                      // divergent lanes: plain vector loads
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < PD; ++j) a = __builtin_fmaf(qr[j], p.w_out[c * PD + j], a);
            orow[c] = rnd16<DT>(a + p.b_out[c]);
        }
    }
}

template <int DT>
int launch_vq_proj(const VqProjK& k, hipStream_t stream) {
    const unsigned grid = (unsigned)vqae::ceil_div(k.N, VP_THREADS);
    const size_t lds_bytes = ((size_t)k.C * (2 * PD + 1) + PD + (size_t)((k.K + 3) & ~3) * PD) * sizeof(float);
    vqae::ProfScope prof(vqae::PROF_VQ_TIER1, stream, (double)k.N * (29.0 * k.K + 4.0 * PD * k.C));
    vq_proj_fused_kernel<DT><<<grid, VP_THREADS, lds_bytes, stream>>>(k);
    prof.done();
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

}  // namespace

extern "C" size_t vqae_vq_projected_workspace_bytes(int64_t n_rows) {
    return (size_t)(256 + 1024 * sizeof(double) + 2 * vqae::round_up(n_rows * 4, 256) + vqae::round_up(n_rows * PD * 4, 256) + 256);
}

extern "C" int vqae_vq_projected_f32(const float* x, const float* wt_in, const float* b_in, const float* embed, const float* w_out,
                                     const float* b_out, int64_t N, int C, int D, int K, float commitment, int dtype,
                                     void* idx_out, int idx_dtype, float* out, float* z_out, float* loss, float* margin,
                                     void* ws, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VQAE_REQUIRE(N >= 0 && N < (1ll << 31), VQAE_ERR_INVALID, "vq_projected: n_rows %lld out of range", (long long)N);
    VQAE_REQUIRE(D == PD, VQAE_ERR_UNSUPPORTED, "vq_projected: projection_dim %d (only %d is fused)", D, PD);
    VQAE_REQUIRE(C >= 4 && C % 4 == 0 && C <= 512, VQAE_ERR_UNSUPPORTED, "vq_projected: channels %d", C);
    VQAE_REQUIRE(K >= 1 && K <= 1024, VQAE_ERR_UNSUPPORTED, "vq_projected: n_codes %d (the codebook must fit LDS: <= 1024)", K);
    VQAE_REQUIRE(dtype >= VQAE_DT_F32 && dtype <= VQAE_DT_F16, VQAE_ERR_INVALID, "vq_projected: dtype %d", dtype);
    VQAE_REQUIRE(idx_dtype != VQAE_IDX_U8 || K <= 256, VQAE_ERR_INVALID, "vq_projected: u8 indices need K <= 256");
    VQAE_REQUIRE(N == 0 || (x && wt_in && b_in && embed && w_out && b_out && idx_out && out && ws), VQAE_ERR_INVALID,
                 "vq_projected: null pointer");
    if (N == 0) {
        if (loss) VQAE_HIP_CHECK(hipMemsetAsync(loss, 0, sizeof(float), stream));
        return VQAE_OK;
    }
    char* w = (char*)ws;
    VqProjK k;
    k.flag_count = (int*)w; w += 256;
    double* partials = (double*)w; w += 1024 * sizeof(double);
    k.flag_list = (int*)w; w += vqae::round_up(N * 4, 256);
    k.idx32 = (int*)w; w += vqae::round_up(N * 4, 256);
    k.z = z_out ? z_out : (float*)w;
    k.x = x; k.wt_in = wt_in; k.b_in = b_in; k.embed = embed; k.w_out = w_out; k.b_out = b_out; k.out = out;
    k.margin = margin; k.N = N; k.C = C; k.K = K;
    k.thr = (4.0f * (float)PD + 16.0f) * 5.9604645e-8f;           // as vqae_vq_forward_f32
    VQAE_HIP_CHECK(hipMemsetAsync(k.flag_count, 0, 16, stream));
    static const bool v1 = getenv("VQAE_VQ_PROJ_V1") && atoi(getenv("VQAE_VQ_PROJ_V1"));
    int rc;
    if (!v1 && C == 128 && K <= 1024)                                // the reference default (conf/model/vq_ae.yaml: 8 * 2^4 channels)
        rc = dtype == VQAE_DT_BF16 ? launch_vq_proj16<128, VQAE_DT_BF16>(k, stream)
           : dtype == VQAE_DT_F16 ? launch_vq_proj16<128, VQAE_DT_F16>(k, stream) : launch_vq_proj16<128, VQAE_DT_F32>(k, stream);
    else
        rc = dtype == VQAE_DT_BF16 ? launch_vq_proj<VQAE_DT_BF16>(k, stream)
           : dtype == VQAE_DT_F16 ? launch_vq_proj<VQAE_DT_F16>(k, stream) : launch_vq_proj<VQAE_DT_F32>(k, stream);
    if (rc) return rc;
    if ((rc = vqae::vq_tier2_run(k.z, embed, K, PD, k.idx32, k.flag_count, k.flag_list, stream))) return rc;
    if (dtype == VQAE_DT_BF16) vq_proj_patch_kernel<VQAE_DT_BF16><<<4, 64, 0, stream>>>(k);
    else if (dtype == VQAE_DT_F16) vq_proj_patch_kernel<VQAE_DT_F16><<<4, 64, 0, stream>>>(k);
    else vq_proj_patch_kernel<VQAE_DT_F32><<<4, 64, 0, stream>>>(k);
    VQAE_LAUNCH_CHECK();
    if (loss && (rc = vqae::vq_loss_from_idx(k.z, embed, k.idx32, N, PD, commitment, partials, loss, stream))) return rc;
    return vqae::vq_write_idx(k.idx32, N, idx_out, idx_dtype, stream);
}
