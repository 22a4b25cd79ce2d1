// Vector-quantiser kernels for gfx950 (MI355X).
//
// Replaces EMAVectorQuantizer.forward (reference vq_ae/layers/vq.py:96-154, eval mode):
//   idx = argmin_k (sum_c |x_c - e_kc|^4)^(1/4)   (torch.cdist p = inputs.dim() = 4, vq.py:121-129)
//   q   = x + (e[idx] - x),  loss = beta * mean((x - e[idx])^2)
//
// The lookup is fp32-VALU bound (3 VALU ops per (row, code, channel) term; no MFMA form exists that
// is index-exact), so the design is:
//   tier 1  vq_tier1_kernel : exact-order fp32 evaluation  acc = fma(d*d, d*d, acc)  with
//           - codebook tile (256 codes x <=128 channels, transposed [c][k]) resident in LDS (128 KB),
//           - each lane owning 4 codes (one ds_read_b128 per channel feeds 4*RM terms),
//           - each wave owning RM rows whose channel values arrive through the SCALAR cache
//             (wave-uniform s_load -> SGPR operands of the VALU ops: no VGPR/LDS traffic for x),
//           - running (best, second-best, argmin) per row, wave butterfly at the end;
//   tier 2  vq_tier2_kernel : rows whose best/second-best gap is inside the fp32 evaluation noise
//           (or exact ties) are re-evaluated with the reference's bit recipe -- per-term
//           RN_f32(d^4) (via fp64), sequential fp32 adds, RN_f32(agg^(1/4)) finish, lowest index on
//           equal finished values (ATen cdist + argmin semantics) -- one wave per flagged row.
#include "common.h"

namespace {

constexpr int VQ_TK = 256;      // codes per LDS tile (4 per lane)
constexpr int VQ_TD = 128;      // channels per LDS tile
constexpr int VQ_WAVES = 16;    // waves per block (4 per SIMD: keeps the VALU issue port full)
constexpr int VQ_RM = 8;        // rows per wave
constexpr int VQ_ROWS_PER_BLOCK = VQ_WAVES * VQ_RM;

// embed [K][D] -> eT [D][Kpad] (zero-filled pad codes), so LDS tiles are straight row copies.
__global__ void vq_transpose_codebook_kernel(const float* __restrict__ embed, float* __restrict__ eT, int K,
                                             int Kpad, int D) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)D * Kpad) return;
    const int c = (int)(i / Kpad), k = (int)(i % Kpad);
    eT[i] = (k < K) ? embed[(int64_t)k * D + c] : 0.0f;
}

__device__ __forceinline__ void merge_best(float& b1, int& i1, float& b2, float ob1, int oi1, float ob2) {
    const bool other_wins = (ob1 < b1) || (ob1 == b1 && oi1 < i1);
    const float loser = other_wins ? b1 : ob1;
    b2 = fminf(fminf(b2, ob2), loser);
    if (other_wins) { b1 = ob1; i1 = oi1; }
}

// P = the Minkowski exponent = inputs.dim() (vq.py:121-129): 4 for the 2-D model's NCHW activations, 3 / 5 for [B, D, L] /
// [B, D, d, h, w] inputs.  One term is |x - e|^P in fp32: P = 4: d2 = d * d, fma(d2, d2, .); 3: fma(d2, |d|, .); 5: fma(d2 * d2, |d|, .).
template <int P>
__device__ __forceinline__ float vq_term(float x, float e, float acc) {
    float d = x - e;
    const float d2 = d * d;
    if (P == 4) return __builtin_fmaf(d2, d2, acc);
    if (P == 3) return __builtin_fmaf(d2, __builtin_fabsf(d), acc);
    return __builtin_fmaf(d2 * d2, __builtin_fabsf(d), acc);
}

template <int P>
__global__ __launch_bounds__(VQ_WAVES * 64)
void vq_tier1_kernel(const float* __restrict__ z, const float* __restrict__ eT, int64_t N, int K, int Kpad,
                     int D, float thr, int* __restrict__ idx32, float* __restrict__ margin,
                     int* __restrict__ flag_count, int* __restrict__ flag_list, const int* __restrict__ run_if) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [<=VQ_TD][VQ_TK]
    if (run_if && *run_if == 0) return;               // the matrix-pipe filter (vq_filter.hip) has done this launch
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    auto load_tile = [&](int kt, int dt, int td) {
        for (int i = threadIdx.x; i < td * (VQ_TK / 4); i += VQ_WAVES * 64) {
            const int c = i >> 6, j4 = i & 63;
            const float4 v = *reinterpret_cast<const float4*>(eT + (int64_t)(dt + c) * Kpad + kt + 4 * j4);
            *reinterpret_cast<float4*>(lds + c * VQ_TK + 4 * j4) = v;
        }
    };
    // Persistent workgroups: when the whole codebook is one LDS tile (K <= 256, D <= 128: cfg A / B) it is
    // loaded once per workgroup and the row groups stream past it with no barrier in the loop.
    const bool single_tile = (Kpad == VQ_TK) && (D <= VQ_TD);
    if (single_tile) {
        load_tile(0, 0, D);
        __syncthreads();
    }
    const int64_t n_groups = (N + VQ_ROWS_PER_BLOCK - 1) / VQ_ROWS_PER_BLOCK;
    for (int64_t rg = blockIdx.x; rg < n_groups; rg += gridDim.x) {
        const int64_t row0 = (rg * VQ_WAVES + wave) * VQ_RM;
        const float* __restrict__ xrow[VQ_RM];
#pragma unroll
        for (int r = 0; r < VQ_RM; ++r) {
            int64_t row = row0 + r;
            if (row > N - 1) row = N - 1;             // clamp (stores are masked below)
            xrow[r] = z + row * D;
        }
        float b1[VQ_RM], b2[VQ_RM];
        int i1[VQ_RM];
#pragma unroll
        for (int r = 0; r < VQ_RM; ++r) { b1[r] = INFINITY; b2[r] = INFINITY; i1[r] = 0; }

        for (int kt = 0; kt < Kpad; kt += VQ_TK) {
            float acc[VQ_RM][4];
#pragma unroll
            for (int r = 0; r < VQ_RM; ++r) { acc[r][0] = 0.f; acc[r][1] = 0.f; acc[r][2] = 0.f; acc[r][3] = 0.f; }

            for (int dt = 0; dt < D; dt += VQ_TD) {
                const int td = (D - dt < VQ_TD) ? (D - dt) : VQ_TD;
                if (!single_tile) {
                    __syncthreads();                  // previous tile fully consumed
                    load_tile(kt, dt, td);
                    __syncthreads();
                }
                // x values of the NEXT 4 channels are fetched (scalar loads) under the math of the current 4
                float4 xn[VQ_RM];
#pragma unroll
                for (int r = 0; r < VQ_RM; ++r) xn[r] = *reinterpret_cast<const float4*>(xrow[r] + dt);
                for (int c4 = 0; c4 < td; c4 += 4) {
                    float4 xv[VQ_RM];
#pragma unroll
                    for (int r = 0; r < VQ_RM; ++r) xv[r] = xn[r];
                    const int cn = (c4 + 4 < td) ? c4 + 4 : c4;     // last step re-reads its own channels
#pragma unroll
                    for (int r = 0; r < VQ_RM; ++r)   // wave-uniform address -> scalar load
                        xn[r] = *reinterpret_cast<const float4*>(xrow[r] + dt + cn);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float4 e = *reinterpret_cast<const float4*>(lds + (c4 + j) * VQ_TK + 4 * lane);
#pragma unroll
                        for (int r = 0; r < VQ_RM; ++r) {
                            const float x = (j == 0) ? xv[r].x : (j == 1) ? xv[r].y : (j == 2) ? xv[r].z : xv[r].w;
                            acc[r][0] = vq_term<P>(x, e.x, acc[r][0]);
                            acc[r][1] = vq_term<P>(x, e.y, acc[r][1]);
                            acc[r][2] = vq_term<P>(x, e.z, acc[r][2]);
                            acc[r][3] = vq_term<P>(x, e.w, acc[r][3]);
                        }
                    }
                }
            }
            // fold this tile's 4 codes into the running (best, second, argmin); k ascends -> strict '<'
#pragma unroll
            for (int r = 0; r < VQ_RM; ++r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = kt + 4 * lane + j;
                    const float s = (k < K) ? acc[r][j] : INFINITY;
                    if (s < b1[r]) { b2[r] = b1[r]; b1[r] = s; i1[r] = k; }
                    else if (s < b2[r]) { b2[r] = s; }
                }
            }
        }

#pragma unroll
        for (int r = 0; r < VQ_RM; ++r) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ob1 = __shfl_xor(b1[r], off, 64);
                const int oi1 = __shfl_xor(i1[r], off, 64);
                const float ob2 = __shfl_xor(b2[r], off, 64);
                merge_best(b1[r], i1[r], b2[r], ob1, oi1, ob2);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < VQ_RM; ++r) {
                const int64_t row = row0 + r;
                if (row < N) {
                    idx32[row] = i1[r];
                    const float gap = b2[r] - b1[r];
                    const float rel = (b2[r] > 0.f) ? gap / b2[r] : 0.f;
                    if (margin) margin[row] = rel;
                    if (!(gap > thr * b2[r])) {        // inside evaluation noise, exact tie, or NaN
                        const int slot = atomicAdd(flag_count, 1);
                        flag_list[slot] = (int)row;
                    }
                }
            }
        }
    }
}

// One wave per flagged row; bit recipe of ATen's scalar cdist loop (see oracle/vq_p4.c).
template <int P>
__global__ __launch_bounds__(256)
void vq_tier2_kernel(const float* __restrict__ z, const float* __restrict__ embed, int K, int D,
                     int* __restrict__ idx32, const int* __restrict__ flag_count,
                     const int* __restrict__ flag_list) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int n_waves = (int)((gridDim.x * blockDim.x) >> 6);
    const int nflag = *flag_count;
    for (int i = wave_global; i < nflag; i += n_waves) {
        const int row = flag_list[i];
        const float* __restrict__ x = z + (int64_t)row * D;
        float bf = INFINITY;
        int bi = 0x7fffffff;
        for (int k = lane; k < K; k += 64) {
            const float* __restrict__ e = embed + (int64_t)k * D;
            float agg = 0.0f;
            for (int c = 0; c < D; c += 4) {
                const float4 xv = *reinterpret_cast<const float4*>(x + c);
                const float4 ev = *reinterpret_cast<const float4*>(e + c);
                const float dx[4] = {xv.x - ev.x, xv.y - ev.y, xv.z - ev.z, xv.w - ev.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double dd = (double)fabsf(dx[j]);
                    const double q = dd * dd;              // exact (48 bits)
                    const float t = (float)(P == 4 ? q * q : (P == 3 ? q * dd : q * q * dd));   // RN_f32(|d|^P)
                    agg = agg + t;                         // sequential fp32 adds, channel order
                }
            }
            const float fin = P == 4 ? (float)sqrt(sqrt((double)agg)) : (float)pow((double)agg, 1.0 / P);   // RN_f32(agg^(1/P))
            if (fin < bf) { bf = fin; bi = k; }                 // k ascends per lane
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float obf = __shfl_xor(bf, off, 64);
            const int obi = __shfl_xor(bi, off, 64);
            if (obf < bf || (obf == bf && obi < bi)) { bf = obf; bi = obi; }
        }
        if (lane == 0 && bi != 0x7fffffff) idx32[row] = bi;
    }
}

// q = x + (e[idx] - x); per-block fp64 partial of sum((x - e[idx])^2).
__global__ __launch_bounds__(256)
void vq_gather_kernel(const float4* __restrict__ z4, const float4* __restrict__ embed4,
                      const int* __restrict__ idx32, int64_t total4, int D4, float4* __restrict__ q4,
                      double* __restrict__ partials) {
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / D4;
        const int c4 = (int)(i - n * D4);
        const int k = idx32[n];
        const float4 e = embed4[(int64_t)k * D4 + c4];
        const float4 x = z4[i];
        float4 d = {x.x - e.x, x.y - e.y, x.z - e.z, x.w - e.w};
        s += d.x * d.x; s += d.y * d.y; s += d.z * d.z; s += d.w * d.w;
        if (q4) {
            float4 q = {x.x + (e.x - x.x), x.y + (e.y - x.y), x.z + (e.z - x.z), x.w + (e.w - x.w)};
            q4[i] = q;
        }
    }
    __shared__ double red[256];
    red[threadIdx.x] = (double)s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256)
void vq_finalize_loss_kernel(const double* __restrict__ partials, int n_partials, double inv_count,
                             float commitment, float* __restrict__ loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n_partials; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(red[0] * inv_count) * commitment;
}

template <typename T>
__global__ void vq_write_idx_kernel(const int* __restrict__ idx32, int64_t N, T* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = (T)idx32[i];
}

template <typename T>
__global__ void embed_code_kernel(const T* __restrict__ idx, const float4* __restrict__ embed4, int64_t total4,
                                  int D4, int K, float4* __restrict__ out4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / D4;
        const int c4 = (int)(i - n * D4);
        int64_t k = (int64_t)idx[n];
        k = k < 0 ? 0 : (k >= K ? K - 1 : k);
        out4[i] = embed4[k * D4 + c4];
    }
}

template <typename T>
__global__ void embed_code_scalar_kernel(const T* __restrict__ idx, const float* __restrict__ embed, int64_t total, int D, int K,
                                         float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / D;
        int64_t k = (int64_t)idx[n];
        k = k < 0 ? 0 : (k >= K ? K - 1 : k);
        out[i] = embed[k * D + (i - n * D)];
    }
}

// ---- training-mode bookkeeping (vq.py:47-74) -------------------------------------------------
// One block per code: deterministic segmented sum.  Each of the 16 waves scans a contiguous slice of
// idx (ballot over 64 rows at a time) and adds the matching rows in ascending row order; the 16
// per-wave partials are then added in wave order.
template <typename T>
__global__ __launch_bounds__(1024)
void vq_code_stats_kernel(const float* __restrict__ z, const T* __restrict__ idx, int64_t N, int D,
                          float* __restrict__ counts, float* __restrict__ dw) {
    extern __shared__ float part[];              // [16][D + 1]
    const int k = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t per = (N + 15) / 16;
    const int64_t lo = wave * per, hi = (lo + per < N) ? lo + per : N;
    const int nd = (D + 63) / 64;                // channel slots per lane (D <= 512)
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float cnt = 0.f;
    for (int64_t base = lo; base < hi; base += 64) {
        const int64_t n = base + lane;
        const bool m = (n < hi) && ((int)idx[n] == k);
        unsigned long long mask = __ballot(m);
        while (mask) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float* __restrict__ row = z + (base + b) * D;
            cnt += 1.f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < nd && lane + 64 * j < D) acc[j] += row[lane + 64 * j];
        }
    }
    float* mine = part + wave * (D + 1);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j < nd && lane + 64 * j < D) mine[lane + 64 * j] = acc[j];
    if (lane == 0) mine[D] = cnt;
    __syncthreads();
    for (int c = threadIdx.x; c <= D; c += 1024) {
        float s = 0.f;
        for (int w = 0; w < 16; ++w) s += part[w * (D + 1) + c];
        if (c < D) dw[(int64_t)k * D + c] = s; else counts[k] = s;
    }
}

// EMA + Laplace smoothing (vq.py:60-74); single block.
__global__ __launch_bounds__(1024)
void vq_ema_update_kernel(float* __restrict__ embed, float* __restrict__ embed_avg, float* __restrict__ cluster_size,
                          const float* __restrict__ counts, const float* __restrict__ dw, int K, int D,
                          float decay, float alpha) {
    __shared__ float red[1024];
    const float omd = 1.f - decay;
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += 1024) {
        const float cs = cluster_size[k] * decay + counts[k] * omd;   // mul_(decay).add_(new, alpha=1-decay)
        cluster_size[k] = cs;
        s += cs;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 512; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float n = red[0];
    const float denom = n + (float)K * alpha;
    for (int64_t i = threadIdx.x; i < (int64_t)K * D; i += 1024) {
        const int k = (int)(i / D);
        const float ea = embed_avg[i] * decay + dw[i] * omd;
        embed_avg[i] = ea;
        const float cs = n * ((cluster_size[k] + alpha) / denom);
        embed[i] = ea / cs;
    }
}

struct VqWorkspace {
    int* flag_count;      // 4 ints (16 B, memset every call)
    int* flag_list;       // N
    int* idx32;           // N
    double* partials;     // 1024
    float* eT;            // D * Kpad
    void* ftab;           // vq_filter.hip's code-side operand table
};

inline VqWorkspace carve(void* ws, int64_t N, int K, int D) {
    VqWorkspace w;
    char* p = (char*)ws;
    w.flag_count = (int*)p; p += 256;
    w.partials = (double*)p; p += 1024 * sizeof(double);
    const int64_t Kpad = vqae::round_up(K, VQ_TK);
    w.eT = (float*)p; p += vqae::round_up((int64_t)D * Kpad * 4, 256);
    w.flag_list = (int*)p; p += vqae::round_up(N * 4, 256);
    w.idx32 = (int*)p; p += vqae::round_up(N * 4, 256);
    w.ftab = (void*)p;
    return w;
}

}  // namespace

namespace vqae {
bool vq_filter_supported(int K, int D);
size_t vq_filter_table_bytes(int K, int D);
int vq_filter_run(const float* z, const float* embed, int64_t N, int K, int D, float thr, int* idx32, int* flags, int* flag_list,
                  void* table, hipStream_t stream);
// shared with the fused projected quantiser (vq_proj.hip)
int vq_tier2_run(const float* z, const float* embed, int K, int D, int* idx32, const int* flag_count, const int* flag_list,
                 hipStream_t stream) {
    vq_tier2_kernel<4><<<256, 256, 0, stream>>>(z, embed, K, D, idx32, flag_count, flag_list);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// *loss = commitment * mean((z - embed[idx])^2) (vq.py:143); partials: >= 1024 doubles of scratch
int vq_loss_from_idx(const float* z, const float* embed, const int* idx32, int64_t N, int D, float commitment, double* partials,
                     float* loss, hipStream_t stream) {
    const int64_t total4 = N * (D / 4);
    const int nblk = (int)std::min<int64_t>(1024, ceil_div(total4, 256));
    vq_gather_kernel<<<nblk, 256, 0, stream>>>((const float4*)z, (const float4*)embed, idx32, total4, D / 4, (float4*)nullptr, partials);
    VQAE_LAUNCH_CHECK();
    vq_finalize_loss_kernel<<<1, 256, 0, stream>>>(partials, nblk, 1.0 / ((double)N * (double)D), commitment, loss);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

int vq_write_idx(const int* idx32, int64_t N, void* idx_out, int idx_dtype, hipStream_t stream) {
    const unsigned gi = (unsigned)ceil_div(N, 256);
    switch (idx_dtype) {
        case VQAE_IDX_I64: vq_write_idx_kernel<int64_t><<<gi, 256, 0, stream>>>(idx32, N, (int64_t*)idx_out); break;
        case VQAE_IDX_U8: vq_write_idx_kernel<uint8_t><<<gi, 256, 0, stream>>>(idx32, N, (uint8_t*)idx_out); break;
        case VQAE_IDX_U16: vq_write_idx_kernel<uint16_t><<<gi, 256, 0, stream>>>(idx32, N, (uint16_t*)idx_out); break;
        case VQAE_IDX_I32: vq_write_idx_kernel<int32_t><<<gi, 256, 0, stream>>>(idx32, N, (int32_t*)idx_out); break;
        default: return fail(VQAE_ERR_INVALID, "vq: bad idx_dtype %d", idx_dtype);
    }
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}
}  // namespace vqae

namespace {
__global__ void vq_pad_rows_kernel(const float* __restrict__ src, int64_t n, int d, int dp, float* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n * dp; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / dp;
        const int c = (int)(i - r * dp);
        dst[i] = c < d ? src[r * d + c] : 0.f;
    }
}
__global__ void vq_unpad_rows_kernel(const float* __restrict__ src, int64_t n, int d, int dp, float* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n * d; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / d;
        dst[i] = src[r * dp + (i - r * d)];
    }
}
}  // namespace

extern "C" size_t vqae_vq_workspace_bytes(int64_t n_rows, int n_codes, int dim) {
    const int64_t Kpad = vqae::round_up(n_codes, VQ_TK);
    const size_t ftab = (dim == 256 || dim == 128) ? (size_t)vqae::round_up((int64_t)vqae::vq_filter_table_bytes(n_codes, dim), 256) : 0;
    return (size_t)(256 + 1024 * sizeof(double) + vqae::round_up((int64_t)dim * Kpad * 4, 256) +
                    2 * vqae::round_up(n_rows * 4, 256) + 256) + ftab;
}

extern "C" int vqae_vq_forward_f32(const float* z, const float* embed, int64_t N, int K, int D, float commitment,
                                   void* idx_out, int idx_dtype, float* q, float* loss, float* margin, void* ws,
                                   void* stream_) {
    return vqae_vq_forward_p_f32(z, embed, N, K, D, 4, commitment, idx_out, idx_dtype, q, loss, margin, ws, stream_);
}

extern "C" int vqae_vq_forward_p_f32(const float* z, const float* embed, int64_t N, int K, int D, int p_norm, float commitment,
                                     void* idx_out, int idx_dtype, float* q, float* loss, float* margin, void* ws,
                                     void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VQAE_REQUIRE(p_norm >= 3 && p_norm <= 5, VQAE_ERR_UNSUPPORTED,
                 "vq_forward: p = %d (p = inputs.dim(): 3-D, 4-D and 5-D inputs are implemented)", p_norm);
    VQAE_REQUIRE(N >= 0 && N < (1ll << 31), VQAE_ERR_INVALID, "vq_forward: n_rows %lld out of range", (long long)N);
    VQAE_REQUIRE(N == 0 || (z && embed && idx_out && ws), VQAE_ERR_INVALID, "vq_forward: null pointer");
    VQAE_REQUIRE(K >= 1 && K <= 65536, VQAE_ERR_UNSUPPORTED, "vq_forward: n_codes %d unsupported", K);
    VQAE_REQUIRE(D >= 1 && D <= 4096, VQAE_ERR_UNSUPPORTED, "vq_forward: dim %d unsupported (1 .. 4096)", D);
    VQAE_REQUIRE(idx_dtype != VQAE_IDX_U8 || K <= 256, VQAE_ERR_INVALID, "vq_forward: u8 indices need K <= 256");
    if (N == 0) {
        if (loss) VQAE_HIP_CHECK(hipMemsetAsync(loss, 0, sizeof(float), stream));
        return VQAE_OK;
    }
    if (D % 4 != 0) {
        // The reference takes any embedding_dim (vq.py:121-129).  The kernels move rows as 16-byte vectors, so such a call runs on
        // zero-padded copies of z and of the codebook (stream-ordered scratch): a zero channel adds fma(0, 0, acc) = acc to every
        // distance -- bit for bit the same sums, indices, q -- and the loss (a mean over N * D elements) is rescaled by Dp / D.
        const int Dp = (int)vqae::round_up(D, 4);
        float *zp = nullptr, *ep = nullptr, *qp = nullptr;
        void* wsp = nullptr;
        VQAE_HIP_CHECK(hipMallocAsync((void**)&zp, (size_t)N * Dp * 4, stream));
        VQAE_HIP_CHECK(hipMallocAsync((void**)&ep, (size_t)K * Dp * 4, stream));
        VQAE_HIP_CHECK(hipMallocAsync((void**)&wsp, vqae_vq_workspace_bytes(N, K, Dp), stream));
        if (q) VQAE_HIP_CHECK(hipMallocAsync((void**)&qp, (size_t)N * Dp * 4, stream));
        vq_pad_rows_kernel<<<(unsigned)std::min<int64_t>(vqae::ceil_div(N * Dp, 256), 65536), 256, 0, stream>>>(z, N, D, Dp, zp);
        vq_pad_rows_kernel<<<(unsigned)std::min<int64_t>(vqae::ceil_div((int64_t)K * Dp, 256), 65536), 256, 0, stream>>>(embed, K, D, Dp, ep);
        VQAE_LAUNCH_CHECK();
        int rc = vqae_vq_forward_p_f32(zp, ep, N, K, Dp, p_norm, commitment * ((float)Dp / (float)D), idx_out, idx_dtype, qp, loss, margin, wsp,
                                       stream_);
        if (rc == VQAE_OK && q) {
            vq_unpad_rows_kernel<<<(unsigned)std::min<int64_t>(vqae::ceil_div(N * D, 256), 65536), 256, 0, stream>>>(qp, N, D, Dp, q);
            if (hipGetLastError() != hipSuccess) rc = vqae::fail(VQAE_ERR_HIP, "vq_forward: unpad launch failed");
        }
        (void)hipFreeAsync(zp, stream); (void)hipFreeAsync(ep, stream); (void)hipFreeAsync(wsp, stream);
        if (qp) (void)hipFreeAsync(qp, stream);
        return rc;
    }
    const int Kpad = (int)vqae::round_up(K, VQ_TK);
    VqWorkspace w = carve(ws, N, K, D);

    VQAE_HIP_CHECK(hipMemsetAsync(w.flag_count, 0, 16, stream));
    {
        const int64_t tot = (int64_t)D * Kpad;
        vq_transpose_codebook_kernel<<<(unsigned)vqae::ceil_div(tot, 256), 256, 0, stream>>>(embed, w.eT, K, Kpad, D);
        VQAE_LAUNCH_CHECK();
    }
    {
        static bool attr_set = false;
        const int td = D < VQ_TD ? D : VQ_TD;
        const size_t lds_bytes = (size_t)td * VQ_TK * sizeof(float);
        if (!attr_set) {
            VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)vq_tier1_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, VQ_TD * VQ_TK * (int)sizeof(float)));
            VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)vq_tier1_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, VQ_TD * VQ_TK * (int)sizeof(float)));
            VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)vq_tier1_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, VQ_TD * VQ_TK * (int)sizeof(float)));
            attr_set = true;
        }
        // evaluation-noise bound between tier-1 sums and the reference recipe's sums; p = 5 has one more rounded product
        const float thr = ((float)(p_norm > 4 ? p_norm : 4) * (float)D + 16.0f) * 5.9604645e-8f;
        // wide codebooks: the matrix-pipe filter + exact evaluation of the survivors (vq_filter.hip); it leaves flag_count[1] != 0
        // (and does nothing) when the codebook is outside the f16 range of its coefficients -- then, and only then, tier 1 runs
        const bool filt = p_norm == 4 && !margin && vqae::vq_filter_supported(K, D);
        if (filt) {
            vqae::ProfScope fprof(vqae::PROF_VQ_TIER1, stream, 3.0 * (double)N * K * D);
            const int frc = vqae::vq_filter_run(z, embed, N, K, D, thr, w.idx32, w.flag_count, w.flag_list, w.ftab, stream);
            fprof.done();
            if (frc) return frc;
        }
        // one workgroup per CU (128 KB LDS tile); persistent over row groups
        const unsigned grid = (unsigned)std::min<int64_t>(vqae::ceil_div(N, VQ_ROWS_PER_BLOCK), 256);
        vqae::ProfScope prof(filt ? vqae::PROF_NONE : vqae::PROF_VQ_TIER1, stream, 3.0 * (double)N * K * D);
        if (p_norm == 3)
            vq_tier1_kernel<3><<<grid, VQ_WAVES * 64, lds_bytes, stream>>>(z, w.eT, N, K, Kpad, D, thr, w.idx32, margin, w.flag_count, w.flag_list, nullptr);
        else if (p_norm == 5)
            vq_tier1_kernel<5><<<grid, VQ_WAVES * 64, lds_bytes, stream>>>(z, w.eT, N, K, Kpad, D, thr, w.idx32, margin, w.flag_count, w.flag_list, nullptr);
        else
            vq_tier1_kernel<4><<<grid, VQ_WAVES * 64, lds_bytes, stream>>>(z, w.eT, N, K, Kpad, D, thr, w.idx32, margin, w.flag_count, w.flag_list,
                                                                           filt ? w.flag_count + 1 : nullptr);
        prof.done();
        VQAE_LAUNCH_CHECK();
    }
    if (p_norm == 3) vq_tier2_kernel<3><<<256, 256, 0, stream>>>(z, embed, K, D, w.idx32, w.flag_count, w.flag_list);
    else if (p_norm == 5) vq_tier2_kernel<5><<<256, 256, 0, stream>>>(z, embed, K, D, w.idx32, w.flag_count, w.flag_list);
    else vq_tier2_kernel<4><<<256, 256, 0, stream>>>(z, embed, K, D, w.idx32, w.flag_count, w.flag_list);
    VQAE_LAUNCH_CHECK();

    if (q || loss) {
        const int64_t total4 = N * (D / 4);
        const int nblk = (int)std::min<int64_t>(1024, vqae::ceil_div(total4, 256));
        vq_gather_kernel<<<nblk, 256, 0, stream>>>((const float4*)z, (const float4*)embed, w.idx32, total4, D / 4,
                                                   (float4*)q, w.partials);
        VQAE_LAUNCH_CHECK();
        if (loss) {
            vq_finalize_loss_kernel<<<1, 256, 0, stream>>>(w.partials, nblk, 1.0 / ((double)N * (double)D),
                                                           commitment, loss);
            VQAE_LAUNCH_CHECK();
        }
    }
    return vqae::vq_write_idx(w.idx32, N, idx_out, idx_dtype, stream);
}

extern "C" int vqae_embed_code_f32(const void* idx, int idx_dtype, const float* embed, int64_t N, int K, int D,
                                   float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VQAE_REQUIRE(idx && embed && out, VQAE_ERR_INVALID, "embed_code: null pointer");
    if (N == 0) return VQAE_OK;
    if (D % 4 != 0) {                                    // any embedding_dim (vq.py:44-45): one element per thread
        const int64_t total = N * D;
        const unsigned nb = (unsigned)std::min<int64_t>(65536, vqae::ceil_div(total, 256));
        switch (idx_dtype) {
            case VQAE_IDX_I64: embed_code_scalar_kernel<int64_t><<<nb, 256, 0, stream>>>((const int64_t*)idx, embed, total, D, K, out); break;
            case VQAE_IDX_U8: embed_code_scalar_kernel<uint8_t><<<nb, 256, 0, stream>>>((const uint8_t*)idx, embed, total, D, K, out); break;
            case VQAE_IDX_U16: embed_code_scalar_kernel<uint16_t><<<nb, 256, 0, stream>>>((const uint16_t*)idx, embed, total, D, K, out); break;
            case VQAE_IDX_I32: embed_code_scalar_kernel<int32_t><<<nb, 256, 0, stream>>>((const int32_t*)idx, embed, total, D, K, out); break;
            default: return vqae::fail(VQAE_ERR_INVALID, "embed_code: bad idx_dtype %d", idx_dtype);
        }
        VQAE_LAUNCH_CHECK();
        return VQAE_OK;
    }
    const int64_t total4 = N * (D / 4);
    const int nblk = (int)std::min<int64_t>(4096, vqae::ceil_div(total4, 256));
    switch (idx_dtype) {
        case VQAE_IDX_I64: embed_code_kernel<int64_t><<<nblk, 256, 0, stream>>>((const int64_t*)idx, (const float4*)embed, total4, D / 4, K, (float4*)out); break;
        case VQAE_IDX_U8: embed_code_kernel<uint8_t><<<nblk, 256, 0, stream>>>((const uint8_t*)idx, (const float4*)embed, total4, D / 4, K, (float4*)out); break;
        case VQAE_IDX_U16: embed_code_kernel<uint16_t><<<nblk, 256, 0, stream>>>((const uint16_t*)idx, (const float4*)embed, total4, D / 4, K, (float4*)out); break;
        case VQAE_IDX_I32: embed_code_kernel<int32_t><<<nblk, 256, 0, stream>>>((const int32_t*)idx, (const float4*)embed, total4, D / 4, K, (float4*)out); break;
        default: return vqae::fail(VQAE_ERR_INVALID, "embed_code: bad idx_dtype %d", idx_dtype);
    }
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

extern "C" int vqae_vq_code_stats_f32(const float* z, const void* idx, int idx_dtype, int64_t N, int K, int D,
                                      float* counts, float* dw, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VQAE_REQUIRE(z && idx && counts && dw, VQAE_ERR_INVALID, "code_stats: null pointer");
    VQAE_REQUIRE(D >= 1 && D <= 512, VQAE_ERR_UNSUPPORTED, "code_stats: dim %d unsupported (<= 512)", D);
    const size_t lds = (size_t)16 * (D + 1) * sizeof(float);
    switch (idx_dtype) {
        case VQAE_IDX_I64: vq_code_stats_kernel<int64_t><<<K, 1024, lds, stream>>>(z, (const int64_t*)idx, N, D, counts, dw); break;
        case VQAE_IDX_U8: vq_code_stats_kernel<uint8_t><<<K, 1024, lds, stream>>>(z, (const uint8_t*)idx, N, D, counts, dw); break;
        case VQAE_IDX_U16: vq_code_stats_kernel<uint16_t><<<K, 1024, lds, stream>>>(z, (const uint16_t*)idx, N, D, counts, dw); break;
        case VQAE_IDX_I32: vq_code_stats_kernel<int32_t><<<K, 1024, lds, stream>>>(z, (const int32_t*)idx, N, D, counts, dw); break;
        default: return vqae::fail(VQAE_ERR_INVALID, "code_stats: bad idx_dtype %d", idx_dtype);
    }
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

extern "C" int vqae_vq_ema_update_f32(float* embed, float* embed_avg, float* cluster_size, const float* counts,
                                      const float* dw, int K, int D, float decay, float alpha, void* ws,
                                      void* stream_) {
    (void)ws;
    hipStream_t stream = (hipStream_t)stream_;
    VQAE_REQUIRE(embed && embed_avg && cluster_size && counts && dw, VQAE_ERR_INVALID, "ema_update: null pointer");
    vq_ema_update_kernel<<<1, 1024, 0, stream>>>(embed, embed_avg, cluster_size, counts, dw, K, D, decay, alpha);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}
