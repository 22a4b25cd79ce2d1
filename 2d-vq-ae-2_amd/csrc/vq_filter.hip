// Matrix-pipe FILTER in front of the exact p = 4 codebook search (EMAVectorQuantizer.forward, reference vq_ae/layers/vq.py:121-129)
// for the wide codebooks (D = 256 channels: cfg C), where vq_tier1_kernel's exact-order evaluation -- 3 vector
// instructions per (row, code, channel) -- is pure VALU work (0.53 ms at N = 262 144, K = 256, D = 128; 4.1 ms at K = 1024, D = 256).
//
//   sum_c (z_c - e_kc)^4 - sum_c z_c^4 = sum_c (-4 e_kc z_c^3 + 6 e_kc^2 z_c^2 - 4 e_kc^3 z_c) + sum_c e_kc^4
// is one f16 GEMM rows x codes with K' = 3 D + 1: row operand [z^3 | z^2 | z | 1], code operand [-4 e | 6 e^2 | -4 e^3 | sum e^4].
// S_k differs from the exact value (minus the row constant) by at most
//   eps = 2^-10 * ((a + b)^4 - a^4),  a = (sum_c z_c^4)^(1/4),  b = max_k (sum_c e_kc^4)^(1/4)
// (every operand is rounded to f16 once: relative 2^-11 each, i.e. 2^-10 on a product; Hoelder bounds the three cross sums;
// the fp32 accumulation of 3 D + 1 terms is 2^-15 of the same sum at D = 256), so every code whose exact distance is within
// the near-tie window of the best satisfies S_k <= min_k S_k + 2 eps (taken with a 1.5x margin).  Those survivors -- one to
// three of K -- are evaluated exactly in fp32 (d = z - e, d^2, fma(d^2, d^2, .)), ties to the lower index, and rows whose two
// best are closer than the fp32 evaluation noise go to vq_tier2_kernel's bit-faithful recipe exactly as before: the filter
// changes which codes are LOOKED AT, never the value a decision is taken on.  Rows with |z| >= 30 after scaling (f16 range of z^3), without
// survivors or cut off by a full list take every code; rows and codes are scaled by 8 / max |e| first (the score is homogeneous:
// the argmin does not move) so the coefficients always fit; a codebook with NaN / Inf hands the whole launch back to
// vq_tier1_kernel (device-side flag, no host round trip).
//
// Kernel: a 256-thread workgroup owns RT rows (128; 64 at D = 256) -- their operand rows in LDS as f16 -- and all K codes: a wave
// owns a quarter of the 32-code tiles and ALL row tiles, so a code fragment (1 KiB from L2, fragment order) feeds RT / 32 MFMAs.
// Pass 1: min S per row (wave minima merged through LDS); pass 2: the MFMAs again (they are ~1/20 of the vector work they
// replace), survivors appended to a (row, code) pair list in LDS; then 8 lanes per pair evaluate the exact distances (several
// pairs per group in flight) and two rounds of 64-bit LDS atomic minima (distance bits << 32 | code) leave the best and the
// second best pair of every row.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
using vqae::lds_barrier;

constexpr int VF_PCAP = 1024;                            // (row, survivor) pairs kept per workgroup tile
typedef unsigned long long u64;

// flags[3] = float bits of max |e| over the codebook: rows and codes are scaled by 8 / max |e| before the f16 operands are formed
// (the score is homogeneous of degree 4, the argmin does not move), so the coefficients -4 e^3 ... always fit
__global__ void vqf_emax_kernel(const float* __restrict__ embed, int64_t n, int* __restrict__ flags) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = __builtin_fabsf(embed[i]);
        m = a > m ? a : (a != a ? INFINITY : m);                    // NaN counts as out of range
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(&flags[3], __float_as_int(m));
}

__device__ __forceinline__ float vqf_scale(const int* flags) {
    const float emax = __int_as_float(flags[3]);
    return (emax > 0.f && emax < INFINITY) ? 8.f / emax : 1.f;
}

// code-side operand, fragment order [Kp / 32][KS][64 lanes][8] f16 (KS = 3 D / 16 + 1 k-steps); flags[1] != 0: coefficient out of
// the f16 range; flags[2] = float bits of max_k sum_c e_kc^4
__global__ void vqf_table_kernel(const float* __restrict__ embed, int K, int Kp, int D, _Float16* __restrict__ tab, int* __restrict__ flags) {
    const int KS = 3 * D / 16 + 1;
    const int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= (int64_t)(Kp / 32) * KS * 64) return;
    const int lp = (int)(ch & 63);
    const int u = (int)((ch >> 6) % KS), ct = (int)((ch >> 6) / KS);
    const int code = 32 * ct + (lp & 31), k0 = 16 * u + 8 * (lp >> 5);
    const float sc = vqf_scale(flags);
    float v[8];
    if (code >= K) {                                     // pad codes: a score no row reaches
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (k0 + j == 3 * D) ? 60000.f : 0.f;
    } else {
        const float* e = embed + (int64_t)code * D;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            float r = 0.f;
            if (k < D) r = -4.f * (sc * e[k]);
            else if (k < 2 * D) { const float t = sc * e[k - D]; r = 6.f * t * t; }
            else if (k < 3 * D) { const float t = sc * e[k - 2 * D]; r = -4.f * t * t * t; }
            else if (k == 3 * D) {
                float e4 = 0.f;
                for (int c = 0; c < D; ++c) { const float t0 = sc * e[c], t = t0 * t0; e4 += t * t; }
                r = e4;
                atomicMax(&flags[2], __float_as_int(e4));            // e4 >= 0: integer order = float order
            }
            v[j] = r;
        }
    }
    bool ok = true;
    f16x8 h;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ok = ok && (__builtin_fabsf(v[j]) <= 60000.f); h[j] = (_Float16)v[j]; }
    if (!ok) flags[1] = 1;                               // NaN too
    *reinterpret_cast<f16x8*>(tab + ch * 8) = h;
}

struct VqfK {
    const float* __restrict__ z;         // [N][D]
    const float* __restrict__ embed;     // [K][D]
    const _Float16* __restrict__ tab;
    int64_t N;
    int K, Kp;
    float thr;
    int* __restrict__ idx32;
    int* __restrict__ flags;             // [0] flagged-row count, [1] coefficient range flag, [2] bits of max sum (s e)^4, [3] bits of max |e|
    int* __restrict__ flag_list;
    int n_tiles;
};

template <int D, int RT>
__global__ __launch_bounds__(256, 2)
void vqf_main_kernel(const VqfK p) {
    constexpr int KS = 3 * D / 16 + 1;                  // k-steps
    constexpr int XS = (3 * D + 16) * 2 + 16;           // operand row bytes in LDS (odd number of 16-B slots)
    constexpr int MTN = RT / 32;                         // row tiles
    constexpr int RING = D == 128 ? 5 : 7;              // code fragments in flight per wave; divides KS (25 / 49)
    static_assert(KS % RING == 0, "ring depth must divide the k-steps");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const Xs = lds;
    float* const epsS = reinterpret_cast<float*>(Xs + RT * XS);      // [RT]
    float* const limS = epsS + RT;                                   // [RT]
    float* const minw = limS + RT;                                   // [4][RT]
    int* const cnt = reinterpret_cast<int*>(minw + 4 * RT);          // [RT]
    int* const rflag = cnt + RT;                                     // [RT]
    u64* const best = reinterpret_cast<u64*>(rflag + RT);            // [RT] (distance bits << 32 | code) of the best / second best
    u64* const second = best + RT;                                   // [RT]
    unsigned* const pairs = reinterpret_cast<unsigned*>(second + RT);   // [VF_PCAP] row << 16 | code
    float* const pdist = reinterpret_cast<float*>(pairs + VF_PCAP);  // [VF_PCAP] exact distances
    int* const npairs = reinterpret_cast<int*>(pdist + VF_PCAP);
    if (p.flags[1] != 0) return;                         // coefficients outside f16: vq_tier1_kernel takes this launch
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const float bq = __builtin_sqrtf(__builtin_sqrtf(__int_as_float(p.flags[2])));
    const float sc = vqf_scale(p.flags);
    const int CTW = p.Kp / 128;                          // 32-code tiles per wave (Kp is a multiple of 128)

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int64_t row0 = (int64_t)tile * RT;
        // ---- operand rows [z^3 | z^2 | z | 1, 0 ..] as f16 ---------------------------------------------------------------------
        constexpr int NZ = RT * (D / 4) / 256;                             // 16: every row of the tile requested before any is used
        f32x4 zv[NZ];
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int idx = tid + 256 * i;
            int64_t row = row0 + idx / (D / 4);
            row = row < p.N ? row : p.N - 1;
            zv[i] = *reinterpret_cast<const f32x4*>(p.z + row * D + 4 * (idx % (D / 4)));
        }
#pragma unroll
        for (int i = 0; i < NZ; ++i) {
            const int idx = tid + 256 * i;
            const int r = idx / (D / 4), c4 = idx % (D / 4);
            const f32x4 v = zv[i] * sc;
            const f32x4 v2 = v * v, v3 = v2 * v;
            char* d = Xs + r * XS + c4 * 8;
            *reinterpret_cast<f16x4*>(d) = __builtin_convertvector(v3, f16x4);
            *reinterpret_cast<f16x4*>(d + 2 * D) = __builtin_convertvector(v2, f16x4);
            *reinterpret_cast<f16x4*>(d + 4 * D) = __builtin_convertvector(v, f16x4);
        }
        for (int r = tid; r < RT; r += 256) {
            const f16x8 one = {(_Float16)1.f, 0, 0, 0, 0, 0, 0, 0}, zero = {0, 0, 0, 0, 0, 0, 0, 0};
            *reinterpret_cast<f16x8*>(Xs + r * XS + 6 * D) = one;
            *reinterpret_cast<f16x8*>(Xs + r * XS + 6 * D + 16) = zero;
            cnt[r] = 0;
            best[r] = ~0ull;
            second[r] = ~0ull;
        }
        if (tid == 0) *npairs = 0;
        lds_barrier();
        for (int r = tid; r < RT; r += 256) {            // row norms from the rounded operands (margin below covers the rounding)
            float x4 = 0.f, amax = 0.f;
            for (int c = 0; c < D; c += 8) {
                const f16x8 q2 = *reinterpret_cast<const f16x8*>(Xs + r * XS + 2 * D + c * 2);
                const f16x8 q1 = *reinterpret_cast<const f16x8*>(Xs + r * XS + 4 * D + c * 2);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float t = (float)q2[j];
                    x4 += t * t;
                    amax = __builtin_fmaxf(amax, __builtin_fabsf((float)q1[j]));
                }
            }
            x4 *= 1.004f;
            const float a = __builtin_sqrtf(__builtin_sqrtf(x4)), t = a + bq, t2 = t * t;
            epsS[r] = (t2 * t2 - x4 * 0.99f) * (1.5f / 1024.f) + 1e-30f;
            rflag[r] = !(amax < 30.f);                   // true for NaN / Inf rows too
        }
        // ---- passes over the codes ---------------------------------------------------------------------------------------------------
        const f16x8* const tabw = reinterpret_cast<const f16x8*>(p.tab) + (int64_t)wave * CTW * KS * 64 + lane;   // + step * 64
        const int n_steps = CTW * KS;
        float lim[MTN];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float m[MTN];
#pragma unroll
            for (int mt = 0; mt < MTN; ++mt) m[mt] = INFINITY;
            f16x8 ring[RING];
#pragma unroll
            for (int s = 0; s < RING; ++s) ring[s] = tabw[(s < n_steps ? s : n_steps - 1) * 64];
#pragma unroll 1
            for (int ct = 0; ct < CTW; ++ct) {
                f32x16 acc[MTN];
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
                // k-steps in chunks of the ring depth (KS is a multiple of it): a real loop keeps the scheduling window -- and the
                // registers hipcc spends on hoisted LDS reads -- bounded; the ring runs RING steps ahead across chunks and tiles
#pragma unroll 1
                for (int u0 = 0; u0 < KS; u0 += RING) {
                    const char* xb = Xs + li * XS + (16 * u0 + 8 * hh) * 2;
#pragma unroll
                    for (int s = 0; s < RING; ++s) {
                        const f16x8 a = ring[s];
                        int tn = ct * KS + u0 + s + RING;
                        tn = tn < n_steps ? tn : n_steps - 1;
                        ring[s] = tabw[tn * 64];
#pragma unroll
                        for (int mt = 0; mt < MTN; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, *reinterpret_cast<const f16x8*>(xb + mt * 32 * XS + 32 * s), acc[mt], 0, 0, 0);
                    }
                }
                const int code0 = 32 * (wave * CTW + ct) + 4 * hh;
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt) {
                    float mn = acc[mt][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) mn = __builtin_fminf(mn, acc[mt][r]);
                    if (pass == 0) {
                        m[mt] = __builtin_fminf(m[mt], mn);
                    } else if (mn <= lim[mt]) {
                        const int r_ = mt * 32 + li;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (acc[mt][r] <= lim[mt]) {
                                atomicAdd(&cnt[r_], 1);
                                const int slot = atomicAdd(npairs, 1);
                                if (slot < VF_PCAP) pairs[slot] = ((unsigned)r_ << 16) | (unsigned)(code0 + (r & 3) + 8 * (r >> 2));
                                else rflag[r_] = 1;                              // list full: the row takes every code
                            }
                        }
                    }
                }
            }
            if (pass == 0) {
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt) {
                    m[mt] = __builtin_fminf(m[mt], __shfl_xor(m[mt], 32));
                    if (hh == 0) minw[wave * RT + mt * 32 + li] = m[mt];
                }
                lds_barrier();
                for (int r = tid; r < RT; r += 256)
                    limS[r] = __builtin_fminf(__builtin_fminf(minw[r], minw[RT + r]), __builtin_fminf(minw[2 * RT + r], minw[3 * RT + r])) + 2.f * epsS[r];
                lds_barrier();
#pragma unroll
                for (int mt = 0; mt < MTN; ++mt)      // rows past N (staged as copies of row N - 1) keep no survivor: nothing of
                    lim[mt] = row0 + mt * 32 + li < p.N ? limS[mt * 32 + li] : -INFINITY;   // theirs is ever addressed beyond z[N][D]
            }
        }
        lds_barrier();
        // ---- exact distances of the survivors: 8 lanes per (row, code) pair, PB pairs per group in flight -------------------------
        const int grp = lane >> 3, gl = lane & 7;
        constexpr int CL = D / 8;                                          // channels per lane
        constexpr int PB = D == 128 ? 4 : 2;                               // pairs per group and trip (register budget: 2 PB CL floats)
        auto pair_dist = [&](const float* __restrict__ zp, const float* __restrict__ ep) -> float {
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < CL; c += 4) {
                const f32x4 zv = *reinterpret_cast<const f32x4*>(zp + c), ev = *reinterpret_cast<const f32x4*>(ep + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float d = zv[j] - ev[j];
                    d = d * d;
                    part = __builtin_fmaf(d, d, part);
                }
            }
            part += __shfl_xor(part, 1);
            part += __shfl_xor(part, 2);
            part += __shfl_xor(part, 4);
            return part;
        };
        auto key_of = [](float dv, unsigned code) -> u64 { return ((u64)__float_as_uint(dv) << 32) | code; };   // d >= 0: bit order = value order
        {
            const int np = *npairs < VF_PCAP ? *npairs : VF_PCAP;
            for (int i0 = wave * 8 + grp; i0 < np; i0 += 32 * PB) {
                float dv[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const int i = i0 + 32 * j < np ? i0 + 32 * j : i0;
                    const unsigned pr = pairs[i];
                    dv[j] = pair_dist(p.z + (row0 + (pr >> 16)) * D + gl * CL, p.embed + (int64_t)(pr & 0xFFFFu) * D + gl * CL);
                }
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    if (gl == 0 && i0 + 32 * j < np) pdist[i0 + 32 * j] = dv[j];
            }
            lds_barrier();
            for (int i = tid; i < np; i += 256) {
                const unsigned pr = pairs[i];
                if (rflag[pr >> 16] == 0) atomicMin(&best[pr >> 16], key_of(pdist[i], pr & 0xFFFFu));
            }
            lds_barrier();
            for (int i = tid; i < np; i += 256) {
                const unsigned pr = pairs[i];
                const u64 kk = key_of(pdist[i], pr & 0xFFFFu);
                if (rflag[pr >> 16] == 0 && kk != best[pr >> 16]) atomicMin(&second[pr >> 16], kk);
            }
            lds_barrier();
        }
        // rows outside the filter's range or without survivors: every code, the whole workgroup per row (rare)
        for (int r = 0; r < RT; ++r) {
            const int64_t row = row0 + r;
            if (row >= p.N) break;
            if (!(rflag[r] != 0 || cnt[r] == 0)) continue;
            u64 lb = ~0ull, ls = ~0ull;                                    // this group's two best keys
            for (int k0 = wave * 8 + grp; k0 < p.K; k0 += 32 * PB) {
                float dv[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    const int k = k0 + 32 * j < p.K ? k0 + 32 * j : k0;
                    dv[j] = pair_dist(p.z + row * D + gl * CL, p.embed + (int64_t)k * D + gl * CL);
                }
#pragma unroll
                for (int j = 0; j < PB; ++j) {
                    if (k0 + 32 * j < p.K) {
                        const u64 kk = key_of(dv[j], (unsigned)(k0 + 32 * j));
                        ls = kk < lb ? lb : (kk < ls ? kk : ls);
                        lb = kk < lb ? kk : lb;
                    }
                }
            }
            if (tid == 0) { best[r] = ~0ull; second[r] = ~0ull; }
            lds_barrier();
            if (gl == 0) atomicMin(&best[r], lb);
            lds_barrier();
            if (gl == 0) atomicMin(&second[r], lb != best[r] ? lb : ls);
            lds_barrier();
            if (tid == 0) { rflag[r] = 0; cnt[r] = 1; }                    // decided below like the other rows
        }
        lds_barrier();
        for (int r = tid; r < RT; r += 256) {                              // the decision per row, on exact values only
            const int64_t row = row0 + r;
            if (row >= p.N) continue;
            const u64 kb = best[r], ks2 = second[r];
            const float b1 = __uint_as_float((unsigned)(kb >> 32));
            const float b2 = ks2 == ~0ull ? 3.0e38f : __uint_as_float((unsigned)(ks2 >> 32));   // a lone survivor: no near tie
            p.idx32[row] = kb == ~0ull ? 0 : (int)(kb & 0xFFFFu);
            const float gap = b2 - b1;
            if (!(gap > p.thr * b2)) {                   // inside evaluation noise, exact tie, or NaN
                const int slot = atomicAdd(&p.flags[0], 1);
                p.flag_list[slot] = (int)row;
            }
        }
        lds_barrier();                                   // the next tile reuses the LDS
    }
}

template <int D, int RT>
int launch_vqf(const VqfK& k, hipStream_t stream) {
    constexpr int XS = (3 * D + 16) * 2 + 16;
    constexpr int lds_bytes = RT * XS + RT * 4 * 2 + 4 * RT * 4 + RT * 4 * 2 + RT * 8 * 2 + VF_PCAP * 4 * 2 + 16;
    static bool attr_set = false;
    if (!attr_set) {
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)vqf_main_kernel<D, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    const unsigned grid = (unsigned)(k.n_tiles < 256 ? k.n_tiles : 256);
    vqf_main_kernel<D, RT><<<grid, 256, lds_bytes, stream>>>(k);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

}  // namespace

namespace vqae {

static int vqf_kp(int K) { return (int)round_up(K, 128); }

bool vq_filter_supported(int K, int D) {
    static const bool off = getenv("VQAE_NO_VQ_FILTER") && atoi(getenv("VQAE_NO_VQ_FILTER"));
    // measured (N = 262 144): D = 256, K = 1024: 4.14 -> 1.86 ms; D = 128, K = 256 with 128-row tiles (one 4-wave workgroup per CU):
    // 0.52 -> 0.48 ms; round 3, 64-row tiles (two workgroups per CU): 0.60 -> 0.46 ms on N(0, 1) data: taken for 128 channels too
    static const bool no128 = getenv("VQAE_NO_VQ_FILTER_128") && atoi(getenv("VQAE_NO_VQ_FILTER_128"));
    return !off && (D == 256 || (D == 128 && !no128)) && K >= 32 && K <= 32768;
}

size_t vq_filter_table_bytes(int K, int D) { return (size_t)vqf_kp(K) * (3 * D + 16) * 2; }

// flags: 4 zeroed ints ([0] = flagged-row counter shared with tier 1 / tier 2).  After this call either idx32 / flag_list are
// filled (flags[1] == 0) or nothing was done and flags[1] != 0 tells vq_tier1_kernel to run.
int vq_filter_run(const float* z, const float* embed, int64_t N, int K, int D, float thr, int* idx32, int* flags, int* flag_list,
                  void* table, hipStream_t stream) {
    VQAE_REQUIRE(vq_filter_supported(K, D) && table, VQAE_ERR_UNSUPPORTED, "vq_filter: K = %d, D = %d", K, D);
    const int Kp = vqf_kp(K), KS = 3 * D / 16 + 1;
    const int64_t chunks = (int64_t)(Kp / 32) * KS * 64;
    vqf_emax_kernel<<<64, 256, 0, stream>>>(embed, (int64_t)K * D, flags);
    VQAE_LAUNCH_CHECK();
    vqf_table_kernel<<<(unsigned)ceil_div(chunks, 256), 256, 0, stream>>>(embed, K, Kp, D, (_Float16*)table, flags);
    VQAE_LAUNCH_CHECK();
    VqfK k;
    k.z = z; k.embed = embed; k.tab = (const _Float16*)table; k.N = N; k.K = K; k.Kp = Kp; k.thr = thr;
    k.idx32 = idx32; k.flags = flags; k.flag_list = flag_list;
    const int rt = 64;
    k.n_tiles = (int)ceil_div(N, rt);
    if (D == 128) return launch_vqf<128, 64>(k, stream);
    return launch_vqf<256, 64>(k, stream);
}

}  // namespace vqae
