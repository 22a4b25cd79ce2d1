// The two 3x3 zero-padded stems (reference vq_ae/model.py: Encoder.in_stem 3 -> C0, Decoder.out_stem C0 -> 3, Conv2d(k = 3,
// padding = 1) with bias) in the 16-bit (torch.autocast) modes, on v_mfma_f32_32x32x16_{bf16,f16}.
//
// conv3x3_direct_kernel (misc_kernels.hip) -- a thread per pixel, 27 C0 fp32 MACs on the vector ALUs -- is VALU-bound at
// 0.55 / 0.73 ms (C0 = 16, 256 x 256, batch 256) where the bytes would allow 0.2 ms.  In the 16-bit modes operands and result
// are rounded to the 16-bit type anyway, so the MACs go to the matrix pipe (the 3 or C0 real output rows padded to the 32 MFMA
// rows: the waste is free) and the vector work left is the operand gather (in-stem) / nothing (out-stem).
//   in-stem   the (8 + 2) x (64 + 2) halo of the 3 input planes (fp32 NHWC / NCHW or uint8 NHWC + normalisation) -> 16-bit in LDS,
//             planar; k = tap * 3 + c (27, padded to 32 = 2 k-steps): a lane gathers its 8 k's as eight 2-byte LDS reads through
//             a per-lane offset table built once; output rows = C0 channels, bias as the accumulator's start value.
//   out-stem  the halo of the C0-channel input (fp32 NHWC) -> 16-bit [pixel][C0] in LDS; k = tap * C0 + c: one 16-byte LDS read
//             per lane and k-step (C0 = 8: a tap pair per step); 3 real output rows; stores NCHW planes or NHWC.
// Out-of-image halo pixels are zeros (Conv2d padding = 1).  Arithmetic: bias and operands rounded to the 16-bit type, products
// exact, fp32 accumulation (from the bias), result rounded -- conv3x3_direct_kernel's recipe; only the summation order differs.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
using vqae::lds_barrier;

template <int DT> struct M16;
template <> struct M16<VQAE_DT_BF16> {
    using el = __bf16; using x8 = bf16x8; using x4 = bf16x4;
    static __device__ __forceinline__ f32x16 mma(const x8& a, const x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(__bf16)v; }
};
template <> struct M16<VQAE_DT_F16> {
    using el = _Float16; using x8 = f16x8; using x4 = f16x4;
    static __device__ __forceinline__ f32x16 mma(const x8& a, const x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(_Float16)v; }
};

constexpr int ST_TH = 8, ST_TW = 64, ST_HC = ST_TW + 2, ST_HP = (ST_TH + 2) * ST_HC;   // tile, halo columns / pixels

struct Norm3s { float mean[4]; float inv[4]; };

// PyTorch [n_out][cin][3][3] fp32 -> fragment order [1 n-tile][KS][64 lanes][8] 16-bit, k = tap * cin + ci (zero beyond 9 cin),
// rows >= n_out zero: lane (r, h) of k-step ks holds w[r][k = 16 ks + 8 h + j]
template <typename EL>
__global__ void stem16_pack_kernel(const float* __restrict__ w, int n_out, int cin, int KS, EL* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KS * 512) return;
    const int j = i & 7, lane = (i >> 3) & 63, ks = i >> 9;
    const int n = lane & 31, k = ks * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    if (n < n_out && k < 9 * cin) v = w[((int64_t)n * cin + k % cin) * 9 + k / cin];
    out[i] = (EL)v;
}

struct IStemK {
    const void* __restrict__ x;          // x_kind 0: fp32 NHWC [B][H][W][3]; 1: fp32 NCHW; 2: uint8 NHWC
    const void* __restrict__ wf;         // fragments [2 k-steps][64][8]
    const float* __restrict__ bias;      // [C0] fp32 (rounded in the kernel)
    float* __restrict__ y;               // [B][H][W][C0] fp32
    Norm3s nrm;
    int x_kind, H, W, tiles_x, tiles_y, n_tiles;
};

template <int C0, int DT>
__global__ __launch_bounds__(256, 2)
void istem16_kernel(const IStemK p) {
    using E = M16<DT>;
    using x8 = typename E::x8;
    using EL = typename E::el;
    constexpr int NQ = C0 / 8;                                            // register quads of a lane that hold real channels
    __shared__ __attribute__((aligned(16))) EL P[3 * ST_HP + 8];          // planes [c][halo pixel]; + a zero slot for k >= 27
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    x8 wv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) wv[u] = *reinterpret_cast<const x8*>((const char*)p.wf + (u * 64 + lane) * 16);
    // this lane's 16 operand offsets (elements, relative to its pixel's halo position): k = 16 u + 8 hh + j -> (tap, c)
    int off[2][8];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * u + 8 * hh + j;
            const int tap = k / 3, c = k - 3 * tap;
            off[u][j] = k < 27 ? c * ST_HP + (tap / 3) * ST_HC + tap % 3 : -1;
        }
    float bq[NQ][4];                                                       // bias of this lane's channels 8 q + 4 hh + e, rounded
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) bq[q][e] = E::rnd(p.bias[8 * q + 4 * hh + e]);
    if (tid < 8) P[3 * ST_HP + tid] = (EL)0.f;

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int txi = tile % p.tiles_x;
        const int tyi = (tile / p.tiles_x) % p.tiles_y;
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int ty0 = tyi * ST_TH, tx0 = txi * ST_TW;
        const int64_t hw = (int64_t)p.H * p.W;
        // ---- stage the halo of the 3 planes, normalised / rounded -----------------------------------------------------------------
        constexpr int NI = (ST_HP + 255) / 256;
        float v[NI][3];
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            int hp = tid + it * 256;
            hp = hp < ST_HP ? hp : ST_HP - 1;
            const int hy = hp / ST_HC, hx = hp - ST_HC * hy;
            const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const int64_t pix = (int64_t)b * hw + (int64_t)iy * p.W + ix;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float t = 0.f;
                if (in) {
                    if (p.x_kind == 0) t = ((const float*)p.x)[pix * 3 + c];
                    else if (p.x_kind == 1) t = ((const float*)p.x)[((int64_t)b * 3 + c) * hw + (int64_t)iy * p.W + ix];
                    else t = ((float)((const unsigned char*)p.x)[pix * 3 + c] - p.nrm.mean[c]) * p.nrm.inv[c];
                }
                v[it][c] = t;
            }
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int hp = tid + it * 256;
            if (hp < ST_HP) {
#pragma unroll
                for (int c = 0; c < 3; ++c) P[c * ST_HP + hp] = (EL)v[it][c];                   // conv input cast
            }
        }
        lds_barrier();
        // ---- 32-pixel row segments --------------------------------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mt = wave * 4 + i;
            const int py = mt >> 1, px = (mt & 1) * 32 + li;
            const int base = py * ST_HC + px;                              // halo position of tap (0, 0)
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * q + e] = bq[q][e];     // accumulation starts from the (rounded) bias
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                unsigned short h[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    h[j] = *reinterpret_cast<const unsigned short*>(&P[off[u][j] >= 0 ? base + off[u][j] : 3 * ST_HP]);
                const u32x4 ob = {(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
                                  (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
                acc = E::mma(wv[u], __builtin_bit_cast(x8, ob), acc);
            }
            float* out = p.y + (((int64_t)b * p.H + ty0 + py) * p.W + tx0 + px) * C0 + 4 * hh;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = E::rnd(acc[4 * q + e]);                       // conv output cast
                *reinterpret_cast<f32x4*>(out + 8 * q) = o;
            }
        }
        lds_barrier();
    }
}

struct OStemK {
    const float* __restrict__ x;         // [B][H][W][C] fp32
    const void* __restrict__ wf;         // fragments [NS k-steps][64][8]
    const float* __restrict__ bias;      // [3]
    float* __restrict__ y;               // NCHW [B][3][H][W] or NHWC [B][H][W][3]
    int y_nchw, H, W, tiles_x, tiles_y, n_tiles;
};

template <int C, int DT>
__global__ __launch_bounds__(256, 2)
void ostem16_kernel(const OStemK p) {
    using E = M16<DT>;
    using x8 = typename E::x8;
    using x4 = typename E::x4;
    constexpr int NS = (9 * C + 15) / 16;                                 // k-steps: 5 (C = 8, tap pairs), 9, 18
    constexpr int PSX = C == 32 ? 80 : C * 2;                             // bytes per halo pixel
    constexpr int C4 = C / 4;
    __shared__ __attribute__((aligned(16))) char T[ST_HP * PSX];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    x8 wv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) wv[s] = *reinterpret_cast<const x8*>((const char*)p.wf + (s * 64 + lane) * 16);
    float b3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) b3[c] = hh == 0 ? E::rnd(p.bias[c]) : 0.f;   // rows 0..2 live in lanes hh = 0, registers 0..2

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int txi = tile % p.tiles_x;
        const int tyi = (tile / p.tiles_x) % p.tiles_y;
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int ty0 = tyi * ST_TH, tx0 = txi * ST_TW;
        const float* const xim = p.x + (int64_t)b * p.H * p.W * C;
        // ---- stage the halo, rounded; all loads of a batch requested before use ---------------------------------------------------------
        constexpr int NITEM = ST_HP * C4;
        constexpr int NIT = (NITEM + 255) / 256;
        constexpr int NB = NIT > 11 ? 11 : NIT;                            // loads in flight per thread
#pragma unroll
        for (int i0 = 0; i0 < NIT; i0 += NB) {
            f32x4 v[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                int idx = tid + (i0 + j) * 256;
                idx = idx < NITEM ? idx : NITEM - 1;
                const int hp = idx / C4, c4 = idx % C4;
                const int hy = hp / ST_HC, hx = hp - ST_HC * hy;
                const int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
                const bool in = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                v[j] = in ? *reinterpret_cast<const f32x4*>(xim + ((int64_t)iy * p.W + ix) * C + 4 * c4) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int idx = tid + (i0 + j) * 256;
                if (i0 + j < NIT && idx < NITEM) *reinterpret_cast<x4*>(T + (idx / C4) * PSX + (idx % C4) * 8) = __builtin_convertvector(v[j], x4);   // conv input cast
            }
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mt = wave * 4 + i;
            const int py = mt >> 1, px = (mt & 1) * 32 + li;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = b3[c];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const char* src;
                if constexpr (C == 8) {
                    int tap = 2 * s + hh;
                    tap = tap > 8 ? 8 : tap;                               // the 10th slot has zero weights: any staged pixel will do
                    src = T + ((py + tap / 3) * ST_HC + px + tap % 3) * PSX;
                } else {
                    constexpr int KC = C / 16;                             // k-steps per tap
                    const int tap = s / KC, ku = s % KC;
                    src = T + ((py + tap / 3) * ST_HC + px + tap % 3) * PSX + (16 * ku + 8 * hh) * 2;
                }
                acc = E::mma(wv[s], *reinterpret_cast<const x8*>(src), acc);
            }
            if (hh == 0) {
                const int oy = ty0 + py, ox = tx0 + px;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float o = E::rnd(acc[c]);                        // conv output cast
                    if (p.y_nchw) p.y[(((int64_t)b * 3 + c) * p.H + oy) * p.W + ox] = o;
                    else p.y[(((int64_t)b * p.H + oy) * p.W + ox) * 3 + c] = o;
                }
            }
        }
        lds_barrier();
    }
}

}  // namespace

namespace vqae {

bool stem16_supported(int c0, int h, int w, int dtype) {
    if (dtype != VQAE_DT_BF16 && dtype != VQAE_DT_F16) return false;
    return (c0 == 8 || c0 == 16 || c0 == 32) && h % ST_TH == 0 && w % ST_TW == 0;
}

size_t stem16_weight_bytes(int cin) { return (size_t)((9 * cin + 15) / 16) * 1024; }

// w: PyTorch [n_out][cin][3][3] fp32 (device) -> 16-bit fragments (device)
int stem16_pack_weight(const float* w_dev, int n_out, int cin, int dtype, void* out_dev, hipStream_t stream) {
    VQAE_REQUIRE(w_dev && out_dev && n_out <= 32 && (dtype == VQAE_DT_BF16 || dtype == VQAE_DT_F16), VQAE_ERR_INVALID, "stem16_pack_weight");
    const int KS = (9 * cin + 15) / 16;
    if (dtype == VQAE_DT_BF16) stem16_pack_kernel<__bf16><<<(unsigned)ceil_div(KS * 512, 256), 256, 0, stream>>>(w_dev, n_out, cin, KS, (__bf16*)out_dev);
    else stem16_pack_kernel<_Float16><<<(unsigned)ceil_div(KS * 512, 256), 256, 0, stream>>>(w_dev, n_out, cin, KS, (_Float16*)out_dev);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// in-stem: x (x_kind 0 NHWC f32 / 1 NCHW f32 / 2 uint8 NHWC + normalisation) -> y [B][H][W][c0] fp32
int istem16(const void* x, int x_kind, const float* mean255, const float* inv_std255, const void* wf, const float* bias, int B,
            int H, int W, int c0, float* y, int dtype, hipStream_t stream) {
    if ((int64_t)B * H * W == 0) return VQAE_OK;
    VQAE_REQUIRE(x && wf && bias && y && stem16_supported(c0, H, W, dtype), VQAE_ERR_UNSUPPORTED, "istem16: C0 = %d, %dx%d, dtype %d", c0, H, W, dtype);
    IStemK k;
    k.x = x; k.wf = wf; k.bias = bias; k.y = y; k.x_kind = x_kind;
    for (int i = 0; i < 4; ++i) {
        k.nrm.mean[i] = (mean255 && i < 3) ? mean255[i] : 0.f;
        k.nrm.inv[i] = (inv_std255 && i < 3) ? inv_std255[i] : 1.f;
    }
    k.H = H; k.W = W; k.tiles_x = W / ST_TW; k.tiles_y = H / ST_TH;
    const int64_t n_tiles = (int64_t)B * k.tiles_x * k.tiles_y;
    VQAE_REQUIRE(n_tiles < (1ll << 31), VQAE_ERR_UNSUPPORTED, "istem16: too many tiles");
    k.n_tiles = (int)n_tiles;
    const unsigned grid = (unsigned)(n_tiles < 256 * 8 ? n_tiles : 256 * 8);
#define VQAE_IS(C_) (dtype == VQAE_DT_BF16 ? (void)(istem16_kernel<C_, VQAE_DT_BF16><<<grid, 256, 0, stream>>>(k)) : (void)(istem16_kernel<C_, VQAE_DT_F16><<<grid, 256, 0, stream>>>(k)))
    if (c0 == 8) VQAE_IS(8); else if (c0 == 16) VQAE_IS(16); else VQAE_IS(32);
#undef VQAE_IS
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// out-stem: x [B][H][W][c] fp32 -> y (NCHW [B][3][H][W] if y_nchw else NHWC) fp32
int ostem16(const float* x, const void* wf, const float* bias, int B, int H, int W, int c, float* y, int y_nchw, int dtype,
            hipStream_t stream) {
    if ((int64_t)B * H * W == 0) return VQAE_OK;
    VQAE_REQUIRE(x && wf && bias && y && stem16_supported(c, H, W, dtype), VQAE_ERR_UNSUPPORTED, "ostem16: C = %d, %dx%d, dtype %d", c, H, W, dtype);
    OStemK k;
    k.x = x; k.wf = wf; k.bias = bias; k.y = y; k.y_nchw = y_nchw;
    k.H = H; k.W = W; k.tiles_x = W / ST_TW; k.tiles_y = H / ST_TH;
    const int64_t n_tiles = (int64_t)B * k.tiles_x * k.tiles_y;
    VQAE_REQUIRE(n_tiles < (1ll << 31), VQAE_ERR_UNSUPPORTED, "ostem16: too many tiles");
    k.n_tiles = (int)n_tiles;
    const unsigned grid = (unsigned)(n_tiles < 256 * 6 ? n_tiles : 256 * 6);
#define VQAE_OS(C_) (dtype == VQAE_DT_BF16 ? (void)(ostem16_kernel<C_, VQAE_DT_BF16><<<grid, 256, 0, stream>>>(k)) : (void)(ostem16_kernel<C_, VQAE_DT_F16><<<grid, 256, 0, stream>>>(k)))
    if (c == 8) VQAE_OS(8); else if (c == 16) VQAE_OS(16); else VQAE_OS(32);
#undef VQAE_OS
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

}  // namespace vqae
