// 16-bit (torch.autocast) form of the high-resolution part of an 'up' Fixup block (reference vq_ae/layers/conv_block.py:196-216,
// mode 'up'; ResizeConv2D = conv1x1(bicubic_x2(.)), vq_ae/layers/conv.py:8-11) in ONE launch:
//   u   = bicubic_x2(t1)                       t1 = ELU(conv1(ELU(x + b1a) + b1b) + b2a) + b2b, fp32, low resolution (head16)
//   t2  = ELU(conv2(u) + b3a) + b3b            conv2: 1x1, C -> C, at the HIGH resolution
//   out = conv3(t2) * scale + b4 + skip_conv(bicubic_x2(x + b1c)) + b1d        conv3, skip_conv: 1x1, C -> C / 2
// Under autocast the resize runs on fp32 tensors and every conv rounds its operand and its result to the 16-bit type, so the
// convs cannot be moved in front of the resize as the fp32 path does (handle.hip): the block is 4 C^2 flop per OUTPUT pixel.
// Unfused that was two resize launches writing 4x-larger fp32 tensors and three generic conv launches reading them back
// (3.9 ms at 32 -> 16 channels, 256 x 256 outputs, batch 256); here the launch reads the two low-resolution tensors (x, t1)
// and writes out.
//
// A 256-thread workgroup owns ROWS x 32 output pixels (ROWS = 4; 2 at C = 128):
//   stage    the (ROWS / 2 + 4) x 20 low-resolution window (clamped indices) of x (+ b1c), then of t1, in LDS as fp32
//   resize   separable, in the order of bicubic_up2_kernel / ATen (out = sum_i wy_i * (sum_j wx_j * v_ij), left to right; the
//            accumulation steps as fmas -- the resize is this kernel's vector-issue bound, and its result is rounded to the
//            16-bit type right after: <= 1 fp32 ulp from the unfused form): a thread owns one output column and 4 channels, interpolates the window rows horizontally once and
//            combines them for the ROWS output rows; results go to LDS as 16-bit (the conv input cast): S (skip), U (branch)
//   convs    v_mfma_f32_32x32x16_{bf16,f16}, weights (fragment order, from L2) as the row operand: conv2 from U -> t2 (16-bit,
//            over U) -> conv3 from t2 and skip_conv from S -> epilogue, fp32 store.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
using vqae::elu_act;
using vqae::lds_barrier;

template <int DT> struct U16;
template <> struct U16<VQAE_DT_BF16> {
    using x8 = bf16x8; using x4 = bf16x4;
    static __device__ __forceinline__ f32x16 mma(const x8& a, const x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(__bf16)v; }
};
template <> struct U16<VQAE_DT_F16> {
    using x8 = f16x8; using x4 = f16x4;
    static __device__ __forceinline__ f32x16 mma(const x8& a, const x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(_Float16)v; }
};

struct Up16K {
    const float* __restrict__ x;         // [B][H][W][C] fp32: the block's input
    const float* __restrict__ t1;        // [B][H][W][C] fp32: ELU(round16(conv1(.)) + b2a) + b2b
    const void* __restrict__ w2;         // 16-bit fragment order (down16_pack_weight): [C][C]
    const void* __restrict__ w3;         //   [C / 2][C]   (rows padded to 32)
    const void* __restrict__ wsk;        //   [C / 2][C]
    float* __restrict__ y;               // [B][2H][2W][C / 2] fp32
    int H, W;                            // low resolution; W a multiple of 16
    int tiles_x, tiles_y;
    float b3a, b3b, b4, scale, b1c, b1d;
};

template <int C, int DT, int ROWS>
__global__ __launch_bounds__(256, 2)
void up16_kernel(const Up16K p) {
    using E = U16<DT>;
    using x8 = typename E::x8;
    using x4 = typename E::x4;
    constexpr int CO = C / 2;
    constexpr int MT = ROWS;                          // 32-pixel m-tiles: one output row each
    constexpr int WRN = ROWS / 2 + 4, WCN = 20;       // window rows / columns (low resolution)
    constexpr int NWP = WRN * WCN;
    constexpr int C4 = C / 4;
    constexpr int PSW = C * 4 + 16;                   // window bytes per pixel
    constexpr int PS = C * 2 + 16;                    // S / U bytes per output pixel (odd number of 16-B slots)
    constexpr int KS = C / 16;                        // k-steps of every conv (K = C)
    constexpr int NT2 = C < 32 ? 1 : C / 32;          // n-tiles of conv2
    constexpr int NQ2 = C >= 32 ? 4 : C / 8;          // real register quads per n-tile (C = 16: rows 16..31 are zero weights)
    constexpr int NT3 = CO < 32 ? 1 : CO / 32;
    constexpr int NQ3 = CO >= 32 ? 4 : CO / 8;
    constexpr int NP2 = MT * NT2 / 4;                 // (m-tile, n-tile) pairs of conv2 per wave
    static_assert(MT * NT2 % 4 == 0 && MT * NT3 == 4, "work split over 4 waves");
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const win = lds;                            // [NWP][PSW] fp32
    char* const S = lds + NWP * PSW;                  // [32 MT][PS] 16-bit: bicubic(x + b1c)
    char* const U = S + 32 * MT * PS;                 // [32 MT][PS] 16-bit: bicubic(t1), then t2
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;

    const int tile = blockIdx.x;
    const int txi = tile % p.tiles_x;
    const int tyi = (tile / p.tiles_x) % p.tiles_y;
    const int64_t b = tile / (p.tiles_x * p.tiles_y);
    const int oy0 = tyi * ROWS, ox0 = txi * 32;
    const int iy0 = oy0 >> 1, jx0 = ox0 >> 1;

    // ---- window items: (window pixel, channel quad) -> this thread's NIT items --------------------------------------------
    constexpr int NITEM = NWP * C4;
    constexpr int NIT = (NITEM + 255) / 256;
    int woff[NIT];                                    // element offset of the item's source (clamped coordinates)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        int idx = tid + it * 256;
        idx = idx < NITEM ? idx : NITEM - 1;
        const int wp = idx / C4, q = idx % C4;
        int iy = iy0 - 2 + wp / WCN, ix = jx0 - 2 + wp % WCN;
        iy = iy < 0 ? 0 : (iy > p.H - 1 ? p.H - 1 : iy);
        ix = ix < 0 ? 0 : (ix > p.W - 1 ? p.W - 1 : ix);
        woff[it] = (iy * p.W + ix) * C + 4 * q;
    }
    const float* const xim = p.x + b * (int64_t)p.H * p.W * C;
    const float* const tim = p.t1 + b * (int64_t)p.H * p.W * C;
    f32x4 xa[NIT], ta[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) xa[it] = *reinterpret_cast<const f32x4*>(xim + woff[it]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) ta[it] = *reinterpret_cast<const f32x4*>(tim + woff[it]);
    auto win_store = [&](const f32x4 (&v)[NIT], float bias) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 256;
            if (NITEM % 256 == 0 || idx < NITEM) *reinterpret_cast<f32x4*>(win + (idx / C4) * PSW + (idx % C4) * 16) = v[it] + bias;
        }
    };

    // ---- separable bicubic x2 of the window -> 16-bit rows of dst ----------------------------------------------------------------
    const float w75[4] = {-0.03515625f, 0.26171875f, 0.87890625f, -0.10546875f};   // even outputs (offset .75)
    const float w25[4] = {-0.10546875f, 0.87890625f, 0.26171875f, -0.03515625f};   // odd outputs  (offset .25)
    auto resize = [&](char* dst) {
        constexpr int NBI = (32 * C4 + 255) / 256;    // (output column, channel quad) items per thread
#pragma unroll
        for (int bi = 0; bi < NBI; ++bi) {
            const int item = tid + bi * 256;
            if (32 * C4 % 256 != 0 && item >= 32 * C4) break;
            const int c = item / C4, q = item % C4;
            const int wcb = (c >> 1) + (c & 1);
            const float* wx = (c & 1) ? w25 : w75;
            f32x4 h[WRN];
#pragma unroll
            for (int wr = 0; wr < WRN; ++wr) {
                const char* src = win + (wr * WCN + wcb) * PSW + q * 16;
                h[wr] = *reinterpret_cast<const f32x4*>(src) * wx[0];
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const f32x4 wk = {wx[k], wx[k], wx[k], wx[k]};
                    h[wr] = __builtin_elementwise_fma(*reinterpret_cast<const f32x4*>(src + k * PSW), wk, h[wr]);
                }
            }
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                const int wrb = (r >> 1) + (r & 1);
                const float* wy = (r & 1) ? w25 : w75;
                f32x4 o = h[wrb] * wy[0];
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    const f32x4 wk = {wy[k], wy[k], wy[k], wy[k]};
                    o = __builtin_elementwise_fma(h[wrb + k], wk, o);
                }
                *reinterpret_cast<x4*>(dst + (r * 32 + c) * PS + q * 8) = __builtin_convertvector(o, x4);   // conv input cast
            }
        }
    };

    auto wfrag = [&](const void* __restrict__ w, int nt, int u) -> x8 {
        return *reinterpret_cast<const x8*>((const char*)w + ((int64_t)(nt * KS + u) * 64 + lane) * 16);
    };

    win_store(xa, p.b1c);                              // skip: bicubic(x + bias1c)
    lds_barrier();
    resize(S);
    lds_barrier();                                     // every thread is done with the x window
    win_store(ta, 0.f);
    lds_barrier();
    resize(U);
    lds_barrier();

    // ---- conv2 (C -> C) from U ------------------------------------------------------------------------------------------------
    f32x16 acc2[NP2];
#pragma unroll
    for (int i = 0; i < NP2; ++i) {
        const int pair = wave + 4 * i;
        const int mt = pair % MT, nt = pair / MT;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][r] = 0.f;
        const char* a0 = U + (mt * 32 + li) * PS + 16 * hh;
#pragma unroll
        for (int u = 0; u < KS; ++u) acc2[i] = E::mma(wfrag(p.w2, nt, u), *reinterpret_cast<const x8*>(a0 + 32 * u), acc2[i]);
    }
    lds_barrier();                                     // every wave is done reading U
#pragma unroll
    for (int i = 0; i < NP2; ++i) {
        const int pair = wave + 4 * i;
        const int mt = pair % MT, nt = pair / MT;
        char* const dst = U + (mt * 32 + li) * PS + (32 * nt + 4 * hh) * 2;
#pragma unroll
        for (int g = 0; g < NQ2; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = elu_act(E::rnd(acc2[i][4 * g + e]) + p.b3a) + p.b3b;   // conv2 output cast
            *reinterpret_cast<x4*>(dst + 16 * g) = __builtin_convertvector(o, x4);                     // conv3 input cast
        }
    }
    lds_barrier();

    // ---- conv3 (t2) and skip_conv (S), C -> C / 2: one (m-tile, n-tile) pair per wave ---------------------------------------------
    const int mt = wave % MT, nt = wave / MT;
    f32x16 acc3, accs;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc3[r] = 0.f; accs[r] = 0.f; }
    {
        const char* a0 = U + (mt * 32 + li) * PS + 16 * hh;
        const char* s0 = S + (mt * 32 + li) * PS + 16 * hh;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            acc3 = E::mma(wfrag(p.w3, nt, u), *reinterpret_cast<const x8*>(a0 + 32 * u), acc3);
            accs = E::mma(wfrag(p.wsk, nt, u), *reinterpret_cast<const x8*>(s0 + 32 * u), accs);
        }
    }
    float* out = p.y + ((b * (2 * p.H) + oy0 + mt) * (int64_t)(2 * p.W) + ox0 + li) * CO + 32 * nt + 4 * hh;
#pragma unroll
    for (int g = 0; g < NQ3; ++g) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = E::rnd(acc3[4 * g + e]) * p.scale;   // branch: conv3 * scale + bias4
            t = t + p.b4;
            o[e] = t + (E::rnd(accs[4 * g + e]) + p.b1d);  // + skip_conv(.) + bias1d
        }
        *reinterpret_cast<f32x4*>(out + 8 * g) = o;
    }
}

template <int C, int DT, int ROWS>
int launch_up16(const Up16K& k, int64_t n_tiles, hipStream_t stream) {
    constexpr int lds_bytes = (ROWS / 2 + 4) * 20 * (C * 4 + 16) + 2 * 32 * ROWS * (C * 2 + 16);
    static bool attr_set = false;
    if (!attr_set) {
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)up16_kernel<C, DT, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    up16_kernel<C, DT, ROWS><<<(unsigned)n_tiles, 256, lds_bytes, stream>>>(k);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

}  // namespace

namespace vqae {

// C (= the block's input channels) in {16, 32, 64, 128}; low-resolution width a multiple of 16
bool up16_supported(int c, int h, int w, int dtype) {
    if (dtype != VQAE_DT_BF16 && dtype != VQAE_DT_F16) return false;
    return (c == 16 || c == 32 || c == 64 || c == 128) && h >= 1 && w % 16 == 0;
}

// x, t1: [B][H][W][c] fp32 -> y [B][2H][2W][c / 2] fp32; weights: down16_pack_weight([c][c]), ([c / 2][c]), ([c / 2][c])
int up16_block(const float* x, const float* t1, const void* w2h, const void* w3h, const void* wskh, int B, int H, int W, int c,
               float b3a, float b3b, float scale, float b4, float b1c, float b1d, int dtype, float* y, hipStream_t stream) {
    if (B == 0) return VQAE_OK;
    VQAE_REQUIRE(x && t1 && w2h && w3h && wskh && y, VQAE_ERR_INVALID, "up16_block: null pointer");
    VQAE_REQUIRE(up16_supported(c, H, W, dtype), VQAE_ERR_UNSUPPORTED, "up16_block: C = %d, %dx%d, dtype %d", c, H, W, dtype);
    VQAE_REQUIRE((int64_t)H * W * c < (1ll << 31), VQAE_ERR_UNSUPPORTED, "up16_block: image too large");
    Up16K k;
    k.x = x; k.t1 = t1; k.w2 = w2h; k.w3 = w3h; k.wsk = wskh; k.y = y;
    k.H = H; k.W = W;
    const int rows = c == 128 ? 2 : 4;
    k.tiles_x = 2 * W / 32; k.tiles_y = 2 * H / rows;
    k.b3a = b3a; k.b3b = b3b; k.scale = scale; k.b4 = b4; k.b1c = b1c; k.b1d = b1d;
    const int64_t n_tiles = (int64_t)B * k.tiles_x * k.tiles_y;
    VQAE_REQUIRE(n_tiles < (1ll << 31) && 2 * H % rows == 0, VQAE_ERR_UNSUPPORTED, "up16_block: tiling");
#define VQAE_U16(C_, R_) (dtype == VQAE_DT_BF16 ? launch_up16<C_, VQAE_DT_BF16, R_>(k, n_tiles, stream) : launch_up16<C_, VQAE_DT_F16, R_>(k, n_tiles, stream))
    if (c == 16) return VQAE_U16(16, 4);
    if (c == 32) return VQAE_U16(32, 4);
    if (c == 64) return VQAE_U16(64, 4);
    return VQAE_U16(128, 2);
#undef VQAE_U16
}

}  // namespace vqae
