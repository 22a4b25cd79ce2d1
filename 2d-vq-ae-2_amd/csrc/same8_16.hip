// A whole 'same' Fixup block at C = 8 channels in the 16-bit (torch.autocast) modes -- the stem-width level of the reference's
// default model (cfg A: 512 x 512 resolution; reference vq_ae/layers/conv_block.py:196-216) -- in ONE launch:
//   t1  = ELU(conv1(ELU(x + b1a) + b1b) + b2a) + b2b        conv1: 1x1, 8 -> 8
//   t2  = ELU(conv2(t1) + b3a) + b3b                        conv2: 3x3 circular, 8 -> 8
//   out = conv3(t2) * scale + b4 + x                        conv3: 1x1
// with autocast's rounding points (every conv operand / result rounded to bf16 / f16, fp32 accumulation, fp32 elsewhere).
//
// The fp32-shaped VALU kernel it replaces in these modes (fixup_fused.hip, fixup_same_c8_kernel: 704 fp32 MACs per pixel on
// the vector ALUs) is VALU-bound at 2.2 ms per block (512 x 512, batch 256).  Here conv2 and conv3 run on
// v_mfma_f32_32x32x16_{bf16,f16} with the 8 output channels padded to the 32 MFMA rows -- 3/4 of the matrix work is zeros, and
// it is still 6 instructions per 32 pixels -- which leaves the activations as the only vector work:
//   P1  a thread per halo pixel: x (32 B) -> pre-activation -> conv1 as 64 fp32 fmas (the operands are 16-bit values, the
//       products exact, the chain order that of the kernel it replaces) -> t1 as 8 x 16 bit = one 16-byte LDS write
//   P2  a wave per 32-pixel row segment: k-step u of conv2 = taps 2u and 2u + 1 (lane half hh picks the tap): ONE 16-byte LDS
//       read per lane and k-step is the whole operand; result layout D[channel][pixel]: lane (pixel, hh) holds channels
//       4 hh .. 4 hh + 3 in its first register quad -> activation -> the two lane halves swap their 4 channels (one cross-lane
//       read) -> conv3 (K = 8, padded) -> epilogue with the residual, both as whole 1 KiB rows per wave instruction.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
using vqae::elu_act;
using vqae::lds_barrier;

template <int DT> struct S16;
template <> struct S16<VQAE_DT_BF16> {
    using x8 = bf16x8; using x4 = bf16x4;
    static __device__ __forceinline__ f32x16 mma(const x8& a, const x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(__bf16)v; }
};
template <> struct S16<VQAE_DT_F16> {
    using x8 = f16x8; using x4 = f16x4;
    static __device__ __forceinline__ f32x16 mma(const x8& a, const x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(_Float16)v; }
};

struct S8K {
    const float* __restrict__ x;         // [B][H][W][8] fp32
    float* __restrict__ y;               // [B][H][W][8] fp32 (not x: halo reads)
    const float* __restrict__ w1;        // packed fp32 [>= 8][8] (rounded to the 16-bit type): w1[co * 8 + k]
    const void* __restrict__ w2f;        // 16-bit fragment order (down16_pack_weight): [32 rows (8 real)][80 (72 real)], k = tap * 8 + c
    const void* __restrict__ w3f;        //   [32 (8)][16 (8)]
    int H, W, tiles_x, tiles_y, n_tiles;
    float b1a, b1b, b2a, b2b, b3a, b3b, b4, scale;
};

constexpr int S8_TH = 8, S8_TW = 64;                     // output tile
constexpr int S8_HC = S8_TW + 2, S8_HP = (S8_TH + 2) * S8_HC;   // halo columns / pixels

template <int DT>
__global__ __launch_bounds__(256, 2)
void same8_16_kernel(const S8K p) {
    using E = S16<DT>;
    using x8 = typename E::x8;
    using x4 = typename E::x4;
    __shared__ __attribute__((aligned(16))) char T1[S8_HP * 16];     // t1 of the halo, 8 x 16 bit per pixel
    __shared__ float W1[64];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    if (tid < 64) W1[tid] = p.w1[tid];
    x8 w2v[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) w2v[u] = *reinterpret_cast<const x8*>((const char*)p.w2f + (u * 64 + lane) * 16);
    const x8 w3v = *reinterpret_cast<const x8*>((const char*)p.w3f + lane * 16);
    __syncthreads();

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int txi = tile % p.tiles_x;
        const int tyi = (tile / p.tiles_x) % p.tiles_y;
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int ty0 = tyi * S8_TH, tx0 = txi * S8_TW;
        const float* const xim = p.x + (int64_t)b * p.H * p.W * 8;
        // ---- P1: t1 on the halo -----------------------------------------------------------------------------------------------
        constexpr int NI = (S8_HP + 255) / 256;
        f32x4 a0[NI], a1[NI];
#pragma unroll
        for (int it = 0; it < NI; ++it) {                                  // every row of the halo requested up front
            int hp = tid + it * 256;
            hp = hp < S8_HP ? hp : S8_HP - 1;
            const int hy = hp / S8_HC, hx = hp - S8_HC * hy;
            int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
            iy = iy < 0 ? iy + p.H : (iy >= p.H ? iy - p.H : iy);
            ix = ix < 0 ? ix + p.W : (ix >= p.W ? ix - p.W : ix);
            const f32x4* src = reinterpret_cast<const f32x4*>(xim + ((int64_t)iy * p.W + ix) * 8);
            a0[it] = src[0];
            a1[it] = src[1];
        }
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int hp = tid + it * 256;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = a0[it][e]; v[4 + e] = a1[it][e]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = E::rnd(elu_act(v[k] + p.b1a) + p.b1b);          // conv1 input cast
            f32x8 t;
#pragma unroll
            for (int co = 0; co < 8; ++co) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = __builtin_fmaf(v[k], W1[co * 8 + k], acc);
                t[co] = elu_act(E::rnd(acc) + p.b2a) + p.b2b;                                   // conv1 output cast
            }
            if (hp < S8_HP) *reinterpret_cast<x8*>(T1 + hp * 16) = __builtin_convertvector(t, x8);   // conv2 input cast
        }
        lds_barrier();
        // ---- P2: conv2 + conv3 per 32-pixel row segment ------------------------------------------------------------------------------
        f32x4 res4[4];                                                     // residual rows of the four segments, requested together (L2 hits)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mt = wave * 4 + i;
            const int py = mt >> 1, px = (mt & 1) * 32 + li;
            res4[i] = *reinterpret_cast<const f32x4*>(p.x + (((int64_t)b * p.H + ty0 + py) * p.W + tx0 + px) * 8 + 4 * hh);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mt = wave * 4 + i;                                   // 16 segments: 8 rows x 2
            const int py = mt >> 1, px = (mt & 1) * 32 + li;
            const int64_t o = (((int64_t)b * p.H + ty0 + py) * p.W + tx0 + px) * 8 + 4 * hh;
            const f32x4 res = res4[i];                                     // residual: consumed after conv3
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                int tap = 2 * u + hh;
                tap = tap > 8 ? 8 : tap;                                   // the 10th slot has zero weights: any finite row will do
                const int ty = tap / 3, tx = tap - 3 * ty;
                acc = E::mma(w2v[u], *reinterpret_cast<const x8*>(T1 + ((py + ty) * S8_HC + px + tx) * 16), acc);
            }
            f32x4 t2;
#pragma unroll
            for (int e = 0; e < 4; ++e) t2[e] = elu_act(E::rnd(acc[e]) + p.b3a) + p.b3b;       // conv2 output cast
            const x4 mine = __builtin_convertvector(t2, x4);                                    // conv3 input cast: channels 4 hh .. + 3
            const u32x2 mb = __builtin_bit_cast(u32x2, mine);
            u32x2 ob;
            ob[0] = __shfl_xor(mb[0], 32);
            ob[1] = __shfl_xor(mb[1], 32);
            // conv3 operand: lanes hh = 0 hold k = 0..7 = channels 0..7 of their pixel; lanes hh = 1 (k = 8..15: zero weights) zeros
            const u32x4 opb = hh == 0 ? u32x4{mb[0], mb[1], ob[0], ob[1]} : u32x4{0u, 0u, 0u, 0u};
            f32x16 acc3;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
            acc3 = E::mma(w3v, __builtin_bit_cast(x8, opb), acc3);
            f32x4 out;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float tv = E::rnd(acc3[e]) * p.scale;                      // conv3 output cast
                tv = tv + p.b4;
                out[e] = tv + res[e];
            }
            *reinterpret_cast<f32x4*>(p.y + o) = out;
        }
        lds_barrier();                                                     // t1 is dead: the next tile may overwrite it
    }
}

// ------------------------------------------------------------------------------------------------------------------
// C = 16 (the 256 x 256 level of cfg A / B): the same one-launch block with conv1 on the MFMA too (K = 16 is one k-step).
// Beside the chained trunk16 launches it replaces at this width (t1 handed from launch to launch as 16-bit: 12 C bytes per
// pixel and one launch per block + a chain-head conv1), a whole block per launch moves 8 C bytes per pixel and recomputes
// conv1 on the 29 % halo -- at 16 channels the matrix work is nothing and the tile's latency chain is two phases instead of six.
//   P1  a wave per 32 halo pixels: lane (pixel, hh) loads channels 8 hh .. + 7, pre-activation, one MFMA, lane holds channels
//       {4 hh .. + 3, 8 + 4 hh .. + 3} -> activation -> t1 (32 B per pixel) in LDS
//   P2  as the C = 8 form with one tap per k-step (9 steps), two register quads per lane, the lane halves swap one quad each.
#ifndef SS16_RES_UP
#define SS16_RES_UP 1
#endif
struct S16K {
    const float* __restrict__ x;         // [B][H][W][C] fp32
    float* __restrict__ y;
    const void* __restrict__ w1f;        // trunk16_pack_weight fragments: [32 rows (C real)][C]
    const void* __restrict__ w2f;        //   [32][9 * C], k = tap * C + c
    const void* __restrict__ w3f;        //   [32][C]
    int H, W, tiles_x, tiles_y, n_tiles;
    float b1a, b1b, b2a, b2b, b3a, b3b, b4, scale;
};

// C = 16 or 32 (one 32-row n-tile).  Lane (pixel, hh) of a result holds NQ = C / 8 register quads: channels 8 q + 4 hh .. + 3.
template <int C, int DT>
__global__ __launch_bounds__(256, 2)                     // 2 waves per SIMD: a 256-register budget keeps the MFMA results in VGPRs (no v_accvgpr_read)
void same_small16_kernel(const S16K p) {
    using E = S16<DT>;
    using x8 = typename E::x8;
    using x4 = typename E::x4;
    constexpr int KS = C / 16;                                            // k-steps of a 1x1 conv
    constexpr int NQ = C / 8;
    constexpr int PSX = C == 16 ? 32 : C * 2 + 16;                        // t1 bytes per halo pixel (C = 32: + one 16-B slot against bank conflicts)
    __shared__ __attribute__((aligned(16))) char T1[S8_HP * PSX];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    x8 w1v[KS], w2v[9 * KS], w3v[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
        w1v[u] = *reinterpret_cast<const x8*>((const char*)p.w1f + (u * 64 + lane) * 16);
        w3v[u] = *reinterpret_cast<const x8*>((const char*)p.w3f + (u * 64 + lane) * 16);
    }
#pragma unroll
    for (int u = 0; u < 9 * KS; ++u) w2v[u] = *reinterpret_cast<const x8*>((const char*)p.w2f + (u * 64 + lane) * 16);

    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int txi = tile % p.tiles_x;
        const int tyi = (tile / p.tiles_x) % p.tiles_y;
        const int b = tile / (p.tiles_x * p.tiles_y);
        const int ty0 = tyi * S8_TH, tx0 = txi * S8_TW;
        const float* const xim = p.x + (int64_t)b * p.H * p.W * C;
        // ---- P1: t1 on the halo, 32 pixels per wave and step -----------------------------------------------------------------------
        constexpr int NG = (S8_HP + 31) / 32;                              // 21 groups
        constexpr int NGW = (NG + 3) / 4;                                  // per wave (the last ones of waves 1..3 repeat group NG - 1)
        constexpr int NGB = C == 16 ? NGW : 3;                             // groups whose rows are requested together (register budget)
#pragma unroll
        for (int it0 = 0; it0 < NGW; it0 += NGB) {
            f32x4 av[NGB][KS][2];
#pragma unroll
            for (int j = 0; j < NGB; ++j) {
                int g = wave + 4 * (it0 + j);
                g = g < NG ? g : NG - 1;
                int hp = g * 32 + li;
                hp = hp < S8_HP ? hp : S8_HP - 1;
                const int hy = hp / S8_HC, hx = hp - S8_HC * hy;
                int iy = ty0 + hy - 1, ix = tx0 + hx - 1;
                iy = iy < 0 ? iy + p.H : (iy >= p.H ? iy - p.H : iy);
                ix = ix < 0 ? ix + p.W : (ix >= p.W ? ix - p.W : ix);
                const float* src = xim + ((int64_t)iy * p.W + ix) * C + 8 * hh;
#pragma unroll
                for (int u = 0; u < KS; ++u) {
                    av[j][u][0] = *reinterpret_cast<const f32x4*>(src + 16 * u);
                    av[j][u][1] = *reinterpret_cast<const f32x4*>(src + 16 * u + 4);
                }
            }
#pragma unroll
            for (int j = 0; j < NGB; ++j) {
                int g = wave + 4 * (it0 + j);
                g = g < NG ? g : NG - 1;
                const int hp = g * 32 + li;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int u = 0; u < KS; ++u) {
                    f32x8 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = elu_act(av[j][u][0][e] + p.b1a) + p.b1b;
                        v[4 + e] = elu_act(av[j][u][1][e] + p.b1a) + p.b1b;
                    }
                    acc = E::mma(w1v[u], __builtin_convertvector(v, x8), acc);    // conv1 input cast
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    f32x4 t;
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = elu_act(E::rnd(acc[4 * q + e]) + p.b2a) + p.b2b;     // conv1 output cast
                    if (hp < S8_HP) *reinterpret_cast<x4*>(T1 + hp * PSX + (8 * q + 4 * hh) * 2) = __builtin_convertvector(t, x4);   // conv2 input cast
                }
            }
        }
        lds_barrier();
        // ---- P2: conv2 + conv3 per 32-pixel row segment ------------------------------------------------------------------------------
        // the residual rows of all four segments are requested up front (L2 hits: P1 just read them): one latency instead of four
        constexpr bool RES_UP = SS16_RES_UP;
        f32x4 res_all[RES_UP ? 4 : 1][NQ];
        if (RES_UP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int mt = wave * 4 + i;
                const int py = mt >> 1, px = (mt & 1) * 32 + li;
                const int64_t o = (((int64_t)b * p.H + ty0 + py) * p.W + tx0 + px) * C + 4 * hh;
#pragma unroll
                for (int q = 0; q < NQ; ++q) res_all[i][q] = *reinterpret_cast<const f32x4*>(p.x + o + 8 * q);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mt = wave * 4 + i;
            const int py = mt >> 1, px = (mt & 1) * 32 + li;
            const int64_t o = (((int64_t)b * p.H + ty0 + py) * p.W + tx0 + px) * C + 4 * hh;
            f32x4 res[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) res[q] = RES_UP ? res_all[i][q] : *reinterpret_cast<const f32x4*>(p.x + o + 8 * q);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ty = tap / 3, tx = tap - 3 * ty;
                const char* tp = T1 + ((py + ty) * S8_HC + px + tx) * PSX + 16 * hh;
#pragma unroll
                for (int u = 0; u < KS; ++u) acc = E::mma(w2v[tap * KS + u], *reinterpret_cast<const x8*>(tp + 32 * u), acc);
            }
            u32x2 qb[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                f32x4 t2;
#pragma unroll
                for (int e = 0; e < 4; ++e) t2[e] = elu_act(E::rnd(acc[4 * q + e]) + p.b3a) + p.b3b;   // conv2 output cast
                qb[q] = __builtin_bit_cast(u32x2, __builtin_convertvector(t2, x4));                   // conv3 input cast
            }
            // conv3 operand of k-step u, lane half hh: channels 16 u + 8 hh .. + 7.  Own quads hold {8 q + 4 hh ..}: hh = 0 keeps
            // quad 2u (16u .. + 3) and receives the partner's quad 2u (16u + 4 ..); hh = 1 receives the partner's quad 2u + 1
            // (16u + 8 ..) and keeps its own quad 2u + 1 (16u + 12 ..)
            f32x16 acc3;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                const u32x2 send = hh == 0 ? qb[2 * u + 1] : qb[2 * u];
                u32x2 recv;
                recv[0] = __shfl_xor(send[0], 32);
                recv[1] = __shfl_xor(send[1], 32);
                const u32x4 opb = hh == 0 ? u32x4{qb[2 * u][0], qb[2 * u][1], recv[0], recv[1]} : u32x4{recv[0], recv[1], qb[2 * u + 1][0], qb[2 * u + 1][1]};
                acc3 = E::mma(w3v[u], __builtin_bit_cast(x8, opb), acc3);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                f32x4 out;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float tv = E::rnd(acc3[4 * q + e]) * p.scale;          // conv3 output cast
                    tv = tv + p.b4;
                    out[e] = tv + res[q][e];
                }
                *reinterpret_cast<f32x4*>(p.y + o + 8 * q) = out;
            }
        }
        lds_barrier();
    }
}

}  // namespace

namespace vqae {

bool same8_16_supported(int c, int h, int w, int dtype) {
    static const bool off = getenv("VQAE_NO_SAME8_16") && atoi(getenv("VQAE_NO_SAME8_16"));
    if (off || (dtype != VQAE_DT_BF16 && dtype != VQAE_DT_F16)) return false;
    return c == 8 && h % S8_TH == 0 && w % S8_TW == 0;
}

// x -> y (x != y), [B][H][W][8] fp32; w1: packed fp32 (rounded) [>= 8][8]; w2h / w3h: down16_pack_weight(w2 [8][72]) / (w3 [8][8]);
// scalars8 = {b1a, b1b, b2a, b2b, b3a, b3b, b4, scale}
int same8_16_block(const float* x, float* y, const float* w1_packed, const void* w2h, const void* w3h, int B, int H, int W,
                   const float* scalars8, int dtype, hipStream_t stream) {
    if (B == 0) return VQAE_OK;
    VQAE_REQUIRE(x && y && x != y && w1_packed && w2h && w3h && scalars8, VQAE_ERR_INVALID, "same8_16_block: bad pointer");
    VQAE_REQUIRE(same8_16_supported(8, H, W, dtype), VQAE_ERR_UNSUPPORTED, "same8_16_block: %dx%d, dtype %d", H, W, dtype);
    S8K k;
    k.x = x; k.y = y; k.w1 = w1_packed; k.w2f = w2h; k.w3f = w3h;
    k.H = H; k.W = W; k.tiles_x = W / S8_TW; k.tiles_y = H / S8_TH;
    const int64_t n_tiles = (int64_t)B * k.tiles_x * k.tiles_y;
    VQAE_REQUIRE(n_tiles < (1ll << 31), VQAE_ERR_UNSUPPORTED, "same8_16_block: too many tiles");
    k.n_tiles = (int)n_tiles;
    k.b1a = scalars8[0]; k.b1b = scalars8[1]; k.b2a = scalars8[2]; k.b2b = scalars8[3];
    k.b3a = scalars8[4]; k.b3b = scalars8[5]; k.b4 = scalars8[6]; k.scale = scalars8[7];
    const unsigned grid = (unsigned)(n_tiles < 256 * 8 ? n_tiles : 256 * 8);
    if (dtype == VQAE_DT_BF16) same8_16_kernel<VQAE_DT_BF16><<<grid, 256, 0, stream>>>(k);
    else same8_16_kernel<VQAE_DT_F16><<<grid, 256, 0, stream>>>(k);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// C = 16 / 32: x -> y (x != y), [B][H][W][C] fp32; w1h / w2h / w3h: trunk16_pack_weight(c = C; taps 1 / 9 / 1)
bool same16_16_supported(int c, int h, int w, int dtype) {
    static const bool off = getenv("VQAE_NO_SAME16_16") && atoi(getenv("VQAE_NO_SAME16_16"));
    static const bool off32 = getenv("VQAE_NO_SAME32_16") && atoi(getenv("VQAE_NO_SAME32_16"));
    if (off || (dtype != VQAE_DT_BF16 && dtype != VQAE_DT_F16)) return false;
    return (c == 16 || (c == 32 && !off32)) && h % S8_TH == 0 && w % S8_TW == 0;
}

int same16_16_block(const float* x, float* y, const void* w1h, const void* w2h, const void* w3h, int B, int H, int W, int c,
                    const float* scalars8, int dtype, hipStream_t stream) {
    if (B == 0) return VQAE_OK;
    VQAE_REQUIRE(x && y && x != y && w1h && w2h && w3h && scalars8, VQAE_ERR_INVALID, "same16_16_block: bad pointer");
    VQAE_REQUIRE(same16_16_supported(c, H, W, dtype), VQAE_ERR_UNSUPPORTED, "same16_16_block: C = %d, %dx%d, dtype %d", c, H, W, dtype);
    S16K k;
    k.x = x; k.y = y; k.w1f = w1h; k.w2f = w2h; k.w3f = w3h;
    k.H = H; k.W = W; k.tiles_x = W / S8_TW; k.tiles_y = H / S8_TH;
    const int64_t n_tiles = (int64_t)B * k.tiles_x * k.tiles_y;
    VQAE_REQUIRE(n_tiles < (1ll << 31), VQAE_ERR_UNSUPPORTED, "same16_16_block: too many tiles");
    k.n_tiles = (int)n_tiles;
    k.b1a = scalars8[0]; k.b1b = scalars8[1]; k.b2a = scalars8[2]; k.b2b = scalars8[3];
    k.b3a = scalars8[4]; k.b3b = scalars8[5]; k.b4 = scalars8[6]; k.scale = scalars8[7];
    const unsigned grid = (unsigned)(n_tiles < 256 * 6 ? n_tiles : 256 * 6);
    if (c == 16) {
        if (dtype == VQAE_DT_BF16) same_small16_kernel<16, VQAE_DT_BF16><<<grid, 256, 0, stream>>>(k);
        else same_small16_kernel<16, VQAE_DT_F16><<<grid, 256, 0, stream>>>(k);
    } else {
        if (dtype == VQAE_DT_BF16) same_small16_kernel<32, VQAE_DT_BF16><<<grid, 256, 0, stream>>>(k);
        else same_small16_kernel<32, VQAE_DT_F16><<<grid, 256, 0, stream>>>(k);
    }
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

}  // namespace vqae
