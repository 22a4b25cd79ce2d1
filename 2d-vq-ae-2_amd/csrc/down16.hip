// 16-bit (torch.autocast) form of the 'down' Fixup block (reference vq_ae/layers/conv_block.py:196-216, mode 'down') in ONE
// launch, on v_mfma_f32_32x32x16_{bf16,f16}:
//   t1  = ELU(conv1(ELU(x + b1a) + b1b) + b2a) + b2b          conv1: 1x1, CI -> CO          (CO = 2 CI = branch width)
//   t2  = ELU(conv2(t1) + b3a) + b3b                          conv2: 2x2 / stride 2, CO -> CO
//   out = conv3(t2) * scale + b4 + skip_conv(x + b1c) + b1d   conv3: 1x1; skip_conv: 2x2 / stride 2, CI -> CO
// with autocast's rounding points: every conv operand and every conv result is rounded (RNE) to the 16-bit type, products
// accumulate in fp32, everything between the convs is fp32.
//
// Why a second kernel beside down_fused.hip: that one runs the 16-bit modes on the fp32 MFMA (exact, but 77 GFLOP per call at
// the stem-side level = 0.5 ms of matrix pipe at 157 TFLOP/s: MFMA-bound at 1.1 ms).  On the 16-bit MFMA the same work is
// 18 instructions per wave and tile, and the block is what it should be: one read of x (fp32) and one write of out.
//
// A 256-thread workgroup owns TPX = 4096 / max(CO, 32) output pixels (4, 4, 2 or 1 output rows of 32 for CI = 8 .. 64) and
// their 2x2 input patches.
// Weights are the MFMA row operand (fragment order, 1 KiB wave-wide loads from L2), so a lane holds 4 consecutive channels
// of one pixel per register quad.  t1 / t2 live in LDS as 16-bit -- the values autocast has already rounded:
//   phase 1  conv1 on the 4 TPX input pixels, rows from global (all of a wave's loads requested up front: the launch's
//            HBM read) -> T1[out pixel][tap * CO + c]
//   phase 2  conv2 from T1 (K = 4 CO); skip_conv from global (L2 hits, requested before conv2) -> T2[out pixel][c]
//   phase 3  conv3 from T2, epilogue, fp32 store.
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
using vqae::elu_act;
using vqae::lds_barrier;

template <int DT> struct D16;
template <> struct D16<VQAE_DT_BF16> {
    using x8 = bf16x8; using x4 = bf16x4;
    static __device__ __forceinline__ f32x16 mma(const x8& a, const x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(__bf16)v; }
};
template <> struct D16<VQAE_DT_F16> {
    using x8 = f16x8; using x4 = f16x4;
    static __device__ __forceinline__ f32x16 mma(const x8& a, const x8& b, const f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float rnd(float v) { return (float)(_Float16)v; }
};

struct Down16K {
    const float* __restrict__ x;         // [B][H][W][CI] fp32
    const void* __restrict__ w1;         // 16-bit fragment order [CO/32][K/16][64][8]:  [CO][CI]
    const void* __restrict__ w2;         //   [CO][4 CO]  (k = tap * CO + c)
    const void* __restrict__ w3;         //   [CO][CO]
    const void* __restrict__ wsk;        //   [CO][4 CI]  (k = tap * CI + c)
    float* __restrict__ y;               // [B][H/2][W/2][CO] fp32
    int H, W;                            // input size; W / 2 is a multiple of 32
    int tiles_x, tiles_y;                // tiles of ROWS x 32 output pixels
    float b1a, b1b, b2a, b2b, b3a, b3b, b4, scale, b1c, b1d;
};

template <int CI, int DT>
__global__ __launch_bounds__(256, 2)
void down16_kernel(const Down16K p) {
    using E = D16<DT>;
    using x8 = typename E::x8;
    using x4 = typename E::x4;
    constexpr int CO = 2 * CI;
    constexpr int COP = CO < 32 ? 32 : CO;            // MFMA rows (CI = 8: 16 channels + 16 zero rows of the packed weights)
    constexpr int TPX = 4096 / COP;                   // output pixels per workgroup
    constexpr int ROWS = TPX / 32;                    // output rows per workgroup
    constexpr int NT = COP / 32;                      // 32-channel output tiles
    constexpr int NQ = CO >= 32 ? 4 : CO / 8;         // register quads of a lane that hold real channels (8 g + 4 hh ..)
    constexpr int PS1 = 4 * CO * 2 + 16;              // T1 bytes per output pixel (odd number of 16-B slots: conflict-free b128 reads)
    constexpr int PS2 = CO * 2 + 16;                  // T2
    extern __shared__ __attribute__((aligned(16))) char lds[];       // T1[TPX][PS1] | T2[TPX][PS2]
    char* const T1 = lds;
    char* const T2 = lds + TPX * PS1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;

    const int tile = blockIdx.x;
    const int txi = tile % p.tiles_x;
    const int tyi = (tile / p.tiles_x) % p.tiles_y;
    const int64_t b = tile / (p.tiles_x * p.tiles_y);
    const int oy0 = tyi * ROWS, ox0 = txi * 32;
    const int Ho = p.H / 2, Wo = p.W / 2;
    const float* const xim = p.x + b * (int64_t)p.H * p.W * CI;

    auto wfrag = [&](const void* __restrict__ w, int ks_total, int ct, int u) -> x8 {
        return *reinterpret_cast<const x8*>((const char*)w + ((int64_t)(ct * ks_total + u) * 64 + lane) * 16);
    };
    auto cvt8 = [](const f32x4& a, const f32x4& c) -> x8 {
        const f32x8 v = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
        return __builtin_convertvector(v, x8);
    };

    // ---- phase 1: conv1 on the 2 ROWS x 64 input pixels -> T1 ------------------------------------------------------------
    // a wave owns PGW groups of 32 consecutive input pixels; lane (li, hh) holds channels 16 u + 8 hh .. + 8 of pixel li
    constexpr int PGW = TPX / 8 / 4;                  // 4, 2, 1
    constexpr int KU = CI < 16 ? 1 : CI / 16;         // k-steps of conv1 (CI = 8: K padded to 16 with zero weights; lanes hh = 1 idle)
    const bool kvalid = CI >= 16 || hh == 0;
    f32x4 xin[PGW][KU][2];
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
        const int pg = wave * PGW + i;
        const int irow = pg >> 1, ix = (pg & 1) * 32 + li;
        const float* src = xim + ((int64_t)(2 * oy0 + irow) * p.W + 2 * ox0 + ix) * CI + 8 * hh;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            xin[i][u][0] = kvalid ? *reinterpret_cast<const f32x4*>(src + 16 * u) : f32x4{0.f, 0.f, 0.f, 0.f};
            xin[i][u][1] = kvalid ? *reinterpret_cast<const f32x4*>(src + 16 * u + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    x8 w1f[NT][KU];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct)
#pragma unroll
        for (int u = 0; u < KU; ++u) w1f[ct][u] = wfrag(p.w1, KU, ct, u);
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
        const int pg = wave * PGW + i;
        const int irow = pg >> 1, ix = (pg & 1) * 32 + li;
        x8 xa[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xin[i][u][0][e] = elu_act(xin[i][u][0][e] + p.b1a) + p.b1b;
                xin[i][u][1][e] = elu_act(xin[i][u][1][e] + p.b1a) + p.b1b;
            }
            xa[u] = cvt8(xin[i][u][0], xin[i][u][1]);                                      // conv1 input cast
        }
        char* const dst = T1 + ((irow >> 1) * 32 + (ix >> 1)) * PS1 + (((irow & 1) * 2 + (ix & 1)) * CO + 4 * hh) * 2;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int u = 0; u < KU; ++u) acc = E::mma(w1f[ct][u], xa[u], acc);             // D[channel][pixel]
#pragma unroll
            for (int g = 0; g < NQ; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = elu_act(E::rnd(acc[4 * g + e]) + p.b2a) + p.b2b;   // conv1 output cast
                *reinterpret_cast<x4*>(dst + (32 * ct + 8 * g) * 2) = __builtin_convertvector(o, x4);  // conv2 input cast
            }
        }
    }

    // ---- phase 2: conv2 (K = 4 CO) from T1; skip_conv (K = 4 CI) from global ---------------------------------------------
    const int pg2 = wave / NT, ct = wave % NT;         // this wave's 32 output pixels and 32 output channels
    const int px = 32 * pg2 + li;                       // output pixel within the tile
    const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
    constexpr int KSK = 4 * CI / 16;
    constexpr int SKB = KSK < 8 ? KSK : 8;             // skip k-steps in flight per lane (2 x 16 B each)
    auto sk_addr = [&](int u) {                         // k = 16 u + 8 hh + j = tap * CI + c
        const int kk = 16 * u + 8 * hh;
        const int tap = kk / CI, c0 = kk % CI;
        return xim + ((int64_t)(2 * oy + (tap >> 1)) * p.W + 2 * ox + (tap & 1)) * CI + c0;
    };
    f32x4 xs[SKB][2];
#pragma unroll
    for (int u = 0; u < SKB; ++u) {
        xs[u][0] = *reinterpret_cast<const f32x4*>(sk_addr(u));
        xs[u][1] = *reinterpret_cast<const f32x4*>(sk_addr(u) + 4);
    }
    constexpr int KS2 = 4 * CO / 16;
    constexpr int WR = KS2 < 4 ? KS2 : 4;              // conv2 weight fragments in flight
    x8 wq[WR];
#pragma unroll
    for (int u = 0; u < WR; ++u) wq[u] = wfrag(p.w2, KS2, ct, u);
    lds_barrier();                                      // T1 complete (LDS-only barrier: the loads above stay in flight)
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    {
        const char* a0 = T1 + px * PS1 + 16 * hh;
        x8 aq[2];
        aq[0] = *reinterpret_cast<const x8*>(a0);
#pragma unroll
        for (int u = 0; u < KS2; ++u) {
            const x8 wv = wq[u % WR];
            if (u + WR < KS2) wq[u % WR] = wfrag(p.w2, KS2, ct, u + WR);
            if (u + 1 < KS2) aq[(u + 1) & 1] = *reinterpret_cast<const x8*>(a0 + 32 * (u + 1));
            acc2 = E::mma(wv, aq[u & 1], acc2);
        }
    }
    {
        char* const dst = T2 + px * PS2 + (32 * ct + 4 * hh) * 2;
#pragma unroll
        for (int g = 0; g < NQ; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = elu_act(E::rnd(acc2[4 * g + e]) + p.b3a) + p.b3b;     // conv2 output cast
            *reinterpret_cast<x4*>(dst + 16 * g) = __builtin_convertvector(o, x4);                    // conv3 input cast
        }
    }
    constexpr int KS3 = CO / 16;
    x8 w3f[KS3];
#pragma unroll
    for (int u = 0; u < KS3; ++u) w3f[u] = wfrag(p.w3, KS3, ct, u);
    f32x16 accs;
#pragma unroll
    for (int r = 0; r < 16; ++r) accs[r] = 0.f;
#pragma unroll
    for (int u = 0; u < KSK; ++u) {
        f32x4 v0 = xs[u % SKB][0], v1 = xs[u % SKB][1];
        if (u + SKB < KSK) {
            xs[u % SKB][0] = *reinterpret_cast<const f32x4*>(sk_addr(u + SKB));
            xs[u % SKB][1] = *reinterpret_cast<const f32x4*>(sk_addr(u + SKB) + 4);
        }
        const x8 wv = wfrag(p.wsk, KSK, ct, u);
        v0 = v0 + p.b1c;
        v1 = v1 + p.b1c;
        accs = E::mma(wv, cvt8(v0, v1), accs);                                                         // skip_conv input cast
    }
    lds_barrier();                                      // T2 complete

    // ---- phase 3: conv3 from T2, epilogue ----------------------------------------------------------------------------------
    f32x16 acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
    {
        const char* a0 = T2 + px * PS2 + 16 * hh;
#pragma unroll
        for (int u = 0; u < KS3; ++u) acc3 = E::mma(w3f[u], *reinterpret_cast<const x8*>(a0 + 32 * u), acc3);
    }
    float* out = p.y + ((b * Ho + oy) * Wo + ox) * CO + 32 * ct + 4 * hh;
#pragma unroll
    for (int g = 0; g < NQ; ++g) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = E::rnd(acc3[4 * g + e]) * p.scale;   // branch: conv3 * scale + bias4
            t = t + p.b4;
            o[e] = t + (E::rnd(accs[4 * g + e]) + p.b1d);  // + skip_conv(x + b1c) + b1d
        }
        *reinterpret_cast<f32x4*>(out + 8 * g) = o;
    }
}

// packed fp32 [>= n_rows][k_src] (vqae_conv_pack_weight_f32: rows beyond cout are zero; already rounded to the 16-bit
// type) -> fragment order [n_rows/32][K/16][64 lanes][8], K = k_src rounded up to 16 with zeros:
// lane (r, h) of k-step ks holds w[32 nt + r][16 ks + 8 h + j], j = 0..7
template <typename EL>
__global__ void pack16_rect_kernel(const float* __restrict__ w, int n_rows, int k_src, int K, EL* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_rows * K) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const int64_t st = i >> 9;
    const int ks = (int)(st % (K / 16)), nt = (int)(st / (K / 16));
    const int n = nt * 32 + (lane & 31), k = ks * 16 + 8 * (lane >> 5) + j;
    out[i] = k < k_src ? (EL)w[(int64_t)n * k_src + k] : (EL)0.f;
}

template <int CI, int DT>
int launch_down16(const Down16K& k, int64_t n_tiles, hipStream_t stream) {
    constexpr int CO = 2 * CI, TPX = 4096 / (CO < 32 ? 32 : CO);
    constexpr int lds_bytes = TPX * ((4 * CO * 2 + 16) + (CO * 2 + 16));
    static bool attr_set = false;
    if (!attr_set) {
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)down16_kernel<CI, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    down16_kernel<CI, DT><<<(unsigned)n_tiles, 256, lds_bytes, stream>>>(k);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

}  // namespace

namespace vqae {

// cin in {8, 16, 32, 64}; output width a multiple of 32, output height a multiple of the tile's rows (4, 4, 2, 1)
bool down16_supported(int cin, int h, int w) {
    if (cin != 8 && cin != 16 && cin != 32 && cin != 64) return false;
    const int rows = (4096 / (2 * cin < 32 ? 32 : 2 * cin)) / 32;
    return h % 2 == 0 && w % 64 == 0 && (h / 2) % rows == 0;
}

static int pad_rows(int n) { return n < 32 ? 32 : n; }
static int pad_k(int k) { return (k + 15) / 16 * 16; }

size_t down16_weight_bytes(int n_rows, int K) { return (size_t)pad_rows(n_rows) * pad_k(K) * 2; }

// packed [>= max(n_rows, 32)][K] fp32 (device; vqae_conv_pack_weight_f32 pads the rows to 128 with zeros) -> 16-bit fragment
// order (device); n_rows % 32 == 0 or n_rows in {8, 16}; K % 8 == 0 (padded to 16 with zeros)
int down16_pack_weight(const float* w_packed_dev, int n_rows, int K, int dtype, void* out_dev, hipStream_t stream) {
    VQAE_REQUIRE((n_rows % 32 == 0 || n_rows == 16 || n_rows == 8) && K % 8 == 0, VQAE_ERR_INVALID, "down16_pack_weight: %d x %d", n_rows, K);
    VQAE_REQUIRE(dtype == VQAE_DT_BF16 || dtype == VQAE_DT_F16, VQAE_ERR_INVALID, "down16_pack_weight: dtype %d", dtype);
    const int nr = pad_rows(n_rows), kp = pad_k(K);
    const int64_t n = (int64_t)nr * kp;
    if (dtype == VQAE_DT_BF16) pack16_rect_kernel<__bf16><<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(w_packed_dev, nr, K, kp, (__bf16*)out_dev);
    else pack16_rect_kernel<_Float16><<<(unsigned)ceil_div(n, 256), 256, 0, stream>>>(w_packed_dev, nr, K, kp, (_Float16*)out_dev);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// x [B][H][W][cin] fp32 -> y [B][H/2][W/2][2 cin] fp32; weights from down16_pack_weight; scalars10 =
// {b1a, b1b, b2a, b2b, b3a, b3b, b4, scale, b1c, b1d}; dtype bf16 / f16
int down16_block(const float* x, const void* w1h, const void* w2h, const void* w3h, const void* wskh, int B, int H, int W,
                 int cin, const float* scalars10, int dtype, float* y, hipStream_t stream) {
    if (B == 0) return VQAE_OK;
    VQAE_REQUIRE(dtype == VQAE_DT_BF16 || dtype == VQAE_DT_F16, VQAE_ERR_INVALID, "down16_block: dtype %d", dtype);
    VQAE_REQUIRE(x && w1h && w2h && w3h && wskh && y && scalars10, VQAE_ERR_INVALID, "down16_block: null pointer");
    VQAE_REQUIRE(down16_supported(cin, H, W), VQAE_ERR_UNSUPPORTED, "down16_block: cin %d, %dx%d", cin, H, W);
    Down16K k;
    k.x = x; k.w1 = w1h; k.w2 = w2h; k.w3 = w3h; k.wsk = wskh; k.y = y;
    k.H = H; k.W = W;
    const int rows = (4096 / (2 * cin < 32 ? 32 : 2 * cin)) / 32;
    k.tiles_x = (W / 2) / 32; k.tiles_y = (H / 2) / rows;
    k.b1a = scalars10[0]; k.b1b = scalars10[1]; k.b2a = scalars10[2]; k.b2b = scalars10[3]; k.b3a = scalars10[4];
    k.b3b = scalars10[5]; k.b4 = scalars10[6]; k.scale = scalars10[7]; k.b1c = scalars10[8]; k.b1d = scalars10[9];
    const int64_t n_tiles = (int64_t)B * k.tiles_x * k.tiles_y;
    VQAE_REQUIRE(n_tiles < (1ll << 31), VQAE_ERR_UNSUPPORTED, "down16_block: too many tiles");
#define VQAE_D16(CI_) (dtype == VQAE_DT_BF16 ? launch_down16<CI_, VQAE_DT_BF16>(k, n_tiles, stream) : launch_down16<CI_, VQAE_DT_F16>(k, n_tiles, stream))
    if (cin == 8) return VQAE_D16(8);
    if (cin == 16) return VQAE_D16(16);
    if (cin == 32) return VQAE_D16(32);
    return VQAE_D16(64);
#undef VQAE_D16
}

}  // namespace vqae
