// A whole 'down' Fixup block (reference vq_ae/layers/conv_block.py:196-216, mode 'down') in ONE launch:
//   t1  = ELU(conv1(ELU(x + b1a) + b1b) + b2a) + b2b          conv1: 1x1, CI -> CO          (CO = 2 CI = branch width)
//   t2  = ELU(conv2(t1) + b3a) + b3b                          conv2: 2x2 / stride 2, CO -> CO
//   out = conv3(t2) * scale + b4 + skip_conv(x + b1c) + b1d   conv3: 1x1; skip_conv: 2x2 / stride 2, CI -> CO
// Unfused the four convs are HBM-bound launches that move 9.7 GB per call at the stem-side level (batch 256): t1 alone is
// written and read back at twice the input's size.  Here the block reads x and writes out: 1.6 GB.
//
// A 256-thread workgroup owns TPX = 4096 / CO output pixels (whole 32-pixel output rows: 4, 2 or 1 of them) and their
// 2x2 input patches.  All four GEMMs run on v_mfma_f32_32x32x2_f32 with the WEIGHTS as the row operand, so a lane holds
// four consecutive channels of one pixel and every LDS / global access of the epilogues is 128 bits wide:
//   phase 1  conv1 on the 4 TPX input pixels straight from global (pre-op in registers) -> t1 into LDS in conv2's operand
//            layout T1[out pixel][tap * CO + c]
//   phase 2  conv2 from T1 (K = 4 CO)                              -> t2 into LDS T2[out pixel][c] (over T1)
//   phase 3  conv3 from T2 and skip_conv from global, epilogue, store.
// Weight matrices are read from L2 in MFMA fragment order (contiguous wave-wide loads, see conv_wino.hip).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
using vqae::elu_act;

struct DownK {
    const float* __restrict__ x;         // [B][H][W][CI]
    const float* __restrict__ w1;        // fragment order, [CO][CI]
    const float* __restrict__ w2;        // fragment order, [CO][4 CO]  (k = tap * CO + c)
    const float* __restrict__ w3;        // fragment order, [CO][CO]
    const float* __restrict__ wsk;       // fragment order, [CO][4 CI]  (k = tap * CI + c)
    float* __restrict__ y;               // [B][H/2][W/2][CO]
    int H, W;                            // input size; W / 2 is a multiple of 32
    int tiles_x, tiles_y;                // tiles of ROWS x 32 output pixels
    float b1a, b1b, b2a, b2b, b3a, b3b, b4, scale, b1c, b1d;
};

// DT: autocast cast points compiled in (every conv operand and conv output rounded to bf16 / f16; weights arrive
// rounded; products of 16-bit values are exact in the fp32 MFMA, accumulation is fp32 as under torch.autocast)
template <int CI, int DT>
__global__ __launch_bounds__(256, 2)
void down_block_kernel(const DownK p) {
    auto rnd = [](float v) -> float {
        if (DT == VQAE_DT_BF16) return (float)(__bf16)v;
        if (DT == VQAE_DT_F16) return (float)(_Float16)v;
        return v;
    };
    constexpr int CO = 2 * CI;
    constexpr int TPX = 4096 / CO;                    // output pixels per workgroup
    constexpr int ROWS = TPX / 32;                    // output rows per workgroup
    constexpr int NT = CO / 32;                       // 32-channel output tiles
    constexpr int LD1 = 4 * CO + 4;                   // T1 row stride (floats)
    constexpr int LD2 = CO + 4;                       // T2 row stride
    extern __shared__ __attribute__((aligned(16))) float lds[];      // T1[TPX][LD1]  /  T2[TPX][LD2]
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

    // fragment-order weights: [n-tile][k-slice][lane][4]
    auto wfrag = [&](const float* __restrict__ w, int ks_total, int ct, int u) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(w + ((int64_t)(ct * ks_total + u) * 64 + lane) * 4);
    };

    // ---- phase 1: conv1 on the 2 ROWS x 64 input pixels -> T1 ----------------------------------------------------------
    // A wave owns PGW groups of 32 consecutive input pixels.  ALL their rows are requested before anything is computed
    // (8 x 16 B per lane, 64 KiB per CU with two workgroups): this is the launch's HBM read, and with one load in flight
    // per wave -- the first version -- the block ran at 1.4 TB/s, bound by latency x bytes in flight, not bandwidth.
    float* const T1 = lds;
    constexpr int PGW = TPX / 8 / 4;                  // 4, 2, 1 pixel groups per wave
    constexpr int KU = CI / 8;                        // k-slices of conv1
    f32x4 xin[PGW][KU];
#pragma unroll
    for (int i = 0; i < PGW; ++i) {
        const int pg = wave * PGW + i;
        const int irow = pg >> 1, ix = (pg & 1) * 32 + li;
        const float* src = xim + ((int64_t)(2 * oy0 + irow) * p.W + 2 * ox0 + ix) * CI + 4 * hh;
#pragma unroll
        for (int u = 0; u < KU; ++u) xin[i][u] = *reinterpret_cast<const f32x4*>(src + 8 * u);
    }
    f32x4 wf[2][KU];                                  // conv1 weight fragments of one channel tile, the next tile's in flight
#pragma unroll
    for (int u = 0; u < KU; ++u) wf[0][u] = wfrag(p.w1, KU, 0, u);
#pragma unroll
    for (int i = 0; i < PGW; ++i)
#pragma unroll
        for (int u = 0; u < KU; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) xin[i][u][e] = rnd(elu_act(xin[i][u][e] + p.b1a) + p.b1b);   // conv1 input cast
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
        if (ct + 1 < NT) {
#pragma unroll
            for (int u = 0; u < KU; ++u) wf[(ct + 1) & 1][u] = wfrag(p.w1, KU, ct + 1, u);
        }
#pragma unroll
        for (int i = 0; i < PGW; ++i) {
            const int pg = wave * PGW + i;
            const int irow = pg >> 1, ix = (pg & 1) * 32 + li;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[ct & 1][u][r], xin[i][u][r], acc, 0, 0, 0);   // D[channel][pixel]
            float* dst = T1 + ((irow >> 1) * 32 + (ix >> 1)) * LD1 + ((irow & 1) * 2 + (ix & 1)) * CO + 32 * ct + 4 * hh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rnd(elu_act(rnd(acc[4 * g + e]) + p.b2a) + p.b2b);   // conv1 output / conv2 input casts
                *reinterpret_cast<f32x4*>(dst + 8 * g) = o;
            }
        }
    }

    // ---- phase 2: conv2 (K = 4 CO) from T1; phase 3 operands -------------------------------------------------------------
    const int pg2 = wave / NT, ct = wave % NT;         // this wave's 32 output pixels and 32 output channels
    const int px = 32 * pg2 + li;                       // output pixel within the tile
    const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
    // skip_conv operands (the 2x2 input patches, L2 hits: phase 1 just read them) requested ahead of conv2
    constexpr int KSK = 4 * CI / 8;
    constexpr int SKB = KSK < 16 ? KSK : 16;           // loads in flight per lane
    auto sk_addr = [&](int u) {
        const int tap = u / (CI / 8), s_ = u % (CI / 8);
        return xim + ((int64_t)(2 * oy + (tap >> 1)) * p.W + 2 * ox + (tap & 1)) * CI + 8 * s_ + 4 * hh;
    };
    f32x4 xs[SKB];
#pragma unroll
    for (int u = 0; u < SKB; ++u) xs[u] = *reinterpret_cast<const f32x4*>(sk_addr(u));
    constexpr int KS2 = 4 * CO / 8;
    constexpr int WR = 4;                               // conv2 weight fragments in flight (L2 round trip ~ 2 k-slices of MFMAs)
    f32x4 wq[WR];
#pragma unroll
    for (int u = 0; u < WR; ++u) wq[u] = wfrag(p.w2, KS2, ct, u);
    vqae::lds_barrier();                                // T1 complete (LDS-only barrier: the loads above stay in flight)
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    {
        const float* a0 = T1 + px * LD1 + 4 * hh;
        f32x4 aq[2];
        aq[0] = *reinterpret_cast<const f32x4*>(a0);
#pragma unroll
        for (int u = 0; u < KS2; ++u) {
            const f32x4 wv = wq[u % WR];
            if (u + WR < KS2) wq[u % WR] = wfrag(p.w2, KS2, ct, u + WR);
            if (u + 1 < KS2) aq[(u + 1) & 1] = *reinterpret_cast<const f32x4*>(a0 + 8 * (u + 1));
#pragma unroll
            for (int r = 0; r < 4; ++r) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[r], aq[u & 1][r], acc2, 0, 0, 0);
        }
    }
    // skip_conv (K = 4 CI) while the other waves finish conv2
    f32x16 accs;
#pragma unroll
    for (int r = 0; r < 16; ++r) accs[r] = 0.f;
    {
#pragma unroll
        for (int u = 0; u < KSK; ++u) {
            f32x4 v = xs[u % SKB];
            if (u + SKB < KSK) xs[u % SKB] = *reinterpret_cast<const f32x4*>(sk_addr(u + SKB));
            const f32x4 wv = wfrag(p.wsk, KSK, ct, u);
            v = v + p.b1c;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rnd(v[e]);                                   // skip_conv input cast
#pragma unroll
            for (int r = 0; r < 4; ++r) accs = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[r], v[r], accs, 0, 0, 0);
        }
    }
    __syncthreads();                                    // every wave is done reading T1
    float* const T2 = lds;
    {
        float* dst = T2 + px * LD2 + 32 * ct + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rnd(elu_act(rnd(acc2[4 * g + e]) + p.b3a) + p.b3b);  // conv2 output / conv3 input casts
            *reinterpret_cast<f32x4*>(dst + 8 * g) = o;
        }
    }
    __syncthreads();

    // ---- phase 3: conv3 from T2, epilogue ------------------------------------------------------------------------------------
    f32x16 acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
    {
        const float* a0 = T2 + px * LD2 + 4 * hh;
        constexpr int KS = CO / 8;
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(a0 + 8 * u);
            const f32x4 wv = wfrag(p.w3, KS, ct, u);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[r], av[r], acc3, 0, 0, 0);
        }
    }
    float* out = p.y + ((b * Ho + oy) * Wo + ox) * CO + 32 * ct + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = rnd(acc3[4 * g + e]) * p.scale;   // branch: conv3 * scale + bias4
            t = t + p.b4;
            o[e] = t + (rnd(accs[4 * g + e]) + p.b1d);  // + skip_conv(x + b1c) + b1d
        }
        *reinterpret_cast<f32x4*>(out + 8 * g) = o;
    }
}

// packed [n][K] (vqae_conv_pack_weight_f32: K = taps * cin, tap-major) -> MFMA fragment order [n-tile][k-slice][lane][4]
__global__ void frag_rect_kernel(const float* __restrict__ w, int n_rows, int K, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * K) return;
    const int n = i / K, k = i % K;
    out[(((n >> 5) * (K / 8) + (k >> 3)) * 64 + ((k >> 2) & 1) * 32 + (n & 31)) * 4 + (k & 3)] = w[i];
}

template <int CI, int DT>
int launch_down_dt(const DownK& k, int64_t n_tiles, hipStream_t stream) {
    constexpr int CO = 2 * CI, TPX = 4096 / CO;
    constexpr int lds_bytes = TPX * (4 * CO + 4) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        VQAE_HIP_CHECK(hipFuncSetAttribute((const void*)down_block_kernel<CI, DT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
        attr_set = true;
    }
    down_block_kernel<CI, DT><<<(unsigned)n_tiles, 256, lds_bytes, stream>>>(k);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

template <int CI>
int launch_down(const DownK& k, int64_t n_tiles, int dtype, hipStream_t stream) {
    if (dtype == VQAE_DT_BF16) return launch_down_dt<CI, VQAE_DT_BF16>(k, n_tiles, stream);
    if (dtype == VQAE_DT_F16) return launch_down_dt<CI, VQAE_DT_F16>(k, n_tiles, stream);
    return launch_down_dt<CI, VQAE_DT_F32>(k, n_tiles, stream);
}

}  // namespace

namespace vqae {

// cin in {16, 32, 64}; output width a multiple of 32, output height a multiple of the tile's rows (4, 2, 1)
bool down_block_supported(int cin, int h, int w) {
    if (cin != 16 && cin != 32 && cin != 64) return false;
    const int rows = (4096 / (2 * cin)) / 32;
    return h % 2 == 0 && w % 64 == 0 && (h / 2) % rows == 0;
}

// packed [n_rows][K] -> fragment order (n_rows % 32 == 0, K % 8 == 0)
int frag_weight_rect(const float* w_packed_dev, int n_rows, int K, float* out_dev, hipStream_t stream) {
    VQAE_REQUIRE(n_rows % 32 == 0 && K % 8 == 0, VQAE_ERR_INVALID, "frag_weight_rect: %d x %d", n_rows, K);
    frag_rect_kernel<<<(unsigned)ceil_div((int64_t)n_rows * K, 256), 256, 0, stream>>>(w_packed_dev, n_rows, K, out_dev);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// x [B][H][W][cin] -> y [B][H/2][W/2][2 cin]; weights in fragment order (frag_weight_rect); scalars10 =
// {b1a, b1b, b2a, b2b, b3a, b3b, b4, scale, b1c, b1d}
int down_block(const float* x, const float* w1f, const float* w2f, const float* w3f, const float* wskf, int B, int H, int W,
               int cin, const float* scalars10, int dtype, float* y, hipStream_t stream) {
    if (B == 0) return VQAE_OK;
    VQAE_REQUIRE(dtype >= VQAE_DT_F32 && dtype <= VQAE_DT_F16, VQAE_ERR_INVALID, "down_block: dtype %d", dtype);
    VQAE_REQUIRE(x && w1f && w2f && w3f && wskf && y && scalars10, VQAE_ERR_INVALID, "down_block: null pointer");
    VQAE_REQUIRE(down_block_supported(cin, H, W), VQAE_ERR_UNSUPPORTED, "down_block: cin %d, %dx%d", cin, H, W);
    DownK k;
    k.x = x; k.w1 = w1f; k.w2 = w2f; k.w3 = w3f; k.wsk = wskf; k.y = y;
    k.H = H; k.W = W;
    const int rows = (4096 / (2 * cin)) / 32;
    k.tiles_x = (W / 2) / 32; k.tiles_y = (H / 2) / rows;
    k.b1a = scalars10[0]; k.b1b = scalars10[1]; k.b2a = scalars10[2]; k.b2b = scalars10[3]; k.b3a = scalars10[4];
    k.b3b = scalars10[5]; k.b4 = scalars10[6]; k.scale = scalars10[7]; k.b1c = scalars10[8]; k.b1d = scalars10[9];
    const int64_t n_tiles = (int64_t)B * k.tiles_x * k.tiles_y;
    VQAE_REQUIRE(n_tiles < (1ll << 31), VQAE_ERR_UNSUPPORTED, "down_block: too many tiles");
    if (cin == 16) return launch_down<16>(k, n_tiles, dtype, stream);
    if (cin == 32) return launch_down<32>(k, n_tiles, dtype, stream);
    return launch_down<64>(k, n_tiles, dtype, stream);
}

}  // namespace vqae
