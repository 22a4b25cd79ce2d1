// MBConv / EfficientNetV2 variant of the conv stack (reference vq_ae/layers/conv_block.py:240-321,
// vq_ae/layers/misc.py:7-30; conf/model/layers/conv_block/mbconv.yaml), inference mode: the BatchNorms are folded
// into the preceding convs by the caller, so a block is
//   1x1 expand (+bias, SiLU)  -> conv_mfma.hip            (MFMA)
//   depthwise conv (+bias, SiLU) + per-(image, channel) sums for the squeeze   -> dw_kernel   (HBM-bound VALU)
//   squeeze-excite gate: mean -> Linear -> SiLU -> Linear -> sigmoid           -> se_gate_kernel (tiny)
//   1x1 project of (x * gate) (+bias) + skip                                   -> conv_mfma.hip, gated operand load
// The three kernels here are byte movers: NHWC, one float4 (4 channels) per lane, consecutive lanes on consecutive
// channel groups / pixels, so every wavefront access is a run of contiguous 16-byte words.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int DW_PPS = 256;                 // output pixels per workgroup ("strip"): one partial sum row per strip

__device__ __forceinline__ float silu1(float v) { return v / (1.0f + expf(-v)); }   // x * sigmoid(x), SiLU
__device__ __forceinline__ f32x4 silu4(f32x4 v) {
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = silu1(v[e]);
    return r;
}

// MODE 0: 3x3 / stride 1 / circular pad      y[oy][ox][c] = sum_{dy,dx} w[dy*3+dx][c] * x[(oy+dy-1) mod H][(ox+dx-1) mod W][c]
// MODE 1: 2x2 / stride 2                     y[oy][ox][c] = sum_{a,b}  w[a*2+b][c]   * x[2oy+a][2ox+b][c]
// MODE 2: ConvTranspose2d 2x2 / stride 2     y[2i+a][2j+b][c] = w[a*2+b][c] * x[i][j][c]
// grid = (strips, B); thread (pl, cg): channel group cg (4 channels), pixels pl, pl + PL, ... of the strip.
template <int MODE>
__global__ __launch_bounds__(256)
void dw_kernel(const f32x4* __restrict__ x, const f32x4* __restrict__ w, const f32x4* __restrict__ bias, int H, int W,
               int C4, int Ho, int Wo, int act, f32x4* __restrict__ y, f32x4* __restrict__ partial, int n_strips) {
    constexpr int TAPS = MODE == 0 ? 9 : 4;
    __shared__ f32x4 red[256];
    const int tid = threadIdx.x;
    const int PL = 256 / C4;                                   // pixel lanes (C4 <= 256)
    const int cg = tid % C4, pl = tid / C4;
    const bool live = pl < PL;
    const int b = blockIdx.y, strip = blockIdx.x;
    const int hw_o = Ho * Wo;
    const int p_end = min((strip + 1) * DW_PPS, hw_o);
    const f32x4* xb = x + (int64_t)b * H * W * C4;
    f32x4* yb = y + (int64_t)b * hw_o * C4;

    f32x4 wt[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) wt[t] = w[t * C4 + cg];
    const f32x4 bv = bias ? bias[cg] : (f32x4)(0.f);
    f32x4 sum = (f32x4)(0.f);
    if (live) {
        for (int p = strip * DW_PPS + pl; p < p_end; p += PL) {
            const int oy = p / Wo, ox = p - oy * Wo;
            f32x4 acc;
            if (MODE == 0) {
                const int ym = oy == 0 ? H - 1 : oy - 1, yp = oy == H - 1 ? 0 : oy + 1;
                const int xm = ox == 0 ? W - 1 : ox - 1, xp = ox == W - 1 ? 0 : ox + 1;
                const int ys[3] = {ym, oy, yp}, xs[3] = {xm, ox, xp};
                f32x4 v[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) v[t] = xb[((int64_t)ys[t / 3] * W + xs[t % 3]) * C4 + cg];
                acc = v[0] * wt[0];
#pragma unroll
                for (int t = 1; t < 9; ++t) acc = acc + v[t] * wt[t];
            } else if (MODE == 1) {
                f32x4 v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = xb[((int64_t)(2 * oy + t / 2) * W + 2 * ox + t % 2) * C4 + cg];
                acc = v[0] * wt[0];
#pragma unroll
                for (int t = 1; t < 4; ++t) acc = acc + v[t] * wt[t];
            } else {
                const f32x4 v = xb[((int64_t)(oy >> 1) * W + (ox >> 1)) * C4 + cg];
                const int t = (oy & 1) * 2 + (ox & 1);
                const f32x4 ws = t == 0 ? wt[0] : (t == 1 ? wt[1] : (t == 2 ? wt[2] : wt[3]));
                acc = v * ws;
            }
            acc = acc + bv;
            if (act) acc = silu4(acc);
            yb[(int64_t)p * C4 + cg] = acc;
            sum = sum + acc;
        }
    }
    if (!partial) return;
    // deterministic reduction over the pixel lanes (fixed order), one row of partial sums per strip
    red[tid] = sum;
    __syncthreads();
    if (pl == 0) {
        f32x4 s = red[cg];
        for (int q = 1; q < PL; ++q) s = s + red[q * C4 + cg];
        partial[((int64_t)b * n_strips + strip) * C4 + cg] = s;
    }
}

// SELayer.forward (layers/misc.py:23-30) for one image per workgroup:
//   mean[c] = sum_strips partial / (Ho*Wo);  hid = SiLU(W0 mean + b0);  gate = sigmoid(W2 hid + b2)
__global__ __launch_bounds__(256)
void se_gate_kernel(const float* __restrict__ partial, int n_strips, float inv_hw, int C, const float* __restrict__ w0,
                    const float* __restrict__ b0, int hidden, const float* __restrict__ w2,
                    const float* __restrict__ b2, float* __restrict__ gate) {
    extern __shared__ float sm[];                              // mean[C] then hid[hidden]
    float* mean = sm;
    float* hid = sm + C;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pb = partial + (int64_t)b * n_strips * C;
    for (int c = tid; c < C; c += 256) {
        float s = 0.f;
        for (int q = 0; q < n_strips; ++q) s += pb[(int64_t)q * C + c];
        mean[c] = s * inv_hw;
    }
    __syncthreads();
    for (int j = wave; j < hidden; j += 4) {                   // one wave per hidden unit, lanes over the inputs
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += w0[(int64_t)j * C + c] * mean[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) hid[j] = silu1(s + b0[j]);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = b2[c];
        for (int j = 0; j < hidden; ++j) s += w2[(int64_t)c * hidden + j] * hid[j];
        gate[(int64_t)b * C + c] = 1.0f / (1.0f + expf(-s));
    }
}

// [B][H][W][(a, b, c)] -> [B][2H][2W][c]: the pixel placement of a stride-2 / kernel-2 transposed conv whose
// channel mixing ran as a 1x1 conv with 4*C outputs.
__global__ void pixel_shuffle2_kernel(const f32x4* __restrict__ x, int H, int W, int C4, int64_t total, f32x4* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // over output float4s
    if (i >= total) return;
    const int c = (int)(i % C4);
    int64_t p = i / C4;
    const int ox = (int)(p % (2 * W)); p /= 2 * W;
    const int oy = (int)(p % (2 * H));
    const int64_t b = p / (2 * H);
    const int t = (oy & 1) * 2 + (ox & 1);
    y[i] = x[(((b * H + (oy >> 1)) * W + (ox >> 1)) * 4 + t) * C4 + c];
}

}  // namespace

extern "C" size_t vqae_dw_partial_floats(int batch, int out_h, int out_w, int channels) {
    return (size_t)(batch > 0 ? batch : 1) * (size_t)vqae::ceil_div((int64_t)out_h * out_w, DW_PPS) * (size_t)channels;
}

extern "C" int vqae_dwconv_f32(const float* x, const float* w_taps, const float* bias, int batch, int h, int w,
                               int channels, int mode, int silu, float* y, float* partial, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0) return VQAE_OK;
    VQAE_REQUIRE(x && w_taps && y, VQAE_ERR_INVALID, "dwconv: null pointer");
    VQAE_REQUIRE(batch > 0 && h >= 1 && w >= 1, VQAE_ERR_INVALID, "dwconv: bad shape");
    VQAE_REQUIRE(channels >= 4 && channels % 4 == 0 && channels <= 1024, VQAE_ERR_UNSUPPORTED,
                 "dwconv: channels %d must be a multiple of 4, at most 1024", channels);
    VQAE_REQUIRE(mode >= VQAE_DW_SAME && mode <= VQAE_DW_UP, VQAE_ERR_INVALID, "dwconv: mode %d", mode);
    if (mode == VQAE_DW_DOWN) VQAE_REQUIRE(h % 2 == 0 && w % 2 == 0, VQAE_ERR_INVALID, "dwconv: odd size for stride 2");
    const int Ho = mode == VQAE_DW_DOWN ? h / 2 : (mode == VQAE_DW_UP ? 2 * h : h);
    const int Wo = mode == VQAE_DW_DOWN ? w / 2 : (mode == VQAE_DW_UP ? 2 * w : w);
    const int strips = (int)vqae::ceil_div((int64_t)Ho * Wo, DW_PPS);
    VQAE_REQUIRE(batch <= 65535, VQAE_ERR_UNSUPPORTED, "dwconv: batch %d > 65535", batch);
    dim3 grid((unsigned)strips, (unsigned)batch);
    const int C4 = channels / 4;
    const f32x4 *x4 = (const f32x4*)x, *w4 = (const f32x4*)w_taps, *b4 = (const f32x4*)bias;
    if (mode == VQAE_DW_SAME)
        dw_kernel<0><<<grid, 256, 0, stream>>>(x4, w4, b4, h, w, C4, Ho, Wo, silu, (f32x4*)y, (f32x4*)partial, strips);
    else if (mode == VQAE_DW_DOWN)
        dw_kernel<1><<<grid, 256, 0, stream>>>(x4, w4, b4, h, w, C4, Ho, Wo, silu, (f32x4*)y, (f32x4*)partial, strips);
    else
        dw_kernel<2><<<grid, 256, 0, stream>>>(x4, w4, b4, h, w, C4, Ho, Wo, silu, (f32x4*)y, (f32x4*)partial, strips);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

extern "C" int vqae_se_gate_f32(const float* partial, int batch, int out_h, int out_w, int channels, const float* fc0_w,
                                const float* fc0_b, int hidden, const float* fc2_w, const float* fc2_b, float* gate,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0) return VQAE_OK;
    VQAE_REQUIRE(partial && fc0_w && fc0_b && fc2_w && fc2_b && gate, VQAE_ERR_INVALID, "se_gate: null pointer");
    VQAE_REQUIRE(batch > 0 && out_h >= 1 && out_w >= 1 && channels >= 1 && hidden >= 1, VQAE_ERR_INVALID, "se_gate: bad shape");
    VQAE_REQUIRE((size_t)(channels + hidden) * 4 <= 64 * 1024, VQAE_ERR_UNSUPPORTED, "se_gate: channels + hidden too large");
    const int strips = (int)vqae::ceil_div((int64_t)out_h * out_w, DW_PPS);
    se_gate_kernel<<<batch, 256, (size_t)(channels + hidden) * 4, stream>>>(
        partial, strips, 1.0f / (float)((int64_t)out_h * out_w), channels, fc0_w, fc0_b, hidden, fc2_w, fc2_b, gate);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

extern "C" int vqae_pixel_shuffle2_f32(const float* x, int batch, int h, int w, int c, float* y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (batch == 0) return VQAE_OK;
    VQAE_REQUIRE(x && y, VQAE_ERR_INVALID, "pixel_shuffle2: null pointer");
    VQAE_REQUIRE(batch > 0 && h >= 1 && w >= 1 && c >= 4 && c % 4 == 0, VQAE_ERR_INVALID, "pixel_shuffle2: bad shape");
    const int64_t total = (int64_t)batch * 4 * h * w * (c / 4);
    pixel_shuffle2_kernel<<<(unsigned)vqae::ceil_div(total, 256), 256, 0, stream>>>((const f32x4*)x, h, w, c / 4, total, (f32x4*)y);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}
