// HBM-bound helper kernels of the VQ-AE hot path (gfx950): the 3-channel stems, bicubic x2,
// boundary layout shuffles, label max-pool and slide-grid stitching.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Direct 3x3 / stride 1 / zero-pad conv + per-channel bias: `in_stem` (3 -> C0, reference
// vq_ae/model.py:198, conv2d.yaml: bias True, padding_mode zeros) and `out_stem` (C0 -> 3, model.py:291).
// One thread per output pixel, weights broadcast from LDS.  4 FLOP/B: HBM-bound.
// x_kind: 0 fp32 NHWC, 1 fp32 NCHW, 2 uint8 NHWC normalised on the fly
//   ((u - mean255[c]) * inv_std255[c]: albumentations Normalize, camelyon16_transforms.yaml:15-23).
// ------------------------------------------------------------------------------------------------
struct Norm3 { float mean[4]; float inv[4]; };

template <int CIN, int COUT>
__global__ __launch_bounds__(256)
void conv3x3_direct_kernel(const void* __restrict__ xin, int x_kind, Norm3 nrm, const float* __restrict__ w,
                           const float* __restrict__ bias, int B, int H, int W, float* __restrict__ y, int y_nchw,
                           int dt) {
    __shared__ float ws[9 * CIN * COUT + COUT];          // [tap][ci][co], then bias
    for (int i = threadIdx.x; i < 9 * CIN * COUT; i += 256) {
        const int co = i % COUT, ci = (i / COUT) % CIN, tap = i / (COUT * CIN);
        ws[i] = vqae::round_dt(w[((int64_t)co * CIN + ci) * 9 + tap], dt);   // PyTorch [co][ci][kh][kw]
    }
    for (int i = threadIdx.x; i < COUT; i += 256) ws[9 * CIN * COUT + i] = vqae::round_dt(bias[i], dt);
    __syncthreads();

    const int64_t npix = (int64_t)B * H * W;
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= npix) return;
    const int hw = H * W;
    const int b = (int)(pix / hw), rem = (int)(pix - (int64_t)b * hw);
    const int oy = rem / W, ox = rem - oy * W;

    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = ws[9 * CIN * COUT + co];

#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy + dy - 1;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int ix = ox + dx - 1;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            const int tap = dy * 3 + dx;
            float xv[CIN];
            if (x_kind == 0) {
                const float* src = (const float*)xin + ((int64_t)b * hw + (int64_t)iy * W + ix) * CIN;
                if constexpr (CIN % 4 == 0) {
#pragma unroll
                    for (int c = 0; c < CIN; c += 4) {
                        const float4 v = *reinterpret_cast<const float4*>(src + c);
                        xv[c] = v.x; xv[c + 1] = v.y; xv[c + 2] = v.z; xv[c + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CIN; ++c) xv[c] = src[c];
                }
            } else if (x_kind == 1) {
#pragma unroll
                for (int c = 0; c < CIN; ++c)
                    xv[c] = ((const float*)xin)[((int64_t)b * CIN + c) * hw + (int64_t)iy * W + ix];
            } else {
                const uint8_t* src = (const uint8_t*)xin + ((int64_t)b * hw + (int64_t)iy * W + ix) * CIN;
#pragma unroll
                for (int c = 0; c < CIN; ++c) xv[c] = ((float)src[c] - nrm.mean[c & 3]) * nrm.inv[c & 3];
            }
            if (dt) {
#pragma unroll
                for (int c = 0; c < CIN; ++c) xv[c] = vqae::round_dt(xv[c], dt);
            }
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float* wr = ws + (tap * CIN + c) * COUT;
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] = __builtin_fmaf(xv[c], wr[co], acc[co]);
            }
        }
    }
    if (dt) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] = vqae::round_dt(acc[co], dt);
    }
    if (y_nchw) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) y[((int64_t)b * COUT + co) * hw + rem] = acc[co];
    } else {
        float* dst = y + pix * COUT;
        if constexpr (COUT % 4 == 0) {
#pragma unroll
            for (int co = 0; co < COUT; co += 4)
                *reinterpret_cast<float4*>(dst + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
        } else {
#pragma unroll
            for (int co = 0; co < COUT; ++co) dst[co] = acc[co];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Bicubic x2 (A = -0.75, align_corners = False, clamped indices): nn.Upsample in ResizeConv2D
// (reference vq_ae/layers/conv.py:8).  For scale 2 the source offset is .75 (even outputs, taps at
// i-2..i+1) or .25 (odd outputs, taps at i-1..i+2) with i = o >> 1; out = sum_i wy_i * (sum_j wx_j * v_ij),
// both sums left to right, unfused (oracle: bicubic_up2_explicit).
// ------------------------------------------------------------------------------------------------
// One thread = one 2x2 output block (oy in {2i+1, 2i+2}, ox in {2j+1, 2j+2}): these four outputs read the same 4x4
// input window (rows i-1..i+2, columns j-1..j+2), so it is loaded once -- 4 loads per output instead of 16.  Blocks
// i = -1 and i = H-1 (j likewise) have one valid row (column).  Per-output arithmetic and its order are unchanged.
__global__ __launch_bounds__(256)
void bicubic_up2_kernel(const float4* __restrict__ x, int B, int H, int W, int C4, float pre_bias,
                        float4* __restrict__ y) {
    const float w75[4] = {-0.03515625f, 0.26171875f, 0.87890625f, -0.10546875f};   // even outputs (offset .75)
    const float w25[4] = {-0.10546875f, 0.87890625f, 0.26171875f, -0.03515625f};   // odd outputs  (offset .25)
    const int OW = 2 * W, OH = 2 * H;
    const int64_t total = (int64_t)B * (H + 1) * (W + 1) * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        int64_t p = i / C4;
        const int bj = (int)(p % (W + 1)) - 1; p /= (W + 1);
        const int bi = (int)(p % (H + 1)) - 1;
        const int b = (int)(p / (H + 1));
        int xs[4], ys[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int v = bj - 1 + j; xs[j] = v < 0 ? 0 : (v > W - 1 ? W - 1 : v);
            int u = bi - 1 + j; ys[j] = u < 0 ? 0 : (u > H - 1 ? H - 1 : u);
        }
        float4 ho[4], he[4];                            // horizontally interpolated rows: odd / even output column
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4* row = x + ((int64_t)b * H + ys[r]) * W * C4 + c4;
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = row[(int64_t)xs[j] * C4];
                v[j].x += pre_bias; v[j].y += pre_bias; v[j].z += pre_bias; v[j].w += pre_bias;
            }
            ho[r].x = v[0].x * w25[0]; ho[r].y = v[0].y * w25[0]; ho[r].z = v[0].z * w25[0]; ho[r].w = v[0].w * w25[0];
            he[r].x = v[0].x * w75[0]; he[r].y = v[0].y * w75[0]; he[r].z = v[0].z * w75[0]; he[r].w = v[0].w * w75[0];
#pragma unroll
            for (int j = 1; j < 4; ++j) {
                ho[r].x = ho[r].x + v[j].x * w25[j]; ho[r].y = ho[r].y + v[j].y * w25[j];
                ho[r].z = ho[r].z + v[j].z * w25[j]; ho[r].w = ho[r].w + v[j].w * w25[j];
                he[r].x = he[r].x + v[j].x * w75[j]; he[r].y = he[r].y + v[j].y * w75[j];
                he[r].z = he[r].z + v[j].z * w75[j]; he[r].w = he[r].w + v[j].w * w75[j];
            }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {                   // a = 0: odd output row 2 bi + 1, a = 1: even row 2 bi + 2
            const int oy = 2 * bi + 1 + a;
            if (oy < 0 || oy >= OH) continue;
            const float* wy = a == 0 ? w25 : w75;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ox = 2 * bj + 1 + c;
                if (ox < 0 || ox >= OW) continue;
                const float4* in = c == 0 ? ho : he;
                float4 out;
                out.x = in[0].x * wy[0]; out.y = in[0].y * wy[0]; out.z = in[0].z * wy[0]; out.w = in[0].w * wy[0];
#pragma unroll
                for (int r = 1; r < 4; ++r) {
                    out.x = out.x + in[r].x * wy[r]; out.y = out.y + in[r].y * wy[r];
                    out.z = out.z + in[r].z * wy[r]; out.w = out.w + in[r].w * wy[r];
                }
                y[(((int64_t)b * OH + oy) * OW + ox) * C4 + c4] = out;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Tail of an 'up' Fixup block at the stem-side levels, fused (fp32, conv-before-resize order of handle.hip):
//   out = conv3(ELU(bicubic_x2(q) + b3a) + b3b) * scale + b4 + bicubic_x2(s)
// q [B][H][W][CB] = branch_conv2 output at the low resolution, s [B][H][W][CO] = skip_conv output, out [B][2H][2W][CO].
// Unfused this is two resize launches that write 4x-larger tensors and a 1x1 conv that reads them back (8 GB per call
// at 256x256x32 for a batch of 256).  A workgroup owns 2 x 16 resize blocks = 4 x 32 output pixels:
//   phase 1  one (block, channel group) item per thread and step: the shared 4x4 input window of the block's 2x2 outputs
//            (as bicubic_up2_kernel) -> resized values, ELU'd for the branch -> LDS
//   phase 2  thread = (pixel, half of the output channels): the CB x CO conv3 from LDS (weights: wave-uniform broadcast
//            reads), + scale / bias4 / resized skip, 128-bit stores.
// Few registers and 31 KB of LDS per workgroup keep 5 workgroups per CU in flight, which hides the load latency that
// bound the one-thread-per-block form (1.87 ms).
// ------------------------------------------------------------------------------------------------
template <int CB, int CO>
__global__ __launch_bounds__(256)
void up_tail_kernel(const float4* __restrict__ q, const float4* __restrict__ sk, const float* __restrict__ w3, int B, int H,
                    int W, int tiles_x, int tiles_y, float b3a, float b3b, float scale, float b4, float* __restrict__ y) {
    constexpr int LB = CB + 4, LS = CO + 4, GB = CB / 4, GS = CO / 4, NG = GB + GS;
    constexpr int NTL = (CO + 15) / 16;                                             // 16-channel output tiles (CO = 8: half a tile)
    static_assert(CB % 16 == 0 && CO % 4 == 0, "conv3 runs on 16x16x4 MFMA tiles");
    const float w75[4] = {-0.03515625f, 0.26171875f, 0.87890625f, -0.10546875f};   // even outputs (offset .75)
    const float w25[4] = {-0.10546875f, 0.87890625f, 0.26171875f, -0.03515625f};   // odd outputs  (offset .25)
    __shared__ __attribute__((aligned(16))) float Tb[128 * LB];                     // ELU'd resized branch [pixel][ci]
    __shared__ __attribute__((aligned(16))) float Ts[128 * LS];                     // resized skip [pixel][co]
    const int tid = threadIdx.x;
    // conv3 weights as the MFMA row operand, kept in registers for the whole (persistent) launch: lane (li, q) of n-tile nt, k-slice s
    // holds w3[co = 16 nt + li][ci = 16 s + 4 q .. + 3] (packed rows are [co][ci])
    float4 w3r[NTL][CB / 16];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int s_ = 0; s_ < CB / 16; ++s_) {
            const int co = 16 * nt + (tid & 15);
            w3r[nt][s_] = co < CO ? *reinterpret_cast<const float4*>(w3 + co * CB + 16 * s_ + 4 * ((tid & 63) >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    const int OW = 2 * W, OH = 2 * H;
    const int n_tiles = B * tiles_x * tiles_y;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int txi = tile % tiles_x;
        const int tyi = (tile / tiles_x) % tiles_y;
        const int64_t b = tile / (tiles_x * tiles_y);
        const int bi0 = 2 * tyi - 1, bj0 = 16 * txi - 1;                            // first resize block of the tile
        __syncthreads();                                                           // previous tile's phase 2 is done
        // ---- phase 1 ------------------------------------------------------------------------------------------------
        for (int it = tid; it < 32 * NG; it += 256) {
            const int g = it % NG, blk = it / NG;
            const int bi = bi0 + (blk >> 4), bj = bj0 + (blk & 15);
            const bool branch = g < GB;
            const float4* src = branch ? q : sk;
            const int C4 = branch ? GB : GS, gg = branch ? g : g - GB;
            int xs[4], ys[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int v = bj - 1 + j; xs[j] = v < 0 ? 0 : (v > W - 1 ? W - 1 : v);
                int u = bi - 1 + j; ys[j] = u < 0 ? 0 : (u > H - 1 ? H - 1 : u);
            }
            float4 ho[4], he[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4* row = src + ((b * H + ys[r]) * W) * C4 + gg;
                float4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = row[(int64_t)xs[j] * C4];
                // accumulation steps as fmas (round 3): the resize is this kernel's vector-issue bound (PMC: 401 M vector instructions per
                // launch against 9.6 M MFMAs) and a * b + c written as two instructions was a third of them; <= 1 fp32 ulp per tap from
                // the unfused form (the fp32 handle's conv-before-resize order is not bit-comparable with the reference's anyway)
                ho[r].x = v[0].x * w25[0]; ho[r].y = v[0].y * w25[0]; ho[r].z = v[0].z * w25[0]; ho[r].w = v[0].w * w25[0];
                he[r].x = v[0].x * w75[0]; he[r].y = v[0].y * w75[0]; he[r].z = v[0].z * w75[0]; he[r].w = v[0].w * w75[0];
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    ho[r].x = __builtin_fmaf(v[j].x, w25[j], ho[r].x); ho[r].y = __builtin_fmaf(v[j].y, w25[j], ho[r].y);
                    ho[r].z = __builtin_fmaf(v[j].z, w25[j], ho[r].z); ho[r].w = __builtin_fmaf(v[j].w, w25[j], ho[r].w);
                    he[r].x = __builtin_fmaf(v[j].x, w75[j], he[r].x); he[r].y = __builtin_fmaf(v[j].y, w75[j], he[r].y);
                    he[r].z = __builtin_fmaf(v[j].z, w75[j], he[r].z); he[r].w = __builtin_fmaf(v[j].w, w75[j], he[r].w);
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float* wy = a == 0 ? w25 : w75;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float4* in = c == 0 ? ho : he;
                    float4 o;
                    o.x = in[0].x * wy[0]; o.y = in[0].y * wy[0]; o.z = in[0].z * wy[0]; o.w = in[0].w * wy[0];
#pragma unroll
                    for (int r = 1; r < 4; ++r) {
                        o.x = __builtin_fmaf(in[r].x, wy[r], o.x); o.y = __builtin_fmaf(in[r].y, wy[r], o.y);
                        o.z = __builtin_fmaf(in[r].z, wy[r], o.z); o.w = __builtin_fmaf(in[r].w, wy[r], o.w);
                    }
                    const int px = (2 * (blk >> 4) + a) * 32 + 2 * (blk & 15) + c;   // pixel of the 4 x 32 tile
                    if (branch) {
                        o.x = vqae::elu_act(o.x + b3a) + b3b; o.y = vqae::elu_act(o.y + b3a) + b3b;
                        o.z = vqae::elu_act(o.z + b3a) + b3b; o.w = vqae::elu_act(o.w + b3a) + b3b;
                        *reinterpret_cast<float4*>(Tb + px * LB + 4 * gg) = o;
                    } else {
                        *reinterpret_cast<float4*>(Ts + px * LS + 4 * gg) = o;
                    }
                }
            }
        }
        __syncthreads();
        // ---- phase 2: conv3 on the fp32 MFMA (round 3; it was CB x CO / 2 broadcast FMAs per thread) ------------------------------------
        // v_mfma_f32_16x16x4_f32 with the weights as the row operand: lane (li, q) feeds Tb[pixel li][16 s + 4 q ..] and receives
        // D[4 q + r][li] = 4 consecutive output channels of its pixel -> 16-byte skip read and store.  A wave owns 2 groups of 16 pixels.
        {
            const int lane = tid & 63, wv = tid >> 6;
            const int li = lane & 15, q = lane >> 4;
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int px = 32 * wv + 16 * gg + li;
                const int oy = 2 * bi0 + 1 + (px >> 5), ox = 2 * bj0 + 1 + (px & 31);
                float4 tb[CB / 16];
#pragma unroll
                for (int s_ = 0; s_ < CB / 16; ++s_) tb[s_] = *reinterpret_cast<const float4*>(Tb + px * LB + 16 * s_ + 4 * q);
#pragma unroll
                for (int nt = 0; nt < NTL; ++nt) {
                    typedef float f32x4_t __attribute__((ext_vector_type(4)));
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s_ = 0; s_ < CB / 16; ++s_) {
                        const float4 wv4 = w3r[nt][s_];
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv4.x, tb[s_].x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv4.y, tb[s_].y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv4.z, tb[s_].z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv4.w, tb[s_].w, acc, 0, 0, 0);
                    }
                    if (oy >= 0 && oy < OH && ox >= 0 && ox < OW && 16 * nt + 4 * q < CO) {
                        const float4 sp = *reinterpret_cast<const float4*>(Ts + px * LS + 16 * nt + 4 * q);
                        float o[4];
                        const float sv[4] = {sp.x, sp.y, sp.z, sp.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float tv = acc[e] * scale;
                            tv = tv + b4;
                            o[e] = tv + sv[e];
                        }
                        *reinterpret_cast<float4*>(y + ((b * OH + oy) * OW + ox) * CO + 16 * nt + 4 * q) = make_float4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
        }
    }
}

// Batched 2-D transpose: in [B][R][S] -> out [B][S][R]  (NCHW <-> NHWC with R/S = C / H*W).
__global__ __launch_bounds__(256)
void transpose_kernel(const float* __restrict__ in, int R, int S, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int64_t base = (int64_t)blockIdx.z * R * S;
    const int s0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int r = r0 + ty + j, s = s0 + tx;
        if (r < R && s < S) tile[ty + j][tx] = in[base + (int64_t)r * S + s];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 32; j += 8) {
        const int s = s0 + ty + j, r = r0 + tx;
        if (r < R && s < S) out[base + (int64_t)s * R + r] = tile[tx][ty + j];
    }
}

__global__ __launch_bounds__(256)
void label_maxpool_kernel(const uint8_t* __restrict__ lab, int B, int H, int W, int O, uint8_t* __restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * O * O) return;
    const int ox = (int)(i % O), oy = (int)((i / O) % O), b = (int)(i / ((int64_t)O * O));
    // adaptive pooling windows: [floor(o*H/O), ceil((o+1)*H/O))
    const int y0 = (oy * H) / O, y1 = ((oy + 1) * H + O - 1) / O;
    const int x0 = (ox * W) / O, x1 = ((ox + 1) * W + O - 1) / O;
    uint8_t m = 0;
    for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) {
            const uint8_t v = lab[((int64_t)b * H + yy) * W + xx];
            m = v > m ? v : m;
        }
    y[i] = m;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256)
void stitch_kernel(const TI* __restrict__ tiles, const int32_t* __restrict__ rc, int64_t total, int th, int tw,
                   TO* __restrict__ grid, int gh, int gw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % tw), yy = (int)((i / tw) % th);
    const int64_t t = i / ((int64_t)tw * th);
    const int r = rc[2 * t], c = rc[2 * t + 1];
    const int64_t gy = (int64_t)r * th + yy, gx = (int64_t)c * tw + x;
    if (gy < gh && gx < gw) grid[gy * gw + gx] = (TO)tiles[i];
}

// ------------------------------------------------------------------------------------------------
// Register-blocked fp32 out_stem (round 3).  The one-thread-per-pixel kernel above issues one LDS weight read per 3 FMAs and 36
// global loads per pixel at C -> 3: 0.65 ms at cfg B where the bytes need 0.2 ms.
//   ostem_rb_kernel<CIN>:  CIN / 4 lanes share a pixel, each with 4 input channels and ITS 9 x 4 x 3 weights in registers for the
//     whole (grid-stride) launch; 9 16-byte loads and 108 FMAs per pixel and lane, a butterfly over the lanes of the pixel, one
//     lane adds the bias and stores the 3 outputs.  (Partial sums per channel quad: a summation order of its own, fp32 rounding.)
//     0.65 -> 0.57 ms.  The same idea for in_stem (a thread = 4 pixels x 4 output channels, weights from LDS once per 16 FMAs)
//     measured SLOWER than the kernel above (0.61 vs 0.54 ms: twice the global load instructions per pixel) and is not kept.
// Same zero padding and bias as conv3x3_direct_kernel; fp32 only (the 16-bit modes have stem16.hip).
// ------------------------------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256)
void ostem_rb_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, int B, int H, int W,
                     float* __restrict__ y, int y_nchw) {
    constexpr int QN = CIN / 4;                                                     // lanes per pixel (a power of two <= 16)
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const int q = threadIdx.x % QN;
    f32x4_t wr[9][3];                                                               // [tap][co] over this lane's 4 input channels
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int co = 0; co < 3; ++co)
#pragma unroll
            for (int e = 0; e < 4; ++e) wr[tap][co][e] = w[((int64_t)co * CIN + 4 * q + e) * 9 + tap];
    const float b0 = bias[0], b1 = bias[1], b2 = bias[2];
    const int hw = H * W;
    const int64_t npix = (int64_t)B * hw;
    const int64_t stride = (int64_t)gridDim.x * (256 / QN);
    // every lane of the wave runs the same number of trips (the butterfly needs all of a pixel's lanes): clamp, store masked
    const int64_t trips = (npix + stride - 1) / stride;
    int64_t pix = (int64_t)blockIdx.x * (256 / QN) + threadIdx.x / QN;
    for (int64_t t = 0; t < trips; ++t, pix += stride) {
        const bool live = pix < npix;
        const int64_t pc = live ? pix : npix - 1;
        const int b = (int)(pc / hw), rem = (int)(pc - (int64_t)b * hw);
        const int oy = rem / W, ox = rem - oy * W;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int64_t src = ((int64_t)b * hw + (ok ? (int64_t)iy * W + ix : rem)) * CIN + 4 * q;
            f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + src);
            if (!ok) v = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0 = __builtin_fmaf(v[e], wr[tap][0][e], a0);
                a1 = __builtin_fmaf(v[e], wr[tap][1][e], a1);
                a2 = __builtin_fmaf(v[e], wr[tap][2][e], a2);
            }
        }
#pragma unroll
        for (int off = 1; off < QN; off <<= 1) {
            a0 += __shfl_xor(a0, off);
            a1 += __shfl_xor(a1, off);
            a2 += __shfl_xor(a2, off);
        }
        if (live && q == 0) {
            a0 += b0; a1 += b1; a2 += b2;
            if (y_nchw) {
                y[((int64_t)b * 3 + 0) * hw + rem] = a0;
                y[((int64_t)b * 3 + 1) * hw + rem] = a1;
                y[((int64_t)b * 3 + 2) * hw + rem] = a2;
            } else {
                y[pix * 3 + 0] = a0; y[pix * 3 + 1] = a1; y[pix * 3 + 2] = a2;
            }
        }
    }
}

template <int CIN, int COUT>
int launch_direct(const void* x, int x_kind, const Norm3& nrm, const float* w, const float* bias, int B, int H, int W,
                  float* y, int y_nchw, int dt, hipStream_t stream) {
    const int64_t npix = (int64_t)B * H * W;
    conv3x3_direct_kernel<CIN, COUT><<<(unsigned)vqae::ceil_div(npix, 256), 256, 0, stream>>>(
        x, x_kind, nrm, w, bias, B, H, W, y, y_nchw, dt);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

template <typename TI>
int stitch_out(const TI* tiles, const int32_t* rc, int64_t total, int th, int tw, void* grid, int gdt, int gh, int gw,
               hipStream_t stream) {
    const unsigned g = (unsigned)vqae::ceil_div(total, 256);
    switch (gdt) {
        case VQAE_IDX_I64: stitch_kernel<TI, int64_t><<<g, 256, 0, stream>>>(tiles, rc, total, th, tw, (int64_t*)grid, gh, gw); break;
        case VQAE_IDX_U8: stitch_kernel<TI, uint8_t><<<g, 256, 0, stream>>>(tiles, rc, total, th, tw, (uint8_t*)grid, gh, gw); break;
        case VQAE_IDX_U16: stitch_kernel<TI, uint16_t><<<g, 256, 0, stream>>>(tiles, rc, total, th, tw, (uint16_t*)grid, gh, gw); break;
        case VQAE_IDX_I32: stitch_kernel<TI, int32_t><<<g, 256, 0, stream>>>(tiles, rc, total, th, tw, (int32_t*)grid, gh, gw); break;
        default: return vqae::fail(VQAE_ERR_INVALID, "stitch: bad grid dtype %d", gdt);
    }
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}


// ------------------------------------------------------------------------------------------------
// 1x1 and 2x2/stride-2 convs with 8 input channels (the stem-width level of the reference default model: conf/model/
// vq_ae.yaml:24 out_channels 8): K = 8 or 32.  The MFMA engine spends a whole 128-pixel workgroup (gather geometry,
// LDS staging, a barrier) on one 8-deep K step and runs 7x under the HBM rate there; here a lane owns one output
// pixel x 4 output channels (lanes of a pixel share its input through L1, stores are contiguous), weights come
// through scalar loads.  Same pre-op / epilogue contract and order as vqae_conv2d_f32 (conv_mfma.hip).
// ------------------------------------------------------------------------------------------------
struct SmallK {
    const float* __restrict__ x;
    const float* __restrict__ w;          // packed [cout_pad][taps * 8]
    const float* __restrict__ bias_vec;
    const float* residual;
    float* y;
    int H, W, Ho, Wo, cout;
    int64_t M;
    int pre_mode, has_scale, has_bias_s, has_act, dt;
    float pre_a, pre_b, scale, bias_s, act_a, act_b;
};

template <int KS>
__global__ __launch_bounds__(256)
void conv_small_k_kernel(const SmallK p) {
    constexpr int K = KS * KS * 8;
    __shared__ __attribute__((aligned(16))) float Wl[K * 64];          // weights [k][cout]: per-lane global weight loads made
    const int cg4 = p.cout >> 2;                                       // this kernel run at 2.5x its HBM time
    for (int j = threadIdx.x; j < K * p.cout; j += 256) Wl[j] = p.w[(int64_t)(j % p.cout) * K + j / p.cout];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.M * cg4) return;
    const int cg = (int)(i % cg4);
    const int64_t m = i / cg4;
    const int ox = (int)(m % p.Wo);
    const int64_t t = m / p.Wo;
    const int oy = (int)(t % p.Ho);
    const int64_t b = t / p.Ho;
    float v[K];
#pragma unroll
    for (int tap = 0; tap < KS * KS; ++tap) {
        const float4* src = reinterpret_cast<const float4*>(
            p.x + (((b * p.H + (int64_t)oy * KS + tap / KS) * p.W) + (int64_t)ox * KS + tap % KS) * 8);
        const float4 a = src[0], c = src[1];
        v[tap * 8 + 0] = a.x; v[tap * 8 + 1] = a.y; v[tap * 8 + 2] = a.z; v[tap * 8 + 3] = a.w;
        v[tap * 8 + 4] = c.x; v[tap * 8 + 5] = c.y; v[tap * 8 + 6] = c.z; v[tap * 8 + 7] = c.w;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (p.pre_mode != VQAE_PRE_NONE) {
            v[k] = v[k] + p.pre_a;
            if (p.pre_mode == VQAE_PRE_BIAS_ELU_BIAS) v[k] = vqae::elu_act(v[k]) + p.pre_b;
        }
        v[k] = vqae::round_dt(v[k], p.dt);
    }
    float out[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float4 wv = *reinterpret_cast<const float4*>(Wl + k * p.cout + 4 * cg);
        out[0] = __builtin_fmaf(v[k], wv.x, out[0]); out[1] = __builtin_fmaf(v[k], wv.y, out[1]);
        out[2] = __builtin_fmaf(v[k], wv.z, out[2]); out[3] = __builtin_fmaf(v[k], wv.w, out[3]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float acc = out[e];
        if (p.bias_vec) acc = acc + p.bias_vec[4 * cg + e];
        acc = vqae::round_dt(acc, p.dt);
        if (p.has_scale) { acc = acc * p.scale; acc = acc + p.bias_s; }
        else if (p.has_bias_s) { acc = acc + p.bias_s; }
        out[e] = acc;
    }
    const int64_t o = m * p.cout + 4 * cg;
    if (p.residual) {
        const float4 r = *reinterpret_cast<const float4*>(p.residual + o);
        out[0] += r.x; out[1] += r.y; out[2] += r.z; out[3] += r.w;
    }
    if (p.has_act == VQAE_ACT_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = out[e] / (1.0f + expf(-out[e]));
    } else if (p.has_act) {
#pragma unroll
        for (int e = 0; e < 4; ++e) out[e] = vqae::elu_act(out[e] + p.act_a) + p.act_b;
    }
    *reinterpret_cast<float4*>(p.y + o) = make_float4(out[0], out[1], out[2], out[3]);
}

}  // namespace

namespace vqae {
// Fused tail of an 'up' block (up_tail_kernel): (branch channels, out channels) in {(64, 32), (32, 16), (16, 8)}.
bool up_tail_supported(int cb, int co) { return (cb == 64 && co == 32) || (cb == 32 && co == 16) || (cb == 16 && co == 8); }

int up_tail(const float* q, const float* s, const float* w3_packed, int B, int H, int W, int cb, int co, float b3a, float b3b,
            float scale, float b4, float* y, hipStream_t stream) {
    VQAE_REQUIRE(q && s && w3_packed && y, VQAE_ERR_INVALID, "up_tail: null pointer");
    VQAE_REQUIRE(up_tail_supported(cb, co), VQAE_ERR_UNSUPPORTED, "up_tail: channels %d -> %d", cb, co);
    if ((int64_t)B * H * W == 0) return VQAE_OK;
    const int tiles_x = (int)ceil_div(W + 1, 16), tiles_y = (int)ceil_div(H + 1, 2);      // 2 x 16 resize blocks per tile
    const int64_t n_tiles = (int64_t)B * tiles_x * tiles_y;
    VQAE_REQUIRE(n_tiles < (1ll << 31), VQAE_ERR_UNSUPPORTED, "up_tail: too many tiles");
    const unsigned grid = (unsigned)std::min<int64_t>(n_tiles, 256 * 40);
    if (cb == 64) up_tail_kernel<64, 32><<<grid, 256, 0, stream>>>((const float4*)q, (const float4*)s, w3_packed, B, H, W, tiles_x, tiles_y, b3a, b3b, scale, b4, y);
    else if (cb == 32) up_tail_kernel<32, 16><<<grid, 256, 0, stream>>>((const float4*)q, (const float4*)s, w3_packed, B, H, W, tiles_x, tiles_y, b3a, b3b, scale, b4, y);
    else up_tail_kernel<16, 8><<<grid, 256, 0, stream>>>((const float4*)q, (const float4*)s, w3_packed, B, H, W, tiles_x, tiles_y, b3a, b3b, scale, b4, y);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// 1x1 / stride 1 or 2x2 / stride 2, no padding, cin == 8, cout % 4 == 0: see conv_small_k_kernel.
bool conv_small_k_supported(const vqae_conv_args* a) {
    return a->cin == 8 && a->pad == 0 && a->ksize == a->stride && (a->ksize == 1 || a->ksize == 2) && a->cout % 4 == 0 &&
           a->cout <= 64 && a->pre_mode <= VQAE_PRE_BIAS_ELU_BIAS && a->in_h % a->ksize == 0 && a->in_w % a->ksize == 0;
}

int conv_small_k(const vqae_conv_args* a, const float* x, const float* w, const float* bias_vec, const float* residual,
                 float* y, hipStream_t stream) {
    SmallK p;
    p.x = x; p.w = w; p.bias_vec = bias_vec; p.residual = residual; p.y = y;
    p.H = a->in_h; p.W = a->in_w; p.Ho = a->in_h / a->ksize; p.Wo = a->in_w / a->ksize; p.cout = a->cout;
    p.M = (int64_t)a->batch * p.Ho * p.Wo;
    p.pre_mode = a->pre_mode; p.has_scale = a->has_scale; p.has_bias_s = a->has_bias_s; p.has_act = a->has_act; p.dt = a->dtype;
    p.pre_a = a->pre_a; p.pre_b = a->pre_b; p.scale = a->scale; p.bias_s = a->bias_s; p.act_a = a->act_a; p.act_b = a->act_b;
    const int64_t threads = p.M * (a->cout / 4);
    if (threads == 0) return VQAE_OK;
    VQAE_REQUIRE(ceil_div(threads, 256) < (1ll << 31), VQAE_ERR_UNSUPPORTED, "conv_small_k: too many pixels");
    const unsigned grid = (unsigned)ceil_div(threads, 256);
    if (a->ksize == 1) conv_small_k_kernel<1><<<grid, 256, 0, stream>>>(p);
    else conv_small_k_kernel<2><<<grid, 256, 0, stream>>>(p);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

// internal entry shared with handle.hip: x_kind 0 NHWC f32 / 1 NCHW f32 / 2 u8 NHWC; y_nchw 0/1
int conv3x3_direct(const void* x, int x_kind, const float* mean255, const float* inv_std255, const float* w,
                   const float* bias, int B, int H, int W, int cin, int cout, float* y, int y_nchw, int dt,
                   hipStream_t stream) {
    Norm3 nrm;
    for (int i = 0; i < 4; ++i) {
        nrm.mean[i] = (mean255 && i < 3) ? mean255[i] : 0.f;
        nrm.inv[i] = (inv_std255 && i < 3) ? inv_std255[i] : 1.f;
    }
    VQAE_REQUIRE(x && w && bias && y, VQAE_ERR_INVALID, "conv3x3_direct: null pointer");
    VQAE_REQUIRE(x_kind != 2 || cin == 3, VQAE_ERR_UNSUPPORTED, "conv3x3_direct: uint8 input needs cin == 3");
    if ((int64_t)B * H * W == 0) return VQAE_OK;
    static const bool no_rb = getenv("VQAE_NO_STEM_RB") && atoi(getenv("VQAE_NO_STEM_RB"));
    if (!no_rb && dt == VQAE_DT_F32) {                                      // register-blocked fp32 stems
        if (cout == 3 && x_kind == 0 && (cin == 8 || cin == 16 || cin == 32)) {
            const int64_t npix = (int64_t)B * H * W;
            const unsigned grid = (unsigned)std::min<int64_t>(vqae::ceil_div(npix, 256 / (cin / 4)), 256 * 12);
            if (cin == 8) ostem_rb_kernel<8><<<grid, 256, 0, stream>>>((const float*)x, w, bias, B, H, W, y, y_nchw);
            else if (cin == 16) ostem_rb_kernel<16><<<grid, 256, 0, stream>>>((const float*)x, w, bias, B, H, W, y, y_nchw);
            else ostem_rb_kernel<32><<<grid, 256, 0, stream>>>((const float*)x, w, bias, B, H, W, y, y_nchw);
            VQAE_LAUNCH_CHECK();
            return VQAE_OK;
        }
    }
#define VQAE_DIRECT_CASE(CI, CO) \
    if (cin == CI && cout == CO) return launch_direct<CI, CO>(x, x_kind, nrm, w, bias, B, H, W, y, y_nchw, dt, stream);
    VQAE_DIRECT_CASE(3, 4) VQAE_DIRECT_CASE(3, 8) VQAE_DIRECT_CASE(3, 16) VQAE_DIRECT_CASE(3, 32) VQAE_DIRECT_CASE(3, 64)
    VQAE_DIRECT_CASE(4, 3) VQAE_DIRECT_CASE(8, 3) VQAE_DIRECT_CASE(16, 3) VQAE_DIRECT_CASE(32, 3) VQAE_DIRECT_CASE(64, 3)
#undef VQAE_DIRECT_CASE
    return fail(VQAE_ERR_UNSUPPORTED, "conv3x3_direct: unsupported channel pair %d -> %d", cin, cout);
}
}  // namespace vqae

namespace vqae {
bool stem16_supported(int c0, int h, int w, int dtype);
size_t stem16_weight_bytes(int cin);
int stem16_pack_weight(const float* w_dev, int n_out, int cin, int dtype, void* out_dev, hipStream_t stream);
int istem16(const void* x, int x_kind, const float* mean255, const float* inv_std255, const void* wf, const float* bias, int B,
            int H, int W, int c0, float* y, int dtype, hipStream_t stream);
int ostem16(const float* x, const void* wf, const float* bias, int B, int H, int W, int c, float* y, int y_nchw, int dtype,
            hipStream_t stream);
}  // namespace vqae

extern "C" int vqae_conv3x3_direct_f32(const float* x, const uint8_t* x_u8, const float* mean255, const float* inv_std255,
                                       const float* w, const float* bias, int B, int H, int W, int cin, int cout,
                                       float* y, int dtype, void* stream) {
    VQAE_REQUIRE(dtype >= VQAE_DT_F32 && dtype <= VQAE_DT_F16, VQAE_ERR_INVALID, "conv3x3_direct: dtype %d", dtype);
    // 16-bit modes, stem shapes the MFMA kernels cover (stem16.hip): the same dispatch the model handle uses
    static const bool no16 = getenv("VQAE_NO_STEM16") && atoi(getenv("VQAE_NO_STEM16"));
    const int c0 = cin == 3 ? cout : (cout == 3 ? cin : 0);
    if (!no16 && dtype != VQAE_DT_F32 && c0 && (cin == 3) != (cout == 3) && vqae::stem16_supported(c0, H, W, dtype) && (cin == 3 || x)) {
        VQAE_REQUIRE((x || x_u8) && w && bias && y, VQAE_ERR_INVALID, "conv3x3_direct: null pointer");
        // fragment scratch: stream-ordered allocation per call (any device, any number of streams per thread; the free is
        // ordered after the kernel that reads it)
        void* wbuf = nullptr;
        VQAE_HIP_CHECK(hipMallocAsync(&wbuf, vqae::stem16_weight_bytes(32), (hipStream_t)stream));
        int rc = vqae::stem16_pack_weight(w, cout, cin, dtype, wbuf, (hipStream_t)stream);
        if (rc == VQAE_OK)
            rc = cin == 3 ? vqae::istem16(x_u8 ? (const void*)x_u8 : (const void*)x, x_u8 ? 2 : 0, mean255, inv_std255, wbuf, bias, B, H, W,
                                          cout, y, dtype, (hipStream_t)stream)
                          : vqae::ostem16(x, wbuf, bias, B, H, W, cin, y, 0, dtype, (hipStream_t)stream);
        const hipError_t fe = hipFreeAsync(wbuf, (hipStream_t)stream);
        if (rc == VQAE_OK && fe != hipSuccess) return vqae::fail(VQAE_ERR_HIP, "conv3x3_direct: hipFreeAsync: %s", hipGetErrorString(fe));
        return rc;
    }
    if (x_u8)
        return vqae::conv3x3_direct(x_u8, 2, mean255, inv_std255, w, bias, B, H, W, cin, cout, y, 0, dtype, (hipStream_t)stream);
    return vqae::conv3x3_direct(x, 0, nullptr, nullptr, w, bias, B, H, W, cin, cout, y, 0, dtype, (hipStream_t)stream);
}

namespace {
__global__ void round_inplace_kernel(float* __restrict__ x, int64_t n, int dt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = vqae::round_dt(x[i], dt);
}
}  // namespace

extern "C" int vqae_round_inplace_f32(float* x, int64_t n, int dtype, void* stream) {
    VQAE_REQUIRE(dtype >= VQAE_DT_F32 && dtype <= VQAE_DT_F16, VQAE_ERR_INVALID, "round_inplace: dtype %d", dtype);
    if (n == 0 || dtype == VQAE_DT_F32) return VQAE_OK;
    VQAE_REQUIRE(x, VQAE_ERR_INVALID, "round_inplace: null pointer");
    round_inplace_kernel<<<(unsigned)vqae::ceil_div(n, 256), 256, 0, (hipStream_t)stream>>>(x, n, dtype);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

extern "C" int vqae_bicubic_up2_f32(const float* x, int B, int H, int W, int C, float pre_bias, float* y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VQAE_REQUIRE(x && y, VQAE_ERR_INVALID, "bicubic_up2: null pointer");
    VQAE_REQUIRE(C % 4 == 0, VQAE_ERR_UNSUPPORTED, "bicubic_up2: channels %d must be a multiple of 4", C);
    if ((int64_t)B * H * W == 0) return VQAE_OK;
    const int64_t total = (int64_t)B * (H + 1) * (W + 1) * (C / 4);        // 2x2 output blocks x channel groups
    const unsigned grid = (unsigned)std::min<int64_t>(vqae::ceil_div(total, 256), 256 * 64);
    bicubic_up2_kernel<<<grid, 256, 0, stream>>>((const float4*)x, B, H, W, C / 4, pre_bias, (float4*)y);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

static int transpose_batched(const float* x, int B, int R, int S, float* y, hipStream_t stream) {
    if ((int64_t)B * R * S == 0) return VQAE_OK;
    VQAE_REQUIRE(B <= 65535, VQAE_ERR_UNSUPPORTED, "transpose: batch %d > 65535", B);
    dim3 grid((unsigned)vqae::ceil_div(S, 32), (unsigned)vqae::ceil_div(R, 32), (unsigned)B);
    transpose_kernel<<<grid, 256, 0, stream>>>(x, R, S, y);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

extern "C" int vqae_nchw_to_nhwc_f32(const float* x, int B, int C, int H, int W, float* y, void* stream) {
    VQAE_REQUIRE(x && y, VQAE_ERR_INVALID, "nchw_to_nhwc: null pointer");
    return transpose_batched(x, B, C, H * W, y, (hipStream_t)stream);
}

extern "C" int vqae_nhwc_to_nchw_f32(const float* x, int B, int C, int H, int W, float* y, void* stream) {
    VQAE_REQUIRE(x && y, VQAE_ERR_INVALID, "nhwc_to_nchw: null pointer");
    return transpose_batched(x, B, H * W, C, y, (hipStream_t)stream);
}

extern "C" int vqae_label_maxpool_u8(const uint8_t* lab, int B, int H, int W, int O, uint8_t* y, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VQAE_REQUIRE(lab && y, VQAE_ERR_INVALID, "label_maxpool: null pointer");
    VQAE_REQUIRE(O >= 1 && O <= H && O <= W, VQAE_ERR_INVALID, "label_maxpool: output %d larger than input", O);
    const int64_t total = (int64_t)B * O * O;
    if (total == 0) return VQAE_OK;
    label_maxpool_kernel<<<(unsigned)vqae::ceil_div(total, 256), 256, 0, stream>>>(lab, B, H, W, O, y);
    VQAE_LAUNCH_CHECK();
    return VQAE_OK;
}

extern "C" int vqae_stitch_tiles(const void* tiles, int idx_dtype, const int32_t* rc, int n_tiles, int th, int tw,
                                 void* grid, int grid_dtype, int gh, int gw, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    VQAE_REQUIRE(tiles && rc && grid, VQAE_ERR_INVALID, "stitch: null pointer");
    const int64_t total = (int64_t)n_tiles * th * tw;
    if (total == 0) return VQAE_OK;
    switch (idx_dtype) {
        case VQAE_IDX_I64: return stitch_out((const int64_t*)tiles, rc, total, th, tw, grid, grid_dtype, gh, gw, stream);
        case VQAE_IDX_U8: return stitch_out((const uint8_t*)tiles, rc, total, th, tw, grid, grid_dtype, gh, gw, stream);
        case VQAE_IDX_U16: return stitch_out((const uint16_t*)tiles, rc, total, th, tw, grid, grid_dtype, gh, gw, stream);
        case VQAE_IDX_I32: return stitch_out((const int32_t*)tiles, rc, total, th, tw, grid, grid_dtype, gh, gw, stream);
        default: return vqae::fail(VQAE_ERR_INVALID, "stitch: bad tile dtype %d", idx_dtype);
    }
}
