/* vqae_hip.h -- C ABI of libvqae_hip.so: the MI355X (gfx950) native VQ-AE inference hot path.
 *
 * Drop-in boundary for sara-nl/2D-VQ-AE-2's conv-encoder -> vector-quantise -> conv-decoder
 * forward pass.  The reference has no native interface: its plugin mechanism is
 * Hydra `_target_` class-path substitution (conf/model/layers/vq/ema_vq.yaml:1,
 * conf/model/layers/conv_block/pre_activation_fixup.yaml:24, conf/model/{encoder,decoder}/default.yaml) over
 * nn.Module.forward contracts.  Each entry point below cites the reference interface it replaces;
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C types only; every `*_dev` pointer is a device (HBM) pointer owned by the caller;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all calls are
 *     asynchronous on it and allocate nothing (handle workspaces grow only inside
 *     vqae_reserve / on the first call with a larger batch, never during steady state);
 *   - activations are fp32, NHWC ("channels-last": [B][H][W][C], C contiguous).  NCHW tensors
 *     (the reference's layout, model.py:189) cross the boundary through vqae_nchw_to_nhwc /
 *     vqae_nhwc_to_nchw or the `layout` argument of the handle-level calls;
 *   - return value: 0 on success, negative vqae_status otherwise; vqae_last_error() returns a
 *     thread-local message.  The Python binding maps them back to the reference's exception
 *     types (AssertionError / NotImplementedError / ValueError), see vqae_status.
 */
#ifndef VQAE_HIP_H
#define VQAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum vqae_status {
    VQAE_OK = 0,
    VQAE_ERR_INVALID = -1,        /* bad argument / failed assert  (reference: AssertionError, vq.py:98, conv_block.py:148) */
    VQAE_ERR_UNSUPPORTED = -2,    /* reference: NotImplementedError (vq.py:100-104) */
    VQAE_ERR_HIP = -3,            /* HIP runtime failure */
    VQAE_ERR_NOMEM = -4,
    VQAE_ERR_NOT_FOUND = -5       /* missing tensor name in a weight set (reference: KeyError in load_state_dict) */
} vqae_status;

enum { VQAE_LAYOUT_NHWC = 0, VQAE_LAYOUT_NCHW = 1 };
/* Compute precision of the convolutions, with torch.autocast semantics (the reference's extraction runs
 * `with torch.autocast('cuda')`, scripts/extract_embeddings/extract_embeddings.py:124-125): conv inputs,
 * weights and conv biases are rounded (RNE) to the 16-bit type, products are accumulated in fp32, the conv
 * output is rounded to the 16-bit type; everything else (Fixup scalar biases/scale, ELU, residual adds,
 * bicubic, the p=4 distance, losses) stays fp32, exactly as type promotion leaves it in the reference. */
enum { VQAE_DT_F32 = 0, VQAE_DT_BF16 = 1, VQAE_DT_F16 = 2 };
enum { VQAE_IDX_I64 = 0, VQAE_IDX_U8 = 1, VQAE_IDX_U16 = 2, VQAE_IDX_I32 = 3 };

const char* vqae_last_error(void);
/* "gfx950;<git describe or build date>" */
const char* vqae_build_info(void);

/* ---------------------------------------------------------------------------------------------
 * 1. Vector quantiser  -- replaces EMAVectorQuantizer.forward, vq_ae/layers/vq.py:96-154
 *    (eval mode) and embed_code, vq.py:44-45.
 * ------------------------------------------------------------------------------------------- */

/* Bytes of scratch vqae_vq_forward_f32 needs for N rows (tie re-check list, loss partials). */
size_t vqae_vq_workspace_bytes(int64_t n_rows, int n_codes, int dim);

/* idx[n] = argmin_k ( sum_c |z[n][c] - embed[k][c]|^4 )^(1/4), lowest k on ties
 *          (vq.py:121-129: torch.cdist(flat, embed, p = inputs.dim() = 4) + argmin(dim=1));
 * q[n]   = z[n] + (embed[idx[n]] - z[n])        (vq.py:130,146: lookup + straight-through value);
 * *loss  = commitment_cost * mean((z - embed[idx])^2)   (vq.py:143).
 *   z_dev     [n_rows][dim] fp32 (the NHWC activation, i.e. the reference's `flat_input`, vq.py:116)
 *   embed_dev [n_codes][dim] fp32 (buffer `embed`, vq.py:27)
 *   idx_dev   [n_rows] of idx_dtype (VQAE_IDX_*); required
 *   q_dev     [n_rows][dim] fp32 or NULL;  loss_dev  one fp32 or NULL
 *   margin_dev [n_rows] fp32 or NULL: relative gap between best and second-best 4th-power sums
 *   workspace_dev: vqae_vq_workspace_bytes(...) bytes.
 *   dim: any 1 .. 4096 like the reference; a dim that is not a multiple of 4 runs on zero-padded copies of z and of the
 *   codebook (stream-ordered scratch): the same sums, indices and q bit for bit.
 * Errors: dim < 1 or > 4096, n_codes < 1 or > 65536 -> VQAE_ERR_UNSUPPORTED. */
int vqae_vq_forward_f32(const float* z_dev, const float* embed_dev, int64_t n_rows, int n_codes, int dim,
                        float commitment_cost, void* idx_dev, int idx_dtype, float* q_dev, float* loss_dev,
                        float* margin_dev, void* workspace_dev, void* stream);

/* The same with the Minkowski exponent spelled out: the reference passes `p = inputs.dim()` to torch.cdist (vq.py:97,121-129), i.e.
 * p = 3 for [B, D, L] inputs, 4 for [B, D, h, w] (vqae_vq_forward_f32), 5 for [B, D, d, h, w]:
 *   idx[n] = argmin_k ( sum_c |z[n][c] - embed[k][c]|^p )^(1/p), lowest k on ties.
 * z_dev is the channel-last flattening of the input (vq.py:107-116) whatever its rank.
 * Errors: p outside 3 .. 5 -> VQAE_ERR_UNSUPPORTED; otherwise as vqae_vq_forward_f32. */
int vqae_vq_forward_p_f32(const float* z_dev, const float* embed_dev, int64_t n_rows, int n_codes, int dim, int p,
                          float commitment_cost, void* idx_dev, int idx_dtype, float* q_dev, float* loss_dev,
                          float* margin_dev, void* workspace_dev, void* stream);

/* ProjectedEMAVectorQuantizer2d.forward (vq.py:190-192), projection_dim = 8 (the reference default,
 * conf/model/layers/vq/projected_ema_vq_2d.yaml): proj_out(VQ(proj_in(x))) in one pass over the activation.
 *   x_dev      [n_rows][channels] fp32 (NHWC activation)
 *   wt_in_dev  [channels][8]   proj_in.weight  ([8][channels][1][1], vq.py:178-182) TRANSPOSED
 *   b_in_dev   [8]             proj_in.bias
 *   embed_dev  [n_codes][8]    buffer `embed` (vq.py:27)
 *   w_out_dev  [channels][8]   proj_out.weight ([channels][8][1][1], vq.py:183-187) as stored
 *   b_out_dev  [channels]      proj_out.bias
 *   dtype      VQAE_DT_*: autocast rounding of the two convolutions (weights / biases pre-rounded by the caller)
 *   idx_dev    [n_rows] idx_dtype; out_dev [n_rows][channels] = proj_out(z + (embed[idx] - z)); z_dev [n_rows][8] or NULL
 *   loss_dev   commitment_cost * mean((z - embed[idx])^2) in the 8-D space, or NULL; margin_dev as vqae_vq_forward_f32
 *   workspace_dev: vqae_vq_projected_workspace_bytes(n_rows) bytes.
 * Errors: projection_dim != 8, channels % 4 != 0 -> VQAE_ERR_UNSUPPORTED. */
size_t vqae_vq_projected_workspace_bytes(int64_t n_rows);
int vqae_vq_projected_f32(const float* x_dev, const float* wt_in_dev, const float* b_in_dev, const float* embed_dev,
                          const float* w_out_dev, const float* b_out_dev, int64_t n_rows, int channels, int projection_dim,
                          int n_codes, float commitment_cost, int dtype, void* idx_dev, int idx_dtype, float* out_dev,
                          float* z_dev, float* loss_dev, float* margin_dev, void* workspace_dev, void* stream);

/* out[n][:] = embed[idx[n]][:]   (embed_code, vq.py:44-45 = F.embedding) */
int vqae_embed_code_f32(const void* idx_dev, int idx_dtype, const float* embed_dev, int64_t n_rows, int n_codes,
                        int dim, float* out_dev, void* stream);

/* Training-mode bookkeeping (vq.py:47-74 `_update_ema`): counts n_k and sums dw_k of the rows
 * assigned to each code.  counts_dev [n_codes] fp32, dw_dev [n_codes][dim] fp32 (both overwritten). */
int vqae_vq_code_stats_f32(const float* z_dev, const void* idx_dev, int idx_dtype, int64_t n_rows, int n_codes,
                           int dim, float* counts_dev, float* dw_dev, void* stream);
/* EMA + Laplace smoothing step of `_update_ema` (vq.py:60-74), after the caller all-reduced
 * counts/dw over ranks (vq.py:57-58): updates cluster_size, embed_avg, embed in place. */
int vqae_vq_ema_update_f32(float* embed_dev, float* embed_avg_dev, float* cluster_size_dev, const float* counts_dev,
                           const float* dw_dev, int n_codes, int dim, float decay, float laplace_alpha,
                           void* workspace_dev /* >= 16 bytes */, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 2. Conv stack primitives -- replace the torch.nn.Conv2d / ELU / Upsample call sites of
 *    PreActFixupResBlock.forward (vq_ae/layers/conv_block.py:196-216) and ResizeConv2D.forward
 *    (vq_ae/layers/conv.py:10-11).
 * ------------------------------------------------------------------------------------------- */

enum { VQAE_PAD_NONE = 0, VQAE_PAD_CIRCULAR = 1, VQAE_PAD_ZEROS = 2 };
enum { VQAE_PRE_NONE = 0, VQAE_PRE_BIAS = 1, VQAE_PRE_BIAS_ELU_BIAS = 2, VQAE_PRE_CHANNEL_GATE = 3 };
enum { VQAE_ACT_NONE = 0, VQAE_ACT_ELU = 1, VQAE_ACT_SILU = 2 };      /* vqae_conv_args.has_act */

/* Floats needed for the packed form of a [cout][cin][k][k] weight (rows padded to 32 couts). */
size_t vqae_conv_packed_floats(int cout, int cin, int ksize);
/* Repack a PyTorch-layout conv weight [cout][cin][k][k] (device) into the kernel's
 * [cout_pad][k*k*cin] layout (tap-major K, zero rows for the pad). */
int vqae_conv_pack_weight_f32(const float* w_oihw_dev, int cout, int cin, int ksize, float* packed_dev,
                              void* stream);

typedef struct vqae_conv_args {
    /* geometry: y[b][oy][ox][:] = sum_{dy,dx,ci} W[:, ci, dy, dx] * pre(x[b][oy*stride+dy-pad][ox*stride+dx-pad][ci]) */
    int batch, in_h, in_w, cin, cout;
    int ksize;            /* 1, 2 or 3 */
    int stride;           /* 1 or 2 */
    int pad;              /* 0 or 1 */
    int pad_mode;         /* VQAE_PAD_* (circular: padding_mode='circular', pre_activation_fixup.yaml:56-58) */
    /* pre-op on the input (Fixup scalar biases, conv_block.py:199-206,211):
     *   VQAE_PRE_BIAS:           x + pre_a
     *   VQAE_PRE_BIAS_ELU_BIAS:  ELU(x + pre_a) + pre_b        (ELU alpha = 1, activation/elu.yaml)
     *   VQAE_PRE_CHANNEL_GATE:   x * gate[b][ci]               (SELayer's `x * y`, layers/misc.py:30; only through
     *                                                           vqae_conv2d_gated_f32: fp32, 1x1, cin % 32 == 0) */
    int pre_mode;
    float pre_a, pre_b;
    /* epilogue, in the reference's rounding order (conv_block.py:208-214):
     *   t = acc; if (bias_vec) t = t + bias_vec[c];  (the conv's own bias)   t = round_dtype(t);
     *   if (has_scale) t = t * scale + bias_s;  else if (has_bias_s) t = t + bias_s;
     *   if (residual) t = t + residual[m][c];
     *   has_act == VQAE_ACT_ELU:  t = ELU(t + act_a) + act_b;   (the NEXT conv's pre-op, fused here)
     *   has_act == VQAE_ACT_SILU: t = t * sigmoid(t)            (MBConv: activation/silu.yaml after the folded BN) */
    int has_scale, has_bias_s, has_act;
    float scale, bias_s, act_a, act_b;
    int dtype;            /* VQAE_DT_*: autocast rounding of operands (after the pre-op) and of acc (+ bias_vec) */
} vqae_conv_args;

/* x_dev [B][H][W][cin], w_packed_dev from vqae_conv_pack_weight_f32, bias_vec_dev [cout] or NULL,
 * residual_dev [B][Ho][Wo][cout] or NULL, y_dev [B][Ho][Wo][cout].  fp32 MFMA implicit GEMM.
 * Requires cin % 8 == 0 (use vqae_conv_small_cin_f32 for the 3-channel stem). */
int vqae_conv2d_f32(const vqae_conv_args* a, const float* x_dev, const float* w_packed_dev,
                    const float* bias_vec_dev, const float* residual_dev, float* y_dev, void* stream);

/* Same conv with the input multiplied by a per-(image, input channel) gate [B][cin] while it is loaded
 * (a->pre_mode == VQAE_PRE_CHANNEL_GATE): conv3 of an MBConv consuming SELayer's output without materialising it
 * (conv_block.py:290-297, layers/misc.py:30). */
int vqae_conv2d_gated_f32(const vqae_conv_args* a, const float* x_dev, const float* gate_dev, const float* w_packed_dev,
                          const float* bias_vec_dev, const float* residual_dev, float* y_dev, void* stream);

/* ---- MBConv pieces (vq_ae/layers/conv_block.py:240-321; BatchNorms folded into weights/bias by the caller) ----
 * Depthwise conv over NHWC x [B][H][W][C] (branch_conv2 with groups = C, conv_block.py:276-281):
 *   VQAE_DW_SAME  3x3 / stride 1 / circular pad     (same2d.yaml + padding_mode circular, mbconv.yaml:60-63)
 *   VQAE_DW_DOWN  2x2 / stride 2                     (down2d.yaml)
 *   VQAE_DW_UP    ConvTranspose2d 2x2 / stride 2     (up2d.yaml)
 * w_taps_dev [k*k][C] (tap-major), bias_dev [C] or NULL, optional SiLU; y_dev [B][Ho][Wo][C].
 * If partial_dev != NULL (vqae_dw_partial_floats() floats) it receives per-(image, 256-pixel strip, channel) sums of
 * y for SELayer's spatial mean, reduced in a fixed order (bit-reproducible run to run). */
enum { VQAE_DW_SAME = 0, VQAE_DW_DOWN = 1, VQAE_DW_UP = 2 };
size_t vqae_dw_partial_floats(int batch, int out_h, int out_w, int channels);
int vqae_dwconv_f32(const float* x_dev, const float* w_taps_dev, const float* bias_dev, int batch, int h, int w,
                    int channels, int mode, int silu, float* y_dev, float* partial_dev, void* stream);
/* SELayer.forward up to the gate (layers/misc.py:23-29): mean over (out_h, out_w) from the partial sums ->
 * Linear(channels, hidden) -> SiLU -> Linear(hidden, channels) -> sigmoid; gate_dev [B][channels].
 * fc*_w are nn.Linear weights [out][in] on the device. */
int vqae_se_gate_f32(const float* partial_dev, int batch, int out_h, int out_w, int channels, const float* fc0_w_dev,
                     const float* fc0_b_dev, int hidden, const float* fc2_w_dev, const float* fc2_b_dev, float* gate_dev,
                     void* stream);
/* x [B][H][W][4*c] with channel order (a, b, c) -> y [B][2H][2W][c], y[2i+a][2j+b] = x[i][j][(a, b, :)]: the pixel
 * placement of ConvTranspose2d(k = 2, s = 2) (skip_conv of an 'up' MBConv) after its channel mixing ran as a 1x1 conv. */
int vqae_pixel_shuffle2_f32(const float* x_dev, int batch, int h, int w, int c, float* y_dev, void* stream);

/* One whole PreActFixupResBlock.forward, mode 'same' (conv_block.py:196-216: 1x1 -> 3x3 circular -> 1x1,
 * in_channels == out_channels == c) in a single launch, for the HBM-bound high-resolution levels.
 * x_dev, y_dev [B][H][W][c] (y != x: neighbouring tiles read halo rows of x); w*_packed_dev from
 * vqae_conv_pack_weight_f32; scalars8 (host) = {bias1a, bias1b, bias2a, bias2b, bias3a, bias3b, bias4, scale}.
 * vqae_fixup_same_supported() tells whether a (c, h, w) has a fused kernel (c in {8, 16, 32}, w % 32 == 0). */
int vqae_fixup_same_supported(int c, int h, int w);
int vqae_fixup_same_block_f32(const float* x_dev, float* y_dev, const float* w1_packed_dev, const float* w2_packed_dev,
                              const float* w3_packed_dev, int batch, int h, int w, int c, const float* scalars8,
                              int dtype /* VQAE_DT_* */, void* stream);
/* Round a device fp32 array in place to bf16/f16-representable values (autocast weight cast). */
int vqae_round_inplace_f32(float* x_dev, int64_t n, int dtype, void* stream);

/* Direct (VALU) 3x3 / stride 1 / zero-pad conv with per-channel bias for tiny channel counts:
 * the stems `in_stem` (3 -> C0, model.py:198) and `out_stem` (C0 -> 3, model.py:291).
 * w_oihw_dev is the PyTorch-layout weight [cout][cin][3][3]; cin, cout <= 64.
 * If x_u8_dev != NULL the input is uint8 NHWC and is normalised on the fly
 * ((u - mean255[c]) * inv_std255[c], albumentations Normalize, camelyon16_transforms.yaml:15-23). */
int vqae_conv3x3_direct_f32(const float* x_dev, const uint8_t* x_u8_dev, const float* mean255, const float* inv_std255,
                            const float* w_oihw_dev, const float* bias_dev, int batch, int h, int w, int cin,
                            int cout, float* y_dev, int dtype /* VQAE_DT_* */, void* stream);

/* y = bicubic_x2(x + pre_bias), A = -0.75, align_corners = False, index-clamped borders
 * (nn.Upsample(mode='bicubic', scale_factor=2), layers/conv.py:8).  x [B][H][W][C] -> y [B][2H][2W][C]. */
int vqae_bicubic_up2_f32(const float* x_dev, int batch, int h, int w, int c, float pre_bias, float* y_dev,
                         void* stream);

/* Layout shuffles at the boundary (reference tensors are NCHW, model.py:189). */
int vqae_nchw_to_nhwc_f32(const float* x_dev, int batch, int c, int h, int w, float* y_dev, void* stream);
int vqae_nhwc_to_nchw_f32(const float* x_dev, int batch, int c, int h, int w, float* y_dev, void* stream);

/* labels [B][H][W] (u8) -> [B][out][out] max over (H/out x W/out) windows
 * (F.adaptive_max_pool2d in run_eval, scripts/extract_embeddings/extract_embeddings.py:127-130). */
int vqae_label_maxpool_u8(const uint8_t* labels_dev, int batch, int h, int w, int out, uint8_t* y_dev, void* stream);

/* Stitch code tiles into a slide grid (get_encodings, extract_embeddings.py:77-84):
 * grid[(r*th + y) * grid_w + c*tw + x] = tiles[t][y][x] for tile t at patch position (r, c) = rc[t].
 * tiles idx_dtype in, grid_dtype out (VQAE_IDX_*; narrowing is the caller's cast_to_lowest_dtype choice). */
int vqae_stitch_tiles(const void* tiles_dev, int idx_dtype, const int32_t* rc_dev, int n_tiles, int th, int tw,
                      void* grid_dev, int grid_dtype, int grid_h, int grid_w, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 3. Whole-model handle -- replaces Encoder.forward (vq_ae/model.py:189-217), Decoder.forward
 *    (:274-291) and VQAE.forward (:41-48) for the single-VQ-level Fixup model that every shipped
 *    config composes.
 * ------------------------------------------------------------------------------------------- */
typedef struct vqae_config {
    int in_channels;      /* 3                 conf/model/vq_ae.yaml:23 */
    int stem;             /* stem out_channels vq_ae.yaml:24 */
    int n_down;           /* vq_ae.yaml:26 */
    int n_pre, n_post;    /* vq_ae.yaml:27-28 */
    int n_enc;            /* n_pre_enc_layers = n_post_enc_layers, vq_ae.yaml:29,39 */
    int num_embeddings;   /* layers/vq/ema_vq.yaml:2 */
    int projection_dim;   /* 0: EMAVectorQuantizer; >0: ProjectedEMAVectorQuantizer2d (vq.py:157-192) */
    float commitment_cost;
    int compute_dtype;    /* VQAE_DT_F32 (default) or autocast bf16 / f16 */
    int block_kind;       /* VQAE_BLOCK_FIXUP (default) | VQAE_BLOCK_MBCONV (conf/model/{encoder,decoder}/efficientnetv2.yaml) */
    int expand_ratio;     /* MBConv: mbconv.yaml:32 (4) */
    int se_divisor;       /* MBConv: layers/misc/se.yaml:5 (4) */
    float bn_eps;         /* MBConv: layers/misc/batchnorm2d.yaml:4 (1e-5) */
} vqae_config;
enum { VQAE_BLOCK_FIXUP = 0, VQAE_BLOCK_MBCONV = 1 };

/* One named fp32 host tensor, named as in the reference's state_dict, e.g.
 * "encoder.pre_enc_layers.0.7.branch_conv2.weight" with PyTorch shapes ([cout][cin][k][k], (1,) ...). */
typedef struct vqae_tensor {
    const char* name;
    const float* data;    /* host pointer */
    int64_t numel;
} vqae_tensor;

typedef struct vqae_handle vqae_handle;

int vqae_create(const vqae_config* cfg, const vqae_tensor* tensors, int n_tensors, vqae_handle** out);
void vqae_destroy(vqae_handle* h);
/* Pre-size the internal workspace for `max_batch` patches of in_h x in_w (optional). */
int vqae_reserve(vqae_handle* h, int max_batch, int in_h, int in_w);
/* Replace the codebook (e.g. after calibration / EMA updates): embed_host [K][D] fp32. */
int vqae_set_codebook(vqae_handle* h, const float* embed_host);

/* Encoder.forward: x [B,3,H,W] (layout per `layout`) -> idx [B][h][w] (idx_dtype), optional
 * q_dev [B][C][h][w] (fp32, `layout`), optional loss_dev (one fp32).  h = H / 2^n_down. */
int vqae_encode(vqae_handle* h, const float* x_dev, int batch, int in_h, int in_w, int layout, void* idx_dev,
                int idx_dtype, float* q_dev, float* loss_dev, void* stream);
/* Same, from raw uint8 NHWC patches normalised on device. */
int vqae_encode_u8(vqae_handle* h, const uint8_t* x_u8_dev, int batch, int in_h, int in_w, void* idx_dev,
                   int idx_dtype, float* q_dev, int q_layout, float* loss_dev, void* stream);
/* Pre-VQ activations z [B][h][w][C] NHWC (for codebook calibration, vq.py:76-94 `_init_ema`);
 * with projection, the projected [B][h][w][D] tensor. */
int vqae_encode_features(vqae_handle* h, const float* x_dev, int batch, int in_h, int in_w, int layout,
                         float* z_dev, void* stream);
/* Decoder.forward: q [B][C][h][w] -> out [B,3,H,W]  (q_h, q_w = latent grid size). */
int vqae_decode(vqae_handle* h, const float* q_dev, int batch, int q_h, int q_w, int layout, float* out_dev,
                void* stream);
/* Decode from code indices: embed_code (+ proj_out) then Decoder.forward. */
int vqae_decode_indices(vqae_handle* h, const void* idx_dev, int idx_dtype, int batch, int q_h, int q_w, int layout,
                        float* out_dev, void* stream);
/* VQAE.forward: out [B,3,H,W], idx (optional), loss (optional). */
int vqae_forward(vqae_handle* h, const float* x_dev, int batch, int in_h, int in_w, int layout, float* out_dev,
                 void* idx_dev, int idx_dtype, float* loss_dev, void* stream);

/* Sub-module calls.  The reference lets a caller run any part of the block stacks on its own
 * (`model.encoder.down_layers[0].layers[l].layers[b](x)`, `model.encoder.pre_enc_layers[0][i:j](x)` --
 * nn.ModuleList / nn.Sequential of PreActFixupResBlock, vq_ae/model.py:160-176,249-264, conv_block.py:196-216).
 * vqae_run_blocks runs residual blocks [first, first + count) of the encoder block list (side 0: the DownBlock
 * levels in order, then pre_enc; model.py:199-208) or of the decoder list (side 1: post_enc, then the UpBlock
 * levels; model.py:278-289) on x_dev [B][in_h][in_w][cin of block `first`] (NHWC fp32) through exactly the kernels
 * the handle-level calls dispatch (including the cross-block fusions when count > 1), and writes
 * y_dev [B][*out_h][*out_w][cout of the last block].  The per-block parity tests are built on it. */
int vqae_block_count(const vqae_handle* h, int side);
int vqae_run_blocks(vqae_handle* h, int side, int first, int count, const float* x_dev, int batch, int in_h, int in_w,
                    float* y_dev, int* out_h, int* out_w, void* stream);

/* Introspection for benchmarks: algorithmic FLOPs (2*MACs) of the conv stacks per patch. */
double vqae_flops_per_patch(const vqae_handle* h, int in_h, int in_w, int encoder, int decoder);

/* ---------------------------------------------------------------------------------------------
 * 4. Measurement hook (bench.py `roofline`): time every launch of one kernel class with HIP events
 *    recorded on the launch stream.  kernel_class: 1 = trunk 3x3 circular conv (MFMA, cin >= 128; incl. its fused conv3/conv1 tail),
 *    2 = trunk 1x1 conv, 3 = VQ tier-1 argmin.  Not thread-safe; off by default.
 * ------------------------------------------------------------------------------------------- */
int vqae_prof_begin(int kernel_class, int max_launches);
/* total_work: algorithmic flops of the timed launches (class 1: 2*M*N*K of the 3x3 conv plus, when the
 * launch also ran the fused conv3 / next-conv1 tail, their 2*M*128*128 each; class 3: 3*N*K*D VALU ops). */
int vqae_prof_end(double* total_ms, int* n_launches, double* total_work);

#ifdef __cplusplus
}
#endif
#endif /* VQAE_HIP_H */
