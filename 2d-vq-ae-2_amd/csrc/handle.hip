// Whole-model handle: the native runtime that walks the Fixup block sequence of
// Encoder.forward / Decoder.forward / VQAE.forward (reference vq_ae/model.py:189-217, 274-291, 41-48)
// and launches the HIP kernels of this library on one stream.  Weights arrive as named host tensors in
// the reference's state-dict naming and are repacked once into the MFMA kernel's
// [cout_pad][tap*cin] layout; activations stay NHWC fp32 in four rotating HBM buffers owned by the
// handle (X = residual stream, P/Q/R = block temporaries).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <unordered_map>
#include <vector>

namespace vqae {
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
int conv_trunk_tail(const vqae_conv_args* a, const float* t1, const float* w2, const float* w3, float t_scale,
                    float t_b4, float* xio, const float* w1n, float n_b1a, float n_b1b, float n_b2a, float n_b2b,
                    float* t1_next, hipStream_t stream);
bool wino_trunk_supported(int c, int h, int w, int dtype);
size_t wino_weight_floats(int c);
int wino_transform_weight(const float* w_oihw_dev, int c, int dtype, float* U_dev, hipStream_t stream);
int wino_frag_weight(const float* w_packed_dev, int c, int sk, float* out_dev, hipStream_t stream);
int conv_tail_kslice(int dtype, int cin);
bool wino43_supported(int c, int h, int w, int dtype);
bool wino43_enabled();
size_t wino43_weight_floats(int c);
int wino43_transform_weight(const float* w_oihw_dev, int c, float* U_dev, hipStream_t stream);
int wino43_trunk_tail(const float* t1, const float* U, const float* w3, float act_a, float act_b, float t_scale, float t_b4,
                      float* xio, const float* w1n, float n_b1a, float n_b1b, float n_b2a, float n_b2b, float* t1_next,
                      int batch, int h, int w, int c, hipStream_t stream);
int wino_trunk_tail(const float* t1, const float* U, const float* w3, float act_a, float act_b, float t_scale, float t_b4,
                    float* xio, const float* w1n, float n_b1a, float n_b1b, float n_b2a, float n_b2b, float* t1_next,
                    int batch, int h, int w, int c, int dtype, hipStream_t stream);
bool trunk16_supported(int c, int h, int w, int dtype);
size_t trunk16_weight_bytes(int c, int taps);
int trunk16_pack_weight(const float* w_packed_dev, int c, int taps, int dtype, void* out_dev, hipStream_t stream);
int trunk16_round_pack(const float* src, void* dst, int64_t n, int dtype, hipStream_t stream);
bool trunk16_head_supported(int c, int64_t m, int dtype);
int trunk16_head(const float* x, const void* w1f, float b1a, float b1b, float b2a, float b2b, void* t1, int64_t m, int c,
                 int dtype, bool out32, hipStream_t stream);
bool stem16_supported(int c0, int h, int w, int dtype);
size_t stem16_weight_bytes(int cin);
int stem16_pack_weight(const float* w_dev, int n_out, int cin, int dtype, void* out_dev, hipStream_t stream);
int istem16(const void* x, int x_kind, const float* mean255, const float* inv_std255, const void* wf, const float* bias, int B,
            int H, int W, int c0, float* y, int dtype, hipStream_t stream);
int ostem16(const float* x, const void* wf, const float* bias, int B, int H, int W, int c, float* y, int y_nchw, int dtype,
            hipStream_t stream);
bool same8_16_supported(int c, int h, int w, int dtype);
int same8_16_block(const float* x, float* y, const float* w1_packed, const void* w2h, const void* w3h, int B, int H, int W,
                   const float* scalars8, int dtype, hipStream_t stream);
bool same16_16_supported(int c, int h, int w, int dtype);
int same16_16_block(const float* x, float* y, const void* w1h, const void* w2h, const void* w3h, int B, int H, int W, int c,
                    const float* scalars8, int dtype, hipStream_t stream);
bool up16_supported(int c, int h, int w, int dtype);
int up16_block(const float* x, const float* t1, const void* w2h, const void* w3h, const void* wskh, int B, int H, int W, int c,
               float b3a, float b3b, float scale, float b4, float b1c, float b1d, int dtype, float* y, hipStream_t stream);
int trunk16_block(const void* t1, const void* w2f, const void* w3f, float act_a, float act_b, float t_scale, float t_b4,
                  float* xio, const void* w1nf, float n_b1a, float n_b1b, float n_b2a, float n_b2b, void* t1_next,
                  int batch, int h, int w, int c, int dtype, hipStream_t stream);
bool fixup_conv1_supported(int c, int64_t m);
int fixup_conv1(const float* x, const float* w1f, float pa, float pb, float aa, float ab, float* y, int64_t m, int c,
                hipStream_t stream);
bool down_block_supported(int cin, int h, int w);
int frag_weight_rect(const float* w_packed_dev, int n_rows, int K, float* out_dev, hipStream_t stream);
bool down16_supported(int cin, int h, int w);
size_t down16_weight_bytes(int n_rows, int K);
int down16_pack_weight(const float* w_packed_dev, int n_rows, int K, int dtype, void* out_dev, hipStream_t stream);
int down16_block(const float* x, const void* w1h, const void* w2h, const void* w3h, const void* wskh, int B, int H, int W,
                 int cin, const float* scalars10, int dtype, float* y, hipStream_t stream);
int down_block(const float* x, const float* w1f, const float* w2f, const float* w3f, const float* wskf, int B, int H, int W,
               int cin, const float* scalars10, int dtype, float* y, hipStream_t stream);
bool up_tail_supported(int cb, int co);
int up_tail(const float* q, const float* s, const float* w3_packed, int B, int H, int W, int cb, int co, float b3a, float b3b,
            float scale, float b4, float* y, hipStream_t stream);
int conv3x3_direct(const void* x, int x_kind, const float* mean255, const float* inv_std255, const float* w,
                   const float* bias, int B, int H, int W, int cin, int cout, float* y, int y_nchw, int dt,
                   hipStream_t stream);
}  // namespace vqae

namespace vqae {
ProfState& prof_state() {
    static ProfState p;
    return p;
}
}  // namespace vqae

extern "C" int vqae_prof_begin(int kernel_class, int max_launches) {
    vqae::ProfState& p = vqae::prof_state();
    VQAE_REQUIRE(kernel_class >= 1 && kernel_class <= 3 && max_launches >= 1, VQAE_ERR_INVALID, "prof_begin: bad args");
    if (p.cap < 2 * max_launches) {
        for (int i = 0; i < p.cap; ++i) (void)hipEventDestroy(p.ev[i]);
        delete[] p.ev;
        p.ev = new hipEvent_t[2 * max_launches];
        p.cap = 0;
        for (int i = 0; i < 2 * max_launches; ++i) {
            VQAE_HIP_CHECK(hipEventCreate(&p.ev[i]));
            p.cap = i + 1;
        }
    }
    p.used = 0;
    p.work = 0.0;
    p.cls = kernel_class;
    return VQAE_OK;
}

extern "C" int vqae_prof_end(double* total_ms, int* n_launches, double* total_work) {
    vqae::ProfState& p = vqae::prof_state();
    p.cls = 0;
    double tot = 0.0;
    for (int i = 0; i + 1 < p.used; i += 2) {
        VQAE_HIP_CHECK(hipEventSynchronize(p.ev[i + 1]));
        float ms = 0.f;
        VQAE_HIP_CHECK(hipEventElapsedTime(&ms, p.ev[i], p.ev[i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (n_launches) *n_launches = p.used / 2;
    if (total_work) *total_work = p.work;
    p.used = 0;
    return VQAE_OK;
}

extern "C" const char* vqae_last_error(void) { return vqae::last_error_buf(); }
extern "C" const char* vqae_build_info(void) { return "gfx950;fp32-mfma;" __DATE__; }

namespace {

enum { MODE_SAME = 0, MODE_DOWN = 1, MODE_UP = 2 };

struct Block {
    int mode, cin, cout, br;
    float b1a, b1b, b2a, b2b, b3a, b3b, b4, scale, b1c, b1d;
    float *w1, *w2, *w3, *wskip;          // packed, device
    float* wU43 = nullptr;                // F(4x4, 3x3)-domain conv2 weights [36][C][C] (fp32, C = 128, conv_wino43.hip)
    float* wU = nullptr;                  // Winograd-domain conv2 weights [16][C][C] (fp32 trunk blocks, C = 64 / 128, conv_wino.hip)
    float *w1f = nullptr, *w3f = nullptr; // conv1 / conv3 weights in MFMA fragment order for the fused tails (both trunk kernels)
    float *w2f = nullptr, *wskf = nullptr;// 'down' blocks: conv2 / skip_conv in fragment order too (down_fused.hip)
    void *w1h = nullptr, *w2h = nullptr, *w3h = nullptr;   // 16-bit modes: conv1 / conv2 / conv3 as 16-bit MFMA fragments (trunk16.hip)
    void *dw1h = nullptr, *dw2h = nullptr, *dw3h = nullptr, *dwskh = nullptr;   // 16-bit modes, 'down' blocks (down16.hip)
    void *uw1h = nullptr, *uw2h = nullptr, *uw3h = nullptr, *uwskh = nullptr;   // 16-bit modes, 'up' blocks (head16 + up16.hip)
    void *s8w2h = nullptr, *s8w3h = nullptr;                                    // 16-bit modes, C = 8 'same' blocks (same8_16.hip)
    // MBConv (conv_block.py:240-321), BatchNorms folded: br = expanded width, w2 = depthwise taps [k*k][br]
    int kind = VQAE_BLOCK_FIXUP, hidden = 0;
    float *bv1 = nullptr, *bv2 = nullptr, *bv3 = nullptr;                        // folded BN shifts
    float *fc0w = nullptr, *fc0b = nullptr, *fc2w = nullptr, *fc2b = nullptr;    // SELayer linears
};

// CAMELYON16 normalisation (conf/transforms/camelyon16_transforms.yaml:15-23), x255
const float kMean255[3] = {0.7279f * 255.0f, 0.5955f * 255.0f, 0.7762f * 255.0f};
const float kInv255[3] = {1.0f / (0.2419f * 255.0f), 1.0f / (0.3083f * 255.0f), 1.0f / (0.1741f * 255.0f)};

}  // namespace

struct vqae_handle {
    vqae_config cfg;
    int C = 0, D = 0, K = 0;
    std::vector<Block> enc, dec;
    float *stem_w = nullptr, *stem_b = nullptr, *ostem_w = nullptr, *ostem_b = nullptr;
    void *stem_wh = nullptr, *ostem_wh = nullptr;   // 16-bit modes: the stems' weights as 16-bit MFMA fragments (stem16.hip)
    bool fuse_stem16 = true;
    float *embed = nullptr;
    float *pin_w = nullptr, *pin_b = nullptr, *pout_w = nullptr, *pout_b = nullptr;
    float *pin_wt = nullptr, *pout_wr = nullptr;   // fused projected VQ (vq_proj.hip): proj_in transposed [C][8], proj_out [C][8], rounded
    bool fuse_vq = true;                   // projection_dim == 8: proj_in + argmin + proj_out in one launch
    std::vector<void*> owned;              // every hipMalloc of the weight set
    // workspace
    float* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t buf_floats = 0;
    void* vq_ws = nullptr;
    size_t vq_ws_bytes = 0;
    float* loss_scratch = nullptr;
    int device = 0;                        // the HIP device the weights / workspaces live on (one handle per process and device)
    bool has_encoder = false, has_decoder = false;
    bool fuse_trunk = true;                // conv2 + conv3 (+ next conv1) in one launch at the 128-channel trunk
    bool t1_ready = false;                 // buf[1] already holds the current block's t1
    bool up_conv_first = true;             // fp32 up blocks: 1x1 convs before the bicubic resize (they commute)
    bool fuse_down = true;                 // 'down' blocks with 16/32/64 input channels: one launch (down_fused.hip)
    bool fuse_down16 = true;               // ... on the 16-bit MFMA in the autocast modes (down16.hip)
    bool fuse_up16 = true;                 // 'up' blocks in the autocast modes: head16 + one launch (up16.hip)
    bool fuse_up_tail = true;              // fp32 up blocks at the stem-side levels: resize + ELU + conv3 + skip in one launch
    bool use_wino = true;                  // fp32 trunk blocks (C = 128 on a 32-wide grid, C = 64 on a 64-wide one): Winograd F(2x2,3x3) conv2
    void* idx_scratch = nullptr;           // indices nobody asked for (vqae_forward with idx == NULL)
    size_t idx_scratch_bytes = 0;
    float* se_ws = nullptr;                // MBConv: SE partial sums, then the gate [B][E]
    size_t se_ws_floats = 0;
    size_t se_gate_off = 0;
};

namespace {

using TensorMap = std::unordered_map<std::string, const vqae_tensor*>;

int find(const TensorMap& tm, const std::string& name, int64_t numel, const float** out) {
    auto it = tm.find(name);
    if (it == tm.end()) return vqae::fail(VQAE_ERR_NOT_FOUND, "missing tensor '%s'", name.c_str());
    if (it->second->numel != numel)
        return vqae::fail(VQAE_ERR_INVALID, "tensor '%s' has %lld elements, expected %lld", name.c_str(),
                          (long long)it->second->numel, (long long)numel);
    *out = it->second->data;
    return VQAE_OK;
}

int dev_alloc(vqae_handle* h, size_t bytes, void** out) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess)
        return vqae::fail(VQAE_ERR_NOMEM, "hipMalloc of %zu bytes failed", bytes);
    h->owned.push_back(p);
    *out = p;
    return VQAE_OK;
}

int upload(vqae_handle* h, const float* host, int64_t numel, float** out) {
    void* p;
    int rc = dev_alloc(h, (size_t)numel * 4, &p);
    if (rc) return rc;
    VQAE_HIP_CHECK(hipMemcpy(p, host, (size_t)numel * 4, hipMemcpyHostToDevice));
    *out = (float*)p;
    return VQAE_OK;
}

int upload_packed(vqae_handle* h, const float* host, int cout, int cin, int ks, float** out) {
    float* raw = nullptr;
    void* tmp = nullptr;
    const int64_t numel = (int64_t)cout * cin * ks * ks;
    if (hipMalloc(&tmp, (size_t)numel * 4) != hipSuccess) return vqae::fail(VQAE_ERR_NOMEM, "hipMalloc failed");
    raw = (float*)tmp;
    hipError_t e = hipMemcpy(raw, host, (size_t)numel * 4, hipMemcpyHostToDevice);
    void* packed = nullptr;
    int rc = (e == hipSuccess) ? dev_alloc(h, vqae_conv_packed_floats(cout, cin, ks) * 4, &packed)
                               : vqae::fail(VQAE_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
    if (rc == VQAE_OK) rc = vqae_conv_pack_weight_f32(raw, cout, cin, ks, (float*)packed, nullptr);
    if (rc == VQAE_OK) rc = vqae_round_inplace_f32((float*)packed, (int64_t)vqae_conv_packed_floats(cout, cin, ks), h->cfg.compute_dtype, nullptr);
    if (rc == VQAE_OK && hipDeviceSynchronize() != hipSuccess) rc = vqae::fail(VQAE_ERR_HIP, "pack sync failed");
    (void)hipFree(tmp);
    *out = (float*)packed;
    return rc;
}

// conv2 weights [c][c][3][3] (host, PyTorch layout) -> Winograd domain on the device
int upload_wino(vqae_handle* h, const float* host, int c, float** out) {
    void* tmp = nullptr;
    const size_t raw = (size_t)c * c * 9 * 4;
    if (hipMalloc(&tmp, raw) != hipSuccess) return vqae::fail(VQAE_ERR_NOMEM, "hipMalloc failed");
    hipError_t e = hipMemcpy(tmp, host, raw, hipMemcpyHostToDevice);
    void* U = nullptr;
    int rc = (e == hipSuccess) ? dev_alloc(h, vqae::wino_weight_floats(c) * 4, &U)
                               : vqae::fail(VQAE_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
    if (rc == VQAE_OK) rc = vqae::wino_transform_weight((const float*)tmp, c, h->cfg.compute_dtype, (float*)U, nullptr);
    if (rc == VQAE_OK && hipDeviceSynchronize() != hipSuccess) rc = vqae::fail(VQAE_ERR_HIP, "winograd weight transform failed");
    (void)hipFree(tmp);
    *out = (float*)U;
    return rc;
}

// conv2 weights [c][c][3][3] (host, PyTorch layout) -> F(4x4, 3x3) domain on the device (conv_wino43.hip)
int upload_wino43(vqae_handle* h, const float* host, int c, float** out) {
    void* tmp = nullptr;
    const size_t raw = (size_t)c * c * 9 * 4;
    if (hipMalloc(&tmp, raw) != hipSuccess) return vqae::fail(VQAE_ERR_NOMEM, "hipMalloc failed");
    hipError_t e = hipMemcpy(tmp, host, raw, hipMemcpyHostToDevice);
    void* U = nullptr;
    int rc = (e == hipSuccess) ? dev_alloc(h, vqae::wino43_weight_floats(c) * 4, &U)
                               : vqae::fail(VQAE_ERR_HIP, "hipMemcpy failed: %s", hipGetErrorString(e));
    if (rc == VQAE_OK) rc = vqae::wino43_transform_weight((const float*)tmp, c, (float*)U, nullptr);
    if (rc == VQAE_OK && hipDeviceSynchronize() != hipSuccess) rc = vqae::fail(VQAE_ERR_HIP, "winograd weight transform failed");
    (void)hipFree(tmp);
    *out = (float*)U;
    return rc;
}

int scalar(const TensorMap& tm, const std::string& name, float* out) {
    const float* p;
    int rc = find(tm, name, 1, &p);
    if (rc) return rc;
    *out = p[0];
    return VQAE_OK;
}

int load_block(vqae_handle* h, const TensorMap& tm, const std::string& pre, int mode, int cin, int cout, Block* b) {
    b->mode = mode; b->cin = cin; b->cout = cout; b->br = cin > cout ? cin : cout;   // conv_block.py:151-155
    b->w1 = b->w2 = b->w3 = b->wskip = nullptr;
    b->b1c = b->b1d = 0.f;
    int rc;
#define S_(field, nm) if ((rc = scalar(tm, pre + "." nm, &b->field))) return rc;
    S_(b1a, "bias1a") S_(b1b, "bias1b") S_(b2a, "bias2a") S_(b2b, "bias2b") S_(b3a, "bias3a") S_(b3b, "bias3b")
    S_(b4, "bias4") S_(scale, "scale")
    const int k2 = mode == MODE_SAME ? 3 : (mode == MODE_DOWN ? 2 : 1);
    const float* p;
    if ((rc = find(tm, pre + ".branch_conv1.weight", (int64_t)b->br * cin, &p))) return rc;
    if ((rc = upload_packed(h, p, b->br, cin, 1, &b->w1))) return rc;
    if ((rc = find(tm, pre + ".branch_conv2.weight", (int64_t)b->br * b->br * k2 * k2, &p))) return rc;
    if ((rc = upload_packed(h, p, b->br, b->br, k2, &b->w2))) return rc;
    b->wU = b->wU43 = b->w1f = b->w3f = nullptr;
    const bool wino = mode == MODE_SAME && cout == cin && h->use_wino &&       // conv_wino.hip: fp32 C = 32/64/128; 16-bit C = 32
                      (h->cfg.compute_dtype == VQAE_DT_F32 ? (cin == 256 || cin == 128 || cin == 64 || cin == 32) : cin == 32);
    if (wino && (rc = upload_wino(h, p, cin, &b->wU))) return rc;
    // F(4x4, 3x3) form (C = 256 / 128 on the 32-wide code grid, 64 on the 64-wide, 32 on the 128-wide level; the grid is not known
    // here, vqae::wino43_supported decides per launch and the F(2x2, 3x3) weights stay for the other grids)
    if (wino && h->cfg.compute_dtype == VQAE_DT_F32 && vqae::wino43_enabled() && vqae::wino43_supported(cin, 8, cin >= 128 ? 32 : (cin == 64 ? 64 : 128), VQAE_DT_F32) &&
        (rc = upload_wino43(h, p, cin, &b->wU43))) return rc;
    if ((rc = find(tm, pre + ".branch_conv3.weight", (int64_t)cout * b->br, &p))) return rc;
    if ((rc = upload_packed(h, p, cout, b->br, 1, &b->w3))) return rc;
    if (wino || (mode == MODE_SAME && (cin == 128 || cin == 64) && cout == cin)) {      // blocks that run a fused-tail kernel
        const int sk = wino ? 8 : vqae::conv_tail_kslice(h->cfg.compute_dtype, cin);
        void *f1, *f3;
        if ((rc = dev_alloc(h, (size_t)cin * cin * 4, &f1)) || (rc = dev_alloc(h, (size_t)cin * cin * 4, &f3))) return rc;
        b->w1f = (float*)f1; b->w3f = (float*)f3;
        if ((rc = vqae::wino_frag_weight(b->w1, cin, sk, b->w1f, nullptr)) || (rc = vqae::wino_frag_weight(b->w3, cin, sk, b->w3f, nullptr))) return rc;
        VQAE_HIP_CHECK(hipDeviceSynchronize());
    }
    b->w1h = b->w2h = b->w3h = nullptr;
    if (mode == MODE_SAME && cout == cin && h->cfg.compute_dtype != VQAE_DT_F32 && (cin == 16 || cin == 32 || cin == 64 || cin == 128 || cin == 256)) {
        struct { float* src; int taps; void** dst; } m[3] = {{b->w1, 1, &b->w1h}, {b->w2, 9, &b->w2h}, {b->w3, 1, &b->w3h}};
        for (auto& e : m) {
            if ((rc = dev_alloc(h, vqae::trunk16_weight_bytes(cin, e.taps), e.dst))) return rc;
            if ((rc = vqae::trunk16_pack_weight(e.src, cin, e.taps, h->cfg.compute_dtype, *e.dst, nullptr))) return rc;
        }
        VQAE_HIP_CHECK(hipDeviceSynchronize());
    }
    if (mode != MODE_SAME) {
        S_(b1c, "bias1c") S_(b1d, "bias1d")
        const int ks = mode == MODE_DOWN ? 2 : 1;
        if ((rc = find(tm, pre + ".skip_conv.weight", (int64_t)cout * cin * ks * ks, &p))) return rc;
        if ((rc = upload_packed(h, p, cout, cin, ks, &b->wskip))) return rc;
    }
    b->w2f = b->wskf = nullptr;
    if (mode == MODE_DOWN && h->fuse_down && cout == 2 * cin &&
        (cin == 16 || cin == 32 || cin == 64)) {             // whole block in one launch (down_fused.hip)
        struct { float* src; int K; float** dst; } m[4] = {{b->w1, cin, &b->w1f}, {b->w2, 4 * cout, &b->w2f},
                                                          {b->w3, cout, &b->w3f}, {b->wskip, 4 * cin, &b->wskf}};
        for (auto& e : m) {
            void* f;
            if ((rc = dev_alloc(h, (size_t)cout * e.K * 4, &f))) return rc;
            *e.dst = (float*)f;
            if ((rc = vqae::frag_weight_rect(e.src, cout, e.K, *e.dst, nullptr))) return rc;
        }
        VQAE_HIP_CHECK(hipDeviceSynchronize());
    }
    if (mode == MODE_DOWN && cout == 2 * cin && h->cfg.compute_dtype != VQAE_DT_F32 && h->fuse_down16 &&
        (cin == 8 || cin == 16 || cin == 32 || cin == 64)) {             // 16-bit MFMA form of the whole block (down16.hip)
        struct { float* src; int K; void** dst; } m16[4] = {{b->w1, cin, &b->dw1h}, {b->w2, 4 * cout, &b->dw2h},
                                                            {b->w3, cout, &b->dw3h}, {b->wskip, 4 * cin, &b->dwskh}};
        for (auto& e : m16) {
            if ((rc = dev_alloc(h, vqae::down16_weight_bytes(cout, e.K), e.dst))) return rc;
            if ((rc = vqae::down16_pack_weight(e.src, cout, e.K, h->cfg.compute_dtype, *e.dst, nullptr))) return rc;
        }
        VQAE_HIP_CHECK(hipDeviceSynchronize());
    }
    if (mode == MODE_SAME && cin == 8 && cout == 8 && h->cfg.compute_dtype != VQAE_DT_F32) {   // same8_16.hip
        if ((rc = dev_alloc(h, vqae::down16_weight_bytes(8, 72), &b->s8w2h)) || (rc = dev_alloc(h, vqae::down16_weight_bytes(8, 8), &b->s8w3h))) return rc;
        if ((rc = vqae::down16_pack_weight(b->w2, 8, 72, h->cfg.compute_dtype, b->s8w2h, nullptr))) return rc;
        if ((rc = vqae::down16_pack_weight(b->w3, 8, 8, h->cfg.compute_dtype, b->s8w3h, nullptr))) return rc;
        VQAE_HIP_CHECK(hipDeviceSynchronize());
    }
    if (mode == MODE_UP && cin == 2 * cout && h->cfg.compute_dtype != VQAE_DT_F32 && h->fuse_up16 &&
        (cin == 16 || cin == 32 || cin == 64 || cin == 128)) {            // 16-bit MFMA form: head16 (conv1) + up16.hip (the rest)
        if ((rc = dev_alloc(h, vqae::trunk16_weight_bytes(cin, 1), &b->uw1h))) return rc;
        if ((rc = vqae::trunk16_pack_weight(b->w1, cin, 1, h->cfg.compute_dtype, b->uw1h, nullptr))) return rc;
        struct { float* src; int rows; void** dst; } mu[3] = {{b->w2, cin, &b->uw2h}, {b->w3, cout, &b->uw3h}, {b->wskip, cout, &b->uwskh}};
        for (auto& e : mu) {
            if ((rc = dev_alloc(h, vqae::down16_weight_bytes(e.rows, cin), e.dst))) return rc;
            if ((rc = vqae::down16_pack_weight(e.src, e.rows, cin, h->cfg.compute_dtype, *e.dst, nullptr))) return rc;
        }
        VQAE_HIP_CHECK(hipDeviceSynchronize());
    }
#undef S_
    return VQAE_OK;
}

// MBConv (conv_block.py:240-321) in eval mode.  Each BatchNorm2d (batchnorm2d.yaml: eps, running statistics) follows a
// bias-free conv, so it folds into that conv: w' = w * g, shift = beta - mean * g, g = gamma / sqrt(var + eps).
int bn_fold(const TensorMap& tm, const std::string& pre, int c, float eps, std::vector<float>* g, std::vector<float>* shift) {
    const float *gamma, *beta, *mean, *var;
    int rc;
    if ((rc = find(tm, pre + ".weight", c, &gamma)) || (rc = find(tm, pre + ".bias", c, &beta)) ||
        (rc = find(tm, pre + ".running_mean", c, &mean)) || (rc = find(tm, pre + ".running_var", c, &var))) return rc;
    g->resize(c); shift->resize(c);
    for (int i = 0; i < c; ++i) {
        (*g)[i] = gamma[i] / std::sqrt(var[i] + eps);
        (*shift)[i] = beta[i] - mean[i] * (*g)[i];
    }
    return VQAE_OK;
}

int load_mbconv(vqae_handle* h, const TensorMap& tm, const std::string& pre, int mode, int cin, int cout, Block* b) {
    *b = Block();
    b->kind = VQAE_BLOCK_MBCONV;
    b->mode = mode; b->cin = cin; b->cout = cout;
    const int e = (cin > cout ? cin : cout) * h->cfg.expand_ratio;               // conv_block.py:255-259
    b->br = e;
    const int div = h->cfg.se_divisor;
    b->hidden = std::max(div, (int)(e + div / 2.0)) / div;                        // make_divisible, train_helpers.py:21-24
    VQAE_REQUIRE(e % 32 == 0 && e <= 1024, VQAE_ERR_UNSUPPORTED, "MBConv: expanded width %d must be a multiple of 32, <= 1024", e);
    const std::string br = pre + ".branch.";
    const float eps = h->cfg.bn_eps;
    std::vector<float> g, sh, w;
    const float* p;
    int rc;
    // 0: 1x1 expand + 1: BN
    if ((rc = find(tm, br + "0.weight", (int64_t)e * cin, &p)) || (rc = bn_fold(tm, br + "1", e, eps, &g, &sh))) return rc;
    w.assign(p, p + (int64_t)e * cin);
    for (int o = 0; o < e; ++o) for (int i = 0; i < cin; ++i) w[(int64_t)o * cin + i] *= g[o];
    if ((rc = upload_packed(h, w.data(), e, cin, 1, &b->w1)) || (rc = upload(h, sh.data(), e, &b->bv1))) return rc;
    // 3: depthwise + 4: BN   (Conv2d weight [e][1][k][k]; ConvTranspose2d weight [e][1][k][k], groups = e)
    const int k2 = mode == MODE_SAME ? 3 : 2;
    if ((rc = find(tm, br + "3.weight", (int64_t)e * k2 * k2, &p)) || (rc = bn_fold(tm, br + "4", e, eps, &g, &sh))) return rc;
    w.assign((size_t)k2 * k2 * e, 0.f);
    for (int c = 0; c < e; ++c) for (int t = 0; t < k2 * k2; ++t) w[(size_t)t * e + c] = p[(size_t)c * k2 * k2 + t] * g[c];
    if ((rc = upload(h, w.data(), (int64_t)k2 * k2 * e, &b->w2)) || (rc = upload(h, sh.data(), e, &b->bv2))) return rc;
    // 6: SELayer
    if ((rc = find(tm, br + "6.fc.0.weight", (int64_t)b->hidden * e, &p)) || (rc = upload(h, p, (int64_t)b->hidden * e, &b->fc0w))) return rc;
    if ((rc = find(tm, br + "6.fc.0.bias", b->hidden, &p)) || (rc = upload(h, p, b->hidden, &b->fc0b))) return rc;
    if ((rc = find(tm, br + "6.fc.2.weight", (int64_t)e * b->hidden, &p)) || (rc = upload(h, p, (int64_t)e * b->hidden, &b->fc2w))) return rc;
    if ((rc = find(tm, br + "6.fc.2.bias", e, &p)) || (rc = upload(h, p, e, &b->fc2b))) return rc;
    // 7: 1x1 project + 8: BN
    if ((rc = find(tm, br + "7.weight", (int64_t)cout * e, &p)) || (rc = bn_fold(tm, br + "8", cout, eps, &g, &sh))) return rc;
    w.assign(p, p + (int64_t)cout * e);
    for (int o = 0; o < cout; ++o) for (int i = 0; i < e; ++i) w[(int64_t)o * e + i] *= g[o];
    if ((rc = upload_packed(h, w.data(), cout, e, 1, &b->w3)) || (rc = upload(h, sh.data(), cout, &b->bv3))) return rc;
    // skip_conv (conv_block.py:303-310): none for 'same' with cin == cout
    if (mode == MODE_DOWN) {
        if ((rc = find(tm, pre + ".skip_conv.weight", (int64_t)cout * cin * 4, &p)) || (rc = upload_packed(h, p, cout, cin, 2, &b->wskip))) return rc;
    } else if (mode == MODE_UP) {
        // ConvTranspose2d weight [cin][cout][2][2] -> a 1x1 conv with 4*cout outputs ordered (a, b, co)
        if ((rc = find(tm, pre + ".skip_conv.weight", (int64_t)cin * cout * 4, &p))) return rc;
        w.assign((size_t)4 * cout * cin, 0.f);
        for (int ci = 0; ci < cin; ++ci) for (int co = 0; co < cout; ++co) for (int t = 0; t < 4; ++t)
            w[((size_t)t * cout + co) * cin + ci] = p[((size_t)ci * cout + co) * 4 + t];
        if ((rc = upload_packed(h, w.data(), 4 * cout, cin, 1, &b->wskip))) return rc;
    } else if (cin != cout) {
        if ((rc = find(tm, pre + ".skip_conv.weight", (int64_t)cout * cin, &p)) || (rc = upload_packed(h, p, cout, cin, 1, &b->wskip))) return rc;
    }
    return VQAE_OK;
}

int load_any(vqae_handle* h, const TensorMap& tm, const std::string& pre, int mode, int cin, int cout, Block* b) {
    return h->cfg.block_kind == VQAE_BLOCK_MBCONV ? load_mbconv(h, tm, pre, mode, cin, cout, b)
                                                  : load_block(h, tm, pre, mode, cin, cout, b);
}

// ---- one conv launch --------------------------------------------------------------------------
// compute dtype of the handle currently executing (set at the top of every entry point; handles are not
// shared across threads)
thread_local int g_dt = VQAE_DT_F32;
struct ConvCall {
    vqae_conv_args a;
    ConvCall(int B, int H, int W, int cin, int cout, int ks, int stride, int pad, int pad_mode, int dt = g_dt) {
        memset(&a, 0, sizeof(a));
        a.dtype = dt;
        a.batch = B; a.in_h = H; a.in_w = W; a.cin = cin; a.cout = cout;
        a.ksize = ks; a.stride = stride; a.pad = pad; a.pad_mode = pad_mode;
    }
    ConvCall& pre(int mode, float pa, float pb) { a.pre_mode = mode; a.pre_a = pa; a.pre_b = pb; return *this; }
    ConvCall& act(float aa, float ab) { a.has_act = 1; a.act_a = aa; a.act_b = ab; return *this; }
    ConvCall& scale_bias(float s, float b) { a.has_scale = 1; a.scale = s; a.bias_s = b; return *this; }
    ConvCall& bias(float b) { a.has_bias_s = 1; a.bias_s = b; return *this; }
};

// MBConv.forward (conv_block.py:316-321), eval mode, on NHWC buffers; on return buf[0] holds the output.
//   X --1x1 (+shift1, SiLU)--> P [E] --depthwise (+shift2, SiLU, strip sums)--> Q [E] --SE gate--> g [B][E]
//   out = conv1x1(Q * g) + shift3 + skip
int run_mbconv(vqae_handle* h, const Block& b, int B, int& H, int& W, hipStream_t st) {
    float *X = h->buf[0], *P = h->buf[1], *Q = h->buf[2], *R = h->buf[3];
    float* partial = h->se_ws;
    float* gate = h->se_ws + h->se_gate_off;
    const int E = b.br;
    int rc;
    const float* skip = X;
    int Ho = H, Wo = W;
    if (b.mode == MODE_DOWN) {
        ConvCall sk(B, H, W, b.cin, b.cout, 2, 2, 0, VQAE_PAD_NONE);
        if ((rc = vqae_conv2d_f32(&sk.a, X, b.wskip, nullptr, nullptr, R, st))) return rc;
        skip = R; Ho = H / 2; Wo = W / 2;
    } else if (b.mode == MODE_UP) {
        ConvCall sk(B, H, W, b.cin, 4 * b.cout, 1, 1, 0, VQAE_PAD_NONE);       // ConvTranspose2d(k2, s2) = 1x1 conv + pixel shuffle
        if ((rc = vqae_conv2d_f32(&sk.a, X, b.wskip, nullptr, nullptr, Q, st))) return rc;
        if ((rc = vqae_pixel_shuffle2_f32(Q, B, H, W, b.cout, R, st))) return rc;
        skip = R; Ho = 2 * H; Wo = 2 * W;
    } else if (b.wskip) {
        ConvCall sk(B, H, W, b.cin, b.cout, 1, 1, 0, VQAE_PAD_NONE);
        if ((rc = vqae_conv2d_f32(&sk.a, X, b.wskip, nullptr, nullptr, R, st))) return rc;
        skip = R;
    }
    ConvCall c1(B, H, W, b.cin, E, 1, 1, 0, VQAE_PAD_NONE);
    c1.a.has_act = VQAE_ACT_SILU;
    if ((rc = vqae_conv2d_f32(&c1.a, X, b.w1, b.bv1, nullptr, P, st))) return rc;
    const int dwm = b.mode == MODE_SAME ? VQAE_DW_SAME : (b.mode == MODE_DOWN ? VQAE_DW_DOWN : VQAE_DW_UP);
    if ((rc = vqae_dwconv_f32(P, b.w2, b.bv2, B, H, W, E, dwm, 1, Q, partial, st))) return rc;
    if ((rc = vqae_se_gate_f32(partial, B, Ho, Wo, E, b.fc0w, b.fc0b, b.hidden, b.fc2w, b.fc2b, gate, st))) return rc;
    ConvCall c3(B, Ho, Wo, E, b.cout, 1, 1, 0, VQAE_PAD_NONE);
    c3.a.pre_mode = VQAE_PRE_CHANNEL_GATE;
    float* out = skip == X ? X : R;                                            // in-place residual add
    if ((rc = vqae_conv2d_gated_f32(&c3.a, Q, gate, b.w3, b.bv3, skip, out, st))) return rc;
    if (out == R) std::swap(h->buf[0], h->buf[3]);
    H = Ho; W = Wo;
    return VQAE_OK;
}

// PreActFixupResBlock.forward (conv_block.py:196-216) on NHWC buffers.  X holds the input and, on
// return, buf[0] holds the output (buffers are swapped for down/up).
int run_block(vqae_handle* h, const Block& b, const Block* next, int B, int& H, int& W, hipStream_t st) {
    if (b.kind == VQAE_BLOCK_MBCONV) return run_mbconv(h, b, B, H, W, st);
    float *X = h->buf[0], *P = h->buf[1], *Q = h->buf[2], *R = h->buf[3];
    int rc;
    if (b.mode == MODE_SAME && b.w2h && h->fuse_trunk && vqae::same16_16_supported(b.cin, H, W, g_dt)) {
        // 16-bit modes, C = 16 / 32: a whole block per launch (csrc/same8_16.hip) beats the chained trunk16 launches at these widths
        const float sc[8] = {b.b1a, b.b1b, b.b2a, b.b2b, b.b3a, b.b3b, b.b4, b.scale};
        if ((rc = vqae::same16_16_block(X, P, b.w1h, b.w2h, b.w3h, B, H, W, b.cin, sc, g_dt, st))) return rc;
        std::swap(h->buf[0], h->buf[1]);
        h->t1_ready = false;
        return VQAE_OK;
    }
    if (b.mode == MODE_SAME && b.w2h && h->fuse_trunk && vqae::trunk16_supported(b.cin, H, W, g_dt)) {
        // 16-bit modes, C = 64 / 128 / 256 (trunk16.hip): t1 travels as 16-bit; one launch per block
        if (!h->t1_ready && vqae::trunk16_head_supported(b.cin, (int64_t)B * H * W, g_dt)) {   // chain head: its own conv1 launch
            if ((rc = vqae::trunk16_head(X, b.w1h, b.b1a, b.b1b, b.b2a, b.b2b, P, (int64_t)B * H * W, b.cin, g_dt, false, st))) return rc;
        } else if (!h->t1_ready) {                   // ... or the generic kernel (fp32 out) + the conv2 input cast
            ConvCall c1(B, H, W, b.cin, b.br, 1, 1, 0, VQAE_PAD_NONE);
            c1.pre(VQAE_PRE_BIAS_ELU_BIAS, b.b1a, b.b1b).act(b.b2a, b.b2b);
            if ((rc = vqae_conv2d_f32(&c1.a, X, b.w1, nullptr, nullptr, Q, st))) return rc;
            if ((rc = vqae::trunk16_round_pack(Q, P, (int64_t)B * H * W * b.cin, g_dt, st))) return rc;
        }
        const bool chain = next && next->mode == MODE_SAME && next->cin == b.cin && next->cout == b.cin && next->w1h;
        if ((rc = vqae::trunk16_block(P, b.w2h, b.w3h, b.b3a, b.b3b, b.scale, b.b4, X, chain ? next->w1h : nullptr,
                                      chain ? next->b1a : 0.f, chain ? next->b1b : 0.f, chain ? next->b2a : 0.f,
                                      chain ? next->b2b : 0.f, chain ? Q : nullptr, B, H, W, b.cin, g_dt, st))) return rc;
        if (chain) std::swap(h->buf[1], h->buf[2]);
        h->t1_ready = chain;
        return VQAE_OK;
    }
    const bool wino = b.mode == MODE_SAME && b.wU && h->fuse_trunk && vqae::wino_trunk_supported(b.cin, H, W, g_dt);
    if (b.mode == MODE_SAME && (wino || ((b.cin == 128 || b.cin == 64) && b.cout == b.cin && h->fuse_trunk))) {
        // trunk: conv1 (unless the previous block's tail already produced t1 in P), then ONE launch for
        // conv2 + conv3 (+ the next block's conv1 when it is another 'same' block of this width)
        if (!h->t1_ready) {
            const int64_t M = (int64_t)B * H * W;
            if (wino && g_dt == VQAE_DT_F32 && vqae::fixup_conv1_supported(b.cin, M)) {
                if ((rc = vqae::fixup_conv1(X, b.w1f, b.b1a, b.b1b, b.b2a, b.b2b, P, M, b.cin, st))) return rc;
            } else {
                ConvCall c1(B, H, W, b.cin, b.br, 1, 1, 0, VQAE_PAD_NONE);
                c1.pre(VQAE_PRE_BIAS_ELU_BIAS, b.b1a, b.b1b).act(b.b2a, b.b2b);
                if ((rc = vqae_conv2d_f32(&c1.a, X, b.w1, nullptr, nullptr, P, st))) return rc;
            }
        }
        const bool chain = next && next->mode == MODE_SAME && next->cin == b.cin && next->cout == b.cin && (!wino || next->w1f);
        if (wino && b.wU43 && vqae::wino43_supported(b.cin, H, W, g_dt)) {
            if ((rc = vqae::wino43_trunk_tail(P, b.wU43, b.w3f, b.b3a, b.b3b, b.scale, b.b4, X, chain ? next->w1f : nullptr,
                                              chain ? next->b1a : 0.f, chain ? next->b1b : 0.f, chain ? next->b2a : 0.f,
                                              chain ? next->b2b : 0.f, chain ? Q : nullptr, B, H, W, b.cin, st))) return rc;
            if (chain) std::swap(h->buf[1], h->buf[2]);
            h->t1_ready = chain;
            return VQAE_OK;
        }
        if (wino) {
            if ((rc = vqae::wino_trunk_tail(P, b.wU, b.w3f, b.b3a, b.b3b, b.scale, b.b4, X, chain ? next->w1f : nullptr,
                                            chain ? next->b1a : 0.f, chain ? next->b1b : 0.f, chain ? next->b2a : 0.f,
                                            chain ? next->b2b : 0.f, chain ? Q : nullptr, B, H, W, b.cin, g_dt, st))) return rc;
            if (chain) std::swap(h->buf[1], h->buf[2]);
            h->t1_ready = chain;
            return VQAE_OK;
        }
        ConvCall c2(B, H, W, b.br, b.br, 3, 1, 1, VQAE_PAD_CIRCULAR);
        c2.act(b.b3a, b.b3b);
        if ((rc = vqae::conv_trunk_tail(&c2.a, P, b.w2, b.w3f, b.scale, b.b4, X, chain ? next->w1f : nullptr,
                                        chain ? next->b1a : 0.f, chain ? next->b1b : 0.f, chain ? next->b2a : 0.f,
                                        chain ? next->b2b : 0.f, chain ? Q : nullptr, st))) return rc;
        if (chain) std::swap(h->buf[1], h->buf[2]);
        h->t1_ready = chain;
        return VQAE_OK;
    }
    h->t1_ready = false;
    if (b.mode == MODE_SAME && b.s8w2h && g_dt != VQAE_DT_F32 && vqae::same8_16_supported(b.cin, H, W, g_dt)) {
        // 16-bit modes, C = 8: the whole block on the 16-bit MFMA (csrc/same8_16.hip), X -> P, swap
        const float sc[8] = {b.b1a, b.b1b, b.b2a, b.b2b, b.b3a, b.b3b, b.b4, b.scale};
        if ((rc = vqae::same8_16_block(X, P, b.w1, b.s8w2h, b.s8w3h, B, H, W, sc, g_dt, st))) return rc;
        std::swap(h->buf[0], h->buf[1]);
        return VQAE_OK;
    }
    if (b.mode == MODE_SAME && b.cin == b.cout && vqae_fixup_same_supported(b.cin, H, W)) {
        // high-resolution levels: the whole block in one launch (csrc/fixup_fused.hip), X -> P, swap
        const float sc[8] = {b.b1a, b.b1b, b.b2a, b.b2b, b.b3a, b.b3b, b.b4, b.scale};
        if ((rc = vqae_fixup_same_block_f32(X, P, b.w1, b.w2, b.w3, B, H, W, b.cin, sc, g_dt, st))) return rc;
        std::swap(h->buf[0], h->buf[1]);
        return VQAE_OK;
    }
    if (b.mode == MODE_SAME) {
        ConvCall c1(B, H, W, b.cin, b.br, 1, 1, 0, VQAE_PAD_NONE);
        c1.pre(VQAE_PRE_BIAS_ELU_BIAS, b.b1a, b.b1b).act(b.b2a, b.b2b);
        if ((rc = vqae_conv2d_f32(&c1.a, X, b.w1, nullptr, nullptr, P, st))) return rc;
        ConvCall c2(B, H, W, b.br, b.br, 3, 1, 1, VQAE_PAD_CIRCULAR);
        c2.act(b.b3a, b.b3b);
        if ((rc = vqae_conv2d_f32(&c2.a, P, b.w2, nullptr, nullptr, Q, st))) return rc;
        ConvCall c3(B, H, W, b.br, b.cout, 1, 1, 0, VQAE_PAD_NONE);
        c3.scale_bias(b.scale, b.b4);
        return vqae_conv2d_f32(&c3.a, Q, b.w3, nullptr, X, X, st);          // + inp, in place
    }
    if (b.mode == MODE_DOWN && b.dw2h && g_dt != VQAE_DT_F32 && vqae::down16_supported(b.cin, H, W)) {
        const float sc[10] = {b.b1a, b.b1b, b.b2a, b.b2b, b.b3a, b.b3b, b.b4, b.scale, b.b1c, b.b1d};
        if ((rc = vqae::down16_block(X, b.dw1h, b.dw2h, b.dw3h, b.dwskh, B, H, W, b.cin, sc, g_dt, R, st))) return rc;
        H /= 2; W /= 2;
        std::swap(h->buf[0], h->buf[3]);
        return VQAE_OK;
    }
    if (b.mode == MODE_DOWN && b.w2f && vqae::down_block_supported(b.cin, H, W)) {
        const float sc[10] = {b.b1a, b.b1b, b.b2a, b.b2b, b.b3a, b.b3b, b.b4, b.scale, b.b1c, b.b1d};
        if ((rc = vqae::down_block(X, b.w1f, b.w2f, b.w3f, b.wskf, B, H, W, b.cin, sc, g_dt, R, st))) return rc;
        H /= 2; W /= 2;
        std::swap(h->buf[0], h->buf[3]);
        return VQAE_OK;
    }
    if (b.mode == MODE_DOWN) {
        ConvCall sk(B, H, W, b.cin, b.cout, 2, 2, 0, VQAE_PAD_NONE);         // skip_conv(inp + bias1c) + bias1d
        sk.pre(VQAE_PRE_BIAS, b.b1c, 0.f).bias(b.b1d);
        if ((rc = vqae_conv2d_f32(&sk.a, X, b.wskip, nullptr, nullptr, R, st))) return rc;
        ConvCall c1(B, H, W, b.cin, b.br, 1, 1, 0, VQAE_PAD_NONE);
        c1.pre(VQAE_PRE_BIAS_ELU_BIAS, b.b1a, b.b1b).act(b.b2a, b.b2b);
        if ((rc = vqae_conv2d_f32(&c1.a, X, b.w1, nullptr, nullptr, P, st))) return rc;
        ConvCall c2(B, H, W, b.br, b.br, 2, 2, 0, VQAE_PAD_NONE);
        c2.act(b.b3a, b.b3b);
        if ((rc = vqae_conv2d_f32(&c2.a, P, b.w2, nullptr, nullptr, Q, st))) return rc;
        H /= 2; W /= 2;
        ConvCall c3(B, H, W, b.br, b.cout, 1, 1, 0, VQAE_PAD_NONE);
        c3.scale_bias(b.scale, b.b4);
        if ((rc = vqae_conv2d_f32(&c3.a, Q, b.w3, nullptr, R, R, st))) return rc;
        std::swap(h->buf[0], h->buf[3]);
        return VQAE_OK;
    }
    // MODE_UP: ResizeConv2D = conv1x1(bicubic_x2(.)) (layers/conv.py:10-11)
    if (g_dt == VQAE_DT_F32 && h->up_conv_first) {
        // A 1x1 conv commutes with the (channel-wise, linear) bicubic resize: run both ResizeConv2D convs at the
        // LOW resolution and upsample their outputs -- 4x fewer MACs and 2.7x less HBM traffic than conv-after-
        // resize.  Mathematically identical; rounding differs at the 1e-7 level (validated <= 1e-5 MSE, SURVEY
        // §8 a5).  fp32 only: under autocast the 16-bit rounding points would move.
        ConvCall sk(B, H, W, b.cin, b.cout, 1, 1, 0, VQAE_PAD_NONE);
        sk.pre(VQAE_PRE_BIAS, b.b1c, 0.f).bias(b.b1d);                                   // skip_conv(inp + b1c) + b1d
        if ((rc = vqae_conv2d_f32(&sk.a, X, b.wskip, nullptr, nullptr, Q, st))) return rc;
        if (h->fuse_up_tail && vqae::up_tail_supported(b.br, b.cout)) {
            // stem-side levels: both resizes, the ELU and conv3 in one launch (misc_kernels.hip up_tail_kernel)
            ConvCall c1(B, H, W, b.cin, b.br, 1, 1, 0, VQAE_PAD_NONE);
            c1.pre(VQAE_PRE_BIAS_ELU_BIAS, b.b1a, b.b1b).act(b.b2a, b.b2b);
            if ((rc = vqae_conv2d_f32(&c1.a, X, b.w1, nullptr, nullptr, P, st))) return rc;
            ConvCall c2(B, H, W, b.br, b.br, 1, 1, 0, VQAE_PAD_NONE);
            if ((rc = vqae_conv2d_f32(&c2.a, P, b.w2, nullptr, nullptr, R, st))) return rc;
            if ((rc = vqae::up_tail(R, Q, b.w3, B, H, W, b.br, b.cout, b.b3a, b.b3b, b.scale, b.b4, X, st))) return rc;
            H *= 2; W *= 2;
            return VQAE_OK;                                                             // output in buf[0]
        }
        if ((rc = vqae_bicubic_up2_f32(Q, B, H, W, b.cout, 0.f, R, st))) return rc;
        ConvCall c1(B, H, W, b.cin, b.br, 1, 1, 0, VQAE_PAD_NONE);
        c1.pre(VQAE_PRE_BIAS_ELU_BIAS, b.b1a, b.b1b).act(b.b2a, b.b2b);
        if ((rc = vqae_conv2d_f32(&c1.a, X, b.w1, nullptr, nullptr, P, st))) return rc;
        ConvCall c2(B, H, W, b.br, b.br, 1, 1, 0, VQAE_PAD_NONE);                          // conv2 at low resolution
        if ((rc = vqae_conv2d_f32(&c2.a, P, b.w2, nullptr, nullptr, Q, st))) return rc;
        if ((rc = vqae_bicubic_up2_f32(Q, B, H, W, b.br, 0.f, P, st))) return rc;
        H *= 2; W *= 2;
        ConvCall c3(B, H, W, b.br, b.cout, 1, 1, 0, VQAE_PAD_NONE);
        c3.pre(VQAE_PRE_BIAS_ELU_BIAS, b.b3a, b.b3b).scale_bias(b.scale, b.b4);
        if ((rc = vqae_conv2d_f32(&c3.a, P, b.w3, nullptr, R, R, st))) return rc;
        std::swap(h->buf[0], h->buf[3]);
        return VQAE_OK;
    }
    if (b.uw2h && vqae::up16_supported(b.cin, H, W, g_dt) && vqae::trunk16_head_supported(b.cin, (int64_t)B * H * W, g_dt)) {
        // 16-bit modes: conv1 at the low resolution (fp32 result of the activation), then the whole high-resolution part in one launch
        if ((rc = vqae::trunk16_head(X, b.uw1h, b.b1a, b.b1b, b.b2a, b.b2b, Q, (int64_t)B * H * W, b.cin, g_dt, true, st))) return rc;
        if ((rc = vqae::up16_block(X, Q, b.uw2h, b.uw3h, b.uwskh, B, H, W, b.cin, b.b3a, b.b3b, b.scale, b.b4, b.b1c, b.b1d, g_dt, R, st))) return rc;
        H *= 2; W *= 2;
        std::swap(h->buf[0], h->buf[3]);
        return VQAE_OK;
    }
    if ((rc = vqae_bicubic_up2_f32(X, B, H, W, b.cin, b.b1c, P, st))) return rc;               // up(inp + bias1c)
    ConvCall sk(B, 2 * H, 2 * W, b.cin, b.cout, 1, 1, 0, VQAE_PAD_NONE);
    sk.bias(b.b1d);
    if ((rc = vqae_conv2d_f32(&sk.a, P, b.wskip, nullptr, nullptr, R, st))) return rc;
    ConvCall c1(B, H, W, b.cin, b.br, 1, 1, 0, VQAE_PAD_NONE);
    c1.pre(VQAE_PRE_BIAS_ELU_BIAS, b.b1a, b.b1b).act(b.b2a, b.b2b);
    if ((rc = vqae_conv2d_f32(&c1.a, X, b.w1, nullptr, nullptr, Q, st))) return rc;
    if ((rc = vqae_bicubic_up2_f32(Q, B, H, W, b.br, 0.f, P, st))) return rc;
    H *= 2; W *= 2;
    ConvCall c2(B, H, W, b.br, b.br, 1, 1, 0, VQAE_PAD_NONE);
    c2.act(b.b3a, b.b3b);
    if ((rc = vqae_conv2d_f32(&c2.a, P, b.w2, nullptr, nullptr, Q, st))) return rc;
    ConvCall c3(B, H, W, b.br, b.cout, 1, 1, 0, VQAE_PAD_NONE);
    c3.scale_bias(b.scale, b.b4);
    if ((rc = vqae_conv2d_f32(&c3.a, Q, b.w3, nullptr, R, R, st))) return rc;
    std::swap(h->buf[0], h->buf[3]);
    return VQAE_OK;
}

size_t max_floats_per_patch(const vqae_handle* h, int in_h, int in_w) {
    // largest NHWC intermediate: the bicubic-upsampled 2C tensor of the last up block / conv1 of the
    // first down block (both 2*stem channels at full resolution), or the code tensor.
    const size_t widen = h->cfg.block_kind == VQAE_BLOCK_MBCONV ? (size_t)h->cfg.expand_ratio : 1;   // MBConv: expanded tensors
    size_t full = (size_t)in_h * in_w * (size_t)(2 * h->cfg.stem) * widen;
    size_t lat = (size_t)(in_h >> h->cfg.n_down) * (in_w >> h->cfg.n_down) * (size_t)h->C * widen;
    return full > lat ? full : lat;
}

int ensure_bufs(vqae_handle* h, size_t need) {
    if (need > h->buf_floats) {
        VQAE_HIP_CHECK(hipDeviceSynchronize());
        for (int i = 0; i < 4; ++i) {
            if (h->buf[i]) (void)hipFree(h->buf[i]);
            h->buf[i] = nullptr;
        }
        h->buf_floats = 0;
        for (int i = 0; i < 4; ++i)
            if (hipMalloc((void**)&h->buf[i], need * sizeof(float)) != hipSuccess)
                return vqae::fail(VQAE_ERR_NOMEM, "workspace hipMalloc of %zu bytes failed", need * sizeof(float));
        h->buf_floats = need;
    }
    return VQAE_OK;
}

int ensure_workspace(vqae_handle* h, int B, int in_h, int in_w) {
    int rc = ensure_bufs(h, max_floats_per_patch(h, in_h, in_w) * (size_t)(B > 0 ? B : 1));
    if (rc) return rc;
    const int64_t rows = (int64_t)B * (in_h >> h->cfg.n_down) * (in_w >> h->cfg.n_down);
    size_t vq_need = vqae_vq_workspace_bytes(rows, h->K, h->D);
    if (h->D == 8) vq_need = std::max(vq_need, vqae_vq_projected_workspace_bytes(rows));
    if (vq_need > h->vq_ws_bytes) {
        VQAE_HIP_CHECK(hipDeviceSynchronize());
        if (h->vq_ws) (void)hipFree(h->vq_ws);
        h->vq_ws = nullptr; h->vq_ws_bytes = 0;
        if (hipMalloc(&h->vq_ws, vq_need) != hipSuccess)
            return vqae::fail(VQAE_ERR_NOMEM, "vq workspace hipMalloc of %zu bytes failed", vq_need);
        h->vq_ws_bytes = vq_need;
    }
    if (h->cfg.block_kind == VQAE_BLOCK_MBCONV) {
        // SE workspace: strip sums of the widest (image, level) + the gate; channels * pixels is largest at full resolution
        const int e_max = 2 * h->cfg.stem * h->cfg.expand_ratio, e_lat = h->C * h->cfg.expand_ratio;
        size_t part = vqae_dw_partial_floats(B, in_h, in_w, e_max);
        const size_t part_lat = vqae_dw_partial_floats(B, in_h >> h->cfg.n_down, in_w >> h->cfg.n_down, e_lat);
        if (part_lat > part) part = part_lat;
        for (int l = 1; l < h->cfg.n_down; ++l) {                                 // intermediate levels
            const size_t pl = vqae_dw_partial_floats(B, in_h >> l, in_w >> l, (2 * h->cfg.stem << l) * h->cfg.expand_ratio);
            if (pl > part) part = pl;
        }
        const size_t gate_floats = (size_t)(B > 0 ? B : 1) * (size_t)(e_lat > e_max ? e_lat : e_max);
        const size_t need_se = (size_t)vqae::round_up((int64_t)part, 64) + gate_floats;
        if (need_se > h->se_ws_floats) {
            VQAE_HIP_CHECK(hipDeviceSynchronize());
            if (h->se_ws) (void)hipFree(h->se_ws);
            h->se_ws = nullptr; h->se_ws_floats = 0;
            if (hipMalloc((void**)&h->se_ws, need_se * 4) != hipSuccess)
                return vqae::fail(VQAE_ERR_NOMEM, "SE workspace hipMalloc of %zu bytes failed", need_se * 4);
            h->se_ws_floats = need_se;
        }
        h->se_gate_off = h->se_ws_floats - gate_floats;
    }
    const size_t idx_need = (size_t)vqae::round_up(rows * 4, 256);
    if (idx_need > h->idx_scratch_bytes) {
        VQAE_HIP_CHECK(hipDeviceSynchronize());
        if (h->idx_scratch) (void)hipFree(h->idx_scratch);
        h->idx_scratch = nullptr; h->idx_scratch_bytes = 0;
        if (hipMalloc(&h->idx_scratch, idx_need) != hipSuccess)
            return vqae::fail(VQAE_ERR_NOMEM, "idx scratch hipMalloc of %zu bytes failed", idx_need);
        h->idx_scratch_bytes = idx_need;
    }
    return VQAE_OK;
}

// A handle's weights and workspaces live on the device that was current in vqae_create; kernels are launched on the
// caller's current device.  One process per GPU is the deployment model: refuse anything else.
int check_device(const vqae_handle* h) {
    int dev = -1;
    VQAE_HIP_CHECK(hipGetDevice(&dev));
    VQAE_REQUIRE(dev == h->device, VQAE_ERR_INVALID, "handle was created on HIP device %d but the current device is %d", h->device, dev);
    return VQAE_OK;
}

int check_geometry(const vqae_handle* h, int B, int in_h, int in_w) {
    const int f = 1 << h->cfg.n_down;
    if (int rc = check_device(h)) return rc;
    VQAE_REQUIRE(B >= 0, VQAE_ERR_INVALID, "negative batch");
    VQAE_REQUIRE(in_h >= f && in_w >= f && in_h % f == 0 && in_w % f == 0, VQAE_ERR_INVALID,
                 "input %dx%d must be a positive multiple of 2^n_down = %d", in_h, in_w, f);
    return VQAE_OK;
}

// in_stem + down blocks + pre_enc blocks: x -> z in buf[0]  (model.py:198-208)
int run_encoder_convs(vqae_handle* h, const void* x, int x_kind, int B, int in_h, int in_w, int* zh, int* zw,
                      hipStream_t st) {
    int rc;
    if (h->stem_wh && vqae::stem16_supported(h->cfg.stem, in_h, in_w, g_dt)) {          // 16-bit modes: the stem on the MFMA (stem16.hip)
        if ((rc = vqae::istem16(x, x_kind, kMean255, kInv255, h->stem_wh, h->stem_b, B, in_h, in_w, h->cfg.stem, h->buf[0], g_dt, st))) return rc;
    } else if ((rc = vqae::conv3x3_direct(x, x_kind, kMean255, kInv255, h->stem_w, h->stem_b, B, in_h, in_w,
                                          h->cfg.in_channels, h->cfg.stem, h->buf[0], 0, g_dt, st))) return rc;
    int H = in_h, W = in_w;
    h->t1_ready = false;
    for (size_t i = 0; i < h->enc.size(); ++i)
        if ((rc = run_block(h, h->enc[i], i + 1 < h->enc.size() ? &h->enc[i + 1] : nullptr, B, H, W, st))) return rc;
    *zh = H; *zw = W;
    return VQAE_OK;
}

// VQ level: z (buf[0]) -> q (NHWC, C channels) in buf[0]; idx/loss to the caller.
// Plain: EMAVectorQuantizer.forward (vq.py:96-154).  Projected: proj_out(VQ(proj_in(z))) (vq.py:190-192).
int run_vq(vqae_handle* h, int B, int zh, int zw, void* idx, int idx_dtype, float* loss, hipStream_t st) {
    const int64_t rows = (int64_t)B * zh * zw;
    int rc;
    if (h->cfg.projection_dim == 8 && h->fuse_vq && h->pin_wt) {
        // reference default: the whole ProjectedEMAVectorQuantizer2d.forward in one pass over z (vq_proj.hip)
        if ((rc = vqae_vq_projected_f32(h->buf[0], h->pin_wt, h->pin_b, h->embed, h->pout_wr, h->pout_b, rows, h->C, h->D, h->K,
                                        h->cfg.commitment_cost, g_dt, idx, idx_dtype, h->buf[1], nullptr, loss, nullptr,
                                        h->vq_ws, st))) return rc;
        std::swap(h->buf[0], h->buf[1]);
        return VQAE_OK;
    }
    if (h->cfg.projection_dim > 0) {
        ConvCall pin(B, zh, zw, h->C, h->D, 1, 1, 0, VQAE_PAD_NONE);
        if ((rc = vqae_conv2d_f32(&pin.a, h->buf[0], h->pin_w, h->pin_b, nullptr, h->buf[1], st))) return rc;
        if ((rc = vqae_vq_forward_f32(h->buf[1], h->embed, rows, h->K, h->D, h->cfg.commitment_cost, idx, idx_dtype,
                                      h->buf[2], loss, nullptr, h->vq_ws, st))) return rc;
        ConvCall pout(B, zh, zw, h->D, h->C, 1, 1, 0, VQAE_PAD_NONE);
        return vqae_conv2d_f32(&pout.a, h->buf[2], h->pout_w, h->pout_b, nullptr, h->buf[0], st);
    }
    if ((rc = vqae_vq_forward_f32(h->buf[0], h->embed, rows, h->K, h->D, h->cfg.commitment_cost, idx, idx_dtype,
                                  h->buf[1], loss, nullptr, h->vq_ws, st))) return rc;
    std::swap(h->buf[0], h->buf[1]);
    return VQAE_OK;
}

// post_enc blocks + up blocks + out_stem: q in buf[0] -> out (model.py:278-291)
int run_decoder_convs(vqae_handle* h, int B, int qh, int qw, int layout, float* out, hipStream_t st) {
    int H = qh, W = qw, rc;
    h->t1_ready = false;
    for (size_t i = 0; i < h->dec.size(); ++i)
        if ((rc = run_block(h, h->dec[i], i + 1 < h->dec.size() ? &h->dec[i + 1] : nullptr, B, H, W, st))) return rc;
    if (h->ostem_wh && vqae::stem16_supported(h->cfg.stem, H, W, g_dt))
        return vqae::ostem16(h->buf[0], h->ostem_wh, h->ostem_b, B, H, W, h->cfg.stem, out, layout == VQAE_LAYOUT_NCHW ? 1 : 0, g_dt, st);
    return vqae::conv3x3_direct(h->buf[0], 0, nullptr, nullptr, h->ostem_w, h->ostem_b, B, H, W, h->cfg.stem,
                                h->cfg.in_channels, out, layout == VQAE_LAYOUT_NCHW ? 1 : 0, g_dt, st);
}

int export_q(vqae_handle* h, int B, int zh, int zw, int layout, float* q, hipStream_t st) {
    if (!q) return VQAE_OK;
    if (layout == VQAE_LAYOUT_NCHW) return vqae_nhwc_to_nchw_f32(h->buf[0], B, h->C, zh, zw, q, st);
    VQAE_HIP_CHECK(hipMemcpyAsync(q, h->buf[0], (size_t)B * zh * zw * h->C * 4, hipMemcpyDeviceToDevice, st));
    return VQAE_OK;
}

}  // namespace

extern "C" int vqae_create(const vqae_config* cfg, const vqae_tensor* tensors, int n_tensors, vqae_handle** out) {
    VQAE_REQUIRE(cfg && tensors && out, VQAE_ERR_INVALID, "vqae_create: null pointer");
    VQAE_REQUIRE(cfg->in_channels == 3, VQAE_ERR_UNSUPPORTED, "in_channels %d (only 3)", cfg->in_channels);
    VQAE_REQUIRE(cfg->stem >= 4 && cfg->stem % 4 == 0 && cfg->stem <= 64, VQAE_ERR_UNSUPPORTED, "stem %d", cfg->stem);
    VQAE_REQUIRE(cfg->stem % 8 == 0, VQAE_ERR_UNSUPPORTED, "stem %d must be a multiple of 8", cfg->stem);
    VQAE_REQUIRE(cfg->n_down >= 1 && cfg->n_down <= 6 && cfg->n_pre >= 0 && cfg->n_post >= 0 && cfg->n_enc >= 0,
                 VQAE_ERR_INVALID, "bad depth parameters");
    VQAE_REQUIRE(cfg->num_embeddings >= 1 && cfg->num_embeddings <= 65536, VQAE_ERR_UNSUPPORTED, "num_embeddings %d",
                 cfg->num_embeddings);
    VQAE_REQUIRE(cfg->projection_dim == 0 || (cfg->projection_dim % 8 == 0), VQAE_ERR_UNSUPPORTED,
                 "projection_dim %d must be 0 or a multiple of 8", cfg->projection_dim);
    VQAE_REQUIRE(cfg->compute_dtype >= VQAE_DT_F32 && cfg->compute_dtype <= VQAE_DT_F16, VQAE_ERR_INVALID,
                 "compute_dtype %d", cfg->compute_dtype);
    VQAE_REQUIRE(cfg->block_kind == VQAE_BLOCK_FIXUP || cfg->block_kind == VQAE_BLOCK_MBCONV, VQAE_ERR_INVALID,
                 "block_kind %d", cfg->block_kind);
    if (cfg->block_kind == VQAE_BLOCK_MBCONV) {
        VQAE_REQUIRE(cfg->compute_dtype == VQAE_DT_F32, VQAE_ERR_UNSUPPORTED, "MBConv blocks run in fp32 only");
        VQAE_REQUIRE(cfg->expand_ratio >= 1 && cfg->se_divisor >= 1 && cfg->bn_eps > 0.f, VQAE_ERR_INVALID,
                     "MBConv: expand_ratio %d, se_divisor %d, bn_eps %g", cfg->expand_ratio, cfg->se_divisor, (double)cfg->bn_eps);
    }
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i) tm[tensors[i].name] = &tensors[i];

    vqae_handle* h = new vqae_handle();
    h->cfg = *cfg;
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return vqae::fail(VQAE_ERR_HIP, "hipGetDevice failed"); }
    h->fuse_trunk = !(getenv("VQAE_NO_TRUNK_FUSION") && atoi(getenv("VQAE_NO_TRUNK_FUSION")));
    h->up_conv_first = !(getenv("VQAE_NO_UP_REORDER") && atoi(getenv("VQAE_NO_UP_REORDER")));
    h->use_wino = !(getenv("VQAE_NO_WINOGRAD") && atoi(getenv("VQAE_NO_WINOGRAD")));
    h->fuse_up_tail = !(getenv("VQAE_NO_UP_TAIL_FUSION") && atoi(getenv("VQAE_NO_UP_TAIL_FUSION")));
    h->fuse_down = !(getenv("VQAE_NO_DOWN_FUSION") && atoi(getenv("VQAE_NO_DOWN_FUSION")));
    h->fuse_down16 = !(getenv("VQAE_NO_DOWN16") && atoi(getenv("VQAE_NO_DOWN16")));
    h->fuse_up16 = !(getenv("VQAE_NO_UP16") && atoi(getenv("VQAE_NO_UP16")));
    h->fuse_stem16 = !(getenv("VQAE_NO_STEM16") && atoi(getenv("VQAE_NO_STEM16")));
    h->fuse_vq = !(getenv("VQAE_NO_VQ_FUSION") && atoi(getenv("VQAE_NO_VQ_FUSION")));
    h->C = cfg->stem << cfg->n_down;
    h->D = cfg->projection_dim > 0 ? cfg->projection_dim : h->C;
    h->K = cfg->num_embeddings;
    int rc = VQAE_OK;
    const float* p;
    auto bail = [&](int code) { vqae_destroy(h); return code; };

    const bool has_enc = tm.count("encoder.in_stem.weight") > 0;
    const bool has_dec = tm.count("decoder.out_stem.weight") > 0;
    if (!has_enc && !has_dec) return bail(vqae::fail(VQAE_ERR_NOT_FOUND, "neither encoder.* nor decoder.* tensors given"));
    h->has_encoder = has_enc;
    h->has_decoder = has_dec;
    int c = cfg->stem << cfg->n_down;
    const std::string vq = "encoder.vq_layers.0.";

    if (has_enc) {
        if ((rc = find(tm, "encoder.in_stem.weight", (int64_t)cfg->stem * 3 * 9, &p)) || (rc = upload(h, p, (int64_t)cfg->stem * 27, &h->stem_w))) return bail(rc);
        if ((rc = find(tm, "encoder.in_stem.bias", cfg->stem, &p)) || (rc = upload(h, p, cfg->stem, &h->stem_b))) return bail(rc);
        if (cfg->compute_dtype != VQAE_DT_F32 && h->fuse_stem16 && cfg->in_channels == 3 && (cfg->stem == 8 || cfg->stem == 16 || cfg->stem == 32)) {
            if ((rc = dev_alloc(h, vqae::stem16_weight_bytes(3), &h->stem_wh))) return bail(rc);
            if ((rc = vqae::stem16_pack_weight(h->stem_w, cfg->stem, 3, cfg->compute_dtype, h->stem_wh, nullptr))) return bail(rc);
        }
        // encoder blocks: DownBlock levels (conv_block.py:35-47) then pre_enc (model.py:173-176)
        c = cfg->stem;
        for (int lvl = 0; lvl < cfg->n_down; ++lvl) {
            const std::string base = "encoder.down_layers.0.layers." + std::to_string(lvl) + ".layers.";
            int bi = 0;
            Block b;
            for (int i = 0; i < cfg->n_pre; ++i, ++bi) {
                if ((rc = load_any(h, tm, base + std::to_string(bi), MODE_SAME, c, c, &b))) return bail(rc);
                h->enc.push_back(b);
            }
            if ((rc = load_any(h, tm, base + std::to_string(bi), MODE_DOWN, c, 2 * c, &b))) return bail(rc);
            h->enc.push_back(b); ++bi;
            for (int i = 0; i < cfg->n_post; ++i, ++bi) {
                if ((rc = load_any(h, tm, base + std::to_string(bi), MODE_SAME, 2 * c, 2 * c, &b))) return bail(rc);
                h->enc.push_back(b);
            }
            c *= 2;
        }
        for (int i = 0; i < cfg->n_enc; ++i) {
            Block b;
            if ((rc = load_any(h, tm, "encoder.pre_enc_layers.0." + std::to_string(i), MODE_SAME, c, c, &b))) return bail(rc);
            h->enc.push_back(b);
        }
    }
    // VQ (codebook is required with an encoder, optional for decode-only handles)
    if (has_enc || tm.count(vq + "embed")) {
        if ((rc = find(tm, vq + "embed", (int64_t)h->K * h->D, &p)) || (rc = upload(h, p, (int64_t)h->K * h->D, &h->embed))) return bail(rc);
        if (cfg->projection_dim > 0) {
            if ((rc = find(tm, vq + "proj_in.weight", (int64_t)h->D * h->C, &p)) || (rc = upload_packed(h, p, h->D, h->C, 1, &h->pin_w))) return bail(rc);
            if ((rc = find(tm, vq + "proj_in.bias", h->D, &p)) || (rc = upload(h, p, h->D, &h->pin_b))) return bail(rc);
            if ((rc = vqae_round_inplace_f32(h->pin_b, h->D, cfg->compute_dtype, nullptr))) return bail(rc);
            if ((rc = find(tm, vq + "proj_out.weight", (int64_t)h->C * h->D, &p)) || (rc = upload_packed(h, p, h->C, h->D, 1, &h->pout_w))) return bail(rc);
            if ((rc = find(tm, vq + "proj_out.bias", h->C, &p)) || (rc = upload(h, p, h->C, &h->pout_b))) return bail(rc);
            if ((rc = vqae_round_inplace_f32(h->pout_b, h->C, cfg->compute_dtype, nullptr))) return bail(rc);
            if (h->D == 8) {                                   // operands of the fused kernel: plain [C][8] matrices, rounded like conv weights
                std::vector<float> wt((size_t)h->C * 8);
                const float* pi = tm.at(vq + "proj_in.weight")->data;      // [8][C]
                for (int j = 0; j < 8; ++j) for (int cc = 0; cc < h->C; ++cc) wt[(size_t)cc * 8 + j] = pi[(size_t)j * h->C + cc];
                if ((rc = upload(h, wt.data(), (int64_t)h->C * 8, &h->pin_wt)) ||
                    (rc = upload(h, tm.at(vq + "proj_out.weight")->data, (int64_t)h->C * 8, &h->pout_wr))) return bail(rc);
                if ((rc = vqae_round_inplace_f32(h->pin_wt, (int64_t)h->C * 8, cfg->compute_dtype, nullptr)) ||
                    (rc = vqae_round_inplace_f32(h->pout_wr, (int64_t)h->C * 8, cfg->compute_dtype, nullptr))) return bail(rc);
            }
        }
    }
    if (has_dec) {
        if ((rc = find(tm, "decoder.out_stem.weight", (int64_t)3 * cfg->stem * 9, &p)) || (rc = upload(h, p, (int64_t)cfg->stem * 27, &h->ostem_w))) return bail(rc);
        if ((rc = find(tm, "decoder.out_stem.bias", 3, &p)) || (rc = upload(h, p, 3, &h->ostem_b))) return bail(rc);
        if (cfg->compute_dtype != VQAE_DT_F32 && h->fuse_stem16 && cfg->in_channels == 3 && (cfg->stem == 8 || cfg->stem == 16 || cfg->stem == 32)) {
            if ((rc = dev_alloc(h, vqae::stem16_weight_bytes(cfg->stem), &h->ostem_wh))) return bail(rc);
            if ((rc = vqae::stem16_pack_weight(h->ostem_w, 3, cfg->stem, cfg->compute_dtype, h->ostem_wh, nullptr))) return bail(rc);
        }
        // decoder blocks: post_enc then UpBlock levels (conv_block.py:72-88)
        c = cfg->stem << cfg->n_down;
        for (int i = 0; i < cfg->n_enc; ++i) {
            Block b;
            if ((rc = load_any(h, tm, "decoder.post_enc_layers.0." + std::to_string(i), MODE_SAME, c, c, &b))) return bail(rc);
            h->dec.push_back(b);
        }
        for (int lvl = 0; lvl < cfg->n_down; ++lvl) {
            const std::string base = "decoder.up_layers.0.layers." + std::to_string(lvl) + ".layers.";
            int bi = 0;
            Block b;
            for (int i = 0; i < cfg->n_pre; ++i, ++bi) {
                if ((rc = load_any(h, tm, base + std::to_string(bi), MODE_SAME, c, c, &b))) return bail(rc);
                h->dec.push_back(b);
            }
            if ((rc = load_any(h, tm, base + std::to_string(bi), MODE_UP, c, c / 2, &b))) return bail(rc);
            h->dec.push_back(b); ++bi;
            for (int i = 0; i < cfg->n_post; ++i, ++bi) {
                if ((rc = load_any(h, tm, base + std::to_string(bi), MODE_SAME, c / 2, c / 2, &b))) return bail(rc);
                h->dec.push_back(b);
            }
            c /= 2;
        }
    }
    void* ls = nullptr;
    if ((rc = dev_alloc(h, 256, &ls))) return bail(rc);
    h->loss_scratch = (float*)ls;
    *out = h;
    return VQAE_OK;
}

extern "C" void vqae_destroy(vqae_handle* h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    for (void* p : h->owned) (void)hipFree(p);
    for (int i = 0; i < 4; ++i)
        if (h->buf[i]) (void)hipFree(h->buf[i]);
    if (h->vq_ws) (void)hipFree(h->vq_ws);
    if (h->idx_scratch) (void)hipFree(h->idx_scratch);
    if (h->se_ws) (void)hipFree(h->se_ws);
    delete h;
}

extern "C" int vqae_reserve(vqae_handle* h, int max_batch, int in_h, int in_w) {
    VQAE_REQUIRE(h, VQAE_ERR_INVALID, "null handle");
    int rc = check_geometry(h, max_batch, in_h, in_w);
    if (rc) return rc;
    return ensure_workspace(h, max_batch, in_h, in_w);
}

extern "C" int vqae_set_codebook(vqae_handle* h, const float* embed_host) {
    VQAE_REQUIRE(h && embed_host && h->embed, VQAE_ERR_INVALID, "set_codebook: null pointer / handle has no codebook");
    VQAE_HIP_CHECK(hipDeviceSynchronize());
    VQAE_HIP_CHECK(hipMemcpy(h->embed, embed_host, (size_t)h->K * h->D * 4, hipMemcpyHostToDevice));
    return VQAE_OK;
}

static int encode_impl(vqae_handle* h, const void* x, int x_kind, int B, int in_h, int in_w, void* idx, int idx_dtype,
                       float* q, int q_layout, float* loss, hipStream_t st) {
    if (h) g_dt = h->cfg.compute_dtype;
    VQAE_REQUIRE(h && x && idx, VQAE_ERR_INVALID, "vqae_encode: null pointer");
    VQAE_REQUIRE(h->has_encoder, VQAE_ERR_INVALID, "vqae_encode: handle was created without encoder.* tensors");
    int rc = check_geometry(h, B, in_h, in_w);
    if (rc) return rc;
    if (B == 0) return VQAE_OK;
    if ((rc = ensure_workspace(h, B, in_h, in_w))) return rc;
    int zh, zw;
    if ((rc = run_encoder_convs(h, x, x_kind, B, in_h, in_w, &zh, &zw, st))) return rc;
    if ((rc = run_vq(h, B, zh, zw, idx, idx_dtype, loss, st))) return rc;
    return export_q(h, B, zh, zw, q_layout, q, st);
}

extern "C" int vqae_encode(vqae_handle* h, const float* x, int B, int in_h, int in_w, int layout, void* idx,
                           int idx_dtype, float* q, float* loss, void* stream) {
    return encode_impl(h, x, layout == VQAE_LAYOUT_NCHW ? 1 : 0, B, in_h, in_w, idx, idx_dtype, q, layout, loss,
                       (hipStream_t)stream);
}

extern "C" int vqae_encode_u8(vqae_handle* h, const uint8_t* x, int B, int in_h, int in_w, void* idx, int idx_dtype,
                              float* q, int q_layout, float* loss, void* stream) {
    return encode_impl(h, x, 2, B, in_h, in_w, idx, idx_dtype, q, q_layout, loss, (hipStream_t)stream);
}

extern "C" int vqae_encode_features(vqae_handle* h, const float* x, int B, int in_h, int in_w, int layout, float* z,
                                    void* stream) {
    if (h) g_dt = h->cfg.compute_dtype;
    hipStream_t st = (hipStream_t)stream;
    VQAE_REQUIRE(h && x && z, VQAE_ERR_INVALID, "vqae_encode_features: null pointer");
    VQAE_REQUIRE(h->has_encoder, VQAE_ERR_INVALID, "vqae_encode_features: handle has no encoder");
    int rc = check_geometry(h, B, in_h, in_w);
    if (rc) return rc;
    if (B == 0) return VQAE_OK;
    if ((rc = ensure_workspace(h, B, in_h, in_w))) return rc;
    int zh, zw;
    if ((rc = run_encoder_convs(h, x, layout == VQAE_LAYOUT_NCHW ? 1 : 0, B, in_h, in_w, &zh, &zw, st))) return rc;
    if (h->cfg.projection_dim > 0) {
        ConvCall pin(B, zh, zw, h->C, h->D, 1, 1, 0, VQAE_PAD_NONE);
        return vqae_conv2d_f32(&pin.a, h->buf[0], h->pin_w, h->pin_b, nullptr, z, st);
    }
    VQAE_HIP_CHECK(hipMemcpyAsync(z, h->buf[0], (size_t)B * zh * zw * h->C * 4, hipMemcpyDeviceToDevice, st));
    return VQAE_OK;
}

extern "C" int vqae_decode(vqae_handle* h, const float* q, int B, int qh, int qw, int layout, float* out, void* stream) {
    if (h) g_dt = h->cfg.compute_dtype;
    hipStream_t st = (hipStream_t)stream;
    VQAE_REQUIRE(h && q && out, VQAE_ERR_INVALID, "vqae_decode: null pointer");
    VQAE_REQUIRE(h->has_decoder, VQAE_ERR_INVALID, "vqae_decode: handle was created without decoder.* tensors");
    VQAE_REQUIRE(B >= 0 && qh >= 1 && qw >= 1, VQAE_ERR_INVALID, "vqae_decode: bad shape");
    if (int rcd = check_device(h)) return rcd;
    if (B == 0) return VQAE_OK;
    int rc;
    if ((rc = ensure_workspace(h, B, qh << h->cfg.n_down, qw << h->cfg.n_down))) return rc;
    if (layout == VQAE_LAYOUT_NCHW) {
        if ((rc = vqae_nchw_to_nhwc_f32(q, B, h->C, qh, qw, h->buf[0], st))) return rc;
    } else {
        VQAE_HIP_CHECK(hipMemcpyAsync(h->buf[0], q, (size_t)B * qh * qw * h->C * 4, hipMemcpyDeviceToDevice, st));
    }
    return run_decoder_convs(h, B, qh, qw, layout, out, st);
}

extern "C" int vqae_decode_indices(vqae_handle* h, const void* idx, int idx_dtype, int B, int qh, int qw, int layout,
                                   float* out, void* stream) {
    if (h) g_dt = h->cfg.compute_dtype;
    hipStream_t st = (hipStream_t)stream;
    VQAE_REQUIRE(h && idx && out, VQAE_ERR_INVALID, "vqae_decode_indices: null pointer");
    VQAE_REQUIRE(h->has_decoder && h->embed, VQAE_ERR_INVALID, "vqae_decode_indices: handle needs decoder.* tensors and a codebook");
    VQAE_REQUIRE(B >= 0 && qh >= 1 && qw >= 1, VQAE_ERR_INVALID, "vqae_decode_indices: bad shape");
    if (int rcd = check_device(h)) return rcd;
    if (B == 0) return VQAE_OK;
    int rc;
    if ((rc = ensure_workspace(h, B, qh << h->cfg.n_down, qw << h->cfg.n_down))) return rc;
    const int64_t rows = (int64_t)B * qh * qw;
    if (h->cfg.projection_dim > 0) {
        if ((rc = vqae_embed_code_f32(idx, idx_dtype, h->embed, rows, h->K, h->D, h->buf[1], st))) return rc;
        ConvCall pout(B, qh, qw, h->D, h->C, 1, 1, 0, VQAE_PAD_NONE);
        if ((rc = vqae_conv2d_f32(&pout.a, h->buf[1], h->pout_w, h->pout_b, nullptr, h->buf[0], st))) return rc;
    } else {
        if ((rc = vqae_embed_code_f32(idx, idx_dtype, h->embed, rows, h->K, h->D, h->buf[0], st))) return rc;
    }
    return run_decoder_convs(h, B, qh, qw, layout, out, st);
}

extern "C" int vqae_block_count(const vqae_handle* h, int side) {
    if (!h) return 0;
    return (int)(side == 0 ? h->enc.size() : h->dec.size());
}

extern "C" int vqae_run_blocks(vqae_handle* h, int side, int first, int count, const float* x, int B, int in_h, int in_w,
                               float* y, int* out_h, int* out_w, void* stream) {
    if (h) g_dt = h->cfg.compute_dtype;
    hipStream_t st = (hipStream_t)stream;
    VQAE_REQUIRE(h && x && y, VQAE_ERR_INVALID, "vqae_run_blocks: null pointer");
    VQAE_REQUIRE(side == 0 || side == 1, VQAE_ERR_INVALID, "vqae_run_blocks: side %d", side);
    std::vector<Block>& v = side == 0 ? h->enc : h->dec;
    VQAE_REQUIRE(first >= 0 && count >= 1 && (size_t)first + (size_t)count <= v.size(), VQAE_ERR_INVALID,
                 "vqae_run_blocks: blocks [%d, %d) of %zu", first, first + count, v.size());
    VQAE_REQUIRE(B >= 0 && in_h >= 1 && in_w >= 1, VQAE_ERR_INVALID, "vqae_run_blocks: bad shape");
    if (int rcd = check_device(h)) return rcd;
    int H = in_h, W = in_w;
    if (B == 0) return VQAE_OK;
    // workspace: the widest tensor any block of the range touches (an 'up' block's upsampled 2C tensor is 4x its input)
    size_t need = 0;
    {
        double hh = in_h, ww = in_w;
        for (int i = first; i < first + count; ++i) {
            const Block& b = v[i];
            VQAE_REQUIRE(b.mode != MODE_DOWN || (((int)hh % 2 == 0) && ((int)ww % 2 == 0)), VQAE_ERR_INVALID,
                         "vqae_run_blocks: odd input size at a 'down' block");
            const size_t wide = (size_t)std::max(std::max(b.cin, b.cout), b.br) * (b.mode == MODE_UP ? 4 : 1);
            need = std::max(need, (size_t)B * (size_t)hh * (size_t)ww * wide);
            if (b.mode == MODE_DOWN) { hh /= 2; ww /= 2; } else if (b.mode == MODE_UP) { hh *= 2; ww *= 2; }
        }
    }
    int rc;
    if ((rc = ensure_bufs(h, need))) return rc;
    if (h->cfg.block_kind == VQAE_BLOCK_MBCONV)
        return vqae::fail(VQAE_ERR_UNSUPPORTED, "vqae_run_blocks: Fixup blocks only");
    VQAE_HIP_CHECK(hipMemcpyAsync(h->buf[0], x, (size_t)B * in_h * in_w * v[first].cin * 4, hipMemcpyDeviceToDevice, st));
    h->t1_ready = false;
    for (int i = first; i < first + count; ++i)
        if ((rc = run_block(h, v[i], i + 1 < first + count ? &v[i + 1] : nullptr, B, H, W, st))) return rc;
    h->t1_ready = false;
    VQAE_HIP_CHECK(hipMemcpyAsync(y, h->buf[0], (size_t)B * H * W * v[first + count - 1].cout * 4, hipMemcpyDeviceToDevice, st));
    if (out_h) *out_h = H;
    if (out_w) *out_w = W;
    return VQAE_OK;
}

extern "C" int vqae_forward(vqae_handle* h, const float* x, int B, int in_h, int in_w, int layout, float* out, void* idx,
                            int idx_dtype, float* loss, void* stream) {
    if (h) g_dt = h->cfg.compute_dtype;
    hipStream_t st = (hipStream_t)stream;
    VQAE_REQUIRE(h && x && out, VQAE_ERR_INVALID, "vqae_forward: null pointer");
    VQAE_REQUIRE(h->has_encoder && h->has_decoder, VQAE_ERR_INVALID, "vqae_forward: handle needs encoder.* and decoder.* tensors");
    int rc = check_geometry(h, B, in_h, in_w);
    if (rc) return rc;
    if (B == 0) return VQAE_OK;
    if ((rc = ensure_workspace(h, B, in_h, in_w))) return rc;
    int zh, zw;
    if ((rc = run_encoder_convs(h, x, layout == VQAE_LAYOUT_NCHW ? 1 : 0, B, in_h, in_w, &zh, &zw, st))) return rc;
    // idx is optional here
    void* idx_out = idx;
    int dt = idx_dtype;
    if (!idx_out) { idx_out = h->idx_scratch; dt = VQAE_IDX_I32; }
    if ((rc = run_vq(h, B, zh, zw, idx_out, dt, loss ? loss : h->loss_scratch, st))) return rc;
    return run_decoder_convs(h, B, zh, zw, layout, out, st);
}

extern "C" double vqae_flops_per_patch(const vqae_handle* h, int in_h, int in_w, int encoder, int decoder) {
    if (!h) return 0.0;
    double fl = 0.0;
    auto blocks = [&](const std::vector<Block>& v, double H, double W) {
        for (const Block& b : v) {
            if (b.kind == VQAE_BLOCK_MBCONV) {                                               // 1x1 expand, depthwise, 1x1 project, skip
                const double k2 = b.mode == MODE_SAME ? 9.0 : 4.0;
                fl += 2.0 * H * W * (double)b.cin * b.br;
                if (b.mode == MODE_DOWN) {
                    fl += 2.0 * (H / 2) * (W / 2) * (k2 * b.br + (double)b.br * b.cout + 4.0 * b.cin * b.cout);
                    H /= 2; W /= 2;
                } else if (b.mode == MODE_UP) {
                    fl += 2.0 * H * W * 4.0 * b.cin * b.cout;
                    H *= 2; W *= 2;
                    fl += 2.0 * H * W * ((double)b.br + (double)b.br * b.cout);
                } else {
                    fl += 2.0 * H * W * (k2 * b.br + (double)b.br * b.cout + (b.wskip ? (double)b.cin * b.cout : 0.0));
                }
                continue;
            }
            if (b.mode == MODE_SAME) {
                fl += 2.0 * H * W * ((double)b.cin * b.br + 9.0 * b.br * b.br + (double)b.br * b.cout);
            } else if (b.mode == MODE_DOWN) {
                fl += 2.0 * H * W * (double)b.cin * b.br;                                   // conv1 @ HxW
                H /= 2; W /= 2;
                fl += 2.0 * H * W * (4.0 * b.br * b.br + (double)b.br * b.cout + 4.0 * b.cin * b.cout);
            } else {
                fl += 2.0 * H * W * (double)b.cin * b.br;                                   // conv1 @ low res
                H *= 2; W *= 2;
                fl += 2.0 * H * W * ((double)b.br * b.br + (double)b.br * b.cout + (double)b.cin * b.cout);
            }
        }
        return std::pair<double, double>(H, W);
    };
    if (encoder) {
        fl += 2.0 * in_h * in_w * 27.0 * h->cfg.stem;
        auto hw = blocks(h->enc, in_h, in_w);
        if (h->cfg.projection_dim > 0) fl += 2.0 * hw.first * hw.second * 2.0 * h->C * h->D;
    }
    if (decoder) {
        blocks(h->dec, in_h >> h->cfg.n_down, in_w >> h->cfg.n_down);
        fl += 2.0 * in_h * in_w * 27.0 * h->cfg.stem;
    }
    return fl;
}
