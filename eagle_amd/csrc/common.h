// Internal declarations shared by the runtime (graph/weights/api) and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/eagle.h"

namespace eagle {

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

struct Err {
    int code;
    std::string msg;
};
// Thrown inside the library only; every extern "C" entry catches it and converts to a status code.
[[noreturn]] void fail(int code, const char* fmt, ...);

#define HIP_CHECK(expr)                                                                                \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) ::eagle::fail(EAGLE_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                            __FILE__, __LINE__);                                       \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): function attributes are per device, and handles on
// different devices / threads may launch the same kernel concurrently.
void ensure_max_dynamic_lds(const void* fn, int bytes);

// A view of an NHWC activation: C channels starting at channel `off` of a buffer whose pixel stride is `cs`.
struct TView {
    void* p = nullptr;   // device base pointer of the underlying buffer
    int n = 0, h = 0, w = 0;
    int c = 0;           // channels of this view (padded to the kernel granularity)
    int cs = 0;          // channel stride (elements per pixel) of the underlying buffer
    int off = 0;         // first channel of the view
    int f32 = 0;         // element type: 0 fp16, 1 fp32, 2 split fp32 (EAGLE_PREC_F32S: per 8 channels 32 bytes, [hi x 8][lo x 8] binary16 with
                         // hi = rn(16 v), lo = rn(16 v - hi); c / cs / off stay in logical channels, multiples of 8)
    size_t esize() const { return f32 ? 4 : 2; }
    TView slice(int o, int cc) const { TView v = *this; v.off = off + o; v.c = cc; return v; }
};

// ---- convolution ------------------------------------------------------------------------------------------
struct ConvConfig {          // tile configuration chosen per layer at graph-build time
    int ks = 3, stride = 1;
    int kc = 0;              // input channels staged per LDS chunk
    int nt = 0;              // 16-wide Cout tiles per workgroup (BN = 16*nt)
    int wx = 2;              // 16-pixel sub-tiles per tile row; tile = (16/wx) rows x (16*wx) cols, 4 per wave
    int cin = 0, cout_pad = 0;
    int variant = 0;         // fp16 only: 0 = register-staged, 1 = persistent LDS-DMA pipeline
};
struct ArgmaxPart { float score; int idx; };
struct ConvLaunch {
    ConvConfig cfg;
    TView x, y, r1, r2;      // r1/r2 optional (p == nullptr)
    const void* w = nullptr; // pre-tiled weights for cfg
    const float* bias = nullptr;
    int pre_act = 0, post_act = 0;
    int out_f32 = 0;         // store fp32 regardless of precision (logits)
    double flop = 0;         // algorithmic 2*MAC (unpadded)
    float descale = 1.0f;    // EAGLE_PREC_F32S: returned by conv_tile_weights for this layer's weight image
    // fp16 family, head convolution: instead of storing the logits, every workgroup reduces sigmoid(logit) of its tile to one
    // (maximum, first index) per channel and writes it to (*am_slot)[(frame * tiles + tile) * cout_pad + channel] (K5 fused into the
    // producer: the fp32 logit tensor never goes to HBM).  The slot is read when the launch is enqueued.
    ArgmaxPart* const* am_slot = nullptr;
    // EAGLE_PREC_F32S: where the step being enqueued counts saturated stores per frame (EagleHandle::cur_sat; read when the launch is enqueued)
    unsigned* const* sat_slot = nullptr;
};
// output tiles per frame of a convolution with this configuration (the number of partials per channel the fused arg-max writes)
int conv_tiles_per_frame(const ConvConfig& cfg, int ho, int wo);
// returns false when no kernel instance exists for cfg
bool conv_supported(int precision, const ConvConfig& cfg);
void conv_launch(int precision, const ConvLaunch& L, hipStream_t s);
size_t conv_weight_elems(int precision, const ConvConfig& cfg);
size_t conv_lds_bytes(int precision, const ConvConfig& cfg);
// w_hwio: folded fp32 weights [ks][ks][cin_real][cout_real]; dst: host buffer of conv_weight_elems elements
// EAGLE_PREC_F32S: the weights are scaled by a power of two (largest magnitude into [2^14, 2^15)) and split into hi / lo binary16 parts;
// *descale receives 2^-(that exponent + 4) (the 4: the activations' own scaling), which the kernel applies to the accumulator
void conv_tile_weights(int precision, const ConvConfig& cfg, const float* w_hwio, int cin_real, int cout_real, void* dst, float* descale = nullptr);
inline int prec_tensor_fmt(int precision) { return precision == EAGLE_PREC_F32 ? 1 : precision == EAGLE_PREC_F32S ? 2 : 0; }
inline bool prec_is_f16_kernels(int precision) { return precision == EAGLE_PREC_F16 || precision == EAGLE_PREC_F32S; }
// plain_epilogue: pre_act none, post_act none/ReLU, at most one residual, fp16 output (what the weight-stationary kernel implements)
ConvConfig conv_choose(int precision, int ks, int stride, int cin_pad, int cout_pad, int wo, bool plain_epilogue = false, bool second_residual = false, bool any_residual = true);

// ---- fused Bottleneck (bneck.hip; EAGLE_PREC_F32S): conv1 1x1 Cin->64 + ReLU, conv2 3x3 64->64 + ReLU, conv3 1x1 64->256 + residual + ReLU in one launch ----
struct BneckLaunch {
    TView x, res, y;         // res: the 256-channel residual operand (x itself for an identity shortcut)
    const void *w1 = nullptr, *w2 = nullptr, *w3 = nullptr;      // fragment-major weight images (bneck_tile_weights)
    const float *b1 = nullptr, *b2 = nullptr, *b3 = nullptr;     // folded-BN biases DIVIDED by the layer's descale (bneck_scale_bias): the accumulators' initial values
    float ds1 = 1.f, ds2 = 1.f, ds3 = 1.f;
    unsigned* const* sat_slot = nullptr;
    unsigned long long* dbg = nullptr;       // developer timing builds only
    bool ds_fused = false;                   // Cin = 64: w3 is the K = 128 image [W3 | downsample weights] (one scale), b3 = (b3 + bd) / ds3, `res` is ignored (pass y)
};
bool bneck_supported(const TView& x, int cmid, int cout);
// w: folded fp32 weights [taps][cin][cout] (taps = 1 or 9)
void bneck_tile_weights(const float* w, int taps, int cin, int cout, std::vector<_Float16>& out, float* descale);
void bneck_scale_bias(std::vector<float>& b, float descale);      // b /= descale (exact: a power of two): what BneckLaunch::b1 / b2 / b3 must hold
void bneck_launch(const BneckLaunch& L, hipStream_t s);

// ---- other kernels ----------------------------------------------------------------------------------------
struct LetterBox { int new_h, new_w, top, left, out_h, out_w; };
LetterBox letterbox_geometry(int h, int w, int imgsz, int square = 0);      // square: EagleConfig::letterbox (auto=False)
// which: bit 0 = write the key-point tensor, bit 1 = write the detector tensor; det_precision >= 0: the detector tensor's family when it differs from `precision`
void preprocess_launch(int precision, const uint8_t* d_bgr, int n, int h, int w, const TView& kp, const TView& det,
                       const LetterBox& lb, hipStream_t s, int which = 3, int det_precision = -1);
struct FuseUp { TView z; };
void fuse_sum_launch(const TView& base, const FuseUp* ups, int n_up, int relu, const TView& y, hipStream_t s, unsigned* sat = nullptr);   // sat: as ConvArgs::sat
void maxpool5_launch(const TView& x, const TView& y, hipStream_t s);
void upsample2_launch(const TView& x, const TView& y, hipStream_t s);
void split_to_f32_launch(const TView& x, const TView& y, hipStream_t s);      // split fp32 -> fp32, exact (the mixed detector's seam)

// logits: fp32 view [n,h,w,64]; parts: [n][chunks][64]
void heat_argmax_launch(const TView& logits, ArgmaxPart* parts, int chunks, hipStream_t s);

struct DetLevel { TView box, cls; int gh, gw; float stride; int a0; };
struct DetScratch {           // per-handle device scratch for decode + NMS, sized for `batch` frames
    int A = 0;                // anchors per frame
    float* boxes = nullptr;   // [n][A][4] xyxy (net-input pixels)
    float* conf = nullptr;    // [n][A]
    int* cls = nullptr;       // [n][A]
    unsigned long long* keys = nullptr;  // [n][A] candidate sort keys
    int* count = nullptr;     // [n]
};
void yolo_decode_launch(const DetLevel* lv, int n_lv, int n, int nc, float conf_floor, const DetScratch& sc, hipStream_t s);

struct PostParams {
    int frame_h, frame_w, in_h, in_w;        // frame size, detector input size
    int hm_h, hm_w, hm_chunks;
    double keypoint_conf, detector_conf, ransac_thresh;
    float nms_iou;
    int ransac_max_iters, lm_iters;
};
void nms_launch(const DetScratch& sc, int n, const PostParams& pp, EagleFrameResult* d_out, hipStream_t s);
void post_launch(const ArgmaxPart* parts, int n, const PostParams& pp, EagleFrameResult* d_out, hipStream_t s);
void reproject_launch(EagleFrameResult* d_recs, const double* d_Hs, const unsigned char* d_flags, int n, int frame_h, int frame_w, hipStream_t s);
void homography_only_launch(const float* d_img, const float* d_world, int npts, double thresh, int max_iters, int lm_iters,
                            double* d_H, uint8_t* d_mask, int* d_ok, hipStream_t s);


// ---- optical-flow key-point cadence (cm.py:188-331, 419-478, 520-555) ---------------------------
// A clip resident in HBM: the caller's BGR frames plus the gray pyramid of every frame (flow.hip, K11).
struct ClipView {
    const uint8_t* bgr = nullptr;
    const uint8_t* g[3] = {nullptr, nullptr, nullptr};
    int lh[3] = {0, 0, 0}, lw[3] = {0, 0, 0};
    int levels = 0;          // highest pyramid level in use (cv2 maxLevel 2, fewer for tiny frames)
    int n = 0, h = 0, w = 0;
};
// mem[i] of the reference loop (cm.py:211): the key-points a model detection (or the first-frame search) produced for frame i,
// in dict order.  n = -1: no entry.
struct MemList { int n; EagleFlowKp kp[EAGLE_N_LANDMARKS]; };
// The loop-carried state of cm.py:207-213 plus the scratch of the current optical-flow step.
struct ChainState {
    int n_prev; EagleFlowKp prev[EAGLE_N_LANDMARKS];       // prev_keypoints, dict order
    int has_H, compute_h; double H[9];                      // homography_matrix, compute_homography
    int stalled;                                            // frame waiting for an on-demand detection (cm.py:317), or -1
    int error;                                              // 1 + frame at which the reference raises (calibration, cm.py:545)
    int lk_valid, lk_n;                                     // this step ran a flow; number of points it tracked
    float lk_prev[2 * EAGLE_N_LANDMARKS], lk_next[2 * EAGLE_N_LANDMARKS];
    unsigned char lk_status[EAGLE_N_LANDMARKS];
    int flow_n; EagleFlowKp flow[EAGLE_N_LANDMARKS];       // operator mode: the filtered dict calculate_optical_flow returns
};
void gray_pyramid_launch(const uint8_t* d_bgr, int n, int h, int w, uint8_t* g0, uint8_t* g1, uint8_t* g2, hipStream_t s);
void lk_launch(const ClipView& cv, int src_frame, int dst_frame, ChainState* st, const MemList* mem, int kint, hipStream_t s);
// ---- ECC camera motion (ecc.hip, K17) ---------------------------------------------------------------------------------------------
struct EccResult { int ok, iters; double rho; float M[6]; };       // M: 2 x 3 warp in the 0.15-scale image's pixels (template -> image)
void ecc_small_launch(const uint8_t* gray, uint8_t* small, int n, int h, int w, int dh, int dw, double inv_scale /* 1 / fx */, hipStream_t s);
// pairs[k] = (template frame or -1 = `carry`, image frame), indices into `small` ([n, h, w] u8)
void ecc_launch(const uint8_t* small, const uint8_t* carry, const int2* pairs, int n_pairs, EccResult* out, int h, int w, int max_iter, double eps, hipStream_t s);
// ---- team colours (teams.hip, K15) ----------------------------------------------------------------------------------------------
void team_colors_launch(const uint8_t* d_bgr, int n_frames, int fh, int fw, const EagleCrop* d_crops, int n_crops, int* d_counts, hipStream_t s);

// ---- appearance embeddings for the tracker (reid.hip, K16): OSNet-x0.25 pieces that are not 1 x 1 convolutions -------------------------------
void reid_crop_launch(const uint8_t* d_bgr, int n_frames, int fh, int fw, const EagleCrop* d_crops, int n, const TView& out, hipStream_t s);
void reid_conv7_launch(const TView& x, const float* w, const float* b, const TView& y, int n, hipStream_t s);
void reid_maxpool3s2_launch(const TView& x, const TView& y, int n, hipStream_t s);
void reid_avgpool2_launch(const TView& x, const TView& y, int n, hipStream_t s);
void reid_dw3_launch(const TView& x, const float* w, const float* b, const TView& y, int n, hipStream_t s);
void reid_gate_launch(const TView* streams, const float* w1, const float* b1, const float* w2, const float* b2, int c_real, int r, float* g, const TView& y, int n, hipStream_t s);
void reid_head_launch(const TView& x, const float* w, const float* b, float* feats, int dim, int n, hipStream_t s);

// ---- track identities (tracker.hip; host side of the library) ----------------------------------------------------------------
struct Tracker;
Tracker* tracker_create(const EagleTrackParams* p);
void tracker_destroy(Tracker* t);
// feats / feat_det / n_feat: optional appearance embeddings (EAGLE_REID_DIM floats each) of the detections feat_det[0..n_feat) of this record
bool tracker_apply(Tracker* t, EagleFrameResult* rec, int frame_h, int frame_w, double detector_conf, const double* warp = nullptr,
                   const float* feats = nullptr, const int* feat_det = nullptr, int n_feat = 0);   // true: the record's persons are now keyed by track id; warp: 2 x 3 camera motion previous frame -> this one
void similarity_ransac(const double* p0, const double* p1, int n, double* W6);       // camera motion from matched points (tracker.hip)
int lk_debug(const char* key, long long value, void* out, long long out_bytes);   // developer diagnostics behind eagle_debug
// heat-map maxima of `n` frames -> mem[first + k*stride] (threshold / pixel mapping / dedup, cm.py:231-251, 500-518)
void decode_mem_launch(const ArgmaxPart* parts, int n, const PostParams& pp, MemList* mem, int first, int stride, hipStream_t s);
// one iteration of the reference loop body for frame `frame` in any cadence (runs after lk_launch of the same frame)
void chain_launch(const ClipView& cv, ChainState* st, const MemList* mem, EagleFrameResult* recs, const PostParams& pp, int frame,
                  int kint, int hint, int calib, hipStream_t s);
// calculate_optical_flow as an operator: filters the flow of lk_launch(src, dst) with hue grids of frame `hue_frame` into st->flow
void flow_filter_launch(const ClipView& cv, ChainState* st, int hue_frame, hipStream_t s);

}  // namespace eagle
