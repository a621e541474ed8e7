/*
 * eagle.h — C ABI of the MI355X-native per-frame inference path for nreHieW/Eagle.
 *
 * The reference has no plugin / FFI interface: the path is reached through three Python call
 * sites of eagle/models/coordinate_model.py, which are the seams this library replaces:
 *   (1) detector      self.detector_model(frame, verbose=False, conf=low_conf)[0].boxes   cm.py:568-572
 *   (2) keypoints     self.keypoint_model.get_keypoints(batch)                            cm.py:226,253,492
 *                                                                                         kh.py:575-595
 *   (3) homography    cv2.findHomography / cv2.perspectiveTransform                       cm.py:355,383,400-403
 * and the per-frame loop body that strings them together (cm.py:277-415) in the stateless configuration
 * (keypoint_interval = homography_interval = 1, tracker off: IDs = detection index, cm.py:598-627).
 *
 * Conventions: every function returns 0 on success or a negative EAGLE_E_* code and never throws; the caller
 * owns the `bgr` and `out` host buffers; the library owns all device memory and streams; one handle per
 * (process, GPU); a handle is not thread-safe; different handles are independent.
 * Plain pointers and sizes only — no torch / numpy types cross this boundary.
 */
#ifndef EAGLE_H
#define EAGLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EAGLE_OK 0
#define EAGLE_E_INVALID (-1)   /* bad argument */
#define EAGLE_E_HIP (-2)       /* HIP runtime error (see eagle_last_error) */
#define EAGLE_E_STATE (-3)     /* call order: weights not finalized, etc. */
#define EAGLE_E_MISSING (-4)   /* a required weight tensor was never loaded */
#define EAGLE_E_NOKERNEL (-5)  /* no kernel instance for a layer shape */
#define EAGLE_E_COMM (-6)      /* RCCL error */
#define EAGLE_E_RANGE (-8)     /* EAGLE_PREC_F32S: activations left the tensor format's range (|v| > 4094) and were clipped.  The records ARE written
                                  (EagleFrameResult.pad[1] = 1 marks the frames) but are not fp32-grade; see EagleConfig.allow_saturation */

#define EAGLE_MAX_DET 300      /* ultralytics max_det */
#define EAGLE_N_LANDMARKS 57   /* eagle/utils/pitch.py:1-60 */
#define EAGLE_MAX_KP 87        /* 57 detected + at most 30 synthesised (cm.py:140) */

#define EAGLE_PREC_F16 0       /* fp16 tensors, fp32 accumulate on v_mfma_f32_16x16x32_f16 (fast path) */
#define EAGLE_PREC_F32 1       /* fp32 tensors on v_mfma_f32_16x16x4_f32: bit-exact vs the oracle's fmaf chain */
#define EAGLE_PREC_F32S 2      /* fp32-grade: every tensor value as a (hi, lo) pair of binary16 numbers (4 bytes, 22+ significant bits), products as
                                  three v_mfma_f32_16x16x32_f16 (hi*hi + hi*lo + lo*hi) into an fp32 accumulator: the error of an fp32 summation in
                                  another order, at 3/16 of the fp32 MFMA's cost per product */

#define EAGLE_AUTO (-1)          /* EagleConfig::use_graph / multi_stream: chosen from `batch` by eagle_create (eagle_resolve_config) */
#define EAGLE_SMALL_BATCH 32     /* steps of at most this many frames replay their network phase as a hipGraph unless the caller says otherwise (8 until the middle of round 6:
                                    with the branch streams' schedule the replay now pays in one-step calls up to 50 frames, +11 % at 8, +5 % at 16, +3 % at 25, +2 % at 32) */
/* (EAGLE_MULTI_STREAM_BATCH, rounds 5 - 6a: branch streams only up to 16 frames per step.  Since the fuse outputs run on parallel streams they pay at every batch: "auto" = on) */
#define EAGLE_DET_PREC_AUTO (-1) /* EagleConfig::det_precision: chosen from `precision` by eagle_create */
#define EAGLE_DET_PREC_MIXED 4   /* EagleConfig::det_precision: the detector's trunk in EAGLE_PREC_F32S; the last C2f of every level (model.15 / 18 / 21), model.16 / 19 and
                                    Detect in EAGLE_PREC_F32.  Measured and NOT the default: it does not keep the exact family's detection ids */

#define EAGLE_LETTERBOX_RECT 0
#define EAGLE_LETTERBOX_SQUARE 1

#define EAGLE_DET_N 0
#define EAGLE_DET_S 1
#define EAGLE_DET_M 2
#define EAGLE_DET_L 3
#define EAGLE_DET_X 4

typedef struct EagleHandle EagleHandle;

/* Replaces the constructor arguments of CoordinateModel (cm.py:49-74) and the constants of cm.py:18-20,567. */
typedef struct EagleConfig {
    int32_t device;            /* HIP device ordinal */
    int32_t frame_h, frame_w;  /* e.g. 720, 1280 */
    int32_t det_variant;       /* EAGLE_DET_* */
    int32_t det_imgsz;         /* 640 (detector_medium/large) or 960 (detector_large_hd), README.md:107-111 */
    int32_t batch;             /* frames processed per device step (>= 1) */
    int32_t precision;         /* EAGLE_PREC_* */
    double keypoint_conf;      /* 0.3   cm.py:49  (double: the reference compares Python floats) */
    double detector_conf;      /* 0.35  cm.py:49 */
    double ransac_thresh;      /* 5.0   cm.py:355 */
    float detector_floor;      /* 0.15  cm.py:567 (compared in fp32 by ultralytics) */
    float nms_iou;             /* 0.7   ultralytics default */
    int32_t ransac_max_iters;  /* 2000  cv2 default */
    int32_t lm_iters;          /* 10    cv2 default */
    int32_t use_graph;         /* 1: capture the per-batch step into a hipGraph and replay it; 0: plain launches; 2: replay only inside calls of at least three
                                  steps (the graph launch of a step then hides behind the previous step on the GPU; a one-step call of a large batch would pay it
                                  up front; a capture costs ~80 ms per (slot, frame count), once); EAGLE_AUTO (default): 1 when batch <= EAGLE_SMALL_BATCH (the per-frame
                                  use of the reference's loop, cm.py:277), else 0 */
    int32_t det_precision;     /* 0: the detector runs in `precision`; EAGLE_PREC_* + 1: that family for the detector alone; EAGLE_DET_PREC_AUTO (what
                                  eagle_default_config sets; resolved by eagle_create): EAGLE_PREC_F32 + 1 when `precision` is EAGLE_PREC_F32S — key-points
                                  in the split family, the detector (1.4 % of the FLOP with yolov8n) in the exact fp32 family, so that boxes, confidences,
                                  classes, NMS order and detection-index ids are the fp32 oracle's bit for bit — and 0 for any other `precision` (a caller
                                  that takes the defaults and only sets precision = EAGLE_PREC_F16 gets BOTH networks in the fast family) */
    int32_t allow_saturation;  /* EAGLE_PREC_F32S: 0 (default): a call in which an activation store was clipped at +-4094 returns EAGLE_E_RANGE;
                                  1: it returns EAGLE_OK and only flags the frames (EagleFrameResult.pad[1]) and counts them (EagleTimings) */
    int32_t multi_stream;      /* 1: HRNet's branches — and, behind their join, the outputs of its fuse layers — on their own HIP streams inside a step;
                                  0: one stream per network; EAGLE_AUTO (default): 1 (every batch since round 6: +13 % at 12 - 16 frames per step, +1.6 %
                                  at 50; EAGLE_MULTI_STREAM=0 in the environment resolves "auto" to 0) */
    int32_t letterbox;         /* detector input geometry (ultralytics LetterBox): EAGLE_LETTERBOX_RECT (0, default) = auto=True, what the .pt predictor
                                  of cm.py:56-57 does — pad only to the next multiple of 32 (1280x720 @640 -> 384 x 640, 5040 anchors);
                                  EAGLE_LETTERBOX_SQUARE (1) = auto=False, what the exported ONNX detector of the reference's CPU default runs with (cm.py:54-55,
                                  detector_medium.onnx: a static det_imgsz x det_imgsz input — 1280x720 @640 -> 640 x 640, rows 140 / 140 of grey padding, 8400 anchors) */
    int32_t reserved[3];
} EagleConfig;

typedef struct EagleDet {      /* one row of boxes.xyxy/.conf/.cls after NMS (cm.py:569-572) + cm.py:598-627 */
    float x1, y1, x2, y2;      /* float boxes in frame pixels, descending-confidence order */
    float conf;
    int32_t cls;               /* 0 Player 1 Goalkeeper 2 Ball 3 Referee 4 Staff (cm.py:61) */
    int32_t id;                /* Player/Goalkeeper: detection index; Ball: enumerate index; -1 otherwise */
    int32_t bx1, by1, bx2, by2;/* "BBox" ints (truncated, clipped for persons) */
    int32_t foot_x, foot_y;    /* "Bottom_center" */
    float pitch_xf, pitch_yf;  /* perspectiveTransform output before .astype(int) */
    int32_t pitch_x, pitch_y;  /* "Transformed_Coordinates" (valid iff in_bounds) */
    uint8_t reported;          /* 1: appears in the reference dict (class kept and conf >= detector_conf) */
    uint8_t in_bounds;         /* 1: H valid and 0<=X<=105, 0<=Y<=68 */
    uint8_t pad[2];
} EagleDet;

typedef struct EagleKeypoint {
    int32_t label;             /* heat-map index 0..56 (label string via eagle/utils/pitch.py:1-60) */
    int32_t x, y;              /* image pixels */
    float score;               /* sigmoid maximum; 0 for synthesised points */
    uint8_t synthesized;       /* 1: added by the line-intersection synthesis (cm.py:140-186) */
    uint8_t on_plane;          /* 1: handed to findHomography (cm.py:338-347) */
    uint8_t inlier;            /* 1: RANSAC inlier (cm.py:359-362); "Keypoints" = inliers when H_valid */
    uint8_t pad;
} EagleKeypoint;

/* Fixed-size per-frame record == res[i] of cm.py:415 (schema docs/data.md:20-41). */
typedef struct EagleFrameResult {
    int32_t n_det;
    int32_t n_kp;
    int32_t n_candidates;      /* boxes above detector_floor before NMS */
    uint8_t H_valid;
    uint8_t bounds_valid;
    uint8_t pad[2];            /* pad[0]: clip sessions, see eagle_clip_fetch; pad[1] = 1: EAGLE_PREC_F32S clipped an activation of this frame (EAGLE_E_RANGE) */
    double H[9];               /* row-major, h33 = 1 */
    double bounds[4];          /* x of [bottom_left(y=0), top_left(68), top_right(68), bottom_right(0)] */
    int32_t hm_idx[EAGLE_N_LANDMARKS];   /* first-maximum flat index per heat-map (kh.py:588) */
    float hm_score[EAGLE_N_LANDMARKS];   /* kh.py:589 */
    EagleKeypoint kp[EAGLE_MAX_KP];
    EagleDet det[EAGLE_MAX_DET];
} EagleFrameResult;

/* sizeof(EagleConfig), sizeof(EagleFrameResult), sizeof(EagleDet), sizeof(EagleKeypoint): binding self-check */
int eagle_abi_sizes(int32_t* out4);
int eagle_default_config(EagleConfig* cfg);
int eagle_create(const EagleConfig* cfg, EagleHandle** out);
void eagle_destroy(EagleHandle* h);
const char* eagle_last_error(EagleHandle* h);   /* h may be NULL: last error of eagle_create */
int eagle_resolve_config(EagleConfig* cfg);     /* in place, no GPU needed: what eagle_create will make of the "auto" fields (det_precision) */
int eagle_get_config(EagleHandle* h, EagleConfig* cfg);   /* the handle's configuration as eagle_create resolved it (det_precision no longer "auto") */

/* Weights: state-dict tensors by their reference names, fp32, PyTorch layouts
 *   HRNet + head: "unnormalized_model.0.<...>", "unnormalized_model.1.{weight,bias}" (kh.py:559-562)
 *   detector:     ultralytics "model.<N>.<...>"                                       (cm.py:54-57)
 * eagle_finalize_weights folds BatchNorm, re-tiles for MFMA and builds the per-batch launch schedule. */
int eagle_load_weights(EagleHandle* h, const char* name, const float* data, const int64_t* shape, int ndim);
int eagle_finalize_weights(EagleHandle* h);

/* The hot path: n BGR uint8 HWC frames (host memory) -> n records.  frame_stride / row_stride in bytes, 0 = dense ([n, h, w, 3] contiguous); a
 * row_stride below 3 * frame_w, a frame_stride below (frame_h - 1) * row_stride + 3 * frame_w or a negative stride is EAGLE_E_INVALID.  This is the call that replaces the reference's
 * per-frame loop over host frames (cm.py:277; frames come from eagle/utils/io.py::read_video).  The upload of batch k+1 overlaps the
 * networks of batch k.  Frames in pinned memory (eagle_host_alloc: what a decoder should write into) are DMA'd in place; frames in
 * pageable memory are first copied into a pinned ring by a few worker threads of the handle (EAGLE_COPY_THREADS, default 8). */
int eagle_process_frames(EagleHandle* h, const uint8_t* bgr, int n, int64_t frame_stride, int64_t row_stride,
                         EagleFrameResult* out);
int eagle_host_alloc(EagleHandle* h, int64_t bytes, void** ptr);   /* pinned host memory for frames */
int eagle_host_free(EagleHandle* h, void* ptr);

/* Same, inputs already resident in HBM (device pointer, dense [n,h,w,3]); records still land on the host.
 * bench.py reports this entry as `resident`; its `value` is eagle_process_frames from pageable host memory. */
int eagle_process_device_frames(EagleHandle* h, const void* d_bgr, int n, EagleFrameResult* out);
int eagle_device_alloc(EagleHandle* h, int64_t bytes, void** dptr);
int eagle_device_free(EagleHandle* h, void* dptr);
int eagle_device_upload(EagleHandle* h, void* dptr, const void* src, int64_t bytes);

/* Reference cadence with homography_interval > 1 (main.py:27 at --fps 5; cm.py:333-415): the caller decides, frame by frame in
 * clip order, which frame's homography each frame uses (scheduled / retry / carried) and hands the records back:
 * flags[i] = 0 keep the record, 1 re-project foot points and boundaries with Hs[9*i..], 2 no homography available yet. */
int eagle_reproject(EagleHandle* h, EagleFrameResult* recs, int n, const double* Hs, const uint8_t* flags);

/* ---- optical-flow key-point cadence: get_coordinates with keypoint_interval > 1, cm.py:188-416 ----------
 * The reference detects key-points with HRNet only every keypoint_interval-th frame and propagates them with
 * cv2.calcOpticalFlowPyrLK in between (cm.py:313-322, 419-478); that makes the loop sequential over a clip.  A clip session
 * keeps the clip, its gray pyramids, every frame's detections and the loop-carried state in HBM:
 *   eagle_clip_open             gray pyramids of all frames                                                      (cm.py:280)
 *   eagle_clip_detect_objects   detector / NMS / object rules on frames [first, first+count), asynchronous      (cm.py:331)
 *   eagle_clip_detect_keypoints HRNet + decode on frames first, first+stride, ... -> their mem[] entries, asynchronous (cm.py:217-276, 285, 317)
 *   eagle_clip_get/set_keypoints read / replace mem[frame]  (the first-frame search of cm.py:289-311 is sequenced by the caller)
 *   eagle_clip_flow             calculate_optical_flow(frames[hue_frame], gray[src], kps, gray[dst]) as an operator (cm.py:419-478)
 *   eagle_clip_run              the loop body for frames [first, last), in order, on the GPU without host round trips, on its own
 *                               stream behind every pass enqueued so far (so the passes of later frames overlap it); it stops itself
 *                               at a frame that needs a model detection it does not have.  wait = 1: block and report that frame
 *                               (*stalled_at, else -1); first > 0 continues with the loop state of the previous call, and with wait = 1 it
 *                               is the resume after an on-demand detection (clears the stall mark first)
 *   eagle_clip_fetch            the n records.  EagleFrameResult.pad[0] = 1 when the frame solved its own homography
 *                               ("Keypoints" = its inliers, cm.py:359-362); kp[].pad bit 0: the value came from the flow, bit 1: moved by the calibration
 *                               (both are numpy integers in the reference's dict, which json.dump(default=float) writes as floats). */
typedef struct EagleFlowKp { int32_t label; int32_t x, y; float score; } EagleFlowKp;
int eagle_clip_open(EagleHandle* h, const void* d_bgr, int n);
int eagle_clip_detect_objects(EagleHandle* h, int first, int count);
int eagle_clip_detect_keypoints(EagleHandle* h, int first, int stride, int count);
int eagle_clip_get_keypoints(EagleHandle* h, int frame, EagleFlowKp* out /* EAGLE_N_LANDMARKS */, int* n /* -1: no entry */);
int eagle_clip_set_keypoints(EagleHandle* h, int frame, const EagleFlowKp* in, int n);
int eagle_clip_flow(EagleHandle* h, int src_frame, int dst_frame, int hue_frame, const EagleFlowKp* in, int n_in,
                    EagleFlowKp* out /* EAGLE_N_LANDMARKS */, int* n_out, float* next_pts /* 2*n_in or NULL */, uint8_t* status /* n_in or NULL */);
int eagle_clip_run(EagleHandle* h, int first, int last, int keypoint_interval, int homography_interval, int calibration, int wait,
                   int* stalled_at);
int eagle_clip_fetch(EagleHandle* h, EagleFrameResult* out);
int eagle_clip_close(EagleHandle* h);
#define EAGLE_E_REFERENCE_RAISES (-7) /* the reference raises IndexError here (calibration grid at the image border, cm.py:545) */

/* ---- track identities: self.tracker.update(dets, frame) of cm.py:66-72, 574-596 -----------------------------------
 * BoT-SORT's motion / IoU association (constant-velocity Kalman filter, high / low confidence sets, three assignments, life cycle);
 * appearance and camera-motion compensation are separate entry points below (eagle_reid_features, eagle_clip_motion_ecc / eagle_clip_motion).  eagle_track_frames walks n records of ONE
 * clip in frame order (call it chunk after chunk; eagle_track_open starts a new clip): Player / Goalkeeper entries become keyed by
 * track id with the filter's boxes and feet (cm.py:577-596; frames on which the tracker reports no player keep the detection-index
 * fallback of cm.py:598-616), then the pitch coordinates of the moved foot points are recomputed on the GPU with each record's H. */
typedef struct EagleCrop { int32_t frame, x1, y1, x2, y2; } EagleCrop;      /* frame[y1:y2, x1:x2] of a clip resident in HBM */
typedef struct EagleTrackParams {
    float track_high_thresh, track_low_thresh, new_track_thresh, match_thresh;   /* 0.5, 0.1, 0.6, 0.8 (boxmot defaults) */
    int32_t track_buffer, frame_rate;                                             /* 30, 30 */
} EagleTrackParams;
int eagle_track_open(EagleHandle* h, const EagleTrackParams* params /* NULL: defaults */);
int eagle_track_frames(EagleHandle* h, EagleFrameResult* recs, int n);
/* Camera-motion compensation (BoT-SORT's gmc step; boxmot's default estimates the warp with ECC, here: a similarity transform from an 8 x 6 grid
 * of points tracked by the key-point cadence's pyramidal LK kernel, RANSAC over point pairs + least squares on the consensus set — stated deviation).
 * eagle_clip_motion: needs an open clip session (eagle_clip_open); warps[6 * i .. +5] = row-major 2 x 3 warp of frame first+i-1 -> first+i
 * (identity for clip frame 0).  eagle_track_frames_cmc: as eagle_track_frames, applying warps[6 * i] to all track states before frame i is associated. */
int eagle_clip_motion(EagleHandle* h, int first, int count, double* warps);
/* boxmot's default estimator (cmc_method "ecc", what the reference's BotSort(...) of cm.py:66-72 runs on the frame passed at cm.py:577):
 * gray -> cv2.resize(fx = fy = 0.15) -> cv2.findTransformECC(prev, cur, MOTION_EUCLIDEAN, 100 iterations / 1e-5), translation rescaled by 1 / 0.15;
 * a failed alignment (cv2 raises: lambda_d <= 0 or NaN correlation) yields the identity and keeps the OLD template, as boxmot does.  Same warps
 * layout as eagle_clip_motion; ok[i] (may be NULL) = 0 where the alignment of frame first+i failed.  carry != 0: the estimator's template
 * survives the clip like boxmot's ECC object survives inside the tracker — clip frame 0 is aligned to the last template of the previous call
 * (eagle_track_open forgets it).  Restated from the published algorithm (oracle/ecc.py); cv2 / boxmot absent: parity unpinned. */
int eagle_clip_motion_ecc(EagleHandle* h, int first, int count, int carry, double* warps, int* ok);
int eagle_track_frames_cmc(EagleHandle* h, EagleFrameResult* recs, int n, const double* warps /* NULL: none */);
/* Appearance (BoT-SORT with_reid, the reference's configuration: cm.py:66-72 builds BotSort with osnet_x0_25 ReID weights and passes the frame at
 * cm.py:577).  eagle_reid_features: OSNet-x0.25 embeddings (EAGLE_REID_DIM floats each) of crops frame[y1:y2, x1:x2] of a clip resident in HBM,
 * prepared as boxmot does (resize to 128 x 256, RGB, ImageNet normalisation); needs the "reid.*" tensors (torchreid's parameter names) loaded
 * before eagle_finalize_weights, EAGLE_E_STATE otherwise.  eagle_track_frames_reid: as eagle_track_frames_cmc with the embeddings of each
 * record's high-confidence detections: record i owns feat_count[i] consecutive rows of `feats`, feat_det lists their detection indices.  The
 * association then is BoT-SORT's: cost = min(IoU distance, embedding distance / 2) with embedding distances above appearance_thresh 0.25 or IoU
 * distances above proximity_thresh 0.5 set to 1; track features are exponential moving averages (alpha 0.9) of the normalised embeddings.
 * Contract: `feats` holds sum(feat_count) rows of EAGLE_REID_DIM floats and `feat_det` as many entries; 0 <= feat_count[i] <= EAGLE_MAX_DET and
 * 0 <= feat_det[.] < recs[i].n_det are checked (EAGLE_E_INVALID) before anything is read. */
#define EAGLE_REID_DIM 512
int eagle_reid_features(EagleHandle* h, const void* d_bgr, int n_frames, const EagleCrop* crops, int n_crops, float* feats);
int eagle_track_frames_reid(EagleHandle* h, EagleFrameResult* recs, int n, const double* warps /* NULL: none */, const float* feats,
                            const int32_t* feat_det, const int32_t* feat_count);

/* ---- team colours: Processor.detect_color, eagle/processor.py:466-503, for player crops of a clip resident in HBM -----
 * counts[12 * i + k]: pixels of crop i's player cluster inside colour range k of proc.py:10-23, k = red (red2 merged), orange, yellow,
 * green, cyan, blue, purple, magenta, white, gray, black; slot 11 = pixels of the player cluster.  Crops are frame[y1:y2, x1:x2]. */
int eagle_team_colors(EagleHandle* h, const void* d_bgr, int n_frames, const EagleCrop* crops, int n_crops, int32_t* counts);

/* Frame-sharded multi-GPU: rank r owns a contiguous chunk; one RCCL all-gather of records.
 * eagle_comm_id fills a 128-byte ncclUniqueId on rank 0; the caller broadcasts it (any channel). */
int eagle_comm_id(void* id128);
int eagle_comm_init(EagleHandle* h, int rank, int world, const void* id128);
int eagle_gather(EagleHandle* h, const EagleFrameResult* local, int n_local, EagleFrameResult* all /* world*n_local */);

/* Timing of the last eagle_process_* call, measured with HIP events on the library's compute stream. */
typedef struct EagleTimings {
    float total_ms;            /* first kernel -> records copied */
    float conv_ms;             /* sum over convolution launches (only when profiling is enabled) */
    int32_t n_launches;
    int32_t n_conv_launches;
    double conv_flop;          /* algorithmic FLOP of the convolutions of the last call (2*MAC) */
    int32_t sat_events;        /* EAGLE_PREC_F32S: lane-level activation stores of the last eagle_process_* / eagle_clip_fetch call that were clipped at +-4094 */
    int32_t sat_frames;        /* ... and the number of frames they occurred in (0 / 0 in every healthy run) */
    int32_t graph_captures;    /* hipGraph captures this handle has made since eagle_create (cumulative: one per pipeline slot and frame count, ~80 ms each; a caller that
                                  sees this number grow call after call is paying for re-captures) */
    int32_t graph_skipped;     /* ... and capture attempts given up because another call of the process held the capture lock (the step then ran as plain launches) */
    int32_t reserved[4];
} EagleTimings;
int eagle_set_profiling(EagleHandle* h, int per_kernel_events);
int eagle_get_timings(EagleHandle* h, EagleTimings* t);
/* Profiling mode also times every non-convolution launch (HIP events on its launch stream) and accumulates, per kernel name, the
 * elapsed time and the ALGORITHMIC HBM bytes of the launches (inputs read once + outputs written once): the HBM-roofline
 * rows of bench.py.  The table is cleared by eagle_set_profiling(h, 1). */
typedef struct EagleKernelTime { char name[40]; float ms; int32_t launches; double bytes; double flop; } EagleKernelTime;   /* convolutions: one row per layer shape, flop = 2*MAC */
int eagle_get_kernel_times(EagleHandle* h, EagleKernelTime* out, int cap, int* n);

/* Operator-level entry points (host buffers in/out) used by the parity tests: each runs ONE kernel of the path.
 * Tensors are dense NHWC fp32 on the host; `precision` selects the fp16 or fp32 kernel family. */
int eagle_op_conv2d(int device, int precision, const float* x, int n, int h, int w, int cin, const float* w_hwio,
                    const float* bias, int cout, int ks, int stride, int pre_act, const float* r1, const float* r2,
                    int post_act, float* y);
/* One fused Bottleneck of HRNet's layer 1 in the split family (bneck.hip; kh.py:101-137): y = relu(conv3(relu(conv2(relu(conv1(x))))) + res), conv1 1x1 Cin->64,
 * conv2 3x3 64->64, conv3 1x1 64->256, BatchNorm already folded into (w, b); weights HWIO ([1][1][Cin][64], [3][3][64][64], [1][1][64][256]); res = NULL: the
 * identity shortcut (Cin = 256).  reps > 0: *ms = average duration of `reps` further launches (HIP events). */
int eagle_op_bottleneck(int device, const float* x, int n, int h, int w, int cin, const float* w1, const float* b1, const float* w2, const float* b2,
                        const float* w3, const float* b3, const float* res, float* y, int reps, float* ms,
                        const float* wd /* NULL, or the 1x1 downsample branch [1][1][64][256] computed inside the launch (Cin = 64, res = NULL) */, const float* bd);
int eagle_op_fuse_sum(int device, int precision, const float* base, int n, int H, int W, int c, int n_up,
                      const float* const* ups, const int* up_h, const int* up_w, int relu, float* y);
int eagle_op_preprocess(int device, int precision, const uint8_t* bgr, int n, int h, int w, int det_imgsz,
                        float* kp_out /* n*540*960*3 */, float* det_out /* n*dh*dw*3 */, int* det_hw /* 2 */);
int eagle_op_preprocess_lb(int device, int precision, const uint8_t* bgr, int n, int h, int w, int det_imgsz, int letterbox /* EAGLE_LETTERBOX_* */,
                           float* kp_out, float* det_out, int* det_hw);
int eagle_op_find_homography(int device, const float* img_pts, const float* world_pts, int n, double thresh,
                             int max_iters, int lm_iters, double* H9, uint8_t* mask, int* ok);

/* Developer diagnostics (process-wide switches and read-backs used by tools/probe_lk_concurrency.py; not part of the data path).
 * Inert (EAGLE_E_STATE) unless the process environment has EAGLE_ENABLE_DEBUG=1: a production caller cannot flip them by accident.
 * Keys: "lk_threads" (64 | 256), "lk_dbg" (1 trace, 2 LDS guard words, 4 end-of-level verification, 8 L1-bypassing loads), "lk_excl_lds" (bytes), "lk_trace", "lk_counters". */
int eagle_debug(const char* key, int64_t value, void* out, int64_t out_bytes);

#ifdef __cplusplus
}
#endif
#endif
