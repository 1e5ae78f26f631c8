/*
 * d2fe.h -- C ABI of libd2fe_hip.so: the MI355X (gfx950) implementation of D2SLAM's d2frontend
 * feature hot path (SuperPoint extraction, NetVLAD global descriptor, brute-force descriptor matching).
 *
 * Every entry point below replaces one call the reference makes into TensorRT / ONNX Runtime / OpenCV;
 * the reference interface it stands in for is cited as file:line relative to the D2SLAM tree.
 * Conventions: plain pointers and sizes only; caller owns every buffer; nothing is retained past return;
 * functions return 0 (D2FE_OK) or a negative d2fe_status and never throw; d2fe_last_error() describes the
 * last failure on the calling thread.  Pointers are HOST pointers unless the parameter name starts with d_.
 */
#ifndef D2FE_H_
#define D2FE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define D2FE_API __attribute__((visibility("default")))
#else
#define D2FE_API
#endif

typedef struct d2fe_context* d2fe_handle;

typedef enum {
  D2FE_OK = 0,
  D2FE_ERR_INVALID = -1,     /* bad argument / size mismatch (reference: assert, superpoint_onnx.cpp:74-75) */
  D2FE_ERR_HIP = -2,         /* HIP runtime failure (reference: infer() returns false, superpoint_tensorrt.cpp:164-170) */
  D2FE_ERR_NOT_READY = -3,   /* weights not loaded */
  D2FE_ERR_TRUNCATED = -4,   /* output capacity too small; n_out holds what was written */
  D2FE_ERR_UNSUPPORTED = -5
} d2fe_status;

/* Post-processing variant. B is the live USE_CUDA path of the reference. */
typedef enum {
  D2FE_POSTPROC_B = 0, /* SuperPoint::processOutput, superpoint_tensorrt.cpp:327-350: threshold, borders, top-K */
  D2FE_POSTPROC_A = 1  /* SuperPointONNX: getKeyPoints + NMS2 + grid_sampler, superpoint_common.cpp:12-177 */
} d2fe_postproc;

typedef enum {
  D2FE_PREC_F32 = 0,   /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32): bitwise equal to the oracle's fmaf chains */
  D2FE_PREC_F16X2 = 1, /* fp16 hi/lo split operands, 3 x v_mfma_f32_32x32x16_f16, fp32 accumulate (~2^-22 rel.) */
  D2FE_PREC_F32_WINO = 2 /* fp32 throughout; the eight 3x3 layers with Cin >= 64 as Winograd F(2x2,3x3) on v_mfma_f32_32x32x2_f32
                            (16 instead of 36 multiplies per output and channel pair).  Bitwise equal to the oracle's restatement
                            of that evaluation order (orc_conv3x3_wino), ~1e-6 relative to the direct chains of D2FE_PREC_F32 */
} d2fe_precision;

/* Mirrors SuperPointConfig (d2frontend/include/d2frontend/CNN/superpoint_tensorrt.h:17-33) plus the
 * SuperPointONNX ctor arguments (superpoint_onnx.h:17-21) and the device/batch geometry. */
typedef struct {
  int32_t struct_size;        /* sizeof(d2fe_config), for forward compatibility */
  int32_t device_id;          /* HIP device ordinal */
  int32_t max_width;          /* largest input width (>= 16)   -- SuperPointConfig::input_width  */
  int32_t max_height;         /* largest input height (>= 16)  -- SuperPointConfig::input_height */
  int32_t max_batch;          /* images per batched call (>= 1) */
  int32_t max_keypoints;      /* SuperPointConfig::max_keypoints: 1..16384 (sorted top-K), or -1 = keep every keypoint above the threshold in raster
                                 order like topKeypoints with k == -1 (superpoint_tensorrt.cpp:241-253)  [params->max_superpoint_cnt].
                                 Keep-all with MORE keypoints in an image than the call's capacity: D2FE_ERR_TRUNCATED, and the strongest
                                 min(capacity, 16384) of them are returned in score order (16384 = the in-LDS sort) */
  int32_t remove_borders;     /* SuperPointConfig::remove_borders (variant B), default 1 */
  float   keypoint_threshold; /* SuperPointConfig::keypoint_threshold, default 0.015 */
  int32_t postproc;           /* d2fe_postproc */
  int32_t nms_dist;           /* variant A: NMS2 dist_thresh (SuperPointONNX::nms_dist) */
  int32_t precision;          /* d2fe_precision */
  int32_t keep_score_map;     /* 1: also write the dense H x W score map ("semi", 1.2 MB/image; read back by the development library's d2fe_debug_read);
                                 variant B does not need it (candidates are emitted by the softmax kernel) */
  int32_t dense_descriptors;  /* 0 (default): the descriptor head is evaluated (convDa, convDb) only at the corner cells of the
                                 selected keypoints when a call carries >= 4 images (below that the dense head is quicker; both give identical bits) -- 5 % fewer
                                 FLOPs; 1: always the dense descriptor map (d2fe_debug_read
                                 "desc_raw" / "convPaDa" need it). */
  int32_t async_tail;         /* 1: d2fe_superpoint_extract_device issues the convolutions on the caller's stream and the post-processing
                                 (softmax .. descriptors) on the handle's tail stream (d2fe_tail_stream), so that it runs UNDER the
                                 convolutions of the next call; outputs are complete on the tail stream -- enqueue consumers there, or
                                 call d2fe_superpoint_wait_tail(h, stream).  Host-pointer calls are unaffected.  Default 0. */
  int32_t reserved[5];
} d2fe_config;

/* One conv layer in PyTorch layout: weight [cout][cin][k][k], bias [cout]. */
typedef struct {
  const float* weight;
  const float* bias;
  int32_t cout, cin, ksize;
} d2fe_conv_params;

/* The 12 SuperPoint layers in the order of d2frontend/superpoint.ipynb:306-321:
 * conv1a conv1b conv2a conv2b conv3a conv3b conv4a conv4b convPa convPb convDa convDb */
#define D2FE_SP_NUM_LAYERS 12
typedef struct {
  d2fe_conv_params layer[D2FE_SP_NUM_LAYERS];
} d2fe_superpoint_weights;

D2FE_API const char* d2fe_last_error(void);
D2FE_API const char* d2fe_version(void);
D2FE_API void d2fe_default_config(d2fe_config* cfg);

/* Lifecycle.  Replaces: LoopCam ctor building the networks (loop_cam.cpp:24-70),
 * SuperPoint::SuperPoint + SuperPoint::build (superpoint_tensorrt.cpp:17-107). */
D2FE_API int d2fe_create(const d2fe_config* cfg, d2fe_handle* out);
D2FE_API void d2fe_destroy(d2fe_handle h);
/* Replaces the ONNX parse / engine deserialisation (superpoint_tensorrt.cpp:109-125,376-398):
 * weights are copied, re-packed into MFMA fragment order and cached on the device. */
D2FE_API int d2fe_load_superpoint(d2fe_handle h, const d2fe_superpoint_weights* w);

/* Optional PCA of the local descriptors (variant A only, as in the reference: computeDescriptors,
 * superpoint_common.cpp:76-85; the variant-B path never applies it).
 * comp: [pca_dims][256] row-major (the CSV layout read by superpoint_onnx.cpp:47-53), mean: [256].
 * pca_dims = 0 disables.  d2fe_desc_dim() returns the per-keypoint descriptor length of extract calls (256 or pca_dims). */
D2FE_API int d2fe_set_superpoint_pca(d2fe_handle h, const float* comp, const float* mean, int pca_dims);
D2FE_API int d2fe_desc_dim(d2fe_handle h);

/* Extractor.  Replaces: bool SuperPoint::infer(const cv::Mat&, std::vector<cv::Point2f>&, std::vector<float>&
 * descriptors, std::vector<float>& scores) (superpoint_tensorrt.h:47-48, .cpp:161-183), called from
 * LoopCam::extractorImgDescDeepnet (loop_cam.cpp:609-610).
 *   gray: u8 image, height rows of `stride` bytes.   kps_xy: cap*2 floats (x,y).   scores: cap floats.
 *   desc: cap*D floats keypoint-major, D = d2fe_desc_dim() (256 unless PCA is set).   *n_out = number of keypoints written (0 on failure).
 * Order of outputs = selection order of the chosen variant (B: raster if K<=N else score-desc; A: score-desc).
 * Not re-entrant per handle (the reference has one caller thread, d2frontend.cpp:155-169). */
D2FE_API int d2fe_superpoint_extract(d2fe_handle h, const uint8_t* gray, int width, int height, int stride,
                                     float* kps_xy, float* scores, float* desc, int cap, int* n_out);

/* Batched form of the same call: n images of identical size, image i at gray + i*image_stride bytes;
 * outputs of image i at kps_xy + i*cap*2, scores + i*cap, desc + i*cap*256, n_out[i]. */
/* async_tail mode: the stream on which the outputs of the last d2fe_superpoint_extract_device call become valid, and a helper
 * that makes another stream wait for them. */
D2FE_API void* d2fe_tail_stream(d2fe_handle h);
D2FE_API int d2fe_superpoint_wait_tail(d2fe_handle h, void* stream);
D2FE_API int d2fe_superpoint_extract_batch(d2fe_handle h, const uint8_t* gray, int n, int width, int height,
                                           int stride, size_t image_stride, float* kps_xy, float* scores,
                                           float* desc, int cap, int* n_out);

/* Device-resident form: inputs already in HBM (d_gray) and outputs left in HBM (d_*), asynchronous on
 * `stream` (a hipStream_t, or NULL for the handle's own stream).  d_n_out: n int32 counts.
 * d_kps_idx (optional, may be NULL): raster indices y*W+x of the keypoints. */
D2FE_API int d2fe_superpoint_extract_device(d2fe_handle h, const uint8_t* d_gray, int n, int width, int height,
                                            int stride, size_t image_stride, float* d_kps_xy, float* d_scores,
                                            float* d_desc, int32_t* d_kps_idx, int cap, int32_t* d_n_out,
                                            void* stream);

/* ---- NetVLAD global descriptor ---------------------------------------------------------------------------------------
 * Replaces: MobileNetVLADONNX (d2frontend/include/d2frontend/CNN/mobilenetvlad_onnx.h:18-74): ctor = ORT session + PCA
 * CSV (:35-46), inference(const cv::Mat&) -> std::vector<float> of 4096 (or PCA dims) (:49-74), called from
 * LoopCam::extractorImgDescDeepnet (loop_cam.cpp:612-616).  The reference's graph file is not in its tree:
 * the network is given as a flat layer list (kinds below), so any MobileNetV2-style trunk + NetVLAD head can be loaded;
 * d2slam_amd/netvlad.py holds the documented stand-in.  Layouts: conv [cout][1][3][3] (3x3 from the 1-channel image, in-graph
 * (x-128)/128), dw [c][3][3], pw [cout][cin]; TF "SAME" padding; act 0 none / 1 ReLU / 2 ReLU6; res = index of the layer whose
 * output is added (-1 none).  Head: 1x1 pre-projection feat_dim -> proj_dim, soft-assignment [K][proj_dim], centroids. */
typedef enum { D2FE_NV_CONV = 0, D2FE_NV_PW = 1, D2FE_NV_DW = 2 } d2fe_nv_kind;
typedef struct {
  int32_t kind, cin, cout, stride, act, res;
  const float* weight;
  const float* bias;
} d2fe_nv_layer;
typedef struct {
  int32_t n_layers;
  const d2fe_nv_layer* layers;
  int32_t feat_dim, proj_dim, n_clusters;
  const float* pre_w;      /* [proj_dim][feat_dim] */
  const float* pre_b;      /* [proj_dim] */
  const float* assign_w;   /* [n_clusters][proj_dim] */
  const float* assign_b;   /* [n_clusters] */
  const float* centroids;  /* [n_clusters][proj_dim] */
} d2fe_netvlad_weights;
D2FE_API int d2fe_load_netvlad(d2fe_handle h, const d2fe_netvlad_weights* w);
/* PCA of the global descriptor: comp [m][G], mean [G], G = n_clusters*proj_dim (the reference's CSV: row 0 = mean,
 * rows 1.. = components, mobilenetvlad_onnx.h:35-41).  m = 0 disables. */
D2FE_API int d2fe_set_netvlad_pca(d2fe_handle h, const float* comp, const float* mean, int m);
D2FE_API int d2fe_netvlad_dim(d2fe_handle h);   /* length of the descriptor written by the calls below */
/* std::vector<float> MobileNetVLADONNX::inference(const cv::Mat&): gray u8 at the network's size (the reference resizes with
 * cv::resize when needed; that stays the caller's job).  out: d2fe_netvlad_dim() floats. */
/* Like the extract calls, the NetVLAD calls are not re-entrant per handle (they reuse the handle's layer buffers and record the slab
 * layout of the call in it); the reference calls MobileNetVLADONNX::inference from its one front-end thread (d2frontend.cpp:155-169). */
D2FE_API int d2fe_netvlad(d2fe_handle h, const uint8_t* gray, int width, int height, int stride, float* out);
D2FE_API int d2fe_netvlad_batch(d2fe_handle h, const uint8_t* gray, int n, int width, int height, int stride,
                                size_t image_stride, float* out);
D2FE_API int d2fe_netvlad_device(d2fe_handle h, const uint8_t* d_gray, int n, int width, int height, int stride,
                                 size_t image_stride, float* d_out, void* stream);

/* Fused form of the two calls LoopCam::extractorImgDescDeepnet makes for one image (loop_cam.cpp:609-616: superpoint_net->infer(...) and then
 * netvlad_onnx->inference(...) on the same frame): ONE upload, SuperPoint and NetVLAD on two streams side by side (at one or two images per call
 * both are latency-bound and leave most of the chip idle), one synchronisation.  Same outputs, bit for bit, as d2fe_superpoint_extract(_batch)
 * followed by d2fe_netvlad(_batch) on the first n_netvlad images (stereo: the left image, loop_cam.cpp:446-451).  netvlad_out: n_netvlad x
 * d2fe_netvlad_dim() floats.  Needs both networks loaded and an image size both accept. */
D2FE_API int d2fe_extract_all(d2fe_handle h, const uint8_t* gray, int width, int height, int stride, float* kps_xy, float* scores, float* desc,
                              int cap, int* n_out, float* netvlad_out);
D2FE_API int d2fe_extract_all_batch(d2fe_handle h, const uint8_t* gray, int n, int width, int height, int stride, size_t image_stride,
                                    float* kps_xy, float* scores, float* desc, int cap, int* n_out, int n_netvlad, float* netvlad_out);

/* Matcher.  Replaces: std::vector<cv::DMatch> matchKNN(const cv::Mat& desc_a, const cv::Mat& desc_b,
 * double knn_match_ratio, pts_a, pts_b, double search_local_dist) (feature_matcher.h:6-11,
 * feature_matcher.cpp:4-42).  a: na x dim row-major, b: nb x dim.  pts_*: n x 2 floats or NULL.
 * radius <= 0 disables the pixel gate.  Outputs ascending in query index; *n_out matches written.
 * Re-entrant (the reference calls it from three threads). */
/* Up to 16384 rows per side (the reference is unbounded; D2FE_ERR_UNSUPPORTED above that). */
D2FE_API int d2fe_match_knn(d2fe_handle h, const float* a, int na, const float* b, int nb, int dim, double ratio,
                            const float* pts_a, const float* pts_b, double radius, int32_t* q_idx,
                            int32_t* t_idx, float* dist, int cap, int* n_out);

/* Replaces: cv::BFMatcher(cv::NORM_L2, true).match(desc_a, desc_b, matches)
 * (loop_cam.cpp:167-170, d2featuretracker.cpp:1141-1142,1175-1176, loop_detector.cpp:576-577). */
D2FE_API int d2fe_match_crosscheck(d2fe_handle h, const float* a, int na, const float* b, int nb, int dim,
                                   int32_t* q_idx, int32_t* t_idx, float* dist, int cap, int* n_out);

/* Both matchers select candidates on a Gram-trick distance (fp32 matrix pipe) whose error against the reference's (OpenCV's) arithmetic
 * is rigorously bounded: every train row that the bound cannot separate from the second-nearest one (2.0-2.3 rows per query on
 * SuperPoint descriptors) is re-ranked in the reference's arithmetic; a query with more than 16 such rows (repeated texture,
 * near-duplicate frames, degenerate sets) is re-evaluated by an exact scan of all train rows.  The result equals the reference's for any
 * input.  d2fe_match_fallback_rows returns how many candidates beyond two per query were re-ranked since the last reset and, in
 * *full_scans (may be NULL), how many queries took the exact scan (diagnostic; synchronises the device). */
D2FE_API long d2fe_match_fallback_rows(d2fe_handle h, int reset, long* full_scans);

/* Batched, device-resident matcher: npairs problems; pair p matches rows [a_off[p], a_off[p]+a_cnt[p]) of
 * d_a against rows [b_off[p], ...) of d_b.  Counts may live on the device (d_a_cnt/d_b_cnt, e.g. the d_n_out
 * of an extract call) -- pass max_n as the upper bound of any count.  mode: 0 = matchKNN, 1 = cross-check.
 * Outputs: pair p writes at most max_n matches at d_q_idx + p*max_n etc., and d_n_out[p]. */
typedef struct {
  const float* d_a; const float* d_b;             /* descriptor pools, row-major dim floats per row */
  const float* d_pts_a; const float* d_pts_b;     /* optional pools of (x,y), may be NULL */
  const int32_t* d_a_off; const int32_t* d_b_off; /* [npairs] first row of each side */
  const int32_t* d_a_cnt; const int32_t* d_b_cnt; /* [npairs] row counts (device) */
  int32_t npairs, dim, max_n, mode;
  double ratio, radius;
  int32_t* d_q_idx; int32_t* d_t_idx; float* d_dist; int32_t* d_n_out;
} d2fe_match_batch;
D2FE_API int d2fe_match_batch_device(d2fe_handle h, const d2fe_match_batch* mb, void* stream);

/* ---- Frames in flight (throughput form of the per-frame work) ------------------------------------------------------------------
 * The reference handles ONE stereo frame at a time on one thread (D2Frontend::processStereoframe, d2frontend.cpp:155-169): per image
 * SuperPoint::infer and, for the main camera, MobileNetVLADONNX::inference (LoopCam::extractorImgDescDeepnet, loop_cam.cpp:589-648,
 * both cameras: generateStereoImageDescriptor :440-470), then matchKNN left<->right and left<->previous left
 * (D2FeatureTracker::trackLocalFrames, d2featuretracker.cpp:403-456,658-695).  A pipe runs exactly that work for `frames` stereo
 * frames per submit with up to `lanes` submits in flight: d2fe_pipe_submit only enqueues (H2D, both networks, ONE matcher launch,
 * ONE D2H, on the lane's own streams), d2fe_pipe_wait returns pointers into the lane's pinned result block.  Results are bit-identical
 * to d2fe_extract_all_batch + d2fe_match_knn on the same frames.  A pipe borrows the handle's packed weights: while a pipe exists d2fe_destroy
 * of its handle is DEFERRED (the handle is released by the last d2fe_pipe_destroy; d2fe_last_error says so) and d2fe_load_* / d2fe_set_*_pca return
 * D2FE_ERR_INVALID -- destroy the pipes first.  A submit whose pass would overwrite a result block with an unreleased device view returns D2FE_ERR_NOT_READY
 * and queues nothing (release the view, submit again).  The first error of a
 * submit or wait is final for the pipe: every later call returns it (a half-enqueued pass cannot be built on); destroy the pipe and create a new one.  One submitting thread per pipe; d2fe_pipe_wait may be called from a second thread (the reference's image
 * callback and its tracker are two threads): the pipe serialises its own bookkeeping and does not hold the lock while a wait blocks.  The caller bounds the
 * frames between its two threads (a queue of at most lanes * coalesce tickets is always safe), as the result blocks are a ring of 2 * lanes passes. */
typedef struct d2fe_pipe_s* d2fe_pipe;
typedef struct {
  int32_t struct_size;      /* sizeof(d2fe_pipe_config) */
  int32_t lanes;            /* submits in flight, 1..16 (each lane owns its activations: ~58 MB per 640x480 image) */
  int32_t frames;           /* stereo frames per submit (>= 1); consecutive in time */
  int32_t width, height;    /* frame size, within the handle's maximum */
  int32_t cap;              /* keypoint capacity per image (rows of the result arrays) */
  int32_t netvlad;          /* 1: NetVLAD of the left images (the main camera, loop_cam.cpp:446-451) */
  int32_t match_lr;         /* 1: matchKNN L_f <-> R_f */
  int32_t match_prev;       /* 1: matchKNN L_f <-> L_(f-1); f = 0: the last left frame of the previous submit (none for the first) */
  int32_t pinned_input;     /* 1: the frame pointers handed to submit are page-locked and stay valid until the ticket was waited for
                               (DMA straight from them); 0: submit copies the frames into the lane's pinned staging first */
  double ratio;             /* knn_match_ratio */
  double radius_lr;         /* search_local_max_dist_lr * width (<= 0: off), d2featuretracker.cpp:669 */
  double radius_prev;       /* search_local_max_dist * width (<= 0: off), d2featuretracker.cpp:21-28 */
  int32_t cu_partition;     /* 1: every lane launches on its own disjoint 1/lanes of the compute units (CU-masked streams): the lanes' kernels run
                               side by side instead of queueing behind each other's full-device launches.  For small `frames`: a lane then
                               behaves like a (256 / lanes)-CU device working on its own frame */
  int32_t netvlad_inline;   /* which stream a pass's NetVLAD call is launched on.  0: the lane's second stream, beside SuperPoint (shortest single pass: 0.72 ms);
                               1: the lane's one stream, in front of SuperPoint (no second streams are created);
                               2 (d2fe_pipe_default_config): auto -- decided per pass: the second stream while at most one OTHER pass is in flight, the lane's own
                               stream beyond that (one or two lanes: always the second stream).  Why: the device runs FOUR busy streams of a process side by side
                               (hardware queue i is served by hardware pipe i mod 4) and makes a fifth take turns.  Single-frame passes on a 4-lane pipe, auto:
                               1390 / 1838 / 1789 / 2044 stereo frames/s with 1 / 2 / 3 / 4 passes in flight; forced second streams: 1627 with 4
                               (profiles/r05_pipe_one_frame.txt (8)).  d2fe_pipe_create measures which streams share a hardware pipe and gives a lane two streams of
                               different pipes (d2fe_pipe_stream_placement); create the pipe on a quiet device.
                               Results are the same bits in every mode */
  int32_t coalesce;         /* > 1 (needs frames == 1): up to this many consecutive submits run as ONE launch sequence when they are submitted before
                               anybody waits for them -- submit() stages the frame (its H2D starts at once) and the pass is launched when it is full
                               or when d2fe_pipe_wait asks for one of its tickets; results per ticket are unchanged (bit-identical).  What a
                               throughput-oriented caller that receives frames one at a time would otherwise do by hand with frames = 2 */
  int32_t lane_cus;         /* > 0: a lane's persistent kernels size their grids for this many compute units (no CU mask: they may run on any CU).  A
                               full-device persistent launch occupies every CU's LDS until it ends, so other lanes' small launches wait for it;
                               with e.g. 128 of 256, two lanes run side by side on every CU.  0: the whole device */
  int32_t netvlad_group;    /* > 1 (needs frames == 1, coalesce == 1, lanes % netvlad_group == 0): the NetVLAD descriptors of this many consecutive
                               submits are computed by ONE call on the pipe's own stream when the last of them has been submitted (or when
                               d2fe_pipe_wait asks for one of them), while SuperPoint and the matches of every submit start at once.  NetVLAD of a
                               single image is ~20 launches of a few workgroups each (0.25 ms for one image, 0.28 ms for four); the descriptor feeds
                               loop detection, not the tracker, so it can trail the keypoints.  Bit-identical results */
  int32_t coalesce_depth;   /* with coalesce > 1.  0: a pass is launched when it is full (or when d2fe_pipe_wait asks).  N > 0: ALSO as soon as fewer than N
                               passes are in flight on the device -- dynamic batching: a caller that waits for every frame before it submits the next
                               gets every frame launched at once (the latency of coalesce = 1), a caller that keeps many frames in flight gets passes
                               that grow up to `coalesce` frames while the device is busy with N others (the throughput of large passes).  2 is the
                               measured choice: one pass running, one queued behind it */
  int32_t reserved[2];
} d2fe_pipe_config;
typedef struct {            /* HOST pointers into the lane's pinned block; valid until 2 * lanes further submits */
  int32_t frames, cap, desc_dim, netvlad_dim;
  const float* kps_xy;      /* [2 frames][cap][2]   image order L_0 .. L_(F-1), R_0 .. R_(F-1) */
  const float* scores;      /* [2 frames][cap] */
  const float* desc;        /* [2 frames][cap][desc_dim] */
  const int32_t* n_kp;      /* [2 frames] */
  const float* netvlad;     /* [frames][netvlad_dim] or NULL */
  const int32_t* lr_q; const int32_t* lr_t; const float* lr_dist; const int32_t* lr_n;            /* [frames][cap] x3, [frames]; NULL when off */
  const int32_t* prev_q; const int32_t* prev_t; const float* prev_dist; const int32_t* prev_n;    /* prev_t indexes the previous left frame's keypoints */
} d2fe_pipe_result;
D2FE_API void d2fe_pipe_default_config(d2fe_pipe_config* cfg);
D2FE_API int d2fe_pipe_create(d2fe_handle h, const d2fe_pipe_config* cfg, d2fe_pipe* out);
D2FE_API void d2fe_pipe_destroy(d2fe_pipe p);
D2FE_API int d2fe_pipe_lanes(d2fe_pipe p);
/* How d2fe_pipe_create placed the lanes' streams: it measures which of its candidate streams take turns on the device (streams of one hardware pipe) and gives every
 * lane two streams of different pipes.  classes[2 k] / classes[2 k + 1] = the class of lane k's own / second stream (-1: no such stream, or CU-masked lanes),
 * *n_classes = the classes told apart (4 on an idle MI355X; 0: the device was not quiet or nothing could be told apart -- the arrangement of a fresh process was used) */
D2FE_API int d2fe_pipe_stream_placement(d2fe_pipe p, int32_t* classes /*[2 * lanes]*/, int32_t* n_classes);
/* d2fe_profile_enable / d2fe_profile_read over all lanes (sums) */
D2FE_API int d2fe_pipe_profile_enable(d2fe_pipe p, int mode);
D2FE_API int d2fe_pipe_profile_read(d2fe_pipe p, float* ms /*[D2FE_PROF_COUNT]*/, int32_t* launches /*[D2FE_PROF_COUNT]*/);
/* left / right: `frames` gray u8 images each, image f at + f * image_stride, rows `stride` bytes apart.  Returns at once with a ticket
 * (0, 1, 2, ...).  Blocks only when the ticket's lane still holds the pass submitted `lanes` passes ago that nobody waited for. */
D2FE_API int d2fe_pipe_submit(d2fe_pipe p, const uint8_t* left, const uint8_t* right, int stride, size_t image_stride, int64_t* ticket);
/* Blocks until the ticket's frame is complete on the host (launching its pass first if coalescing still holds it back).  Tickets may be waited
 * for in any order, each within 2 * lanes passes (a pass = `coalesce` submits). */
D2FE_API int d2fe_pipe_wait(d2fe_pipe p, int64_t ticket, d2fe_pipe_result* out);

/* Device-side consumers of a ticket's results: the cross-agent exchange (pack -> all-gather -> gate -> remote matching; the reference broadcasts
 * the frame it has just extracted, loop_net.cpp:24-87, d2featuretracker.cpp:237-310) runs on a stream of its OWN, behind the extraction of the ticket and beside the
 * pipe's later passes -- the lanes' convolutions never wait for a collective.
 *   d2fe_pipe_device_view     launches the ticket's pass if coalescing still holds it back, makes `stream` (a hipStream_t, not NULL) wait for the ticket's SuperPoint and
 *                             NetVLAD results (not for its matcher or its D2H) and returns DEVICE pointers into the lane's result block: the same arrays, in the same
 *                             row order, as d2fe_pipe_result.  Read-only.  Valid until the matching release, at most 2 * lanes passes.
 *   d2fe_pipe_device_release  everything queued on `stream` so far is what read the view: the lane's next write of that block waits for it (an event, no host wait).
 * A block whose view has not been released when its lane comes round again (2 * lanes passes later) fails that submit with D2FE_ERR_INVALID (and, like every failed
 * submit, ends the pipe).  Use ONE consumer stream per pipe (the lane waits for the LAST release recorded for a block: consumers on several streams would have to order
 * those streams among themselves).  Both calls may come from a thread other than the submitting one.  Not available with netvlad_group > 1. */
typedef struct {
  int32_t frames, cap, desc_dim, netvlad_dim;
  const float* d_kps_xy;    /* [2 frames][cap][2] */
  const float* d_scores;    /* [2 frames][cap] */
  const float* d_desc;      /* [2 frames][cap][desc_dim] */
  const int32_t* d_n_kp;    /* [2 frames] */
  const float* d_netvlad;   /* [frames][netvlad_dim] or NULL */
} d2fe_pipe_device_result;
/* Where a stream of the CALLER's sits relative to the pipe's streams (same measurement as d2fe_pipe_stream_placement, against one lane stream per class; the pipe must be
 * idle; ~1 ms): *cls = the class it takes turns with, or -1 = none of the classes the lanes use.  A consumer that may choose among several streams (torch.cuda.Stream()
 * hands out pool streams) takes one of class -1, else one that only meets second (NetVLAD) streams. */
D2FE_API int d2fe_pipe_classify_stream(d2fe_pipe p, void* stream, int32_t* cls);
D2FE_API int d2fe_pipe_device_view(d2fe_pipe p, int64_t ticket, void* stream, d2fe_pipe_device_result* out);
D2FE_API int d2fe_pipe_device_release(d2fe_pipe p, int64_t ticket, void* stream);

/* The stream of the lane that ran the ticket's pass (a hipStream_t): work queued on it now runs behind that pass's matcher and D2H and in front of the lane's next
 * pass (lanes - 1 submits later) -- where the cross-agent exchange below goes by default.  d2fe_pipe_geometry / d2fe_pipe_handle: what such a consumer sizes its
 * buffers from, and the handle whose kernels it launches. */
D2FE_API int d2fe_pipe_lane_stream(d2fe_pipe p, int64_t ticket, void** stream);
D2FE_API int d2fe_pipe_geometry(d2fe_pipe p, int32_t* frames, int32_t* cap, int32_t* desc_dim, int32_t* netvlad_dim);
D2FE_API d2fe_handle d2fe_pipe_handle(d2fe_pipe p);

/* ---- Cross-agent exchange behind a pipe -------------------------------------------------------------------------------------
 * Replaces, per submitted stereo frame set: the LCM broadcast of the frame an agent has just extracted (LoopNet::broadcastVisualImageDescArray,
 * d2frontend/src/loop_net.cpp:24-87; wire precision VisualImageDesc::toLCM, d2common/include/d2common/d2frontend_types.h:228-268) and, on every receiver,
 * D2FeatureTracker::trackRemoteFrames (d2frontend/src/d2featuretracker.cpp:237-310: the NetVLAD gate of getMatchedPrevKeyframe :185-203, then matchKNN of the
 * local frame against the remote one).  One sequence per ticket, asynchronous, on ONE stream of the exchange's own (own_stream = 1, the default and the measured best
 * beside a two-lane pipe: profiles/r06_exchange_placement_ab.txt) or on the stream of the lane that produced the ticket (own_stream = 0: no further stream, but that
 * lane's next pass waits for the sequence):
 *   device view of the ticket -> pack one block per left frame (fp32, or the reference's int8 wire form) -> ONE all-gather over the communicator -> [int8: decode
 *   as the receiving constructor does, :319-338] -> counts -> NetVLAD gate of every (local frame f, remote frame f of rank r) pair -> ONE matcher launch (local
 *   descriptors read in place in the lane's result block, remote ones in place in the gathered blocks) -> release of the view -> ONE D2H into pinned slot `slot`.
 * The communicator is an RCCL ncclComm_t made by the caller (ncclCommInitRank in D2SLAM's own start-up code, or d2fe_rccl_comm_init_rank below: librccl is loaded
 * with dlopen, only when these entry points are used); every rank must enqueue its tickets in the same order.  A caller without RCCL (tests over gloo on one GPU)
 * passes comm = NULL and an all_gather callback.  Pair p = (rank-major over the OTHER ranks r, then frame f): local left frame f against frame f of rank r. */
typedef enum { D2FE_WIRE_FP32 = 0, D2FE_WIRE_INT8 = 1, D2FE_WIRE_INT8_RENORM256 = 2 } d2fe_wire;
typedef int (*d2fe_all_gather_fn)(void* user, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream);   /* 0 = ok; must be complete or stream-ordered on return */
typedef struct {
  int32_t struct_size;
  int32_t world, rank;          /* of the communicator */
  int32_t wire;                 /* d2fe_wire: fp32 blocks, the reference's int8 LCM precision (hard-coded 32-float renormalisation), or int8 + 256-float renormalisation */
  int32_t loopback;             /* the rank's OWN gathered blocks count as a remote agent too (how a one-rank communicator exercises the whole sequence) */
  int32_t slots;                /* ring of pinned result slots (>= the exchanges the caller keeps in flight) */
  int32_t own_stream;           /* 1 (default): one stream of the exchange's own; 0: each ticket's sequence on its lane's stream */
  int32_t timing;               /* 1: HIP events around the five phases (d2fe_exchange_result.phase_ms) */
  double gate_thres;            /* track_remote_netvlad_thres (d2featuretracker.cpp:199) */
  double ratio;                 /* knn_match_ratio */
  d2fe_all_gather_fn all_gather; void* all_gather_user;      /* used when the communicator is NULL */
  int32_t reserved[6];
} d2fe_exchange_config;
typedef struct {
  int64_t ticket;
  int32_t npairs, cap;
  const int32_t* q_idx;         /* [npairs][cap] local keypoint index   } host pointers into the pinned slot, valid until the slot is enqueued again */
  const int32_t* t_idx;         /* [npairs][cap] remote keypoint index  } */
  const float* dist;            /* [npairs][cap] */
  const int32_t* n_match;       /* [npairs] */
  const int32_t* gate_pass;     /* [npairs] 1 = the reference would have tracked this pair (NULL without NetVLAD) */
  const float* gate_sims;       /* [npairs] */
  int32_t gate_n;               /* pairs passing the gate */
  float phase_ms[5];            /* timing = 1: pack, all-gather, decode + counts + gate, remote matchKNN, release + D2H */
} d2fe_exchange_result;
typedef struct d2fe_exchange_s* d2fe_exchange;
D2FE_API void d2fe_exchange_default_config(d2fe_exchange_config* c);
D2FE_API int d2fe_exchange_create(d2fe_pipe p, void* nccl_comm, const d2fe_exchange_config* cfg, d2fe_exchange* out);
D2FE_API void d2fe_exchange_destroy(d2fe_exchange x);      /* before the pipe */
D2FE_API int d2fe_exchange_enqueue(d2fe_exchange x, int64_t ticket, int slot);       /* asynchronous; within 2 * lanes passes of the ticket's submit */
D2FE_API int d2fe_exchange_collect(d2fe_exchange x, int slot, d2fe_exchange_result* out);      /* blocks until the slot's results are in host memory */
D2FE_API int d2fe_exchange_pairs(d2fe_exchange x);
D2FE_API int d2fe_exchange_block_bytes(d2fe_exchange x);   /* bytes one frame contributes to the all-gather */
D2FE_API void* d2fe_exchange_stream(d2fe_exchange x);      /* own_stream = 1: that stream (hipStream_t), else NULL */
/* RCCL without any other dependency: rank 0 makes a 128-byte id, every rank gets it by whatever channel D2SLAM has (its LCM bus, a file, MPI), then all ranks call
 * comm_init_rank.  path: a librccl to load (NULL: one the process already holds, else librccl.so.1 / librccl.so on the loader path, else /opt/rocm/lib). */
D2FE_API int d2fe_rccl_load(const char* path);
D2FE_API const char* d2fe_rccl_path(void);
D2FE_API int d2fe_rccl_unique_id(void* id128);
D2FE_API int d2fe_rccl_comm_init_rank(const void* id128, int world, int rank, int device, void** comm_out);
D2FE_API int d2fe_rccl_comm_destroy(void* comm);

/* Half-image filter for quadcam neighbour matching.  Replaces getFeatureHalfImg
 * (d2featuretracker.cpp:1051-1075): map[c] = source index of the c-th kept keypoint; returns count in *n_out. */
D2FE_API int d2fe_half_image_filter(const float* pts_xy, int n, int require_left, int width_undistort,
                                    double undistort_fov, int32_t* map, int* n_out);

/* A1, variant A / NetVLAD image prep.  Replaces the cv::cvtColor(COLOR_BGR2GRAY) + cv::resize(INTER_LINEAR) block in front of
 * SuperPointONNX::infer and MobileNetVLADONNX::inference (superpoint_onnx.cpp:76-83, mobilenetvlad_onnx.h:51-59): channels = 1
 * (gray) or 3 (BGR, interleaved); the output is dw x dh gray u8, tight rows.  Same size and 1 channel = a copy.  Fused, one
 * pass; OpenCV's 8-bit fixed-point arithmetic (incl. its silent INTER_AREA for an exact 2x decimation). */
D2FE_API int d2fe_prepare_gray(d2fe_handle h, const uint8_t* src, int channels, int sw, int sh, int sstride, int dw, int dh,
                               uint8_t* dst);
D2FE_API int d2fe_prepare_gray_device(d2fe_handle h, const uint8_t* d_src, int n, int channels, int sw, int sh, int sstride,
                                      size_t src_image_stride, int dw, int dh, uint8_t* d_dst, void* stream);

/* ---- the components either side of the hot path --------------------------------------------------
 * (f)-1 Fisheye undistort + photometric gain.  Replaces FisheyeUndist::undist_id_cuda
 * (d2common/include/d2common/fisheye_undistort.h:152-176: cv::cuda::remap(INTER_LINEAR, constant 0 border) -> convertTo(32F)
 * -> cv::cuda::multiply(gain) -> convertTo(8U)) by one fused kernel.  mapx/mapy: dh*dw floats (source coordinates, the
 * reference's undistMapsGPUX/Y); gain: dh*dw floats or NULL.  Host-pointer form and device-resident form (n frames
 * sharing the maps, frame i at d_src + i*src_image_stride, output i at d_dst + i*dh*dw). */
D2FE_API int d2fe_undistort(d2fe_handle h, const uint8_t* src, int sw, int sh, int sstride, const float* mapx,
                            const float* mapy, const float* gain, int dw, int dh, uint8_t* dst);
D2FE_API int d2fe_undistort_device(d2fe_handle h, const uint8_t* d_src, int n, int sw, int sh, int sstride,
                                   size_t src_image_stride, const float* d_mapx, const float* d_mapy, const float* d_gain,
                                   int dw, int dh, uint8_t* d_dst, void* stream);

/* (f)-1, map generation.  FisheyeUndist::generateCylinderMap + genOneUndistMap (fisheye_undistort.h:458-500,559-613): a
 * virtual camodocal::CylindricalCamera (fx = fy = width / (fov_deg * pi/180), cx = width/2, cy = height/2) is lifted
 * (CylindricalCamera.cc:207-220) and projected through the fisheye model; the pinhole form is the other genOneUndistMap
 * (:615-660, the five virtual cameras of generateAllUndistMap :346-456): objPoint = q * (x - width/2, y - height/2, f).
 * The fisheye model is camodocal's CataCamera ("omni" + "radtan" in config/quadcam/quad_cam_calib-*.yaml; spaceToPlane
 * CataCamera.cc:495-515).  Maps are width*height floats each, written to HBM (device form) or copied back (host form). */
typedef struct {
  double xi, k1, k2, p1, p2, gamma1, gamma2, u0, v0;   /* kalibr order: intrinsics [xi fu fv pu pv], distortion [k1 k2 p1 p2] */
} d2fe_mei_camera;
D2FE_API int d2fe_gen_cylinder_map(d2fe_handle h, const d2fe_mei_camera* cam, int width, int height, double fov_deg,
                                   float* mapx, float* mapy);
D2FE_API int d2fe_gen_cylinder_map_device(d2fe_handle h, const d2fe_mei_camera* cam, int width, int height, double fov_deg,
                                          float* d_mapx, float* d_mapy, void* stream);
D2FE_API int d2fe_gen_pinhole_map(d2fe_handle h, const d2fe_mei_camera* cam, const double* q_wxyz, int width, int height,
                                  double f, float* mapx, float* mapy);
D2FE_API int d2fe_gen_pinhole_map_device(d2fe_handle h, const d2fe_mei_camera* cam, const double* q_wxyz, int width, int height,
                                         double f, float* d_mapx, float* d_mapy, void* stream);

/* (f)-2 NetVLAD keyframe database.  Replaces faiss::IndexFlatIP (members d2frontend/include/d2frontend/loop_detector.h:71-72;
 * add d2frontend/src/loop_detector.cpp:254-263; search :318) and the gate of LoopDetector::queryIndexFromDatabase
 * (:300-350).  Vectors live in HBM; a search is one streaming pass over the database. */
typedef struct d2fe_db* d2fe_db_handle;
D2FE_API int d2fe_db_create(d2fe_handle h, int dim, int capacity, d2fe_db_handle* out);
D2FE_API void d2fe_db_destroy(d2fe_db_handle db);
D2FE_API int d2fe_db_ntotal(d2fe_db_handle db);
D2FE_API int d2fe_db_add(d2fe_db_handle db, const float* vecs, int n);        /* IndexFlatIP::add; returns first new label or <0 */
/* IndexFlatIP::search(nq, q, k, sims, labels): k best by inner product, descending (ties: lower label); -1 pads. */
D2FE_API int d2fe_db_search(d2fe_db_handle db, const float* q, int nq, int k, float* sims, int32_t* labels);
/* queryIndexFromDatabase (loop_detector.cpp:300-350) minus the ROS bookkeeping: k = min(5 + max_index, ntotal) nearest;
 * the first with label <= ntotal - max_index and similarity > thres is returned in *label (else -1) with its similarity. */
D2FE_API int d2fe_db_query_gated(d2fe_db_handle db, const float* q, int max_index, double thres, int32_t* label, float* sim);

/* (f)-3 int8 wire codec.  Replaces the quantisation in VisualImageDesc::toLCM (d2common/include/d2common/d2frontend_types.h:
 * 228-237 landmark descriptors, float max; 260-268 NetVLAD, double max: pass double_max = 1) and the decode of the LCM
 * constructor (:313-351): x = q/127.0; landmark_num >= 0: the first landmark_num 32-float segments are re-normalised
 * (the reference's hard-coded 32); landmark_num < 0: whole-vector L2 (global descriptor). */
D2FE_API int d2fe_quantize_int8(d2fe_handle h, const float* x, int n, int double_max, int8_t* out);
D2FE_API int d2fe_dequantize_int8(d2fe_handle h, const int8_t* q, int n, int landmark_num, float* out);

/* (f)-4 LK optical-flow tracker (d2frontend/src/opticaltrack_utils.cpp).  A d2fe_lk_frame is the device-resident image
 * pyramid the reference keeps in LKImageInfoGPU::pyr (opticaltrack_utils.h:16-23): level 0 = the gray frame, level l+1 =
 * cv::cuda::pyrDown(level l) (buildImagePyramid, opticaltrack_utils.cpp:526-542; PYR_LEVEL = 2, opticaltrack_utils.h:10). */
typedef struct d2fe_lk_frame_s* d2fe_lk_frame;
D2FE_API int d2fe_lk_frame_create(d2fe_handle h, const uint8_t* gray, int width, int height, int stride, int levels,
                                  d2fe_lk_frame* out);
D2FE_API int d2fe_lk_frame_create_device(d2fe_handle h, const uint8_t* d_gray, int width, int height, int stride, int levels,
                                         void* stream, d2fe_lk_frame* out);
D2FE_API void d2fe_lk_frame_destroy(d2fe_lk_frame f);
/* copy pyramid level `level` (tight rows) to the host; returns bytes or <0 */
D2FE_API long d2fe_lk_frame_read_level(d2fe_lk_frame f, int level, uint8_t* dst, size_t max_bytes, int* width, int* height);
/* The tracking block of opticalflowTrackPyr (opticaltrack_utils.cpp:236-272) in one launch: SparsePyrLKOpticalFlow(win, levels,
 * iters, useInitialFlow).calc(prev, cur) from cur_init, reverse calc(cur, prev) from the result shifted back by move_cols
 * (type 1 LEFT_RIGHT_IMG_MATCH: -move_cols, 2 RIGHT_LEFT_IMG_MATCH: +move_cols, 0 WHOLE_IMG_MATCH: none), status[i] = both
 * succeeded && |prev - reverse| <= 0.5 && inBorder(cur) (:35-41).  cur_pts[n][2] and status[n] are written for every point;
 * the reduceVector() compaction stays with the caller.  Reference constants: win 21 (WIN_SIZE, :25), iters 30 (:239). */
D2FE_API int d2fe_lk_track(d2fe_handle h, d2fe_lk_frame prev, d2fe_lk_frame cur, const float* prev_pts, const float* cur_init,
                           int n, int type, float move_cols, int win, int iters, float* cur_pts, uint8_t* status);
/* Batched form: all the tracks of one frame set in ONE launch and one H2D/D2H pair (quadcam: 4 temporal tracks + the
 * left/right neighbour tracks of trackLK, d2featuretracker.cpp:472-621).  Points are concatenated; pair p owns points
 * [first, first + count); every point must belong to exactly one pair. */
typedef struct {
  d2fe_lk_frame prev, cur;
  int32_t first, count;
  int32_t type;          /* 0 WHOLE_IMG_MATCH, 1 LEFT_RIGHT_IMG_MATCH, 2 RIGHT_LEFT_IMG_MATCH */
  float move_cols;
} d2fe_lk_pair;
D2FE_API int d2fe_lk_track_batch(d2fe_handle h, const d2fe_lk_pair* pairs, int npairs, const float* prev_pts,
                                 const float* cur_init, int n_total, int win, int iters, float* cur_pts, uint8_t* status);
/* detectFastByRegion (opticaltrack_utils.cpp:444-493): cv::cuda::FastFeatureDetector(threshold, nonmax, TYPE_9_16,
 * max_npoints = features) on each of the cols x rows regions of level 0, sorted by response, top `features`.
 * response (optional) receives the FAST scores. */
D2FE_API int d2fe_detect_fast_by_region(d2fe_handle h, d2fe_lk_frame f, int features, int cols, int rows, int threshold,
                                        float* pts_xy, int32_t* response, int cap, int* n_out);
/* cv::cuda::createGoodFeaturesToTrackDetector(type, max_corners, quality, min_dist)->detect (detectPoints,
 * opticaltrack_utils.cpp:404-412): min-eigenvalue corners (blockSize 3, Sobel 3), eig > quality * max, 3x3 local maxima,
 * sorted by eigenvalue, host min-distance grid filter. */
D2FE_API int d2fe_good_features_to_track(d2fe_handle h, d2fe_lk_frame f, int max_corners, double quality, double min_dist,
                                         float* pts_xy, int cap, int* n_out);

/* ---- cross-agent exchange --------------------------------------------------------------------------------
 * Replaces the LCM broadcast of a keyframe's image descriptor (d2frontend/src/loop_net.cpp:24-87: landmark positions, scores,
 * SuperPoint descriptors, NetVLAD descriptor) by one fixed-capacity block per frame, so that the exchange is ONE all-gather:
 *   block (float words) = desc[cap][256] | kps[cap][2] | scores[cap] | netvlad[G] | n (int32) | zero padding to a multiple of 256.
 * Descriptors come first and the size is a multiple of 256 words: a gathered block's descriptors are addressable by
 * d2fe_match_batch_device (offsets in rows of `dim` floats) without unpacking. */
D2FE_API int d2fe_block_words(int cap, int netvlad_dim);                 /* words of one block */
D2FE_API int d2fe_block_field_offset(int cap, int netvlad_dim, int field);   /* word offset of field 0 desc, 1 kps, 2 scores, 3 netvlad, 4 n */
/* Packs nframes frames of a d2fe_superpoint_extract_device result (dense [rows][cap][...] arrays; frame f is row row0 + f*row_step)
 * and of a d2fe_netvlad_device result (d_netvlad [nframes][G], may be NULL) into d_blocks [nframes][d2fe_block_words]. */
D2FE_API int d2fe_pack_blocks_device(d2fe_handle h, const float* d_desc, const float* d_kps_xy, const float* d_scores,
                                     const int32_t* d_n, const float* d_netvlad, int row0, int row_step, int nframes, int cap,
                                     int netvlad_dim, float* d_blocks, void* stream);
/* The same block in the reference's WIRE precision.  VisualImageDesc::toLCM quantises the descriptors to int8 before the broadcast
 * (d2common/include/d2common/d2frontend_types.h:228-237: one float maximum over the frame's landmark descriptors; :260-268: the NetVLAD
 * vector with a double maximum; scores are not sent) and the receiving constructor decodes them (:319-338: q / 127.0, the first
 * landmark_num 32-float segments re-normalised -- the reference's hard-coded 32 --, the NetVLAD vector normalised as a whole).
 *   int8 block (bytes) = desc_q[cap][256] | netvlad_q[G] | kps f32[cap][2] | n int32 | zero padding to a multiple of 64   (3.9x smaller)
 * d2fe_pack_blocks_int8_device = the quantisation, on the sender; d2fe_unpack_blocks_int8_device expands gathered int8 blocks into the
 * fp32 block layout above (scores = 0) with the decode arithmetic: renorm 0 = as the reference (landmark_num = n), 1 = every descriptor
 * re-normalised over its 256 floats.  Gate and matcher then read exactly what a receiving agent of the reference would hold. */
D2FE_API int d2fe_block_bytes_int8(int cap, int netvlad_dim);
D2FE_API int d2fe_pack_blocks_int8_device(d2fe_handle h, const float* d_desc, const float* d_kps_xy, const int32_t* d_n,
                                          const float* d_netvlad, int row0, int row_step, int nframes, int cap, int netvlad_dim,
                                          int8_t* d_blocks, void* stream);
D2FE_API int d2fe_unpack_blocks_int8_device(d2fe_handle h, const int8_t* d_blocks_int8, int nblocks, int cap, int netvlad_dim, int renorm,
                                            float* d_blocks, void* stream);
/* NetVLAD gate of a pair list.  Replaces the similarity test of D2FeatureTracker::getMatchedPrevKeyframe
 * (d2frontend/src/d2featuretracker.cpp:185-203: `vlad_desc.dot(vlad_desc_remote) < track_remote_netvlad_thres` rejects) and of
 * LoopDetector::queryIndexFromDatabase (loop_detector.cpp:339).  Pair p compares row d_pair_q[p] of d_q (rows q_stride words apart)
 * with row d_pair_db[p] of d_db.  Outputs (each may be NULL): d_pass[p] = 1/0, d_sims[p], *d_n_pass += passing pairs (the caller
 * zeroes it), and d_cnt_inout[p] = 0 for a rejected pair -- with the matcher's a_cnt array there, d2fe_match_batch_device returns
 * no matches for pairs the reference would not have tracked. */
D2FE_API int d2fe_gate_pairs_device(d2fe_handle h, const float* d_q, size_t q_stride, const float* d_db, size_t db_stride, int dim,
                                    const int32_t* d_pair_q, const int32_t* d_pair_db, int npairs, double thres,
                                    int32_t* d_cnt_inout, int32_t* d_pass, float* d_sims, int32_t* d_n_pass, void* stream);

/* The same gate for a FOURCORNER_FISHEYE (quadcam) agent: getMatchedPrevKeyframe's second branch (d2featuretracker.cpp:212-233) compares
 * view 2 of the REMOTE quad frame with the local keyframe's views in the order dirs = {2, 3, 0, 1} and stops at the first whose similarity
 * is not below thres (dir_b = dirs[j]); trackRemoteFrames (:282-297) then tracks the four view pairs (remote view a = (2+k)%4, local view
 * (dir_b - 2 + a) % 4).  Job j = (local quad frame, remote quad frame): the NetVLAD vector of local view v is row d_job_local_row0[j] +
 * v*local_view_step of d_local (rows local_stride words apart), of remote view v row d_job_remote_row0[j] + v*remote_view_step of d_remote.
 * Outputs (each may be NULL): d_dir_prev[j] = dir_b or -1; d_sims[j][4] = the similarities for dirs[0..3]; *d_n_pass += passing jobs;
 * d_cnt_inout[j*16 + local_view*4 + remote_view] = 0 for every view pair the reference would NOT track (the 16 view pairs of a job laid
 * out as 16 consecutive matcher problems: with the matcher's a_cnt there, only the reference's four pairs are matched). */
D2FE_API int d2fe_quad_gate_device(d2fe_handle h, const float* d_local, size_t local_stride, const float* d_remote, size_t remote_stride,
                                   int dim, const int32_t* d_job_local_row0, const int32_t* d_job_remote_row0, int local_view_step,
                                   int remote_view_step, int njobs, double thres, int32_t* d_dir_prev, float* d_sims,
                                   int32_t* d_cnt_inout, int32_t* d_n_pass, void* stream);

/* ---- quadcam neighbour matching on the device (A12) -------------------------------------------------------------------------------
 * d2fe_half_image_compact_device = getFeatureHalfImg (d2frontend/src/d2featuretracker.cpp:1051-1075) for a batch of jobs, plus the
 * a-side shift of matchLocalFeatures (:1161-1170).  Job j reads row d_job_row[j] of the dense extract outputs (d_desc [rows][cap][dim],
 * d_pts_xy [rows][cap][2], d_n [rows]), keeps x < W_u - move_cols (d_job_left[j] != 0) or x >= move_cols (move_cols = (float)(W_u * 90.0 /
 * undistort_fov), as the reference computes it), and writes, in order: d_out_desc [njobs][cap][dim], d_out_pts [njobs][cap][2] with
 * d_job_shift_x[j] added to x (pass +move_cols / -move_cols / 0), d_out_map [njobs][cap] (compacted index -> original index), d_out_n [njobs].
 * d2fe_half_move_cols returns that float.  cap <= 1024.
 * d2fe_remap_matches_device = the index remap of :1178-1181 for a batch of pairs: q_idx[p][i] = map[d_map_a_job[p]][q_idx[p][i]], same for
 * t_idx with d_map_b_job (d_maps [njobs][cap_map], match arrays [npairs][cap_match]). */
D2FE_API float d2fe_half_move_cols(int width_undistort, double undistort_fov);
D2FE_API int d2fe_half_image_compact_device(d2fe_handle h, const float* d_desc, const float* d_pts_xy, const int32_t* d_n,
                                            const int32_t* d_job_row, const int32_t* d_job_left, const float* d_job_shift_x, int njobs,
                                            int cap, int dim, int width_undistort, double undistort_fov, float* d_out_desc,
                                            float* d_out_pts, int32_t* d_out_map, int32_t* d_out_n, void* stream);
D2FE_API int d2fe_remap_matches_device(d2fe_handle h, int32_t* d_q_idx, int32_t* d_t_idx, const int32_t* d_n_match,
                                       const int32_t* d_map_a_job, const int32_t* d_map_b_job, const int32_t* d_maps, int npairs,
                                       int cap_match, int cap_map, void* stream);

/* Test hooks and kernel diagnostics (d2fe_debug_*) are NOT part of this library: they live in the development library
 * (lib/libd2fe_hip_dev.so, built with -DD2FE_DEVTOOLS) and are declared in include/d2fe_debug.h. */

/* Stage timing with HIP events recorded on the stream the kernels run on.
 * mode 0 = off, 1 = bracket only the dominant kernel (conv1b), 2 = bracket every stage.
 * d2fe_profile_read synchronises, returns the accumulated milliseconds and launch counts per stage since the
 * last d2fe_profile_enable call, and clears them.  Stage order: D2FE_PROF_* below. */
enum {
  D2FE_PROF_CONV1A = 0, D2FE_PROF_CONV1B, D2FE_PROF_CONV2A, D2FE_PROF_CONV2B, D2FE_PROF_CONV3A, D2FE_PROF_CONV3B,
  D2FE_PROF_CONV4A, D2FE_PROF_CONV4B, D2FE_PROF_CONVPADA, D2FE_PROF_CONVPB, D2FE_PROF_CONVDB, D2FE_PROF_SOFTMAX,
  D2FE_PROF_SELECT, D2FE_PROF_SAMPLE, D2FE_PROF_MATCH, D2FE_PROF_NETVLAD, D2FE_PROF_COUNT
};
D2FE_API int d2fe_profile_enable(d2fe_handle h, int mode);
D2FE_API int d2fe_profile_read(d2fe_handle h, float* ms /*[D2FE_PROF_COUNT]*/, int32_t* launches /*[D2FE_PROF_COUNT]*/);

/* Synchronise the handle's stream (for timing with device-resident calls). */
D2FE_API int d2fe_sync(d2fe_handle h);

#ifdef __cplusplus
}
#endif
#endif /* D2FE_H_ */
