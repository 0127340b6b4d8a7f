/*
 * mi355_yolo.h -- C ABI of the MI355X-native YOLO detect/pose engine (libmi355yolo.so).
 *
 * This is the drop-in boundary for the ONE hot path of
 * cthadeufaria/computer-vision-shoplifting-detection: the per-frame Ultralytics call made at
 *   /root/reference/model.py:18   self.model = YOLO("./models/yolov5mu.pt")
 *   /root/reference/model.py:38   self.model.track(frame, persist=True, show=False, classes=[0], verbose=False)
 *   /root/reference/model.py:40   results[0].boxes
 * (driven per decoded frame by /root/reference/preprocess.py:38-47).  The arithmetic behind those
 * calls lives in the un-vendored ultralytics==8.3.225 (requirements.txt:121); each entry point below
 * names the Ultralytics function it replaces.  Plain pointers and sizes only: no torch / numpy /
 * C++ types cross this boundary.  The Python facade (cvsd_amd.YOLO) binds it with ctypes; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (MI355_OK) or a negative MI355_E* code; mi355_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - the caller owns every input/output buffer; the engine owns device memory, its HIP stream and
 *     its weight copies.  One handle = one device = one in-order stream; infer calls on one handle
 *     are NOT re-entrant (the reference is single-threaded; Ultralytics serialises predict with a lock).
 *   - zero detections is not an error: the image's count is 0.
 */
#ifndef MI355_YOLO_H
#define MI355_YOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_OK            0
#define MI355_EINVAL       -1   /* bad argument (null pointer, non-positive size, capacity too small ...) */
#define MI355_EIO          -2   /* weights file missing / unreadable */
#define MI355_EFORMAT      -3   /* not a .mi355w file, or unsupported version / program */
#define MI355_EHIP         -4   /* a HIP runtime call failed (message has the hipError string) */
#define MI355_ENOMEM       -5

#define MI355_TASK_DETECT   0
#define MI355_TASK_POSE     1
#define MI355_MAX_KPT_FLOATS 51   /* 17 keypoints x (x, y, conf) */

typedef struct mi355_yolo mi355_yolo;   /* opaque engine handle */

/* Engine options; zero-initialise and set struct_size = sizeof(mi355_opts). 0 / NULL means "default".  A caller built against an
 * earlier header (smaller struct_size) keeps working: fields beyond its struct_size take their defaults.
 * Every behaviour-changing knob of the engine is here; the MI355_* environment variables listed in INTEGRATION.md section 5 are
 * A/B overrides for kernel experiments (tools/), not configuration. */
#define MI355_OPT_NO_FUSE_UPSAMPLE  0x01   /* always launch the nearest-2x upsample kernel (never read through the consuming 1x1 conv) */
#define MI355_OPT_NO_FUSE_1X1       0x02   /* never run a Conv3x3 -> Conv1x1 pair as one launch */
#define MI355_OPT_NO_FUSE_TAIL      0x04   /* never run a C2f tail (last Bottleneck conv + C2f.cv2) as one launch */
#define MI355_OPT_NO_GROUPS         0x08   /* no grouped launches / step schedule in the latency-bound regime (<= 5 frames per pass) */
#define MI355_OPT_NO_MEM_REUSE      0x10   /* one region per graph tensor instead of the liveness-shared arena */
#define MI355_OPT_HIP_GRAPH         0x20   /* replay stem..decode of a chunk as a hipGraph (measured slower on ROCm 7.2; off) */
#define MI355_OPT_NO_DIRECT_ROWS    0x40   /* small calls: copy rows to the host instead of writing them from the NMS kernel */
#define MI355_OPT_NO_PASS_TUNE      0x80   /* autotuner: skip the whole-pass re-check of fusion / group decisions */
typedef struct mi355_opts {
    int struct_size;
    int batch_chunk;      /* frames pushed through the net per pass (default 64); larger batches are looped */
    int half;             /* 1 = Ultralytics' half=True (engine/predictor.py: model.half(), im.half()): activations (the /255
                           * input included) and weights stored as fp16, fp32 accumulate / bias / SiLU, head logits, decode and NMS in fp32.
                           * Results then differ from the fp32 path by fp16 rounding (not a bit-exact mode). */
    int fast_act;         /* fp32 path only. 0 (default): the canonical SiLU x / (1 + exp(-x)) with a reproducible exp and IEEE division --
                           * results are bit-identical to oracle/det_oracle.c.  1: v_exp_f32 / v_rcp_f32 SiLU in the conv epilogues (as the
                           * half path has it): ~1e-6 relative per activation, NOT bit-exact, tolerance-tested against float64. */
    int autotune;         /* candidate launch plans timed per conv when a shape is first seen: 0 = default (32; 28 for half), -1 = off
                           * (the planner's static guess), n > 0 = n */
    int streams;          /* HIP streams the ops of a pass of >= 6 frames are dealt to along the dependency DAG: 0 = default (4), 1 = one */
    int flags;            /* MI355_OPT_* */
    int reserved;
    const char* plan_dir;        /* read-only directory of SHIPPED launch-plan files (cvsd_amd/plans): looked up first, so that every process
                                  * on the same GPU model runs the same launch sequence without timing anything.  NULL = none */
    const char* plan_cache_dir;  /* writable directory where freshly timed choices are kept (and found again): NULL = ~/.cache/mi355yolo,
                                  * "" = do not persist */
} mi355_opts;

/* One post-NMS detection, coordinates in ORIGINAL-image pixels (after scale_boxes / scale_coords).
 * Replaces one row of Results.boxes.data (+ Results.keypoints.data) -- ultralytics/engine/results.py. */
typedef struct mi355_det {
    float x1, y1, x2, y2;
    float conf;
    int   cls;
    int   anchor_idx;                      /* index of the source anchor in [0, A): P3 row-major, then P4, then P5 */
    float kpt[MI355_MAX_KPT_FLOATS];       /* pose only: x, y, conf per keypoint; zeros for detect models */
} mi355_det;

typedef struct mi355_model_info {
    int task;             /* MI355_TASK_* */
    int nc;               /* number of classes */
    int nkpt, kdim;       /* keypoint shape (17, 3) for pose, (0, 0) for detect */
    int reg_max;          /* 16 */
    int n_levels;
    int strides[4];       /* 8, 16, 32 */
    int n_convs, n_ops, n_buffers;
    long long n_params;   /* fused parameter count incl. the 16 DFL weights (what Ultralytics' model.info() prints) */
    long long macs_640;   /* multiply-accumulates of all convs for one 640x640 frame */
    char family[8];       /* "v8" | "v5u" */
    char scale;           /* 'n','s','m','l','x' */
    char pad_[7];
} mi355_model_info;

/* Timing of the last infer call on this handle (HIP events on the engine's own stream). */
typedef struct mi355_timing {
    float total_ms;       /* preprocess .. NMS, device time, whole call */
    float conv_ms;        /* sum over the implicit-GEMM conv launches (the dominant kernel) */
    float stem_ms, pool_ms, upsample_ms, letterbox_ms, decode_ms, nms_ms;
    int   conv_launches;
    int   frames;
} mi355_timing;

const char* mi355_last_error(void);

/* YOLO(path) -- ultralytics/engine/model.py:Model.__init__ + nn/tasks.py load + autobackend fuse.
 * `weights_path` is a .mi355w file (fused weights + op program, written by cvsd_amd.weights). */
int  mi355_yolo_create(const char* weights_path, int device_id, const mi355_opts* opts, mi355_yolo** out);
/* Same, from an in-memory image of the file (what a rank receives from the RCCL weight broadcast). */
int  mi355_yolo_create_from_memory(const void* blob, size_t nbytes, int device_id, const mi355_opts* opts,
                                   mi355_yolo** out);
void mi355_yolo_destroy(mi355_yolo* h);
int  mi355_yolo_info(const mi355_yolo* h, mi355_model_info* info);

/* model(frames, conf=, iou=, classes=, max_det=, imgsz=) -- engine/predictor.py:stream_inference:
 * LetterBox + BGR->RGB + /255 (preprocess), the fused conv graph (nn/tasks.py:_predict_once), Detect/Pose
 * decode (nn/modules/head.py), utils/nms.py:non_max_suppression, utils/ops.py:scale_boxes/scale_coords.
 *   bgr_nhwc           n frames of h x w x 3 uint8, BGR (cv2 layout), HOST memory; consecutive rows are
 *                      row_stride_bytes apart (0 = w*3) and consecutive frames h*row_stride_bytes apart
 *   conf, iou          thresholds (Ultralytics predict defaults 0.25 / 0.7; .track() uses conf 0.1)
 *   classes,n_classes  keep only detections whose best class is listed (NULL / 0 = all)   [classes=[0] at model.py:38]
 *   max_det            at most this many rows per image (default 300)
 *   imgsz              letterbox target (default 640; rectangular "auto" padding to a multiple of 32)
 *   out_rows           [n * out_capacity_per_image] rows, image i at out_rows + i*out_capacity_per_image,
 *                      confidence-descending (the NMS keep order)
 *   out_counts         [n] rows written per image
 */
int  mi355_yolo_infer(mi355_yolo* h, const uint8_t* bgr_nhwc, int n, int height, int width, int row_stride_bytes,
                      float conf, float iou, const int* classes, int n_classes, int max_det, int imgsz,
                      mi355_det* out_rows, int out_capacity_per_image, int* out_counts);
/* Same with the frames already resident in DEVICE memory (dense n x h x w x 3); outputs still go to host. */
int  mi355_yolo_infer_device(mi355_yolo* h, const uint8_t* bgr_nhwc_dev, int n, int height, int width,
                             float conf, float iou, const int* classes, int n_classes, int max_det, int imgsz,
                             mi355_det* out_rows, int out_capacity_per_image, int* out_counts);

/* Asynchronous device-output form (multi-GPU pipelines: the per-rank rows are gathered over RCCL straight from HBM while the
 * next batch computes; replaces the reference's per-frame `results[0].boxes` hand-over, model.py:40, for a whole shard).
 * Frames AND outputs live in DEVICE memory owned by the caller; nothing crosses PCIe and the call returns as soon as the work
 * is enqueued on the engine's stream:
 *   rows_dev    [n * max_det] rows, PACKED in frame order: frame 0's counts[0] rows, then frame 1's, ...
 *   counts_dev  [n] rows kept per frame;   total_dev [1] = sum(counts) = number of valid packed rows
 * The buffers are valid once the stream has passed this call: wait with mi355_yolo_sync(), or order another HIP stream
 * behind mi355_yolo_stream() (a hipStream_t; e.g. torch.cuda.ExternalStream).  The next infer call on the handle waits for
 * a pending asynchronous one before it reuses the engine's scratch, so callers double-buffer the output buffers only. */
int  mi355_yolo_infer_device_async(mi355_yolo* h, const uint8_t* bgr_nhwc_dev, int n, int height, int width,
                                   float conf, float iou, const int* classes, int n_classes, int max_det, int imgsz,
                                   mi355_det* rows_dev, int* counts_dev, int* total_dev);
void* mi355_yolo_stream(mi355_yolo* h);
int  mi355_yolo_sync(mi355_yolo* h);

/* The decoded pre-NMS head tensor, exactly the layout Detect/Pose.forward returns: out[n][4+nc+nk][A] fp32
 * (xywh in letterboxed pixels, sigmoid class scores, decoded keypoints).  For parity tests.
 * out may be NULL to query *out_channels / *out_anchors for the given frame size. */
int  mi355_yolo_raw_head(mi355_yolo* h, const uint8_t* bgr_nhwc, int n, int height, int width, int row_stride_bytes,
                         int imgsz, float* out, int* out_channels, int* out_anchors);

/* Launch plans of the shape last run: plan_hash identifies the candidate lists AND the choice per conv (two runs with the same
 * hash launch the same kernels with the same grids); source 0 = static guess (autotune off), 1 = this process's memory,
 * 2 = this machine's plan cache, 3 = timed now, 4 = shipped plan file (opts.plan_dir); launches = kernel launches of one pass (stem .. last conv; decode and NMS follow);
 * activation_bytes = device memory held by the activation arena (buffers whose lifetimes cannot overlap under any legal
 * schedule share bytes), activation_bytes_unshared = what one region per graph tensor would take.  Any out pointer may be NULL. */
int  mi355_yolo_plan_info(const mi355_yolo* h, unsigned long long* plan_hash, int* source, int* launches,
                          long long* activation_bytes, long long* activation_bytes_unshared);
/* Host-only: where the activation buffers of a weight image would live for n frames of height x width per pass (byte offsets
 * into the engine's arena and sizes, per program buffer, in .mi355w buffer order) -- the liveness-based placement the engine
 * uses, computed without a device.  reuse = 0: one region per buffer.  For the memory planner's unit tests. */
int  mi355_memory_plan(const void* blob, size_t nbytes, int n, int height, int width, int imgsz, int half, int reuse,
                       long long* offsets, long long* sizes, int cap, int* n_buffers, long long* arena_bytes,
                       long long* unshared_bytes);
/* Per-kernel-kind timing costs two HIP events per launch; off by default (total_ms is always measured). */
int  mi355_yolo_set_profiling(mi355_yolo* h, int on);
int  mi355_yolo_last_timing(const mi355_yolo* h, mi355_timing* t);

/* ---- single-operator entry points (host pointers; used by the parity tests to isolate a kernel) -------- */

/* Fused conv + bias (+SiLU) (+residual), NHWC fp32 -- ultralytics/nn/modules/conv.py:Conv.forward_fuse.
 * x[n][h][w][cin] dense, w_oihw[cout][cin][k][k], bias[cout], residual (or NULL) and y dense [n][ho][wo][cout];
 * k in {1,3}, stride in {1,2}, pad = k/2.  plan_index selects one of the engine's candidate launch plans (tile shape,
 * wave arrangement, v1 / v2 staging; 0 = the default guess, taken modulo the number of candidates, which is returned
 * through n_plans when non-NULL): every plan must give the same bits. */
int  mi355_op_conv2d(int device_id, const float* x, int n, int h, int w, int cin, const float* w_oihw,
                     const float* bias, int cout, int k, int stride, int silu, const float* residual, float* y,
                     int plan_index, int* n_plans);
/* The same operator on the half=True path: x, w and residual are rounded to fp16 (nearest-even) on the way in, the
 * kernel accumulates in fp32 and rounds y to fp16 once (out_f32 = 1: y is written as fp32, as for the head's final
 * convs); y is handed back as fp32 values. */
int  mi355_op_conv2d_f16(int device_id, const float* x, int n, int h, int w, int cin, const float* w_oihw,
                         const float* bias, int cout, int k, int stride, int silu, const float* residual, float* y,
                         int out_f32, int plan_index, int* n_plans);
/* Pointwise conv over the channel concatenation of a nearest-2x upsampled half-resolution tensor and a full-resolution one, with the
 * upsample fused into the conv's read side, as the engine runs the neck's Upsample -> Concat -> C2f.cv1 (fp32; ultralytics
 * nn/modules: nn.Upsample(None, 2, "nearest"), Concat, C2f.cv1): x_half[n][h/2][w/2][up_c] (up_c a multiple of 16),
 * x_skip[n][h][w][skip_c], w[cout][up_c + skip_c][1][1] -> y[n][h][w][cout].  Must equal mi355_op_conv2d on the materialised
 * concatenation, bit for bit, for every plan_index. */
int  mi355_op_conv1x1_upcat(int device_id, const float* x_half, const float* x_skip, int n, int h, int w, int up_c, int skip_c,
                            const float* w_oihw, const float* bias, int cout, int silu, float* y, int plan_index, int* n_plans);
/* The same on the half=True path (operands rounded to fp16 on the way in, fp32 accumulation, y rounded to fp16 once): must equal
 * mi355_op_conv2d_f16 on the materialised concatenation, bit for bit, for every plan_index. */
int  mi355_op_conv1x1_upcat_f16(int device_id, const float* x_half, const float* x_skip, int n, int h, int w, int up_c, int skip_c,
                                const float* w_oihw, const float* bias, int cout, int silu, float* y, int plan_index, int* n_plans);
/* Conv3x3 (stride 1|2, pad 1) + bias + SiLU -> Conv1x1 + bias (+SiLU if silu2) run as ONE fused launch, the way the engine runs
 * a 3x3 conv whose only reader is a pointwise conv (the first conv's output never leaves the chip).  x[n][h][w][cin],
 * w1[c1][cin][3][3], b1[c1], w2[c2][c1][1][1], b2[c2] -> y[n][h/stride][w/stride][c2]; must equal the two convs run separately,
 * bit for bit, for every plan_index. */
int  mi355_op_conv2d_fused(int device_id, const float* x, int n, int h, int w, int cin, const float* w1_oihw, const float* b1,
                           int c1, int stride, const float* w2_oihw, const float* b2, int c2, int silu2, float* y,
                           int plan_index, int* n_plans);
/* The tail of a C2f block as ONE fused launch (fp32): the last Bottleneck's second conv, y1 = SiLU(conv3x3(x) + b1) + residual,
 * and C2f.cv2 over the concatenation, y = SiLU(conv1x1(cat(lead, y1)) + b2) -- lead[n][h][w][lead_c] are the earlier slices of
 * the block's concat buffer (lead_c a multiple of 16, as is c1), w2[c2][lead_c + c1][1][1]; residual (or NULL) [n][h][w][c1].
 * Must equal mi355_op_conv2d (3x3, residual) followed by mi355_op_conv2d (1x1 on the concatenation), bit for bit, for every plan. */
int  mi355_op_c2f_tail(int device_id, const float* x, int n, int h, int w, int cin, const float* w1_oihw, const float* b1, int c1,
                       const float* residual, const float* lead, int lead_c, const float* w2_oihw, const float* b2, int c2,
                       float* y, int plan_index, int* n_plans);
/* The same fused pair on the half=True path (operands rounded to fp16 on the way in, the intermediate image rounded to fp16 as the
 * unfused launch would have stored it, fp32 accumulation; out_f32 = 1: y written as fp32, as for the head's final convs): must
 * equal mi355_op_conv2d_f16 of the 3x3 conv followed by mi355_op_conv2d_f16 of the 1x1, bit for bit, for every plan_index. */
int  mi355_op_conv2d_fused_f16(int device_id, const float* x, int n, int h, int w, int cin, const float* w1_oihw, const float* b1,
                               int c1, int stride, const float* w2_oihw, const float* b2, int c2, int silu2, float* y, int out_f32,
                               int plan_index, int* n_plans);
/* Two independent convs of one input x[n][h][w][cin] (weights wa / wb, SiLU, no residual) as ONE grouped launch -- the way the
 * engine runs independent convs of one step of the op DAG in the latency-bound regime (the head branches beside the neck).
 * plan_a / plan_b pick among each conv's candidate plans whose kernel instance is on the group kernel's menu (their counts come
 * back through n_menu_a / n_menu_b); ya / yb must equal mi355_op_conv2d of each conv alone, bit for bit, for every pair.
 * cout2_a > 0: conv a is a 3x3 conv with a pointwise conv w2a[cout2_a][cout_a][1][1] / b2a (no activation) fused behind it, as in
 * mi355_op_conv2d_fused, and ya holds that pointwise conv's output [n][h/stride_a][w/stride_a][cout2_a]. */
int  mi355_op_conv2d_group(int device_id, const float* x, int n, int h, int w, int cin, const float* wa, const float* ba,
                           int cout_a, int k_a, int stride_a, const float* wb, const float* bb, int cout_b, int k_b,
                           int stride_b, float* ya, float* yb, int plan_a, int plan_b, int* n_menu_a, int* n_menu_b,
                           const float* w2a, const float* b2a, int cout2_a);
/* Device-resident timing of one conv launch plan on random data (diagnostics / tuning): average milliseconds over
 * `iters` back-to-back launches of candidate plan `plan_index`; plan_desc (optional) receives a description. */
int  mi355_bench_conv2d(int device_id, int n, int h, int w, int cin, int cout, int k, int stride, int silu, int residual,
                        int plan_index, int iters, float* avg_ms, int* n_plans, char* plan_desc, int plan_desc_len);
int  mi355_bench_conv2d_f16(int device_id, int n, int h, int w, int cin, int cout, int k, int stride, int silu,
                            int residual, int plan_index, int iters, float* avg_ms, int* n_plans, char* plan_desc,
                            int plan_desc_len);
/* Host-only query of the launch planner (nothing is launched or allocated): the kernel version of every candidate launch
 * plan for a conv of this shape over buffers with these pixel strides (elements); versions[i] = 1 conv_igemm (LDS-staged),
 * 3 streaming pointwise, 4 pipelined pointwise, 6 split-K, + 100 when the plan carries the fused pointwise stage
 * (f2_cout > 0).  res_cs = 0: no residual.  For the planner's unit tests (address-range guards). */
int  mi355_plan_query(int n, int h, int w, int cin, int cout, int k, int stride, int src_cs, int dst_cs, int res_cs,
                      int f2_cout, int f2_dst_cs, int half, int* versions, int cap, int* n_plans);
/* HOST computation (no GPU): pyramidal Lucas-Kanade optical flow of n points between two gray uint8 frames [height][width] --
 * the cv2.calcOpticalFlowPyrLK step of BoT-SORT's global motion compensation (ultralytics/trackers/utils/gmc.py, reached
 * from /root/reference/model.py:38).  win x win windows (odd), max_level + 1 pyramid levels, at most max_iters iterations or
 * |step| <= eps, points whose window's minimum eigenvalue is below min_eig or which leave the frame get status 0.
 * pts / next_pts: [n][2] (x, y).  Returns 0, or -1 on a bad argument. */
int  mi355_gmc_pyr_lk(const uint8_t* prev, const uint8_t* cur, int height, int width, const float* pts, int n, int win,
                      int max_level, int max_iters, double eps, double min_eig, float* next_pts, uint8_t* status);
/* The same on GPU `device` (csrc/gmc_kernels.hip: one wavefront per point, float64, pyramids built on the GPU; host buffers in and
 * out).  On the host the 1000-corner budget costs 10-27 ms per frame -- thirty times the detector pass -- so model.track() hands
 * its tracker the engine's device.  Results agree with the host routine to rounding (window sums are associated differently).
 * win <= 21 (the default window).  Returns 0, -1 (bad argument) or -2 (HIP error). */
int  mi355_gmc_pyr_lk_device(int device, const uint8_t* prev, const uint8_t* cur, int height, int width, const float* pts, int n, int win,
                             int max_level, int max_iters, double eps, double min_eig, float* next_pts, uint8_t* status);
/* Frame preparation of the same motion compensation on GPU `device`: BGR frame -> gray plane of oh x ow (cv2.cvtColor(BGR2GRAY) +
 * cv2.resize(INTER_LINEAR) in their fixed-point arithmetic; xtab / ytab = per output column / row {source index, tap 0, tap 1} with
 * 11-bit taps, unused when oh x ow is the frame size), the float32 min-eigenvalue map of cv2.goodFeaturesToTrack (3x3 Sobel, 3x3
 * block) and the 0/1 mask of the corners it keeps (quality threshold against the map's maximum, 3x3 non-maximum suppression,
 * border excluded); the caller orders the kept corners by strength.  Host buffers in and out.  0, -1 (bad argument), -2 (HIP error). */
int  mi355_gmc_prepare_device(int device, const uint8_t* bgr, int height, int width, int oh, int ow, const int* xtab, const int* ytab,
                              double quality, uint8_t* gray_out, float* eig_out, uint8_t* ok_out);
/* One motion-compensation step as two calls, so that it runs BESIDE the detector pass on a stream of its own: model.track() enqueues
 * the step for a frame before the detector runs on it and collects it when the tracker asks for the warp.  The object keeps the
 * previous frame's pyramid on the device.  step_begin = mi355_gmc_prepare_device of `bgr` plus, when n_prev > 0, Lucas-Kanade tracking
 * of prev_pts from the previous step's plane into this one (that step must have prepared a plane of the same oh x ow); it returns at
 * once and reads nothing of its arguments afterwards.  step_finish waits and fills gray / eig / ok [oh * ow] and, when the step had
 * points, next_pts [n_prev][2] and status [n_prev].  One step may be pending per object.  0, -1 (bad argument / order), -2 (HIP). */
/* The two small host stages behind the GPU step, as gmc.py states them in numpy: goodFeaturesToTrack's ordering of the kept corners
 * (strongest first, raster order among equals, at most max_corners; returns the count written to xy_out [max_corners][2]) and
 * cv2.estimateAffinePartial2D (RANSAC over 2-point similarities + refit; src / dst float64 [n][2]; returns 1 and fills H_out [6] and
 * inliers_out [n] (may be NULL), 0 when no transform was found, -1 on bad arguments; draws from its own seeded generator). */
int  mi355_gmc_order_corners(const float* eig, const uint8_t* ok, int height, int width, int max_corners, float* xy_out);
int  mi355_gmc_affine_partial(const double* src, const double* dst, int n, double threshold, double confidence, int max_iters,
                              unsigned long long seed, double* H_out, uint8_t* inliers_out);
typedef struct mi355_gmc mi355_gmc;
int  mi355_gmc_create(int device, mi355_gmc** out);
void mi355_gmc_destroy(mi355_gmc* g);
int  mi355_gmc_step_begin(mi355_gmc* g, const uint8_t* bgr, int height, int width, int oh, int ow, const int* xtab, const int* ytab,
                          double quality, const float* prev_pts, int n_prev, int win, int max_level, int max_iters, double eps, double min_eig);
int  mi355_gmc_step_finish(mi355_gmc* g, uint8_t* gray_out, float* eig_out, uint8_t* ok_out, float* next_pts, uint8_t* status);
/* The pending step's frame as it sits on the device (dense BGR [height][width][3]; valid until the next step_begin / track_begin): the
 * detector pass of the same frame reads this copy through mi355_yolo_infer_device instead of uploading the frame a second time, which is
 * also what lets the two overlap on the GPU (model.track, /root/reference/model.py:38).  Blocks the host until the upload has landed. */
int  mi355_gmc_pending_frame(mi355_gmc* g, const uint8_t** dev_bgr, int* height, int* width);
/* ---- the tracker behind model.track (/root/reference/model.py:38-46), host C++ (csrc/tracker_host.cpp) -------------------------------------
 * Ultralytics' default botsort.yaml tracker, one call per frame: BYTETracker.update's two-stage association on IoU cost fused with the
 * detection score, KalmanFilterXYWH, STrack.multi_gmc's warp of the predicted states, lap.lapjv(extend_cost=True, cost_limit=thresh).
 *   det       [n][6] float32 rows x1, y1, x2, y2, conf, cls (Results.boxes.data of the frame, conf >= 0.1 under .track)
 *   warp      6 doubles, row-major 2 x 3: the camera motion since the previous frame (GMC.apply), or NULL = none
 *   out_rows  [cap][8] float32 rows x1, y1, x2, y2, id, score, cls, idx (idx = row of det); the box is the Kalman state
 * mi355_tracker_update returns the number of confirmed tracks (rows beyond cap are not written), or -1 on a bad argument.  It must be
 * called on EVERY frame, empty ones included (frame counter, lost-track ageing against track_buffer 30). */
typedef struct mi355_tracker mi355_tracker;
int  mi355_tracker_create(int frame_rate, mi355_tracker** out);
void mi355_tracker_destroy(mi355_tracker* t);
int  mi355_tracker_update(mi355_tracker* t, const float* det, int n, const double* warp, float* out_rows, int cap);
int  mi355_tracker_last_rows(const mi355_tracker* t, float* out_rows, int cap);   /* the rows of the last update again (count returned) */
int  mi355_tracker_state(const mi355_tracker* t, int* frame_id, int* ids_issued, int* n_tracked, int* n_lost);
/* which = 0: tracked (confirmed or awaiting confirmation), 1: lost.  Rows of 16 doubles: id, state (1 tracked, 2 lost, 3 removed),
 * confirmed, frame of birth, last matched frame, score, cls, idx, mean[8] = cx cy w h + velocities.  Returns the count. */
int  mi355_tracker_tracks(const mi355_tracker* t, int which, double* out, int cap);
/* The pieces, for known-answer tests: the filter on (mean[8], cov[8][8] row-major) in place, and the assignment solver --
 * lap.lapjv(cost [n_rows][n_cols], extend_cost=True, cost_limit): x_out[i] = column of row i or -1, y_out[j] = row of column j or -1. */
int  mi355_kalman_initiate(const double* z_xywh, double* mean, double* cov);
int  mi355_kalman_predict(double* mean, double* cov);
int  mi355_kalman_update(double* mean, double* cov, const double* z_xywh);
int  mi355_kalman_warp(double* mean, double* cov, const double* warp);
int  mi355_lapjv(const double* cost, int n_rows, int n_cols, double cost_limit, int* x_out, int* y_out);
/* Host form of mi355_gmc_prepare_device (csrc/gmc_host.cpp): the same expressions in the same order, hence the same plane and corners. */
int  mi355_gmc_prepare_host(const uint8_t* bgr, int height, int width, int oh, int ow, const int* xtab, const int* ytab, double quality,
                            uint8_t* gray_out, float* eig_out, uint8_t* ok_out);
/* The whole step of GMC.apply_sparseoptflow (ultralytics/trackers/utils/gmc.py) on the object, which keeps the previous frame's plane and
 * ordered corners: track_begin enqueues frame preparation + Lucas-Kanade from the previous frame's corners (downscale 2 is Ultralytics'),
 * track_finish collects, orders the new corners, estimates the partial affine transform (RANSAC over the tracked pairs when more than 4
 * survive) and writes the 2 x 3 matrix, translation in frame pixels (the identity on a first frame / too few points).  An object made by
 * mi355_gmc_create(-1) runs every stage in host C++ and never touches a GPU.  track_state: the previous frame as held (tests). */
int  mi355_gmc_track_begin(mi355_gmc* g, const uint8_t* bgr, int height, int width, int downscale);
int  mi355_gmc_track_finish(mi355_gmc* g, double* H_out);
/* n consecutive frames in ONE call (a batched sweep holds a detector batch's frames before the tracker needs their warps): all n frame
 * preparations as one set of launches, all n Lucas-Kanade steps as one launch, corner ordering and RANSAC on host threads.  H_out [n][6] is
 * bit for bit what n track_begin / track_finish steps return; the object's previous frame is continued from and left behind. */
int  mi355_gmc_track_batch(mi355_gmc* g, const uint8_t* const* frames, int n, int height, int width, int downscale, double* H_out);
/* The frames a mi355_gmc_track_batch call (on another thread, or returned) has uploaded, for the detector pass of the same batch to read in
 * place: waits up to timeout_ms for an upload numbered above after_seq (= mi355_gmc_batch_seq taken before that call was started), then
 * frame f is at dev + f * stride on the object's GPU (dense BGR; stride == height * width * 3 when that is a multiple of 16).  Valid until
 * the next mi355_gmc_track_batch on the object.  0 = ok, 1 = nothing new within the timeout (cvsd_amd/sweep.py: the production sweep of
 * /root/reference/preprocess.py:36-47). */
unsigned long long mi355_gmc_batch_seq(mi355_gmc* g);
int  mi355_gmc_batch_frames(mi355_gmc* g, unsigned long long after_seq, int timeout_ms, const uint8_t** dev, int* n, int* height, int* width,
                            long long* stride);
int  mi355_gmc_track_reset(mi355_gmc* g);
int  mi355_gmc_track_state(const mi355_gmc* g, int* oh, int* ow, int* n_pts, uint8_t* gray_out, float* pts_out, int pts_cap);
/* The u8 stem: letterboxed BGR frames -> (x/255, RGB) -> conv k x k stride s (pad k/2, or 2 for k=6) + bias + SiLU. */
int  mi355_op_stem(int device_id, const uint8_t* bgr, int n, int h, int w, const float* w_oihw, const float* bias,
                   int cout, int k, int stride, float* y);
/* The half=True form of the same op (input, weights and output rounded to fp16 as the half predictor does, fp32 accumulation); y holds
 * fp16 bit patterns.  variant: 0 = the kernel the engine launches, 1 = the general kernel, 2 = the k 3 / stride 2 kernel. */
int  mi355_op_stem_f16(int device_id, const uint8_t* bgr, int n, int h, int w, const float* w_oihw, const float* bias,
                       int cout, int k, int stride, int variant, uint16_t* y);
/* data/augment.py:LetterBox on uint8 BGR frames (cv2.resize INTER_LINEAR fixed-point + 114 border).
 * out must hold n * out_h * out_w * 3 bytes where (out_h,out_w) = mi355_letterbox_shape(). */
int  mi355_letterbox_shape(int height, int width, int imgsz, int* out_h, int* out_w);
int  mi355_op_letterbox(int device_id, const uint8_t* bgr, int n, int height, int width, int imgsz, uint8_t* out);
/* utils/nms.py:non_max_suppression on a decoded head tensor pred[n][4+nc+extra][A] (Ultralytics layout).
 * Rows come back in letterboxed pixels (no scale-back): x1,y1,x2,y2,conf,cls,anchor_idx, kpt = the `extra` columns. */
int  mi355_op_nms(int device_id, const float* pred, int n, int nc, int extra, int anchors, float conf, float iou,
                  const int* classes, int n_classes, int max_det, mi355_det* out_rows, int out_capacity_per_image,
                  int* out_counts);

#ifdef __cplusplus
}
#endif
#endif /* MI355_YOLO_H */
