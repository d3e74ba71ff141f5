/*
 * openglottal_hip.h — C-ABI of the MI355X-native glottal segmentation hot path.
 *
 * The reference (hari-krishnan/openglottal) is pure Python and has no FFI; the
 * "plugin/operator API" for this path is the Python call surface of
 *   openglottal/models/unet.py:36-88      UNet(in_ch,out_ch,features), .load_state_dict, __call__
 *   openglottal/utils.py:218-241          unet_segment_frame(frame_gray, model, device, threshold)
 *   openglottal/features.py:234-245       per-frame area = sum(mask>0) [inside the YOLO box]
 *   scripts/benchmark_video_speed.py:89-109  timed frame loop
 * Each entry point below names the reference lines it replaces.  A maintainer
 * binds them with ctypes/cffi (INTEGRATION.md shows the stub); the in-tree host
 * mirror (openglottal_amd/unet.py, utils.py, features.py) does exactly that.
 *
 * Conventions
 *  - every call returns 0 on success or a negative OG_E* code; nothing throws
 *    across the ABI; og_last_error() returns a thread-local string.  Every kernel
 *    launch reports its own failure (OG_EHIP from the call that issued it) and every
 *    launch's grid / LDS / workspace / counter bounds are decided on the host before
 *    anything runs (og_unet_plan() walks those decisions without a device).  What the
 *    library cannot turn into a code: a GPU memory fault -- the HIP runtime answers
 *    it with abort() in the process.  A call that fails half-way leaves the handle
 *    usable (arrival counters restored on every error path).
 *  - the calling thread's current HIP device is restored on every exit path.
 *  - host buffers are caller-owned and only read/written during the call.
 *  - *_dev entry points take DEVICE pointers (hipMalloc'd or a torch tensor's
 *    data_ptr()), enqueue on the handle's stream and return without syncing;
 *    call og_unet_sync() before reading results.
 *  - a handle is not thread-safe; distinct handles are independent.
 *  - tensors are plain C arrays: NCHW float32 for og_unet_forward_f32 (as the
 *    reference's torch tensors), [B,H,W] uint8 for frames and masks.
 */
#ifndef OPENGLOTTAL_HIP_H
#define OPENGLOTTAL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OG_OK 0
#define OG_EINVAL (-1)   /* bad argument / shape / key */
#define OG_ESTATE (-2)   /* call order (e.g. forward before finalize) */
#define OG_EHIP (-3)     /* HIP runtime error, see og_last_error() */
#define OG_ENOMEM (-4)
#define OG_ENODEV (-5)   /* no usable gfx950 device */
#define OG_ERANGE (-6)   /* split precision only: an activation left the f16 range; the call's results are invalid */

#define OG_DTYPE_F32 0
#define OG_DTYPE_I64 1

typedef struct og_unet og_unet;

/* Library / device ------------------------------------------------------- */
const char* og_last_error(void);
const char* og_version(void);
int og_device_count(void);                 /* >=0, or negative error */
int og_init(int device);                   /* select device (replaces torch.device(...), cli.py:56) */

/* Device memory helpers (plumbing for callers without torch) -------------- */
void* og_malloc(size_t bytes);             /* NULL on failure */
int og_free(void* dptr);
int og_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int og_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);

/* U-Net handle ------------------------------------------------------------ */
/* = UNet.__init__ (unet.py:50-72).  features[n_levels]; in_ch and out_ch must be 1
 * (the only configuration the reference's pipelines construct, cli.py:61). */
og_unet* og_unet_create(const int* features, int n_levels, int in_ch, int out_ch);
void og_unet_destroy(og_unet* h);

/* = load_state_dict, one tensor per call (cli.py:62-64).  The library COPIES.
 * key: reference state_dict key, e.g. "downs.0.net.0.weight".  Unknown key or
 * wrong shape -> OG_EINVAL.  "*.num_batches_tracked" (int64) accepted, ignored. */
int og_unet_set_tensor(og_unet* h, const char* key, const void* host, const int64_t* shape, int ndim, int dtype);

/* Strict completeness check, BatchNorm fold (eval mode, eps 1e-5, in float64),
 * repack into the kernels' MFMA fragment layout, upload.  = .eval().to(device) */
int og_unet_finalize(og_unet* h);

/* = UNet.forward (unet.py:74-88): x [B,1,H,W] f32 (host) -> logits [B,1,H,W] f32 (host).
 * H and W must be multiples of 2^n_levels (the reference's bilinear fallback,
 * unet.py:84-85, never fires there).  Synchronous. */
int og_unet_forward_f32(og_unet* h, const float* x_nchw, int B, int H, int W, float* logits_nchw);

/* = unet_segment_frame (utils.py:218-241) over a batch of frames already at
 * network resolution, fused with the area count of features.py:238 / :241-245.
 *   gray   [B,H,W] u8              (required)
 *   boxes  [B,4] int32 x1,y1,x2,y2 or NULL; x1<0 marks "no detection" -> area 0
 *   mask   [B,H,W] u8 in {0,255}   or NULL
 *   area   [B] int32               or NULL   (count of mask>0, inside box if given)
 *   logits [B,H,W] f32             or NULL   (debug / parity)
 * Host-pointer variant is synchronous (H2D, run, D2H). */
int og_unet_segment_u8(og_unet* h, const uint8_t* gray, int B, int H, int W, float threshold,
                       const int32_t* boxes, uint8_t* mask, int32_t* area, float* logits);
/* The frame loop of extract_features_unet with the video on the host (features.py:226,234-245), STREAMED: frames [B,H,W]
 * gray (channels = 1) or [B,H,W,3] BGR (channels = 3; cv2.cvtColor(BGR2GRAY) of features.py:235 runs on the device) go up
 * in micro-batches through a ring of pinned buffers, the copy of micro-batch k+1 and the return of micro-batch k-1 under the
 * kernel chain of micro-batch k; device memory is bounded by (lanes + 2) micro-batches whatever B is (the reference loads
 * every frame first, utils.py:43-54).  Pageable or pinned (hipHostMalloc / torch pin_memory) host memory; synchronous;
 * boxes / mask / area as og_unet_segment_u8.  og_unet_segment_u8 itself runs on this engine (option "stream" 0: one-shot). */
int og_unet_stream_u8(og_unet* h, const uint8_t* frames, int B, int H, int W, int channels, float threshold,
                      const int32_t* boxes, uint8_t* mask, int32_t* area);
/* The same engine for a video held the way the reference holds it: a LIST of separately allocated frames (`frames_bgr`,
 * utils.py:43-54 / features.py:226,234).  frame_ptrs[i] -> frame i, [H,W] gray or [H,W,3] BGR u8, contiguous.  Each frame is
 * copied ONCE, straight into the pinned ring slot of its micro-batch, while the device works on the micro-batches before it:
 * no stacked copy of the video is ever made (a Python np.stack of 502 frames costs more than a fifth of the whole pass). */
int og_unet_stream_frames_u8(og_unet* h, const uint8_t* const* frame_ptrs, int B, int H, int W, int channels, float threshold,
                             const int32_t* boxes, uint8_t* mask, int32_t* area);
/* Same as og_unet_segment_u8, DEVICE pointers, asynchronous on the handle's stream. */
int og_unet_segment_u8_dev(og_unet* h, const uint8_t* gray_dev, int B, int H, int W, float threshold,
                           const int32_t* boxes_dev, uint8_t* mask_dev, int32_t* area_dev, float* logits_dev);

/* Box-gated recount on masks already on the device (features.py:244-245 when the
 * boxes only become known after the sequential TemporalDetector pass). */
int og_mask_area_dev(og_unet* h, const uint8_t* mask_dev, int B, int H, int W, const int32_t* boxes_dev, int32_t* area_dev);

/* `cv2.cvtColor(frm_bgr, cv2.COLOR_BGR2GRAY)` (openglottal/features.py:235) on the HOST, for callers that keep the reference's
 * per-frame loop (one gray frame per `unet_segment_frame` call): n_pixels BGR triples -> n_pixels gray bytes, OpenCV's published u8
 * path (15-bit fixed point, R 9798 / G 19235 / B 3735, (x + 2^14) >> 15) -- the arithmetic of k_bgr2gray.  No device involved.
 * PARITY UNPINNED against OpenCV (absent from the reference's snapshot and from the build image). */
int og_bgr2gray_host(const uint8_t* bgr, long long n_pixels, uint8_t* gray);

/* = cv2.cvtColor(frame, COLOR_BGR2GRAY) (features.py:235) for [B,H,W,3] u8 on the device. */
int og_bgr2gray_dev(og_unet* h, const uint8_t* bgr_dev, int B, int H, int W, uint8_t* gray_dev);

/* YOLO-Crop+UNet on the device (scripts/eval_girafe.py:127-159 `unet_on_crop`, utils.py:97-186): per frame,
 * crop boxes[b] from gray[b], letterbox it NEAREST into a size x size tile (zeros around), segment, project
 * the tile mask back NEAREST and paste it into a zero frame.  geom[b] = {pad_top, pad_left, content_h,
 * content_w} as `letterbox_with_info` returns them (host computes them: Python round()).  Boxes must be
 * inside the frame; x1 < 0 or an empty box gives an all-zero mask.  size must be a multiple of 2^n_levels.
 * The host variant returns OG_EINVAL for a box that reaches outside the frame or a geom that does not fit the tile; the
 * device variant cannot read them on the host, so its kernels treat such a record as "no detection" (all-zero mask)
 * instead of indexing out of bounds. */
int og_unet_segment_crops_u8(og_unet* h, const uint8_t* gray, int B, int H, int W, const int32_t* boxes, const int32_t* geom,
                             int size, float threshold, uint8_t* out_masks);
int og_unet_segment_crops_u8_dev(og_unet* h, const uint8_t* gray_dev, int B, int H, int W, const int32_t* boxes_dev,
                                 const int32_t* geom_dev, int size, float threshold, uint8_t* tiles_scratch_dev,
                                 uint8_t* tile_masks_scratch_dev, uint8_t* out_masks_dev);

/* BAGLS front end (scripts/eval_bagls.py:46-70,153-155 `letterbox`): frames of mixed sizes, packed back to back in one
 * buffer (frame b = shapes[b] = {H, W} pixels of `channels` bytes starting at byte offsets[b]), are scaled so that the
 * longest side equals `size` and padded symmetrically with `value`: INTER_NEAREST for channels = 1 (gray frames, GT masks),
 * INTER_LINEAR for channels = 3 (BGR), as the reference's letterbox chooses.  geom[b] = {pad_top, pad_left, content_h,
 * content_w} as the host computes them (Python round()).  out [B,size,size,channels].  OpenCV's index / coefficient rules
 * as restated in openglottal_amd/geometry.py (parity unpinned: cv2 is absent). */
int og_canvas_letterbox_u8(og_unet* h, const uint8_t* packed, const int64_t* offsets, const int32_t* shapes, int B, int channels,
                           int size, const int32_t* geom, int value, uint8_t* out);
int og_canvas_letterbox_u8_dev(og_unet* h, const uint8_t* packed_dev, const int64_t* offsets_dev, const int32_t* shapes_dev, int B,
                               int channels, int size, const int32_t* geom_dev, int value, uint8_t* out_dev);

/* Per-frame confusion counts for `frame_metrics` (scripts/eval_bagls.py:75-87, eval_girafe.py:113-124) on resident masks:
 * stats[b] = {tp, n_pred, n_gt} int32 (fp = n_pred - tp, fn = n_gt - tp).  boxes (or NULL): the prediction counts only
 * inside the box, x1 < 0 = empty prediction -- the "yolo+unet" row (eval_bagls.py:201-207). */
int og_mask_stats_dev(og_unet* h, const uint8_t* pred_dev, const uint8_t* gt_dev, int B, int H, int W, const int32_t* boxes_dev,
                      int32_t* stats_dev);

int og_unet_sync(og_unet* h);                /* also reports OG_ERANGE of asynchronous split-precision work */
void* og_unet_stream(og_unet* h);          /* hipStream_t the handle launches on */

/* Micro-batch the frame loop uses per kernel chain (default 32); >=1. */
int og_unet_set_chunk(og_unet* h, int frames_per_launch);
/* Allocate the activation arenas of every lane for micro-batches of up to `frames_per_launch` frames of H x W ahead of
 * time.  Optional: the entry points grow them on demand (one hipFree/hipMalloc + graph re-capture when a call needs more
 * bytes than any before it); calls at other frame sizes or smaller micro-batches re-plan inside the existing allocation and
 * keep their captured graphs, so reserving the largest shape once makes a mixed-size stream allocation-free. */
int og_unet_reserve(og_unet* h, int frames_per_launch, int H, int W);
/* 1 = replay captured hipGraphs for repeated shapes (default), 0 = eager launches. */
int og_unet_set_graphs(og_unet* h, int enable);

/* Kernel-selection knobs for A/B measurements (results are bit-identical across them, except "precision", "wino" and "splitk"):
 * "conv_impl" 0|1, "tps_nt1" 1|3|9, "tps_nt2" 1|3, "wg_per_cu" 1|2, "prio_mode" 0|1|2|3, "tile_h" 0|8|16,
 * "splitk_fused" 0|1 (the last-arriving K part of a tile reduces all parts in split order and runs the epilogue; 0: separate reduce launch),
 * "splitk_occ" 0|1 (K parts of a split launch on the occupancy kernel; 0: persistent kernel), "splitk_slots" 1..4 and "splitk_div" 1..8
 * (its target workgroups per CU / split when the launch fills less than 1/div of them),
 * "occ_min_pct" 0..400 (occupancy kernel when a launch has at least that many workgroups per 100 CUs), "convt_occ" 0|1, "fuse_first" 0|1 (first layer computed inside downs.0's second conv), "fuse_head" 0|1 (head +
 * threshold + area inside the last conv's epilogue), "keep_taps" 0|1, "precision" 0|1 (NOT bit-identical: 0 = exact f32, the default and the parity reference; 1 = opt-in split precision -- activations and weights as f16 hi/lo pairs, three v_mfma_f32_32x32x16_f16 per f32 product, f32 accumulation; passes the reference fixtures at the f32 tolerance, 2.5x faster; an activation beyond the f16 range makes the call fail with OG_ERANGE), "h_square" 0|1 (its wave tiling), "stream" 0|1 (og_unet_segment_u8 through the streaming engine, default 1; 0 = whole batch staged at once), "dual" 0|1 (micro-batches of one call alternate over extra lanes = streams/arenas, so that launch tails overlap) with "lanes" 0..3 (0 = 3 lanes up to 16 frames per launch, else 2), and
 * "wino_w" 0..4 [1] (under-filled Winograd launches -- one frame per kernel chain, the reference's call pattern utils.py:235-237 --
 * on finer tiles with the 16 positions split over the four WAVES of a workgroup, V transformed in registers, accumulators exchanged
 * through LDS (k_conv_wino_w), deep layers with a tile's four position rows on four workgroups (k_conv_wino_wp); the same sums as
 * k_conv_wino, bit for bit; 0 off, 1 auto -- by tiles, channels and the number of lanes in flight --, 2 / 3 force k_conv_wino_w with
 * one / two 32-window blocks per workgroup, 4 forces k_conv_wino_wp), "convt_w" 0|1 [1] (under-filled transposed convs on
 * 32-pixel x 32-column wave tiles, k_convt_w: bit-identical),
 * "wino_ps" 0..4 [1] (round 3's form of the same, used when "wino_w" is 0: a tile's 16 positions over 16 / PN workgroups,
 * k_conv_wino_ps: the same sums, bit for bit; 0 off, 1 auto, 2 / 3 / 4 force PN = 4 / 2 / 1),
 * "zero_copy" 0|1 [1] (host entry points, calls of ONE micro-batch of at most 4 frames -- the reference's per-frame call,
 * utils.py:235-237: the first kernel reads the frame from, and the last ones write the mask / area to, the engine's pinned host
 * buffers directly instead of through H2D / D2H / memset commands, each of which costs a ~5 us launch floor on the one stream:
 * 0.370 -> 0.358 ms per unet_segment_frame; results identical),
 * "inject_fault" n (TEST HOOK: the n-th conv launch from now on fails with OG_EHIP after scribbling over the arrival counters of the
 * fused reduces; every error path restores them -- tests/test_gpu_recovery.py).
 *
 * The two options that DO change the arithmetic:
 * "wino" 0|1 [1]: the form of the HANDLE.  1 = every 3x3 conv whose map tiles (H, W multiples of 16; 32 rows for 32-column layers)
 * runs in Winograd F(2x2,3x3) form, all f32 (16 MFMA multiplies per 2x2 output window and channel pair instead of 36, transforms in
 * f32 adds) AT EVERY MICRO-BATCH SIZE, one frame per call included; 0 = the direct kernels everywhere.  Which layers take which form
 * depends on the options and (H, W) only, never on B, the lane, the shard or the entry point: a frame's mask is a function of the frame
 * alone, as in the reference's per-frame loop (features.py:234-238).  Both forms pass the same reference fixtures inside the
 * reference's own run-to-run noise band.  "wino_first" 0|1 (first layer unfused so that the second conv takes the Winograd kernel).
 * "splitk" 0|1 [0]: OPT-IN, non-canonical.  1 = launches that would fill < 1/div of the chip split K across workgroups ("wino" 0 layers
 * and transposed convs); sums are taken in a fixed order, so results are deterministic, but they differ in the last bits from the
 * unsplit order -- a frame's logits then depend on how many frames share its launch.  "splitk_nt1" 0|1 (split launches of 64-column
 * layers on 32-column tiles), "splitk_min_steps" 1..9. */
int og_unet_set_option(og_unet* h, const char* name, int value);

/* HIP-event timing on the handle's stream (bench.py's roofline leg). */
int og_timer_start(og_unet* h);
int og_timer_stop(og_unet* h, float* elapsed_ms);   /* records, synchronises, returns ms since start */

/* Parity/debug: copy a layer-boundary activation of the LAST forward/segment call
 * (first `B` frames of the last chunk) to host as NCHW f32.  Names follow
 * tests/golden: "downs.0.a", "downs.0.b", "pool0", "bottleneck.a", "ups.0", "ups.1.a", ...
 * Writes C,H,W into dims[3]; returns OG_EINVAL if capacity_floats is too small. */
int og_unet_get_activation(og_unet* h, const char* name, int B, float* out_nchw, size_t capacity_floats, int* dims);

/* Roofline leg of bench.py: run the chain for B frames `reps` times EAGERLY with a HIP event
 * pair around every kernel launch (on the handle's stream) and return, per launch in chain
 * order: layer name and kernel symbol (64-byte slots), mean duration in ms, and the launch's
 * algorithmic FLOPs (2 x MACs with the true, unpadded channel counts). */
int og_unet_profile(og_unet* h, const uint8_t* gray_dev, int B, int H, int W, int reps, int max_entries,
                    char* layers, char* kernels, float* ms, double* flops, int* n_entries);

/* Diagnostic: run one eager chain with {s_memtime, s_memrealtime} stamps at entry/exit of every
 * persistent conv workgroup and return the median in-kernel shader clock (MHz) per launch, in
 * chain order (entry 0 = first layer: 0).  Tells "pipe saturated at a DVFS-lowered clock" from
 * "pipe idle" (MI355X_MICROARCH.md, DVFS give-back item 6).  Not used on the product path. */
int og_unet_clock_probe(og_unet* h, const uint8_t* gray_dev, int B, int H, int W, int max_entries, double* mhz, int* n_entries);

/* Raw stamps {memtime0, realtime0, memtime1, realtime1} x 1024 workgroups of launch `entry` of the last probe. */
int og_unet_clock_probe_raw(og_unet* h, int entry, unsigned long long* out4x1024);

/* Algorithmic work of one forward at HxW (conv + convT + head MACs x2), for rooflines. */
double og_unet_flops_per_frame(og_unet* h, int H, int W);

/* Dry run of the kernel chain of one micro-batch (B u8 frames of H x W, areas wanted) for a net of the given widths, WITHOUT a
 * device: the library's own launch decisions (the code path of og_unet_segment_u8_dev's run_chunk) are executed on a host-only
 * handle with every launch recorded instead of issued.  `options` = "name=value,name=value" (og_unet_set_option names), `lanes` =
 * lanes of the call the micro-batch belongs to (1..3: a scheduling hint of the under-filled launches).  `out` receives one line per
 * launch, "kernel|grid.x|grid.y|grid.z|block|lds_bytes|workspace_bytes|arrival_counters"; returns the number of launches or a
 * negative error code.  Nothing here has a counterpart in the reference (it launches nothing by hand); it exists so that the bounds
 * every launch must respect -- og_workspace_limit() -- can be checked for every micro-batch size, layer shape and forced option on a
 * machine without a GPU (tests/test_launch_plan.py), instead of being found by a faulting kernel. */
int og_unet_plan(const int* features, int n_levels, int B, int H, int W, int lanes, const char* options, char* out, size_t cap,
                 long long* arena_bytes);
/* which: 0 split-K / position-split workspace bytes per lane, 1 arrival counters per lane, 2 largest grid.y / grid.z, 3 LDS bytes per workgroup */
long long og_workspace_limit(int which);

/* YOLOv8 detector ----------------------------------------------------------
 * Replaces `self.model(frame_bgr, conf=self.conf, verbose=False)` + `boxes.conf.argmax()` /
 * `boxes.xyxy[idx]` of TemporalDetector.detect (openglottal/models/detector.py:58-64), i.e. the
 * ultralytics predictor (BGR->RGB, /255, YOLOv8n, Detect decode) reduced to what the reference
 * consumes: the highest-confidence box per frame.  ultralytics is a third-party dependency that
 * the reference neither vendors nor pins: architecture restated from its published sources,
 * PARITY UNPINNED.  Tensors use ultralytics' own state_dict keys ("model.0.conv.weight", ...),
 * with BatchNorm (eps 1e-3) or already fused; widths/depths are inferred from the shapes. */
typedef struct og_yolo og_yolo;
og_yolo* og_yolo_create(int nc);                      /* nc: number of classes (1 for the glottis detector) */
void og_yolo_destroy(og_yolo* h);
int og_yolo_set_tensor(og_yolo* h, const char* key, const void* host, const int64_t* shape, int ndim, int dtype);
int og_yolo_finalize(og_yolo* h);
/* Tuning knobs (defaults in brackets).  The detector's arithmetic is a property of the HANDLE (round 4): a conv whose one-frame
 * launch would leave most of the chip idle sums its K range as `ks` parts combined in split order -- ks decided from the layer, the
 * frame size and these options, never from B.  "latency_batch" [1]: calls of at most this many frames (the reference's
 * one-frame-per-call pattern, detector.py:58) run the parts on separate workgroups with a fused reduce (the latency path); larger
 * calls run them in one workgroup and keep them apart in registers -- THE SAME BITS, so `detect(frame)` and `detect_frames(video)[i]`
 * return the same five floats and detector.py:68-69,94-95 truncate the same numbers (0: every call takes the batched kernels).
 * "splitk_max" [8], "splitk_min_steps" [3], "splitk_slots" [1], "splitk_div" [2], "latency_nt1" [1] shape the split (they change the
 * summation order, for every call of the handle alike); "head_fused" [1]: a Detect level's box and class branches as one chain;
 * "zero_copy" [1]: latency-path calls without `pred` read the frame from, and write `best` to, the handle's pinned host buffer directly
 * (no H2D / D2H command on the 58-launch chain's one stream). */
int og_yolo_set_option(og_yolo* h, const char* name, int value);
int og_yolo_num_anchors(og_yolo* h, int H, int W);    /* (H/8)(W/8)+(H/16)(W/16)+(H/32)(W/32) */
/* frames [B,H,W,3] u8 BGR at network resolution (H,W multiples of 32; the caller letterboxes).
 *   best [B,5] f32: x1,y1,x2,y2,conf of the arg-max-confidence candidate with conf > conf_thres,
 *                   clipped to the frame; conf = -1 when there is none  (detector.py:61-64)
 *   pred [B,A,5] f32 or NULL: every decoded candidate (for NMS on the host / parity tests) */
int og_yolo_detect_u8(og_yolo* h, const uint8_t* bgr, int B, int H, int W, float conf_thres, float* best, float* pred);
int og_yolo_detect_u8_dev(og_yolo* h, const uint8_t* bgr_dev, int B, int H, int W, float conf_thres, float* best_dev, float* pred_dev);
/* The same call in two halves, for callers that have something to run in between (round 4).  In the reference's frame loop
 * (features.py:235-245) the detector's box only gates the COUNT of the U-Net's mask, so `detector.detect(frame)` (detector.py:58) and
 * `unet_segment_frame(gray)` (utils.py:218-241) of one frame are independent: begin enqueues the detector's chain on its own stream
 * and returns, the caller runs the U-Net call, end waits and delivers `best [B,5]`.  One call in flight per handle (anything else
 * on the handle in between: OG_EINVAL); calls that do not take the latency path run inside begin. */
int og_yolo_detect_u8_begin(og_yolo* h, const uint8_t* bgr, int B, int H, int W, float conf_thres);
int og_yolo_detect_u8_end(og_yolo* h, float* best);
int og_yolo_sync(og_yolo* h);
/* Parity/debug: "model.0" ... "model.21" (module outputs), "box0..2", "cls0..2" (Detect branches), NCHW f32. */
int og_yolo_get_activation(og_yolo* h, const char* name, int B, float* out_nchw, size_t capacity_floats, int* dims);

#ifdef __cplusplus
}
#endif
#endif /* OPENGLOTTAL_HIP_H */
