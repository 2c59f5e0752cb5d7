/*
 * ctpn_hip.h -- C ABI of libctpn_hip.so, the MI355X (gfx950) CTPN inference hot path.
 *
 * Every entry point is `extern "C"`, takes plain pointers and sizes, returns an int status
 * (0 = CTPN_OK, negative = error; ctpn_last_error() gives the text) and never throws.
 * One ctpn_ctx per GPU / per host thread; a ctx owns its HIP stream and its HBM arena.
 *
 * Each declaration cites the reference interface (paths relative to the upstream tree of
 * eragonruan/text-detection-ctpn) that it replaces; INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add to bind it.
 */
#ifndef CTPN_HIP_H
#define CTPN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTPN_ABI_VERSION 10

/* status codes */
#define CTPN_OK            0
#define CTPN_ERR_ARG      -1   /* bad argument (null pointer, size out of range, K not tile-aligned ...) */
#define CTPN_ERR_HIP      -2   /* a HIP runtime call failed; text has hipGetErrorString */
#define CTPN_ERR_STATE    -3   /* call order violated (e.g. forward before weights are loaded) */
#define CTPN_ERR_CAPACITY -4   /* caller buffer or ctx arena too small for the request */
#define CTPN_ERR_NODEVICE -5   /* no usable gfx950 device: the product path never falls back to CPU */
#define CTPN_ERR_UNSUPPORTED -6 /* a well-formed input of a kind this entry point does not handle (ctpn_decode_jpeg_batch: CMYK / 4:1:1 / arithmetic-coded / incomplete files) */

/* arithmetic of the conv stack / LSTM input projection (BiLSTM recurrence and heads are fp32 in all of them) */
#define CTPN_PREC_FP32  0      /* exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): correctness gate, config 2 */
#define CTPN_PREC_BF16  1      /* bf16 MFMA, fp32 accumulate (v_mfma_f32_32x32x16_bf16): configs 3-5 (BASELINE.json's dtype) */
#define CTPN_PREC_FP16  2      /* IEEE fp16 MFMA, fp32 accumulate (v_mfma_f32_32x32x16_f16): the bf16 mode's rate, three more mantissa bits
                                  (activations of this network stay far below 65504) */
#define CTPN_PREC_SPLIT 3      /* parity-grade at the matrix cores' 16-bit rate / 3: every activation and weight is a (hi, lo) pair of bf16 and
                                  a product is three bf16 MFMAs (x_hi w_hi + x_lo w_hi + x_hi w_lo, fp32 accumulate; the dropped term is
                                  ~2^-17 of the product): holds north_star's 1e-3 / +-1 px against the fp32 path like CTPN_PREC_FP32 does */
/* (value 4 was CTPN_PREC_FP16W, ABI versions 6 - 7: fp16 with three layers through a 1-D Winograd transform. +2 % images/s for a fifth
 * arithmetic; removed in ABI version 8, ctpn_create answers CTPN_ERR_ARG for it. Measurements: profiles/r04_layers_fp16w.txt.) */

/* text-line connector mode: cfg.TEST.DETECT_MODE, lib/fast_rcnn/config.py:150 */
#define CTPN_MODE_H 0
#define CTPN_MODE_O 1

/* fixed geometry of the path (lib/rpn_msr/generate_anchors.py:24-32, lib/fast_rcnn/config.py:175-183) */
#define CTPN_NUM_ANCHORS      10
#define CTPN_FEAT_STRIDE      16
#define CTPN_WEIGHT_FLOATS    17893244  /* fp32 scalars in the flat weight arena, see ctpn_weight_manifest */

typedef struct ctpn_ctx ctpn_ctx;

/* ---- library ---------------------------------------------------------------------------- */

int         ctpn_abi_version(void);
/* thread-local text of the last error raised on this thread ("" if none) */
const char* ctpn_last_error(void);
/* number of visible HIP devices (0 if none); never fails */
int         ctpn_device_count(void);

/* ---- context ------------------------------------------------------------------------------
 * Replaces: tf.Session + get_network("VGGnet_test") (ctpn/demo.py:79-82,
 * lib/networks/VGGnet_test.py:7-55) and _set_device (lib/utils/nms_kernel.cu:80-89).
 * Allocates every activation buffer for up to max_batch images of max_h x max_w once; no
 * allocation happens on the forward path afterwards. */
int ctpn_create(ctpn_ctx** out, int device_id, int max_batch, int max_h, int max_w, int precision);
/* Post-processing-only ctx: the proposal layer (ctpn_proposals_from_host, ctpn_proposal_anchors) for up to max_batch feature
 * maps of max_hf x max_wf cells, WITHOUT the VGG activation arena and the weights (what lib/rpn_msr/proposal_layer_tf.py
 * needs when the network ran elsewhere, ctpn/demo_pb.py:91-92). ctpn_forward / ctpn_load_weights_* fail with CTPN_ERR_STATE. */
int ctpn_create_postproc(ctpn_ctx** out, int device_id, int max_batch, int max_hf, int max_wf);
int ctpn_destroy(ctpn_ctx* ctx);
/* Behaviour switches of ONE ctx (ABI 5; they were process-wide environment variables before): integer options by name, settable between
 * calls (the ctx drains its streams first; CTPN_ERR_STATE while a submitted batch is uncollected). ctpn_option_count / ctpn_option_name
 * enumerate them. None has a counterpart in the reference, whose only knobs are cfg.TEST.* (lib/fast_rcnn/config.py:147-183).
 *   keep_acts       0 | 1  also store the full-resolution output of pool-fused convs and lstm_o (layer-wise parity via ctpn_get_tensor)
 *   conv1_kernel    0..2   16-bit modes: conv1_1 as 2 = exact integer pixels x 16-bit weights, one MFMA term, from the q-image (8-byte pixels
 *                          (q_B, q_G, q_R, 1.0), q = p - round(mean); uint8 feed; default), 1 = split-bf16 operands, three terms (fp32-class; the
 *                          float-blob feed always), 0 = fp32 VALU
 *   conv1_fuse      0 | 1  with conv1_kernel = 2 and keep_acts = 0: conv1_1 is computed inside conv1_2's window stage and never stored
 *                          (default); 0 = stored by a stand-alone kernel and read back -- the same bytes downstream either way
 *   lstm_split      0 | 1  BiLSTM recurrent product h Wh on split-bf16 MFMAs (three bf16 terms per product, fp32 state / gates / accumulation:
 *                          |d| < 3e-5 vs the exact-fp32 MFMA kernel, 2 x faster). Default 1 in CTPN_PREC_BF16 / FP16 / SPLIT (split precision: since
 *                          ABI 9 -- it is that mode's own arithmetic), 0 in FP32, which never uses it
 *   nms_columns     0 .. 3 proposal-layer NMS: 1 (default) the column decomposition -- one workgroup per image, and for batches of up to four
 *                          images one COLUMN per wave over a quarter as many workgroups per image (a lone image's tail used one CU of 256);
 *                          0 the generic kernel; 2 / 3 pin the one-workgroup / the multi-workgroup form. Identical keep lists in all four
 *   nms_check       0 | 1  debug: run both and fail with CTPN_ERR_STATE on a mismatch (synchronises)
 *   connect_device  0 | 1  text-line connector of ctpn_detect_*: host C++ worker pool (default) or connect_kernel on the GPU: identical lines
 *   tail_overlap    0 | 1  ctpn_detect_submit: BiLSTM + heads of batch k on the proposal stream next to conv1_1 of batch k + 1
 *   conv_p64        0 | 1  CTPN_PREC_SPLIT: conv1_2 (Co = 64) through the persistent kernel's 64-channel form (default 1; 0 = the non-persistent
 *                          kernel of ABI 8). Other last bits than ABI 8 (kx-major K order), same tolerance class
 *   split_edge      0 | 1  CTPN_PREC_SPLIT: ragged tile columns (W mod 16 = 1, 2; pooled layers: 2, 4) through conv3x3_edge_kernel's split form, as in
 *                          the 16-bit modes (default 1; 0 = a padded tile column as in ABI 9: other last bits in those columns, same tolerance class)
 *   tail_confine    0 | 1  ctpn_detect_submit: the forward of batch k + 1 waits, behind its conv1_1, for the proposal tail of batch k (default 0).
 *                          Round 6 shipped 1 for CTPN_PREC_SPLIT while a cross-batch race of the conv kernels was open; the race is fixed in the
 *                          kernels (ABI 10), the switch stays for A/B runs
 *   nms_prefix      0 | 1  proposal-layer column NMS: look at the 4096 best-scored candidates first and at all of them only if those hold fewer
 *                          than post_nms_topn survivors (default 1). Identical keep lists
 *   debug_hog       0 .. 200000  diagnostic: microseconds a kernel with the one-workgroup NMS's footprint (1024 threads, 84 KB of LDS, one
 *                          workgroup per image) spins, without memory traffic, in front of the proposal NMS (default 0: not launched); values
 *                          above 50000: it also keeps writing its LDS, for (value - 50000) microseconds; above 100000: twice the workgroups gather random 16-byte
 *                          pieces of the largest activation buffer for (value - 100000) microseconds (memory-system load beside EVERY layer of
 *                          the next batch). It delays the NMS of batch k into later
 *                          layers of batch k + 1: the stress under which tests/test_gpu_round6.py checks that batches in flight do not
 *                          change each other's bits
 *   debug_nms       0 .. 15  diagnostic, WRONG proposals: parts of the one-workgroup proposal NMS switched off (1 no greedy pass, 2 no output
 *                          stores, 4 no box loads in the greedy pass, 8 return after the column lists); tools/r6_nms_parts.sh */
int         ctpn_set_option(ctpn_ctx* ctx, const char* key, int value);
int         ctpn_get_option(ctpn_ctx* ctx, const char* key, int* value_out);
int         ctpn_option_count(void);
const char* ctpn_option_name(int index);
/* Host worker threads of a ctx (per-image connector work of ctpn_detect_collect, staging copies of pageable images): one
 * persistent pool per ctx, created in ctpn_create. Size = ctpn_host_thread_budget(hardware cores, LOCAL_WORLD_SIZE of the
 * launcher (torchrun), CTPN_HOST_THREADS): `requested` if > 0, else cores / ranks-on-this-node clamped to [1, 32].
 * CTPN_AFFINITY=1 pins the pool to the cores [LOCAL_RANK * budget, (LOCAL_RANK + 1) * budget). Pure function, needs no GPU.
 * (The reference runs its connector on the one Python thread of ctpn/demo.py:63-64.) */
int ctpn_host_thread_budget(int cpu_count, int local_world_size, int requested);
int ctpn_host_threads(ctpn_ctx* ctx, int* threads_out);
/* block until everything queued on the ctx stream has finished */
int ctpn_sync(ctpn_ctx* ctx);
/* the hipStream_t the ctx launches on (as void*), for callers that bracket it with events */
int ctpn_stream(ctpn_ctx* ctx, void** stream_out);

/* ---- weights ------------------------------------------------------------------------------
 * Replaces: tf.train.Saver().restore (ctpn/demo.py:85-93) / Network.load (lib/networks/network.py:40-53).
 * The arena is CTPN_WEIGHT_FLOATS fp32 values: the reference's TF variables, each in its TF
 * layout (conv HWIO [3,3,Ci,Co], LSTMCell kernel [640,512] gate order i,j,f,o, matmul [in,out]),
 * concatenated in the order ctpn_weight_manifest() reports. */
int ctpn_weight_manifest(int index, const char** name, int* rank, int shape4[4], size_t* offset_floats);
int ctpn_weight_count(void);
int ctpn_load_weights_host(ctpn_ctx* ctx, const float* arena_host);
/* arena already in this device's HBM (e.g. a torch tensor filled by an RCCL broadcast): packs in place */
int ctpn_load_weights_device(ctpn_ctx* ctx, const void* arena_dev);

/* ---- weight broadcast (RCCL over xGMI) ----------------------------------------------------------
 * The reference has no collective at all (no NCCL / MPI anywhere; batch asserted to 1). This build shards
 * images over GPUs with ONE collective: the fp32 arena is broadcast once at start-up (north_star: "RCCL broadcast of weights over
 * xGMI only"), nothing per batch. librccl.so is loaded on first use (dlopen by soname; CTPN_RCCL_LIB overrides the path).
 *
 * ctpn_broadcast_weights: one process, n ctxs on n DIFFERENT devices (SURVEY section 8b export list): handles[0] must have its weights
 * loaded; every other ctx receives the arena over RCCL (ncclCommInitAll + grouped ncclBroadcast on each ctx's stream) and packs it.
 *
 * One process per GPU (torchrun; bench.py): the root calls ctpn_comm_unique_id and hands the CTPN_COMM_ID_BYTES bytes to the other
 * ranks by any side channel (bench.py: the torch.distributed store), then EVERY rank calls ctpn_broadcast_weights_rank, which
 * joins the communicator (collective call), receives / sends the arena on the ctx stream and packs. The root loads its weights first. */
#define CTPN_COMM_ID_BYTES 128
int ctpn_broadcast_weights(ctpn_ctx** handles, int n);
int ctpn_comm_unique_id(char* id_out, size_t capacity);
int ctpn_broadcast_weights_rank(ctpn_ctx* ctx, const char* unique_id, int rank, int world, int root);
/* ctpn_broadcast_weights_rank is a COLLECTIVE with no timeout of its own: a rank whose local pre-checks fail (librccl not loadable, a
 * post-processing-only ctx, a root without weights) returns an error BEFORE joining the communicator, and every other rank then blocks in
 * ncclCommInitRank. The caller must either make sure all ranks pass the same pre-checks (same library, same kind of ctx, root loaded) or
 * run the call under its own watchdog -- bench.py does both (180 s timer, and the ranks agree on success over the side channel before
 * anyone proceeds; on failure all of them fall back to a host broadcast). */

/* ---- network forward ----------------------------------------------------------------------
 * Replaces: _get_image_blob + sess.run of conv1_1 .. rpn_cls_prob_reshape / rpn_bbox_pred
 * (lib/fast_rcnn/test.py:7-51, lib/networks/VGGnet_test.py:20-52, lib/networks/network.py:88-196,
 * 269-277, 332-337). images: n x h x w x 3 uint8, BGR, already at network resolution (the two
 * reference resizes are identity there); PIXEL_MEANS (lib/fast_rcnn/config.py:200) are subtracted
 * on device. images_on_device != 0 means the pointer is HBM, else host (copied on the ctx stream).
 * Asynchronous: returns after enqueueing. */
int ctpn_forward(ctpn_ctx* ctx, const uint8_t* images, int images_on_device, int n, int h, int w);
/* Same, fed with the reference's own `net.data` blob (lib/fast_rcnn/test.py:47-49): n x h x w x 3 float32,
 * BGR, PIXEL_MEANS already subtracted (what _get_image_blob returns after its cv2.resize).
 * Feed dtype and conv1_1 (CTPN_PREC_BF16 / CTPN_PREC_FP16; CTPN_PREC_FP32 and CTPN_PREC_SPLIT compute both feeds identically): the uint8 feed runs
 * conv1_1 as EXACT integer pixels x bf16-rounded weights (one MFMA term), the float feed -- arbitrary floats -- as split-bf16
 * operands (three terms, fp32-class). The same image through the two feeds therefore differs by the bf16 rounding of conv1_1's 27
 * weights per channel, the same class of error every other layer of the bf16 path carries: rpn_cls_prob within 3e-2 max / 2e-3
 * mean of each other at 600x900 (tests/test_gpu_round3.py::test_float_blob_feed_tracks_uint8_feed_in_bf16_mode). In bf16 mode the
 * BiLSTM gates use v_exp_f32 / v_rcp_f32 (|diff| < 2e-5 on lstm_out against the exact path, same file). */
int ctpn_forward_blob(ctpn_ctx* ctx, const float* blob, int blob_on_device, int n, int h, int w);
/* feature-map geometry of the last forward: hf = h/16 (VALID pools), wf = w/16 */
int ctpn_feat_shape(ctpn_ctx* ctx, int* n, int* hf, int* wf);
/* copy a named activation of the last forward to the host as dense fp32 NHWC (layer-wise parity).
 * names: conv1_1 .. conv5_3, pool1..pool4, rpn_conv/3x3, lstm_pre, lstm_out, lstm_o, heads,
 * rpn_cls_prob_reshape (n,hf,wf,20), rpn_bbox_pred (n,hf,wf,40). Synchronises the stream. */
int ctpn_get_tensor(ctpn_ctx* ctx, const char* name, float* out_host, size_t capacity_floats, int shape4[4]);

/* ---- proposal layer -----------------------------------------------------------------------
 * Replaces: Network.proposal_layer / proposal_layer (lib/networks/network.py:207-222,
 * lib/rpn_msr/proposal_layer_tf.py:14-157) incl. bbox_transform_inv, clip_boxes, _filter_boxes
 * (lib/fast_rcnn/bbox_transform.py:36-80) and the nms call at :144. Runs on the activations of
 * the last ctpn_forward. im_info: n x 3 floats [H, W, scale] (lib/fast_rcnn/test.py:44-46).
 * rois_out: n x post_nms_topn x 5 floats [score,x1,y1,x2,y2] (column 0 is the score, as in the
 * reference, proposal_layer_tf.py:155), counts_out: n ints. Tie order: descending score, equal
 * scores by ascending anchor index (h, w, a) -- documented deviation from numpy's unstable sort.
 * Synchronous (returns after the D2H copy). */
int ctpn_proposals(ctpn_ctx* ctx, const float* im_info, int pre_nms_topn, int post_nms_topn,
                   float nms_thresh, float min_size, float* rois_out, int* counts_out);
/* same computation from caller-supplied head outputs (the demo_pb.py seam, ctpn/demo_pb.py:91-92):
 * cls_prob n x hf x wf x 20, bbox_pred n x hf x wf x 40, fp32 host arrays */
int ctpn_proposals_from_host(ctpn_ctx* ctx, const float* cls_prob, const float* bbox_pred,
                             int n, int hf, int wf, const float* im_info, int pre_nms_topn,
                             int post_nms_topn, float nms_thresh, float min_size,
                             float* rois_out, int* counts_out);

/* Second return value of proposal_layer (lib/rpn_msr/proposal_layer_tf.py:133-157: bbox_deltas[order][keep]): for every roi row of
 * the LAST ctpn_proposals / ctpn_proposals_from_host call, the index of the anchor that produced it, (y * wf + x) * 10 + a --
 * i.e. the row of rpn_bbox_pred.reshape(-1, 4). anchors_out: n x post_nms_topn ints (rows >= counts_out[i] are undefined);
 * post_nms_topn must equal that call's. */
int ctpn_proposal_anchors(ctpn_ctx* ctx, int* anchors_out, int post_nms_topn);

/* ---- input pipeline ------------------------------------------------------
 * Replaces: cv2.resize(im, None, None, fx=f, fy=f, interpolation=cv2.INTER_LINEAR) as called by
 * resize_im (ctpn/demo.py:21-25, uint8 BGR image) and by _get_image_blob (lib/fast_rcnn/test.py:17-27,
 * float32 image after the PIXEL_MEANS subtraction). OpenCV itself (opencv_python==3.4.0.12) is a
 * third-party dependency outside the reference tree: the kernel follows its published algorithm
 * (dsize = round-half-even(src * f); sample at (d + 0.5) / f - 0.5; uint8 in 11-bit fixed point,
 * float32 in fp32) -- parity with the real cv2 is UNPINNED, parity with oracle/resize_ref.py is
 * bit-exact. src: n x h x w x 3, dst: n x out_h x out_w x 3 (capacity in elements); either side may
 * be a device pointer (src_on_device / dst_on_device), e.g. to feed ctpn_forward without a round
 * trip. out_h / out_w are always written (call with dst == NULL to size the output). */
int ctpn_resize_dims(int h, int w, double fx, double fy, int* out_h, int* out_w);
int ctpn_resize(int device_id, const void* src, int src_is_f32, int src_on_device, int n, int h, int w,
                double fx, double fy, void* dst, int dst_on_device, long long dst_capacity,
                int* out_h, int* out_w);

/* ---- NMS ----------------------------------------------------------------------------------
 * Replaces: void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
 *                     int boxes_dim, float nms_overlap_thresh, int device_id)
 * (lib/utils/gpu_nms.hpp:1-2, lib/utils/nms_kernel.cu:91-143). Same argument order and meaning:
 * boxes_host is boxes_num x boxes_dim (>= 4) fp32 rows [x1,y1,x2,y2,...] already sorted by
 * descending score; keep_out (capacity boxes_num) receives positions into that array in
 * ascending order. Suppress iff fp32 IoU("+1" areas) > fp32(thresh) (nms_kernel.cu:24-32,71).
 * Differences: returns a status instead of printing CUDA errors; device buffers are cached per
 * device instead of malloc/free per call; only keep[] leaves the GPU (no 18 MB mask D2H). */
int ctpn_nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num, int boxes_dim,
             float nms_overlap_thresh, int device_id);

/* ---- text-line connector ------------------------------------------------------------------
 * Replaces: TextDetector.detect (lib/text_connector/detectors.py:19-49) with the graph builder
 * (text_proposal_graph_builder.py:6-78), Graph.sub_graphs_connected (other.py:20-29) and
 * get_text_lines H (text_proposal_connector.py:21-64) / O (text_proposal_connector_oriented.py:24-105).
 * boxes: r x 4 fp32 [x1,y1,x2,y2], scores: r fp32, (im_h, im_w) = `size`. recs_out: capacity x 9
 * float64 [x1,y1,x2,y2,x3,y3,x4,y4,score]; count_out = number of lines (may exceed capacity ->
 * CTPN_ERR_CAPACITY with count_out set). The NMS(0.2) inside runs on device_id when
 * device_id >= 0 (reference: nms_wrapper.nms -> gpu_nms, detectors.py:29). Host C++ otherwise. */
int ctpn_text_lines(const float* boxes, const float* scores, int r, int im_h, int im_w, int mode,
                    int device_id, double* recs_out, int capacity, int* count_out);
/* The connector's constants as compiled in (the reference's TextLineCfg, lib/text_connector/text_connect_cfg.py:4-12, reads them at run time):
 * out8 = {TEXT_PROPOSALS_WIDTH * MIN_NUM_PROPOSALS, MIN_RATIO, LINE_MIN_SCORE, MAX_HORIZONTAL_GAP, TEXT_PROPOSALS_MIN_SCORE,
 * TEXT_PROPOSALS_NMS_THRESH, MIN_V_OVERLAPS, MIN_SIZE_SIM}. The Python mirror's TextDetector compares its Config with them and raises if a
 * caller has edited one: an edit that would silently do nothing is worse than an error. Needs no device. */
int ctpn_connector_constants(double* out8);

/* ---- result files ---------------------------------------------------------------------------
 * Replaces draw_boxes (ctpn/demo.py:28-52), host C++: the text of data/results/res_<stem>.txt -- one "min_x,min_y,max_x,max_y\r\n" line
 * per text line, coordinates int(coord / scale) (truncation, :43-46), lines skipped where |box[0] - box[1]| < 5 or |box[3] - box[0]| < 5
 * (the reference's scalar comparison, :32) -- and the outline rendering into the BGR image (cv2.line(..., 2): green for score >= 0.9).
 * recs: n_lines x 9 float64 as returned by ctpn_text_lines / ctpn_detect. ctpn_result_text with out == NULL only sizes the text. */
int ctpn_result_text(const double* recs, int n_lines, double scale, char* out, size_t capacity, size_t* bytes_out, int* lines_out);
int ctpn_write_result_file(const char* path, const double* recs, int n_lines, double scale, int* lines_out);
int ctpn_draw_boxes(uint8_t* img_bgr, int h, int w, const double* recs, int n_lines);

/* ---- whole path, batched ------------------------------------------------------------------
 * Replaces the body of ctpn() in ctpn/demo.py:55-68 between imread/resize and draw_boxes for a
 * batch: forward -> proposals -> (boxes / scale) -> text lines. scales: n floats (im_scales[0] of
 * lib/fast_rcnn/test.py:57, 1.0 when the image is already at network resolution).
 * recs_out: n x line_capacity x 9 float64, line_counts: n ints; rois_out/roi_counts may be NULL. */
int ctpn_detect(ctpn_ctx* ctx, const uint8_t* images, int images_on_device, int n, int h, int w,
                const float* scales, int mode, double* recs_out, int line_capacity, int* line_counts,
                float* rois_out, int* roi_counts);

/* Asynchronous form of ctpn_detect for throughput: submit enqueues the whole device part of a batch (forward on the
 * ctx stream; proposal layer + connector front end + D2H on a second stream) into slot 0 or 1 and returns at once;
 * collect waits for that slot, runs the host part of the connector and fills the outputs. With two slots the host
 * part of batch k and the latency-bound sort / NMS kernels overlap the convolutions of batch k+1. A slot must be
 * collected before it is submitted to again; images must stay valid until the forward of that submit has run
 * (device pointers: until ctpn_sync or the matching collect). */
int ctpn_detect_submit(ctpn_ctx* ctx, const uint8_t* images, int images_on_device, int n, int h, int w,
                       const float* scales, int slot);
int ctpn_detect_collect(ctpn_ctx* ctx, int slot, int mode, double* recs_out, int line_capacity, int* line_counts,
                        float* rois_out, int* roi_counts);

/* ---- measurement --------------------------------------------------------------------------
 * When enabled, every kernel launch on the ctx stream is bracketed by hipEvents; ctpn_profile_read
 * returns, per kernel kind, the accumulated milliseconds, launch count and algorithmic work
 * (flops for MFMA kernels, bytes for HBM-bound kernels) since the last reset. Used by bench.py for
 * the roofline object; costs two hipEventRecord per launch. */
#define CTPN_KIND_CONV_FIRST  0   /* preprocess + conv1_1 (direct, VALU)        */
#define CTPN_KIND_CONV_GEMM   1   /* implicit-GEMM conv3x3 (MFMA)               */
#define CTPN_KIND_POOL        2   /* 2x2/2 VALID max-pool                       */
#define CTPN_KIND_GEMM        3   /* LSTM input projection, lstm_o FC, heads    */
#define CTPN_KIND_BILSTM      4   /* persistent recurrent kernel                */
#define CTPN_KIND_DECODE      5   /* softmax + anchor decode + clip + filter    */
#define CTPN_KIND_SORT        6   /* per-image key sort + gather                */
#define CTPN_KIND_NMS         7   /* greedy NMS                                 */
#define CTPN_KIND_COUNT       8
/* on = 1: a hipEvent pair around every stage (CTPN_KIND_*); on = 2: ONE pair around the 13 conv3x3 launches of a forward and
 * nothing else (42 event records per step cost ~0.2 ms of bubbles at 11 ms / step); on = 0: off. */
int ctpn_profile_enable(ctpn_ctx* ctx, int on);
int ctpn_profile_reset(ctpn_ctx* ctx);
int ctpn_profile_read(ctpn_ctx* ctx, int kind, double* ms, long long* launches, double* work);

/* ---- diagnostics ---------------------------------------------------------------------------
 * One conv3x3 + bias + ReLU (+ 2x2/2 VALID max-pool when fuse_pool) on dense fp32 host tensors, through the same
 * kernels the forward uses (impl 1 = tap-reuse conv3x3.hip, 0 = im2col igemm.hip + pool kernel). in: n x h x w x ci,
 * w_hwio: 3 x 3 x ci x co (TF layout), out_full: n x h x w x co or NULL, out_pool: n x h/2 x w/2 x co or NULL.
 * ci must be a multiple of 32 (fp32) / 64 (bf16, fp16, split), co of 8 (split: <= 64 or a multiple of 128). precision: CTPN_PREC_*.
 * impl 1 = the product kernels (conv3x3), impl 0 = the im2col GEMM as an independent reference (its pool taken on the host).
 * Unit-test hook for shapes VGG never produces. */
/* fp32 -> bf16 exactly as the kernels' epilogues do it: use_hw_instruction 1 = v_cvt_pk_bf16_f32, 0 = integer
 * round-to-nearest-even formula (the two must agree bit for bit on finite inputs). */
int ctpn_debug_cvt_bf16(int device_id, const float* in, uint16_t* out, int n, int use_hw_instruction);
/* The kernels' two LDS-DMA helper forms (conv3x3_impl.h: c3_glds16_saddr declares m0 clobbered, c3_glds16_asm saves and restores it around
 * global_load_lds_dwordx4) on the same `bytes` (a multiple of 1024) of src: both outputs must equal src. Guards the clobber form's
 * contract against a compiler that starts to keep state in m0. */
int ctpn_debug_lds_dma(int device_id, const uint8_t* src, size_t bytes, uint8_t* out_clobber, uint8_t* out_keep);
int ctpn_debug_conv3x3(int device_id, const float* in_nhwc, const float* w_hwio, const float* bias, int n, int h, int w,
                       int ci, int co, int precision, int impl, int fuse_pool, float* out_full, float* out_pool);
/* TextDetector.detect (lib/text_connector/detectors.py:19-49) for ONE image with every step on the device -- score > 0.7 prefix and
 * boxes / scale (lines_prep_kernel), NMS 0.2 (nms_kernel), graph build / chains / line fit / filter_boxes (connect_kernel) -- i.e.
 * the device-connector form of the asynchronous detect path (option connect_device = 1). rois: r x 5 fp32 [score,x1,y1,x2,y2] in
 * descending score order (what proposal_layer returns), r <= 1000. Test hook for the connector kernel. */
int ctpn_debug_connect(int device_id, const float* rois, int r, int im_h, int im_w, float scale, int mode, double* recs_out,
                       int capacity, int* count_out);

/* ---- cv2.imread for JPEG files (reference ctpn/demo.py:59), split where the work splits: marker parsing and Huffman decoding on the host
 * (the ctx's worker pool, one image per thread), dequantisation + inverse DCT + chroma upsampling + YCbCr -> BGR on the device. The pixel
 * arithmetic is libjpeg's integer arithmetic (islow IDCT, h2v2 / h2v1 / h1v2 "fancy" upsampling, 16-bit fixed-point colour conversion): the
 * images equal what cv2 / Pillow (libjpeg-turbo) return, bit for bit. Supported: 8-bit Huffman-coded files, sequential (SOF0 / SOF1) and
 * progressive (SOF2: it differs in the host half only), 1 component or YCbCr 4:4:4 / 4:4:0 / 4:2:2 / 4:2:0, restart intervals, and the
 * eight EXIF orientations, applied the way cv2.imread applies them (an index map in the colour kernel: every size below is the TURNED
 * image's) -- i.e. every JPEG file of the reference's own data/demo. Anything else (CMYK, 4:1:1, arithmetic coding, 12-bit, three components
 * that store RGB by libjpeg's marker rule) and every INCOMPLETE file (entropy data that ends early, a progressive file without its last
 * scans: libjpeg smooths / zero-fills those by rules of its own) is CTPN_ERR_UNSUPPORTED and the caller decodes that file another way
 * (lib/utils/image.py).
 *   ctpn_jpeg_probe            size (as cv2.imread returns it), components and layout of one file: layout & 0xff = luma sampling (1: 4:4:4 /
 *                              gray, 2: 4:2:0, 0x21: 4:2:2 = 2 horizontally, 1 vertically, 0x12: 4:4:0), layout >> 8 = EXIF orientation - 1.
 *                              Host only.
 *   ctpn_jpeg_coef_capacity    int16 elements one h x w image can need in ctpn_jpeg_entropy_decode's coefficient buffer
 *   ctpn_jpeg_entropy_decode   the host half alone (no device needed: the seam the CPU tests use): quantised DCT blocks, natural order,
 *                              component after component, [block rows][block columns][64] each; qt = 3 x 64 quantisation values (natural
 *                              order); layout8 = {STORED h, w, ncomp, horizontal luma sampling | (EXIF orientation - 1) << 8, block columns
 *                              of component 0, 1, block rows of component 0, 1} (vertical luma sampling = block rows of component 0 / 1)
 *   ctpn_decode_jpeg_batch     resize_im(cv2.imread(f)) (reference ctpn/demo.py:59-60) for n files of one size h x w (turned), one layout and
 *                              one orientation: decode,
 *                              then -- unless fx = fy = 1 (or <= 0) -- cv2.resize(fx, fy, INTER_LINEAR) in the same queue. Result: n x
 *                              out_h x out_w x 3 BGR uint8 in device memory owned by the ctx; *images_dev_out is valid for ctpn_forward /
 *                              ctpn_detect_submit(images_on_device = 1) until the second-next call of this function (two buffer sets; the
 *                              ctx orders the decode, the forward that reads it and the buffer's reuse by events: no host wait). The call
 *                              returns when the host half is done and the device half is queued. The buffers grow with the largest
 *                              n x h x w seen; h x w is the FILE size and is not bound by the ctx's max_h x max_w (the output is, once it
 *                              is passed to ctpn_forward).
 *   ctpn_decode_jpeg_files     the same from n PATHS (the reference's cv2.imread(im_name) takes a path): the files are read inside the worker
 *                              threads, right before their entropy decoding.
 *   ctpn_jpeg_probe_files      header scan of n paths on `threads` host threads (<= 0: up to 16): info4[i] = {h, w, components, layout}
 *                              as ctpn_jpeg_probe returns them; h = 0 marks a file ctpn_decode_jpeg_files does not take (unreadable, not a JPEG, or a kind
 *                              that is CTPN_ERR_UNSUPPORTED) -- per-file outcomes are data here, not errors: the caller routes those
 *                              files to its other decoder. Host only, needs no device.
 *   ctpn_jpeg_batch_fetch      copy a batch returned by ctpn_decode_jpeg_batch (still live) to the host, n x out_h x out_w x 3 bytes: for
 *                              callers that also draw on the image (reference ctpn/demo.py:28-52). Waits for that batch's decode. */
int    ctpn_jpeg_probe(const uint8_t* data, size_t len, int* h, int* w, int* ncomp, int* luma_sampling);
size_t ctpn_jpeg_coef_capacity(int h, int w);
int    ctpn_jpeg_entropy_decode(const uint8_t* data, size_t len, int16_t* coef, size_t coef_capacity, uint16_t* qt, int* layout8);
int    ctpn_decode_jpeg_batch(ctpn_ctx* ctx, const uint8_t* const* files, const size_t* sizes, int n, int h, int w, double fx, double fy,
                              const uint8_t** images_dev_out, int* out_h, int* out_w);
int    ctpn_decode_jpeg_files(ctpn_ctx* ctx, const char* const* paths, int n, int h, int w, double fx, double fy,
                              const uint8_t** images_dev_out, int* out_h, int* out_w);
int    ctpn_jpeg_probe_files(const char* const* paths, int n, int* info4, int threads);
int    ctpn_jpeg_batch_fetch(ctpn_ctx* ctx, const uint8_t* images_dev, uint8_t* host_out, size_t capacity);

/* ---- cv2.imread for PNG files (reference ctpn/demo.py:59; data/demo holds .jpg and .png). HOST ONLY, by the nature of the format: one
 * DEFLATE stream (zlib's inflate, the library libpng itself sits on) and row filters that chain from row to row -- nothing a GPU is for. One
 * file per host thread, straight into the caller's batch buffer, which ctpn_detect_submit / ctpn_forward take as host images (one
 * host-to-device copy per batch). libpng's IMREAD_COLOR transforms restated: RGB / RGBA -> BGR with the alpha dropped, gray 1 / 2 / 4 / 8
 * bit and gray + alpha -> B = G = R, palette images through PLTE (tRNS ignored), Adam7 interlacing; critical-chunk CRCs and the zlib
 * checksum are verified. 16-bit files are CTPN_ERR_UNSUPPORTED. Byte-equal to Pillow's decode (tests/test_png.py). Needs no device.
 *   ctpn_png_probe          size, colour type and bit depth from the header
 *   ctpn_png_decode         one file in memory -> h x w x 3 BGR uint8 (capacity in bytes)
 *   ctpn_png_probe_files    info4[i] = {h, w, colour type, bit depth} of n paths on `threads` host threads (<= 0: up to 16); h = 0 marks a
 *                           file ctpn_decode_png_files does not take (unreadable, not a PNG, 16-bit): data, not an error
 *   ctpn_decode_png_files   n files of one size h x w -> n x h x w x 3 BGR uint8 in the caller's host buffer, one file per thread
 *                           (threads <= 0: up to 32); the first failing file is the call's error, its path in ctpn_last_error()
 *   ctpn_debug_png_backend  which DEFLATE back end decodes: returns 1 = libdeflate (dlopen'ed libdeflate.so.0), 0 = zlib's inflate();
 *                           zlib_only = 1 / 0 forces zlib / lifts that (process-wide; < 0 only asks). Test hook: both give the same bytes. */
int    ctpn_debug_png_backend(int zlib_only);
int    ctpn_png_probe(const uint8_t* data, size_t len, int* h, int* w, int* color_type, int* bit_depth);
int    ctpn_png_decode(const uint8_t* data, size_t len, uint8_t* bgr_out, size_t capacity);
int    ctpn_png_probe_files(const char* const* paths, int n, int* info4, int threads);
int    ctpn_decode_png_files(const char* const* paths, int n, int h, int w, uint8_t* bgr_out, int threads);

#ifdef __cplusplus
}
#endif
#endif /* CTPN_HIP_H */
