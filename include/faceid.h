/*
 * faceid.h -- C ABI of libfaceid.so: the MI355X (gfx950) implementation of the
 * detect -> align -> embed -> match hot path of Kumar2421/scrfd_arcface_facerecognition.
 *
 * The reference has no native boundary of its own: its hot path is Python that calls into
 * onnxruntime / OpenCV / scikit-image / numpy.  Each entry point below replaces one of those
 * third-party calls (cited as reference file:line) and is what the ctypes binding in
 * scrfd_arcface_facerecognition_amd/_lib.py loads (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 (FID_OK) or a negative FID_E_* code; no exception crosses the ABI;
 *     fid_last_error() gives the message for the calling thread's last failure.
 *   - pointers named *_dev are DEVICE pointers (from fid_malloc, or any HIP allocation such as a
 *     torch tensor's data_ptr()); everything else is host memory owned by the caller.
 *   - all work is enqueued on the context's HIP stream and is asynchronous unless the function
 *     copies to host memory (fid_memcpy_d2h, fid_*_read) or is fid_sync().
 *   - a context is not re-entrant: calls on the same fid_ctx are serialised by an internal mutex;
 *     different contexts are independent (one process per GPU in the multi-GPU runs).
 *   - images are uint8, H x W x 3, BGR, dense (the layout cv2.imread / VideoCapture.read return,
 *     reference main.py:95,176).
 */
#ifndef FACEID_H
#define FACEID_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FID_ABI_VERSION 2   /* 2: fid_face_gates takes the pose angles as float64 (round 4) */

#define FID_OK 0
#define FID_E_INVALID (-1)   /* bad argument / shape / table */
#define FID_E_HIP (-2)       /* HIP runtime error (message has the hipError string) */
#define FID_E_CAPACITY (-3)  /* caller-provided capacity too small */
#define FID_E_NOMEM (-4)
#define FID_E_STATE (-5)     /* object used in the wrong state */

typedef struct fid_ctx fid_ctx;         /* device + stream + scratch */
typedef struct fid_net fid_net;         /* a compiled conv net (layer table + packed weights) */
typedef struct fid_gallery fid_gallery; /* L2-normalised fp16 gallery resident in HBM */
typedef struct fid_comm fid_comm;       /* one rank of an RCCL communicator (one process per GPU) */

/* ---- library / context -------------------------------------------------------------------- */
int fid_abi_version(void);
const char *fid_last_error(void);
int fid_device_count(int *count);
/* stream: an existing hipStream_t to enqueue on (e.g. torch's current stream), or NULL to let
 * the context create its own non-blocking stream. */
int fid_ctx_create(int device, void *stream, fid_ctx **out);
int fid_ctx_destroy(fid_ctx *ctx);
int fid_sync(fid_ctx *ctx);
int fid_device_name(fid_ctx *ctx, char *buf, int buflen);

/* ---- device memory (so a host without torch can drive the library) -------------------------- */
int fid_malloc(fid_ctx *ctx, size_t bytes, void **dptr);
int fid_free(fid_ctx *ctx, void *dptr);
int fid_memcpy_h2d(fid_ctx *ctx, void *dst_dev, const void *src, size_t bytes);
int fid_memcpy_d2h(fid_ctx *ctx, void *dst, const void *src_dev, size_t bytes); /* synchronises */
int fid_memset(fid_ctx *ctx, void *dst_dev, int value, size_t bytes);

/* ---- pinned host staging + an upload stream: double-buffered H2D for the video front-end (the step before the
 * path: reference main.py:174-184 reads and uploads one frame at a time).  fid_upload_async copies on a separate
 * HIP stream, ordered after the compute work enqueued so far (the destination may still be in use);
 * fid_upload_wait makes the compute stream wait for it -- no host synchronisation in either. */
int fid_pinned_alloc(fid_ctx *ctx, size_t bytes, void **hptr);
int fid_pinned_free(fid_ctx *ctx, void *hptr);
int fid_upload_async(fid_ctx *ctx, void *dst_dev, const void *src_pinned, size_t bytes);
int fid_upload_wait(fid_ctx *ctx);
/* Per-buffer ordering for double buffering (slot = one device staging buffer, 0 <= slot < FID_UPLOAD_SLOTS):
 * fid_upload_release(slot) marks the compute work enqueued so far as the last reader of that buffer;
 * fid_upload_async_slot copies into it after ONLY that release (so the upload of batch i+1 overlaps the compute of
 * batch i, which reads the other buffer); fid_upload_wait_slot makes the compute stream wait for that upload. */
#define FID_UPLOAD_SLOTS 4
int fid_upload_release(fid_ctx *ctx, int slot);
int fid_upload_async_slot(fid_ctx *ctx, int slot, void *dst_dev, const void *src_pinned, size_t bytes);
int fid_upload_wait_slot(fid_ctx *ctx, int slot);

/* ---- timing with HIP events on the context's stream (bench.py roofline leg) ----------------- */
#define FID_MAX_EVENTS 64
int fid_event_record(fid_ctx *ctx, int slot);
int fid_event_elapsed_ms(fid_ctx *ctx, int slot_start, int slot_stop, float *ms); /* synchronises */

/* ---- conv nets: replaces onnxruntime.InferenceSession(...).run ------------------------------
 * reference models/scrfd.py:59-62,83 and models/arcface.py:18-21,51.
 * The net is described by a layer table produced by scrfd_arcface_facerecognition_amd/lower.py
 * (format documented in csrc/net.h): `ops` is n_ops x FID_OP_WORDS int32, `tensors` is
 * n_tensors x FID_TENSOR_WORDS int32, `blob` holds packed fp16 weights + fp32 epilogue tables.
 * Input of fid_net_run: uint8 BGR images [batch, H, W, 3]; the blob conversion
 * (cv2.dnn.blobFromImage(s), scrfd.py:76-82 / arcface.py:44-50) is fused into the first conv.
 * A net belongs to the context it was created with: its activation slots, lazily built weight repackings and kernel plans
 * are per net and unsynchronised, so run one net from ONE context (several host threads may share that context: its mutex
 * serialises them); a second stream needs its own fid_net (bench.py: one per lane). */
#define FID_OP_WORDS 32
#define FID_TENSOR_WORDS 8
int fid_net_create(fid_ctx *ctx, const int32_t *ops, int n_ops, const int32_t *tensors,
                   int n_tensors, const void *blob, size_t blob_bytes, int in_h, int in_w,
                   int max_batch, fid_net **out);
int fid_net_destroy(fid_ctx *ctx, fid_net *net);
int fid_net_run(fid_ctx *ctx, fid_net *net, const uint8_t *images_dev, int batch);
/* Execute depth-first over sub-batches of this many images (0 = the whole batch layer by layer), so
 * that a layer's input is still in L2 / Infinity Cache when the next layer reads it. */
int fid_net_set_sub_batch(fid_net *net, int sub_batch);
/* device address + geometry of a tensor of the last run: dims = {H, W, C_logical, C_stored},
 * dtype 0 = fp16, 1 = fp32; layout [batch, H, W, C_stored]. */
int fid_net_tensor(fid_net *net, int tensor_id, void **dptr, int dims[4], int *dtype);
/* per-op device time of the last fid_net_run_profiled (ms per op, n_ops floats) */
int fid_net_run_profiled(fid_ctx *ctx, fid_net *net, const uint8_t *images_dev, int batch,
                         float *op_ms);
/* Kernel plans (per conv op and batch size the executor times its candidate kernels at first use and keeps the fastest).
 * Text lines keyed by ISA name + CU count + layer-table hash (incl. the library's candidate-set revision); every line is checked
 * against this library's candidates before its first use; a loaded plan replaces the timing, so two boxes run the same
 * kernels / fp32 summation orders and return bit-identical outputs.  Environment FID_PLAN=<file>: load at
 * fid_net_create, append every new pick; FID_PLAN_RO=<file>: load only.  (No reference analogue: onnxruntime picks its kernels internally.) */
int fid_net_plan_save(fid_net *net, const char *path);
int fid_net_plan_load(fid_net *net, const char *path, int *n_loaded);
/* algorithmic cost of one image through the net: multiply-accumulates (true channel counts) */
int fid_net_macs(fid_net *net, double *macs_per_image);

/* ---- letterbox: replaces cv2.resize + zero paste, reference models/scrfd.py:123-138 ---------
 * frames [B,H,W,3] -> out [B,in_h,in_w,3]; det_scale (new_h / H, as a double) is returned. */
int fid_letterbox(fid_ctx *ctx, const uint8_t *frames_dev, int B, int H, int W,
                  uint8_t *out_dev, int in_h, int in_w, double *det_scale);

/* ---- SCRFD post-process: replaces the numpy code of reference models/scrfd.py:89-178,180-207
 * and utils/helpers.py:62-107 (threshold, distance2bbox/kps decode, sort, greedy NMS, max_num).
 * The 9 head tensors (scores, bbox, kps for strides 8/16/32) are given as strided views so both
 * the ONNX layout ([H*W*A,1|4|10]) and the executor's fused NHWC head tensor can be read:
 *   element(frame b, anchor i, component c) =
 *       ptr[k][ b*batch_stride[k] + (i / A)*pix_stride[k] + (i % A)*anc_stride[k] + c ]
 * Outputs (device): det [B,cap,5] (x1,y1,x2,y2,score), kps [B,cap,10], counts [B].
 * All frames of one call share the original image size (img_h,img_w) -- they are one dense array.
 * metric: 0 = "max" (area), 1 = area - 2*centre_dist^2 (scrfd.py:169-172).
 * Returns FID_E_CAPACITY (after the fact, via fid_scrfd_check) if a frame had more survivors
 * than `cap` or more candidates than cand_cap. */
int fid_scrfd_postprocess(fid_ctx *ctx, const float *const head_dev[9], const int32_t pix_stride[9],
                          const int32_t anc_stride[9], const int64_t batch_stride[9], int B,
                          int in_h, int in_w, int num_anchors, int img_h, int img_w, float conf_thres,
                          float iou_thres, int max_num, int metric, float *det_dev, float *kps_dev,
                          int32_t *counts_dev, int cap);
/* candidates per frame the sort/NMS workspace is sized for (default 4096; max 16800 = all anchors
 * of a 640x640 input) */
int fid_scrfd_set_candidate_capacity(fid_ctx *ctx, int cand_cap);
/* synchronises and reports overflow of the last fid_scrfd_postprocess: max candidates seen */
int fid_scrfd_check(fid_ctx *ctx, int *max_candidates);
/* SCRFD.forward's decode loop alone (reference models/scrfd.py:89-119): the candidates with
 * score >= conf_thres in anchor order (levels 8,16,32 concatenated), NOT divided by det_scale.
 * rec_dev: [B, cand_cap, 16] floats = x1 y1 x2 y2 score kps[10] flat-anchor-index(int bits). */
int fid_scrfd_decode(fid_ctx *ctx, const float *const head_dev[9], const int32_t pix_stride[9],
                     const int32_t anc_stride[9], const int64_t batch_stride[9], int B, int in_h,
                     int in_w, int num_anchors, float conf_thres, float *rec_dev, int32_t *counts_dev);
/* distance2bbox / distance2kps on plain device arrays (reference utils/helpers.py:62-107):
 * points [n,2], distance [n,4] / [n,ncol] -> out [n,4] / [n,ncol] */
int fid_distance2bbox(fid_ctx *ctx, const float *points_dev, const float *dist_dev, int n, float *out_dev);
int fid_distance2kps(fid_ctx *ctx, const float *points_dev, const float *dist_dev, int n, int ncol,
                     float *out_dev);
/* SCRFD.nms(dets, iou_thres) itself (reference models/scrfd.py:180-207): dets [K,5] device,
 * keep_dev receives indices into dets in keep order, count_dev[0] their number. */
int fid_nms(fid_ctx *ctx, const float *dets_dev, int K, float iou_thres, int32_t *keep_dev,
            int32_t *count_dev);

/* ---- alignment: replaces skimage SimilarityTransform.estimate + cv2.warpAffine +
 * cv2.dnn.blobFromImages' layout step; reference utils/helpers.py:18-59, arcface.py:54-57.
 * For frame b and face slot f < faces_per_frame: uses kps[b, f, :] if f < counts[b], else the
 * crop is zero-filled.  crops: uint8 [B*faces_per_frame, 112,112,3] BGR.  M_dev (optional, may
 * be NULL): the 2x3 double matrices estimate_norm would return, [B*faces_per_frame, 6]. */
int fid_align_crops(fid_ctx *ctx, const uint8_t *frames_dev, int B, int H, int W,
                    const float *kps_dev, const int32_t *counts_dev, int cap, int faces_per_frame,
                    uint8_t *crops_dev, double *M_dev);

/* ---- face gates of the reference's product layer: replaces smart_face_recognition.py:1145-1216
 * (assess_face_quality), :1218-1297 (get_face_pose_angles / is_side_face), :1299-1399 (analyze_bbox_for_side_face) and the
 * best-face selection with its four rejections (:1473-1519), for every face of a batch in one launch on the post-process's own
 * device arrays.  The thresholds are the reference's config.json blocks face_quality / side_face_detection / face_detection. */
typedef struct fid_gate_config {
    float size_normalization;
    float w_detection, w_size, w_blur, w_pose, w_lighting;
    float ar_extreme_profile, ar_very_strong_profile, ar_strong_profile, ar_very_wide, ar_wide, ar_moderately_wide;
    float area_extremely_small, area_very_small, area_small, area_very_large, area_large;
    float compactness_very_low, compactness_low;
    float confidence_very_low, confidence_low;
    float edge_position_threshold;
    int32_t decision_threshold;
    float yaw_threshold, pitch_threshold;              /* degrees */
    float confidence_threshold, min_quality_threshold;
} fid_gate_config;
enum { FID_GATE_ACCEPT = 0, FID_GATE_NO_FACE = 1, FID_GATE_LOW_CONFIDENCE = 2, FID_GATE_SIDE_FACE = 3, FID_GATE_LOW_QUALITY = 4 };
/* det [B,cap,5], kps [B,cap,10], counts [B] as fid_scrfd_postprocess writes them; the first min(counts[b], faces_per_frame) slots
 * of frame b are faces.  pose (optional, may be NULL): FLOAT64 [B, faces_per_frame, 2] yaw / pitch in radians (the reference hands python floats to
 * math.degrees, :1226-1240: float32 here would move the verdict of an angle at the threshold), 0 = not available (then the
 * bbox analysis decides, as in the reference).  Outputs (device): quality [B, faces_per_frame, 5] = overall, blur, pose, lighting,
 * size (zeros in empty slots); side [B, faces_per_frame] = analyze_bbox_for_side_face's score | is_side_face << 16; best [B, 2] =
 * index of the FIRST face with the highest det_score (-1: no face) and its verdict (FID_GATE_*), checked in the reference's
 * order: confidence_threshold, side face, min_quality_threshold. */
int fid_face_gates(fid_ctx *ctx, const float *det_dev, const float *kps_dev, const int32_t *counts_dev, int B, int cap,
                   int faces_per_frame, const double *pose_dev, const fid_gate_config *cfg, float *quality_dev,
                   int32_t *side_dev, int32_t *best_dev);

/* ---- embeddings -> unit fp16 rows: the norm half of reference utils/helpers.py:120-123 ------ */
int fid_l2_normalize_f16(fid_ctx *ctx, const float *emb_dev, int n, int dim, void *out_f16_dev);
/* the same for the n = B * faces_per_frame face slots of a batch: slot (b, f) with f >= counts[b] holds no face (reference
 * main.py:132 iterates detected faces only) and is written as a zero row, which scores 0 against every gallery row and so never
 * matches (fid_match: idx -1, score 0).  Its FIRST element is -0.0 (fp16 bit pattern 0x8000), every other +0.0: the same numbers, but
 * told apart from the all +0.0 row a degenerate embedding (zero / NaN / inf norm) of a DETECTED face gets.  The unit-embedding matrix
 * thereby carries the face counts exactly: after the all-gather of SURVEY.md 8e every rank can tell another rank's faces from its
 * empty slots (row == {-0.0, +0.0 ...} <=> no face; pipeline.gathered_face_counts). */
int fid_l2_normalize_f16_slots(fid_ctx *ctx, const float *emb_dev, int n, int dim, const int32_t *counts_dev,
                               int faces_per_frame, void *out_f16_dev);

/* ---- gallery match: replaces the per-target python loop of reference main.py:136-142 --------
 * gallery: host fp32 [G, dim] raw embeddings (as build_targets collects them, main.py:102-103).
 * fid_match: for every query row (unit fp16) the first index of the maximum cosine, provided
 * it is > max(0, thresh); otherwise idx = -1 and score = 0 (strict '>' like main.py:140). */
int fid_gallery_create(fid_ctx *ctx, const float *gallery, int G, int dim, fid_gallery **out);
int fid_gallery_destroy(fid_ctx *ctx, fid_gallery *g);
int fid_match(fid_ctx *ctx, fid_gallery *g, const void *query_f16_dev, int n, float thresh,
              int32_t *idx_dev, float *score_dev);
int fid_gallery_info(fid_gallery *g, int *G, int *G_padded, int *dim);
/* device address of the unit fp16 rows [G_padded, dim] (read-only view; owned by the gallery) */
int fid_gallery_data(fid_gallery *g, void **unit_rows_dev);
/* Vector-store use of the gallery -- the product layer's QdrantManager (reference qdrant_manager.py:91-212,
 * smart_face_recognition.py:1619-1643): top-k cosine search with a score threshold (k in {1,2,4,5,8};
 * results score-descending, index-ascending on ties, -1/0 beyond the last hit) and in-place upsert / delete
 * of rows (an all-zero embedding deletes: a zero row can never match). */
int fid_gallery_topk(fid_ctx *ctx, fid_gallery *g, const void *query_f16_dev, int n, int k, float thresh,
                     int32_t *idx_dev, float *score_dev);
int fid_gallery_set_rows(fid_ctx *ctx, fid_gallery *g, const int32_t *rows_host, const float *emb_host, int n);
/* Gallery sharded over ranks by contiguous row blocks (the 1 M-entry variant of main.py:136-142):
 * fid_match_keys scans THIS rank's rows (global index of its row 0 = first_row) for all n queries and writes one
 * packed key per query, (order-preserving bits of the score << 32) | ~global_index; after the ranks' key arrays
 * have been all-gathered ([parts, n], 8 bytes per query and rank) fid_match_merge takes the maximum key per query
 * = the best score, lowest global index on ties, and applies the strict '>' threshold like fid_match. */
int fid_match_keys(fid_ctx *ctx, fid_gallery *g, const void *query_f16_dev, int n, int first_row,
                   uint64_t *keys_dev);
int fid_match_merge(fid_ctx *ctx, const uint64_t *keys_dev, int parts, int n, int G_total, float thresh,
                    int32_t *idx_dev, float *score_dev);

/* ---- the path's one collective (no reference analogue: the reference is single-process; spec = BASELINE.json
 * north_star "a single RCCL all-gather over xGMI of per-rank embeddings before the gallery match").
 * One process per GPU.  Rank 0 calls fid_comm_unique_id and the host hands the FID_COMM_ID_BYTES bytes to every
 * rank (file, environment, MPI, a torch.distributed store ...); every rank then calls fid_comm_init_rank (collective,
 * blocks until all ranks arrived).  fid_allgather enqueues ncclAllGather on the context's stream:
 * recv_dev [nranks][bytes_per_rank] <- each rank's send_dev, in rank order. */
#define FID_COMM_ID_BYTES 128
int fid_comm_unique_id(void *id_out, size_t bytes);
int fid_comm_init_rank(fid_ctx *ctx, int nranks, int rank, const void *id, size_t bytes, fid_comm **out);
int fid_comm_destroy(fid_ctx *ctx, fid_comm *comm);
int fid_comm_info(fid_comm *comm, int *nranks, int *rank);
int fid_allgather(fid_ctx *ctx, fid_comm *comm, const void *send_dev, void *recv_dev, size_t bytes_per_rank);

/* full cosine matrix fp32 [n, G_padded] (row stride = G_padded, a multiple of 32; columns >= G are
 * 0): tests / compute_similarity parity.  Caller-allocated device memory. */
int fid_cosine_matrix(fid_ctx *ctx, fid_gallery *g, const void *query_f16_dev, int n,
                      float *out_dev);

#ifdef __cplusplus
}
#endif
#endif /* FACEID_H */
