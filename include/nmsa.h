/*
 * nmsa.h — C ABI of libnmsa_hip.so: the MI355X (gfx950) implementation of the
 * dense-prediction hot path of nicr-mt-scene-analysis.
 *
 * The reference is pure Python (no FFI of its own); each entry point below names
 * the reference Python symbol whose ATen-op chain it replaces
 * (paths relative to src/nicr_mt_scene_analysis/ of the reference).  The Python
 * mirror of the reference API (package nicr_mt_scene_analysis_amd) binds these
 * with ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (contiguous, NCHW /
 *    row-major as noted), unless the parameter name ends in _host;
 *  - all work is enqueued on `stream` (a hipStream_t); nothing synchronises,
 *    nothing allocates; scratch is passed in by the caller and sized with the
 *    matching *_workspace_bytes() query;
 *  - return value: NMSA_OK (0) or a negative NMSA_ERR_* code; nothing throws.
 *  - "bool" tensors are uint8 (torch.bool storage), 0 / non-0.
 */
#ifndef NMSA_H
#define NMSA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nmsa_stream_t; /* hipStream_t */

enum {
    NMSA_OK = 0,
    NMSA_ERR_ARG = -1,         /* invalid argument (shape, null pointer, range)      */
    NMSA_ERR_LAUNCH = -2,      /* HIP launch / runtime error (see nmsa_last_hip_error) */
    NMSA_ERR_WORKSPACE = -3,   /* workspace too small                                  */
    NMSA_ERR_UNSUPPORTED = -4  /* valid in the reference, not implemented here        */
};

/* floating-point element types of prediction tensors */
enum { NMSA_F32 = 0, NMSA_BF16 = 1, NMSA_F16 = 2 };
/* integer element types of label / id tensors */
enum { NMSA_U8 = 0, NMSA_I16 = 1, NMSA_I32 = 2, NMSA_I64 = 3 };
/* nmsa_resize_nearest also moves f32 maps (scores): */
#define NMSA_ELEM_F32 8

#define NMSA_MAX_INSTANCE_IDS 256 /* instance ids are uint8 in the reference (instance.py:236) */

int nmsa_version(void);
const char* nmsa_strerror(int code);
int nmsa_last_hip_error(void); /* last hipError_t seen by this library (thread-local) */
/* What the library sizes its grids from: compute units, XCDs and LDS bytes per CU of the CURRENT
 * device, queried once per device from the HIP runtime (no literal 256 / 8: a partitioned MI355X
 * shows fewer).  The environment variables NMSA_ASSUME_CUS / NMSA_ASSUME_XCDS override the
 * queried numbers (tests).  Any pointer may be NULL.  Host pointers. */
int nmsa_device_geometry(int* cus_host, int* xcds_host, size_t* lds_per_cu_host);

/* ---------------------------------------------------------------------------
 * a2  InstancePostprocessing._get_instance_centers
 *     model/postprocessing/instance.py:79-168
 * threshold -> k x k max-pool NMS (first maximum in the window wins, border of
 * (k-1)/2 px never a center) -> per-image top-k value (before the optional
 * foreground mask) -> keep >= kth (ties kept) -> raster-ordered coordinates.
 *   center        f32 [B,H,W]           (the [B,1,H,W] head output)
 *   fg            u8  [B,H,W] or NULL   (required iff apply_fg)
 *   centers_yx    i32 [B,max_centers,2] (y,x) raster order, first n_centers[b] valid
 *   n_centers     i32 [B]  TRUE count (may exceed max_centers: caller re-runs larger)
 *   scores        f32 [B,max_centers]   raw heatmap value at each center (meta 'score')
 *   center_mask   u8  [B,H,W] or NULL   the boolean map the reference also returns
 * ------------------------------------------------------------------------- */
size_t nmsa_center_nms_workspace_bytes(int B, int H, int W);
int nmsa_center_nms_topk(const float* center, const uint8_t* fg,
                         int B, int H, int W,
                         float threshold, int ksize, int topk, int apply_fg,
                         int max_centers,
                         int32_t* centers_yx, int32_t* n_centers, float* scores,
                         uint8_t* center_mask,
                         void* workspace, size_t workspace_bytes,
                         nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * a3  InstancePostprocessing._get_instance_segmentation (grouping part)
 *     model/postprocessing/instance.py:187-253
 * For every foreground pixel: loc = (y,x) + offset*(scale_y,scale_x); id =
 * 1 + argmin_i ||center_i - loc||_2 (fp32, lowest index on ties, uint8 wrap);
 * optional `min_dist > dist_thr -> 0`.
 *   offset  f32 [B,2,H,W] (dy,dx); scale = (H,W) for normalized offsets, else 1
 *   inst    u8  [B,H,W]   (0 outside fg)
 *   area    i32 [B,256]   bincount of ids over fg (meta 'area'); may be NULL
 * ------------------------------------------------------------------------- */
int nmsa_group_offsets(const float* offset, const uint8_t* fg,
                       const int32_t* centers_yx, const int32_t* n_centers,
                       int B, int H, int W, int max_centers,
                       float scale_y, float scale_x,
                       int use_dist_thr, float dist_thr,
                       uint8_t* inst, int32_t* area,
                       nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * a1  SemanticPostprocessing._postprocess_inference (argmax + score)
 *     model/postprocessing/semantic.py:52-53
 *   logits  f32|bf16|f16 [B,C,H,W];  any of the outputs may be NULL
 *   idx_u8 [B,H,W] (C <= 256), idx_i64 [B,H,W], score f32 [B,H,W] = max softmax
 * ------------------------------------------------------------------------- */
int nmsa_semantic_argmax(const void* logits, int logits_dtype,
                         int B, int C, int H, int W,
                         uint8_t* idx_u8, int64_t* idx_i64, float* score,
                         nmsa_stream_t stream);

/* softmax over C (semantic.py:52 'semantic_softmax_scores'), f32 out [B,C,H,W] */
int nmsa_semantic_softmax(const void* logits, int logits_dtype,
                          int B, int C, int H, int W, float* probs,
                          nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * a1+a3+a4+a5  PanopticPostprocessing._postprocess_inference (core)
 *     model/postprocessing/panoptic.py:77-168 with
 *     utils/panoptic_merge.py:172-225 (deeplab_merge_semantic_and_instance)
 *
 * nmsa_panoptic_fused: one pass over the logits: per-pixel argmax (a1),
 * foreground = is_thing[class] (panoptic.py:123-127), offset grouping (a3) and
 * the per-instance class-vote histogram of the merge (panoptic_merge.py:201).
 *   is_thing   u8 [C]
 *   sem_u8     u8 [B,H,W]   class index 0..C-1 (required: paint reads it)
 *   inst       u8 [B,H,W]
 *   fg_out     u8 [B,H,W] or NULL ('panoptic_foreground_mask')
 *   score      f32 [B,H,W] or NULL
 *   votes      u32 [B,256,C+1]; votes[b,id,c] = #px of instance id whose (class+1) == c.
 *              Zeroed by this call unless votes_are_zero != 0 (the caller keeps a
 *              persistent table that nmsa_panoptic_assign(clear_votes=1) left clean)
 *   vote_rows_hint  kept for ABI stability, ignored: the per-workgroup vote counters are a
 *              fixed-size LDS hash table keyed by (instance id, class); 0 is fine
 * nmsa_panoptic_assign: per instance: class = mode (smallest on ties), running
 * per-class counter in ascending id order -> panoptic id (panoptic_merge.py:
 * 192-210).
 *   pan_of_inst i64 [B,256]  (void_label where the instance is dropped)
 *   area        i32 [B,256]  row sums of votes (meta 'area'; index 0 unused)
 *   ids_pan/ids_ins i64 [B,256], n_ids i32 [B]: the id dict in insertion order
 * nmsa_panoptic_paint: pan[p] = inst ? pan_of_inst[inst] :
 *                               (is_thing[sem] ? void : (sem+1)*max_inst)
 *   pan i64 [B,H,W]; pan_sem i64 [B,H,W] or NULL (= pan // max_inst, panoptic.py:160)
 * ------------------------------------------------------------------------- */
int nmsa_panoptic_fused(const void* logits, int logits_dtype, const float* offset,
                        const int32_t* centers_yx, const int32_t* n_centers,
                        const uint8_t* is_thing,
                        int B, int C, int H, int W, int max_centers,
                        float scale_y, float scale_x,
                        int use_dist_thr, float dist_thr,
                        uint8_t* sem_u8, uint8_t* inst, uint8_t* fg_out, float* score,
                        uint32_t* votes, int votes_are_zero, int vote_rows_hint,
                        nmsa_stream_t stream);

int nmsa_panoptic_assign(uint32_t* votes, int B, int n_vote_classes, int clear_votes,
                         int64_t max_instances_per_category, int64_t void_label,
                         int64_t* pan_of_inst, int32_t* area,
                         int64_t* ids_pan, int64_t* ids_ins, int32_t* n_ids,
                         nmsa_stream_t stream);

int nmsa_panoptic_paint(const uint8_t* sem_u8, const uint8_t* inst,
                        const int64_t* pan_of_inst, const uint8_t* is_thing,
                        int B, int C, int H, int W,
                        int64_t max_instances_per_category, int64_t void_label,
                        int64_t* pan, int64_t* pan_sem,
                        nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * a5  deeplab_merge_batch   utils/panoptic_merge.py:18-40,172-225
 * Generic entry (GT path / InstanceTaskHelper): arbitrary integer dtypes.
 *   sem        [B,H,W] sem_dtype, class values 0..n_classes-1 (0 = void)
 *   ins        [B,H,W] ins_dtype, instance ids 0..255
 *   thing_seg  u8 [B,H,W]
 *   is_thing_class u8 [n_classes]  (class value c is in thing_ids)
 *   votes      u32 [B,256,n_classes] scratch, zeroed by this call
 * Outputs as for assign + paint.
 * ------------------------------------------------------------------------- */
int nmsa_panoptic_merge(const void* sem, int sem_dtype, const void* ins, int ins_dtype,
                        const uint8_t* thing_seg, const uint8_t* is_thing_class,
                        int B, int n_classes, int H, int W,
                        int64_t max_instances_per_category, int64_t void_label,
                        uint32_t* votes, int64_t* pan_of_inst,
                        int64_t* pan, int64_t* ids_pan, int64_t* ids_ins, int32_t* n_ids,
                        nmsa_stream_t stream);

/* Same merge for instance ids beyond uint8 (ground-truth maps: uint16 ids stored as
 * int32, e.g. > 256 instances per image; task_helper/instance.py:61).  Ids 0..65535
 * are ranked per image first (ascending order preserved), at most max_segments
 * (<= 65536: every id such a map can hold; n_classes <= 65535) distinct thing ids per image.
 *   ids_pan / ids_ins  i64 [B, cap] with cap = max_segments rounded up to 1024
 *   status  i32 [1]: NMSA_ST_TABLE_OVERFLOW (more ids than max_segments),
 *                    32 = instance id outside [0, 65535]
 */
size_t nmsa_panoptic_merge_wide_workspace_bytes(int B, int n_classes, int max_segments);
int nmsa_panoptic_merge_wide(const void* sem, int sem_dtype, const void* ins, int ins_dtype,
                             const uint8_t* thing_seg, const uint8_t* is_thing_class,
                             int B, int n_classes, int H, int W,
                             int64_t max_instances_per_category, int64_t void_label,
                             int max_segments,
                             int64_t* pan, int64_t* ids_pan, int64_t* ids_ins, int32_t* n_ids,
                             int32_t* status, void* workspace, size_t workspace_bytes,
                             nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * f2  DensePostprocessingBase._crop_to_valid_region_and_resize_prediction
 *     model/postprocessing/dense_base.py:15-58   (the full-resolution step)
 * Every plane of `src` ([planes,Hs,Ws], planes = B or B*C) is cropped to the valid
 * region [y0:y0+h, x0:x0+w] and resized to [Ho,Wo]; dst is [planes,Ho,Wo].
 * The arithmetic is bit-for-bit the one of ATen's CPU kernels (see csrc/resize.hip):
 *   nearest : src = min(int(floorf(dst * float(in)/float(out))), in-1); i32 / i64 maps
 *             take the reference's float32 round trip (dense_base.py:38-40, :52)
 *   bilinear: align_corners=False, width first; dst has the dtype of src
 *   elem_type: NMSA_U8 (also bool) | NMSA_I16 | NMSA_I32 | NMSA_I64 | NMSA_ELEM_F32
 * ------------------------------------------------------------------------- */
int nmsa_resize_nearest(const void* src, int elem_type, int planes, int Hs, int Ws,
                        int y0, int x0, int h, int w, int Ho, int Wo, void* dst,
                        nmsa_stream_t stream);
int nmsa_resize_bilinear(const void* src, int dtype, int planes, int Hs, int Ws,
                         int y0, int x0, int h, int w, int Ho, int Wo, void* dst,
                         nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * f2  SemanticPostprocessing full-resolution argmax   model/postprocessing/semantic.py:61-80
 * crop + bilinear resize + argmax + max-softmax score of the logits in ONE pass; the
 * [B,C,Ho,Wo] full-resolution logits are not materialised (nmsa_resize_bilinear makes
 * them when `semantic_output_fullres` itself is read).  Same outputs / NULL rules as
 * nmsa_semantic_argmax; logits [B,C,Hs,Ws].
 * ------------------------------------------------------------------------- */
int nmsa_semantic_argmax_resized(const void* logits, int logits_dtype, int B, int C,
                                 int Hs, int Ws, int y0, int x0, int h, int w, int Ho, int Wo,
                                 uint8_t* idx_u8, int64_t* idx_i64, float* score,
                                 nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * f3  PanopticPostprocessing `compute_scores` branch   model/postprocessing/panoptic.py:171-239
 * Dense semantic / instance / panoptic score maps and the per-instance mean semantic score,
 * without the [B,C,H,W] softmax tensor and without the per-instance Python loop.
 *   logits      f32|bf16|f16 [B,C,H,W]
 *   sem_idx     u8  [B,H,W]  argmax class (nmsa_panoptic_fused / nmsa_semantic_argmax)
 *   sem_prob    f32 [B,H,W]  its softmax probability (the `score` output of the same calls)
 *   inst        u8  [B,H,W]; pan i64 [B,H,W] = class_value * max_instances_per_category + k
 *   pan_of_inst i64 [B,256]  panoptic id of every instance id (nmsa_panoptic_assign)
 *   inst_score_tab f32 [B,256]  instance score by instance id (meta 'score'; entry 0 unused)
 * outputs (f32 [B,H,W]):
 *   semantic score = softmax probability of the pixel's PANOPTIC class (0 on void),
 *   instance score = the instance's score on its painted pixels, else 0,
 *   panoptic score = mean semantic score of the instance x instance score on painted pixels,
 *                    else the semantic score;
 *   mean_semantic_score f32 [B,256] (meta 'semantic_score'; NaN for unused ids), may be NULL
 * ------------------------------------------------------------------------- */
size_t nmsa_panoptic_scores_workspace_bytes(int B);
int nmsa_panoptic_scores(const void* logits, int logits_dtype,
                         const uint8_t* sem_idx, const float* sem_prob,
                         const uint8_t* inst, const int64_t* pan,
                         const int64_t* pan_of_inst, const float* inst_score_tab,
                         int B, int C, int H, int W, int64_t max_instances_per_category,
                         float* out_semantic_score, float* out_instance_score,
                         float* out_panoptic_score, float* mean_semantic_score,
                         void* workspace, size_t workspace_bytes, nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * next-1  InstancePostprocessing._get_instance_orientation
 *     model/postprocessing/instance.py:271-319
 *   orientation f32 [B,2,H,W]; inst u8; mask u8 or NULL
 *   sums f64 [B,256,2] (sum of channel 0 / 1 per id), count i32 [B,256]; both
 *   zeroed by this call
 *   (angle = atan2(sum1, sum0) is taken on the host over <= 255 entries)
 * ------------------------------------------------------------------------- */
int nmsa_instance_orientation(const float* orientation, const uint8_t* inst,
                              const uint8_t* mask, int B, int H, int W,
                              double* sums, int32_t* count, nmsa_stream_t stream);
/* the same for ground-truth instance maps (ids 0..65535 in any integer dtype, at most
 * max_instances <= 4096 distinct masked ids per image; instance.py:432-438 passes
 * batch['instance']): ids i32 [B,cap] ascending, n_ids i32 [B], sums f64 [B,cap,2],
 * count i32 [B,cap] by position in `ids` (cap = max_instances rounded up to 1024);
 * status bits: 1 too many ids, 32 id out of range;
 * workspace: nmsa_targets_workspace_bytes(B, 1, max_instances) */
int nmsa_instance_orientation_wide(const float* orientation, const void* instance, int ins_dtype,
                                   const uint8_t* mask, int B, int H, int W, int max_instances,
                                   int32_t* ids, int32_t* n_ids, double* sums, int32_t* count,
                                   int32_t* status, void* workspace, size_t workspace_bytes,
                                   nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * f4  ground-truth target generation on the device (per batch instead of per sample in the
 *     dataloader workers).  Label maps in their on-wire dtypes (data/preprocessing/torch.py:60-66:
 *     semantic uint8, instance uint16 -> int32); instance ids in [0, 65535], at most
 *     `max_instances` (<= 4096) distinct ids per image.
 *  status bits (OR-ed into *status): 1 too many distinct ids, 32 id out of range,
 *     64 semantic label outside [0, n_classes), 128 id table (max_segments) overflow.
 *  workspace: nmsa_targets_workspace_bytes(B, n_classes, max_instances), 8-byte aligned (16-byte
 *     aligned for the one-launch front end of the on-wire layout, see below).
 *     workspace_is_clean (nmsa_instance_targets / nmsa_panoptic_targets): 1 = this very workspace
 *     (same B, n_classes, max_instances) was last used by one of these two calls on the on-wire
 *     layout and nothing else wrote to it since — they leave their tables zeroed, so the call
 *     skips its memset; 0 = unknown contents (first use): the call zeroes what it needs.
 *  status: on the on-wire layout the word is SET by the call (no need to zero it); on any other
 *     layout the bits are OR-ed into it (zero it first).
 *  launches: the on-wire layout (semantic uint8, instance int32, 16-byte aligned rows of 4 pixels,
 *     W % 4 == 0, n_classes <= 16384) runs as [memset +] ONE scan launch (presence, statistics,
 *     rank, decide / naive ranks: k_tg_scan) + the paint launch; any other layout as memset + 5.
 *
 *  nmsa_instance_clear_stuff   InstanceClearStuffIDs._preprocess   data/preprocessing/instance.py:46-93
 *     instance[is_stuff_class[semantic]] = 0, in place (is_stuff_class includes void)
 *  nmsa_instance_targets       InstanceTargetGenerator._preprocess data/preprocessing/instance.py:157-286
 *     is_thing_class u8 [n_classes] or NULL (every instance encoded); is_stuff_class u8 [n_classes]
 *     or NULL (center mask = foreground); gauss_lut f32 [2*(3*sigma+1)^2 + 1]: heat-map value by
 *     integer squared distance to the center, float32(exp(-d2 / (2 sigma^2)));
 *     center f32 [B,H,W]; offset [B,2,H,W] (dy,dx): f32 divided by (H,W) when normalized_offset,
 *     else i16; foreground / center_mask u8 [B,H,W] (center_mask may be NULL);
 *     encoded_ids / skipped_ids i32 [B, cap] ascending (cap = max_instances rounded up to 1024),
 *     n_encoded / n_skipped i32 [B]   (the reference's dynamic parameters; any may be NULL)
 *  nmsa_panoptic_targets       PanopticTargetGenerator._preprocess data/preprocessing/panoptic.py:48-85
 *                              = naive_merge_semantic_and_instance_np utils/panoptic_merge.py:43-107
 *     panoptic i64 [B,H,W]; ids_pan / ids_ins i64 [B,max_segments] in the reference's dict order
 *     (instance ascending, then class ascending), n_ids i32 [B]
 *  nmsa_dve_targets            DenseVisualEmbeddingTargetGenerator  data/preprocessing/
 *                              dense_visual_embedding.py:22-93
 *     keys i64 [B,K] (first n_keys[b] valid), embeddings f32 [B,K,D], image_embedding f32 [B,D];
 *     lut f32 [B,K,D] = normalise(embedding - diff_factor * image_embedding) (NULL: indices only);
 *     indices i32 [B,H,W] = 1 + position of the pixel's panoptic id in keys (0 = none)
 * ------------------------------------------------------------------------- */
size_t nmsa_targets_workspace_bytes(int B, int n_classes, int max_instances);
int nmsa_instance_clear_stuff(const void* semantic, int sem_dtype, void* instance, int ins_dtype,
                              const uint8_t* is_stuff_class, int n_classes, int64_t n_px,
                              nmsa_stream_t stream);
int nmsa_instance_targets(const void* semantic, int sem_dtype, const void* instance, int ins_dtype,
                          const uint8_t* is_thing_class, const uint8_t* is_stuff_class,
                          int B, int n_classes, int H, int W,
                          int sigma, const float* gauss_lut, int normalized_offset, int max_instances,
                          float* center, void* offset, uint8_t* foreground, uint8_t* center_mask,
                          int32_t* encoded_ids, int32_t* n_encoded,
                          int32_t* skipped_ids, int32_t* n_skipped,
                          int32_t* status, void* workspace, size_t workspace_bytes, int workspace_is_clean,
                          nmsa_stream_t stream);
int nmsa_panoptic_targets(const void* semantic, int sem_dtype, const void* instance, int ins_dtype,
                          const uint8_t* is_thing_class, int B, int n_classes, int H, int W,
                          int64_t max_instances_per_category, int64_t void_label,
                          int max_instances, int max_segments,
                          int64_t* panoptic, int64_t* ids_pan, int64_t* ids_ins, int32_t* n_ids,
                          int32_t* status, void* workspace, size_t workspace_bytes, int workspace_is_clean,
                          nmsa_stream_t stream);
int nmsa_dve_targets(const int64_t* panoptic, const int64_t* keys, const int32_t* n_keys,
                     const float* embeddings, const float* image_embedding, float diff_factor,
                     int B, int K, int D, int H, int W,
                     float* lut, int32_t* indices, nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * a11  MeanIntersectionOverUnion.update   metric/miou.py:44-56
 *   confmat[t, p] += 1 for every element (bincount(t*n + p, minlength=n*n)).
 *   preds / target: any integer dtype, n_px elements each.
 *   pred_div : preds are used as raw // pred_div (1 = as is; 65536 gives the
 *              `pan // max_instances_per_category` of task_helper/panoptic.py:123)
 *   mode 0   : every element counts
 *   mode 1   : elements with target == 0 are skipped and target-1 is used
 *              (the void masking of task_helper/semantic.py:124-128)
 *   confmat  : i64 [n_classes, n_classes], accumulated into (not zeroed)
 *   status   : i32 [1] device word, OR-ed with NMSA_ST_* bits (value out of range
 *              = what makes the reference's bincount/reshape raise)
 * ------------------------------------------------------------------------- */
#define NMSA_ST_TABLE_OVERFLOW 1
#define NMSA_ST_CATEGORY_RANGE 2
#define NMSA_ST_MISSING_KEY 4
#define NMSA_ST_VALUE_RANGE 8
#define NMSA_ST_SENTINEL_KEY 16
size_t nmsa_confmat_workspace_bytes(int n_classes);
int nmsa_confmat_update(const void* preds, int pred_dtype, int64_t pred_div,
                        const void* target, int target_dtype,
                        int64_t n_px, int n_classes, int mode,
                        int64_t* confmat, int32_t* status,
                        void* workspace, size_t workspace_bytes, nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * a12/a13  compare_and_accumulate + PanopticQuality.update
 *     metric/pq.py:60-179, :262-296
 *   pred, target : i64 [B,H,W] panoptic maps (category*max_inst + instance)
 *   iou/tp/fn/fp : f64 [num_categories] states, accumulated into (image order;
 *                  the IoU sums are bit-identical to the reference's fp64 sums)
 *   matches      : i64 [B,match_capacity,2] (gt id, pred id) of the TP pairs in
 *                  ascending intersection-id order, or NULL; n_matches i32 [B]
 *   status       : i32 [1] device word, OR-ed with NMSA_ST_* bits
 *   limits       : <= 2048 distinct ids per image and side, num_categories <= 1024;
 *                  distinct (target, pred) intersections per image <= cap / 2 with
 *                  cap = H*W / 24 rounded up to a power of two in [4096, 131072]
 *                  (640x480: 8192 intersections, 1024x768: 16384); beyond that
 *                  NMSA_ST_TABLE_OVERFLOW is raised
 *   workspace_is_clean : non-zero when `workspace` was last used by a completed
 *                  nmsa_pq_update of the same B, H, W (which leaves the tables empty); the
 *                  per-call table initialisation is then skipped
 * ------------------------------------------------------------------------- */
size_t nmsa_pq_workspace_bytes(int B, int H, int W, int num_categories);
int nmsa_pq_update(const int64_t* pred, const int64_t* target, int B, int H, int W,
                   int num_categories, int64_t ignored_label,
                   int64_t max_instances_per_category, int64_t offset,
                   int64_t void_segment_id,
                   double* iou_per_class, double* tp_per_class,
                   double* fn_per_class, double* fp_per_class,
                   int64_t* matches, int match_capacity, int32_t* n_matches,
                   int32_t* status, void* workspace, size_t workspace_bytes,
                   int workspace_is_clean, nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * Host hand-over of the per-image tables of one pipeline run (the Python objects of
 *     panoptic.py:118-167 — id dicts, instance meta — are built on the host): packs
 *     [n_centers, n_ids, centers_yx[kc][2], scores[kc], area[min(kc+1,256)],
 *      ids_pan[min(kc,256)], ids_ins[min(kc,256)]], kc = min(columns, max_centers), into one
 *     f64 row per image (exact: ids < 2^53, f32 scores) -> ONE device->host copy per batch.
 *     n_ids / ids_pan / ids_ins may be NULL together (instance-only postprocessing): the row
 *     then ends after the areas and its second entry is 0.
 *     `out` may also be pinned host memory (hipHostMalloc: mapped into the device's address space
 *     at the same address): the kernel then stores the rows where the host reads them — valid
 *     once an event recorded behind the call has completed — and no copy is needed at all.
 * ------------------------------------------------------------------------- */
int nmsa_pack_tables(const int32_t* n_centers, const int32_t* n_ids,
                     const int32_t* centers_yx, const float* scores,
                     const int32_t* area, const int64_t* ids_pan, const int64_t* ids_ins,
                     int B, int max_centers, int columns, double* out, nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * a11 + a12 in ONE pass over the prediction (PanopticTaskHelper.validation_step,
 *     task_helper/panoptic.py:104-126, updates both metrics from the same panoptic map):
 *     PQ exactly as nmsa_pq_update, plus confmat[target_semantic, pred // pred_div] += 1
 *     exactly as nmsa_confmat_update(mode 0) — the i64 prediction map is read once.
 *   target_semantic u8 [B,H,W]; confmat i64 [n,n] with n = confmat_classes <= 64;
 *   confmat_status: the status word of the mIoU metric (label outside [0, n) -> bit 8);
 *   confmat_workspace: nmsa_pq_confmat_workspace_bytes(B,H,W,n) bytes (0 = not supported).
 * ------------------------------------------------------------------------- */
size_t nmsa_pq_confmat_workspace_bytes(int B, int H, int W, int n_classes);
int nmsa_pq_update_with_confmat(
    const int64_t* pred, const int64_t* target, const uint8_t* target_semantic,
    int B, int H, int W,
    int num_categories, int64_t ignored_label, int64_t max_instances_per_category, int64_t offset,
    int64_t void_segment_id,
    double* iou_per_class, double* tp_per_class, double* fn_per_class, double* fp_per_class,
    int64_t* matches, int match_capacity, int32_t* n_matches,
    int32_t* status, void* workspace, size_t workspace_bytes, int workspace_is_clean,
    int confmat_classes, int64_t pred_div, int64_t* confmat, int32_t* confmat_status,
    void* confmat_workspace, size_t confmat_workspace_bytes, nmsa_stream_t stream);
/* The same update with the prediction given as the PARTS nmsa_panoptic_paint paints the map from
 * (the fused pipeline's outputs) instead of the painted int64 map: the predicted panoptic id of
 * a pixel is pan_of_inst[b][pred_instance] where pred_instance != 0, else (pred_semantic + 1) *
 * max_instances_per_category when that class is stuff, else void_label — 2 B/px read instead of
 * 8 B/px; results are bit-identical to nmsa_pq_update_with_confmat on the painted map
 * (task_helper/panoptic.py:104-126 at network resolution; no match list).
 *   pred_semantic u8 [B,H,W] class index 0..n_sem_classes-1 (<= 255), pred_instance u8 [B,H,W],
 *   pan_of_inst i64 [B,256], is_thing_class u8 [n_sem_classes].
 *   confmat_classes = 0 with target_semantic = confmat = confmat_status = NULL: the PQ update
 *   alone (nmsa_pq_update on the painted map, read as its parts) — for class counts beyond the
 *   fused matrix (64) */
int nmsa_pq_update_with_confmat_parts(
    const uint8_t* pred_semantic, const uint8_t* pred_instance, const int64_t* pan_of_inst,
    const uint8_t* is_thing_class, int n_sem_classes, int64_t void_label,
    const int64_t* target, const uint8_t* target_semantic,
    int B, int H, int W,
    int num_categories, int64_t ignored_label, int64_t max_instances_per_category, int64_t offset,
    int64_t void_segment_id,
    double* iou_per_class, double* tp_per_class, double* fn_per_class, double* fp_per_class,
    int32_t* status, void* workspace, size_t workspace_bytes, int workspace_is_clean,
    int confmat_classes, int64_t pred_div, int64_t* confmat, int32_t* confmat_status,
    void* confmat_workspace, size_t confmat_workspace_bytes, nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * a6-a10  per-pixel multi-task losses (forward + backward)
 * Common conventions: predictions f32|bf16|f16 (code in `dtype`), targets f32,
 * labels / masks u8, planar NCHW.  Forward returns device scalars
 * loss_sum (f64) and the element count (i64) — the task helpers divide
 * (task_helper/base.py:161-182); nothing is copied to the host.  Backward
 * recomputes from the inputs and multiplies by the upstream gradient read from
 * the device scalar *grad_scale; gradients are written in the prediction dtype.
 * workspace: nmsa_loss_workspace_bytes(B,H,W) bytes (block partials; the final
 * sum is taken in a fixed order -> run-to-run deterministic).
 *
 * nmsa_loss_ce_*       CrossEntropyLossSemantic._compute_loss  loss/ce.py:40-68
 *     target u8 [B,H,W], 0 = void (ignored); weights f32 [C] or NULL;
 *     weight_sum (f64, may be NULL) = sum over non-void px of w[label]: the divisor
 *     of the ESANet `weighted_reduction` (ce.py:57-68);
 *     lse2 f32 [B,H,W] (optional, NULL = off): the forward pass stores -log2(sum_c exp(x_c))
 *     per pixel and the backward pass, given the same buffer, reads the logits ONCE
 *     instead of twice (what autograd's saved log_softmax buys the reference)
 * nmsa_loss_masked_*   MSELoss / L1Loss (reduction='sum') loss/mse.py:21-41, l1.py:21-41
 *     with the masking of task_helper/instance.py:129-139,154-167:
 *     sum_px mean_c f(pred*mask - target), n = sum(mask); mask u8 [B,H,W] or NULL;
 *     kind 0 = MSE, 1 = L1; C = 1 for [B,H,W] inputs
 *     kind 2 = center focal loss — EXTENSION, not in the reference: CenterNet's penalty-reduced
 *     focal loss (alpha 2, beta 4, pred clamped to [1e-4, 1-1e-4]) over the masked pixels;
 *     n_mask then counts the positive (target == 1) masked pixels
 * nmsa_loss_vonmises_* VonMisesLossBiternion  loss/vonmises.py:27-51 over the px
 *     where mask != 0 (gather of task_helper/instance.py:186-216), pred/target [B,2,H,W]
 * nmsa_loss_*_fwd_grad / nmsa_loss_*_bwd_unless  forward and gradient in ONE pass:
 *     the reference's losses are sums that the caller divides (`loss / n`,
 *     task_helper/base.py:161-182: sum_scales loss / sum_scales n), so the upstream gradient
 *     of a loss sum is known before backward runs once n is (nmsa_count_u8 over the labels /
 *     mask, 1 B/px).  *_fwd_grad takes that EXPECTED scale as a device scalar and writes
 *     expected * d loss_sum / d pred into grad_* while it computes the forward sum: the
 *     prediction is read once and the gradient written once (CE at C = 40, bf16: 162 B/px
 *     instead of 250).  Backward then calls *_bwd_unless with the real upstream gradient:
 *     when *grad_scale is bit-equal to *computed_for the gradient buffer is already right and
 *     the kernel returns at once (counters[0]++), otherwise it recomputes the gradient from
 *     the inputs (counters[1]++), so the result never depends on the expectation.
 *     The CE variant keeps a pixel's whole class column in registers up to C = 48; for 49..256
 *     classes the four waves of a workgroup share the column (wave w holds a quarter of the
 *     classes of the same pixels; maximum and sum of exponentials are exchanged through LDS):
 *     logits read once, gradient written once.  A recomputing backward launch (*_bwd_unless on
 *     a wrong expectation) is the same single pass: it never costs more than nmsa_loss_ce_bwd.
 *     Beyond 256 classes the launch walks the column twice through the caches.
 *     nmsa_loss_ce_fwd_grad_supported(dtype, C): 1 for every valid dtype and C <= 4096.
 * nmsa_count_u8        *count = #{i : lo <= values[i] <= hi}  (labels 1..C, mask bytes 1..255);
 *     *mean_scale (optional) = weight / (float)count, the correctly rounded fp32 division
 *     autograd performs for `weight * loss_sum / count`;
 *     workspace: nmsa_count_workspace_bytes()
 * nmsa_loss_cos_emb_*  CosineEmbeddingLoss  loss/cos_emb.py:21-56 with the LUT gather of
 *     task_helper/dense_visual_embedding.py:110-171: pred [B,D,H,W], indices i32
 *     [B,H,W] (0 = no target), lut f32 [B,L,D];
 *     dots f32 [B,2,H,W] (optional, NULL = off; needs H*W % 8 == 0 and 16-B aligned
 *     pointers, NMSA_ERR_ARG otherwise — see nmsa_loss_cos_emb_can_keep_dots): the forward
 *     pass stores x.y and |x|^2 per pixel and the backward pass, given the same buffer, reads
 *     the prediction ONCE (for the gradient) instead of twice
 * ------------------------------------------------------------------------- */
size_t nmsa_loss_workspace_bytes(int B, int H, int W);
int nmsa_loss_ce_fwd(const void* logits, int dtype, const uint8_t* target, const float* weights,
                     int B, int C, int H, int W, float label_smoothing,
                     double* loss_sum, int64_t* n_elements, double* weight_sum, float* lse2_out,
                     int32_t* status, void* workspace, size_t workspace_bytes, nmsa_stream_t stream);
int nmsa_loss_ce_bwd(const void* logits, int dtype, const uint8_t* target, const float* weights,
                     int B, int C, int H, int W, float label_smoothing,
                     const float* grad_scale, const float* lse2, void* grad_logits,
                     nmsa_stream_t stream);
/* the same two kernels for more than 255 classes (reference ce.py:40-68 takes any number): labels
 * as int16 (0 = void, 1..C), C <= 4096 */
int nmsa_loss_ce_fwd_i16(const void* logits, int dtype, const int16_t* target,
                         const float* weights, int B, int C, int H, int W,
                         float label_smoothing,
                         double* loss_sum, int64_t* n_elements, double* weight_sum,
                         float* lse2_out,
                         int32_t* status, void* workspace, size_t workspace_bytes,
                         nmsa_stream_t stream);
int nmsa_loss_ce_bwd_i16(const void* logits, int dtype, const int16_t* target,
                         const float* weights, int B, int C, int H, int W,
                         float label_smoothing, const float* grad_scale, const float* lse2,
                         void* grad_logits, nmsa_stream_t stream);
int nmsa_loss_ce_fwd_grad_supported(int dtype, int C);
int nmsa_loss_ce_fwd_grad(const void* logits, int dtype, const uint8_t* target, const float* weights,
                          int B, int C, int H, int W, float label_smoothing,
                          const float* expected_grad_scale,
                          double* loss_sum, int64_t* n_elements, double* weight_sum,
                          void* grad_logits, int32_t* status,
                          void* workspace, size_t workspace_bytes, nmsa_stream_t stream);
int nmsa_loss_ce_bwd_unless(const void* logits, int dtype, const uint8_t* target,
                            const float* weights, int B, int C, int H, int W,
                            float label_smoothing, const float* grad_scale, void* grad_logits,
                            const float* computed_for, int32_t* counters, nmsa_stream_t stream);
size_t nmsa_count_workspace_bytes(void);
int nmsa_count_u8(const uint8_t* values, int64_t n, int lo, int hi, int64_t* count,
                  float* mean_scale, float weight,
                  void* workspace, size_t workspace_bytes, nmsa_stream_t stream);
int nmsa_loss_masked_fwd(const void* pred, int dtype, const float* target, const uint8_t* mask,
                         int B, int C, int H, int W, int kind,
                         double* loss_sum, int64_t* n_mask,
                         void* workspace, size_t workspace_bytes, nmsa_stream_t stream);
int nmsa_loss_masked_bwd(const void* pred, int dtype, const float* target, const uint8_t* mask,
                         int B, int C, int H, int W, int kind,
                         const float* grad_scale, void* grad_pred, nmsa_stream_t stream);
int nmsa_loss_masked_fwd_grad(const void* pred, int dtype, const float* target,
                              const uint8_t* mask, int B, int C, int H, int W, int kind,
                              const float* expected_grad_scale,
                              double* loss_sum, int64_t* n_mask, void* grad_pred,
                              void* workspace, size_t workspace_bytes, nmsa_stream_t stream);
int nmsa_loss_masked_bwd_unless(const void* pred, int dtype, const float* target,
                                const uint8_t* mask, int B, int C, int H, int W, int kind,
                                const float* grad_scale, void* grad_pred,
                                const float* computed_for, int32_t* counters, nmsa_stream_t stream);
int nmsa_loss_vonmises_fwd(const void* pred, int dtype, const float* target, const uint8_t* mask,
                           int B, int H, int W, float kappa,
                           double* loss_sum, int64_t* n_rows,
                           void* workspace, size_t workspace_bytes, nmsa_stream_t stream);
int nmsa_loss_vonmises_bwd(const void* pred, int dtype, const float* target, const uint8_t* mask,
                           int B, int H, int W, float kappa,
                           const float* grad_scale, void* grad_pred, nmsa_stream_t stream);
int nmsa_loss_vonmises_fwd_grad(const void* pred, int dtype, const float* target,
                                const uint8_t* mask, int B, int H, int W, float kappa,
                                const float* expected_grad_scale,
                                double* loss_sum, int64_t* n_rows, void* grad_pred,
                                void* workspace, size_t workspace_bytes, nmsa_stream_t stream);
int nmsa_loss_vonmises_bwd_unless(const void* pred, int dtype, const float* target,
                                  const uint8_t* mask, int B, int H, int W, float kappa,
                                  const float* grad_scale, void* grad_pred,
                                  const float* computed_for, int32_t* counters,
                                  nmsa_stream_t stream);
int nmsa_loss_cos_emb_fwd(const void* pred, int dtype, const int32_t* indices, const float* lut,
                          int B, int D, int H, int W, int L,
                          double* loss_sum, int64_t* n_rows, float* dots_out, int32_t* status,
                          void* workspace, size_t workspace_bytes, nmsa_stream_t stream);
int nmsa_loss_cos_emb_bwd(const void* pred, int dtype, const int32_t* indices, const float* lut,
                          int B, int D, int H, int W, int L,
                          const float* grad_scale, const float* dots, void* grad_pred,
                          nmsa_stream_t stream);
/* forward + gradient in ONE pass over the prediction: the D-column of a pixel stays in the
 * registers of ceil(D / 64) waves between the reduction and the gradient.  H*W a multiple of 4
 * (2 for f32), 8-byte aligned planes, and either D % 64 == 0, D <= 512 with the image's fp32 LUT +
 * the exchange buffers within the CU's LDS (k_cos_split: one workgroup per column) or D <= 1024,
 * any D (k_cos_parts: the column over ceil(D / 256) cooperating workgroups that exchange their
 * partial sums as 8-byte {value, tag} granules through the workspace; DVEFormer's D = 768; a
 * ragged last wave when D % 64 != 0).  The gradient
 * is written for *expected_gscale; nmsa_loss_cos_emb_bwd_unless confirms it (bit-equal
 * *grad_scale) or recomputes — it takes the forward call's workspace (NULL is fine for
 * D <= 512).  Status bit 32: a cooperating workgroup did not answer within ~2 s (sum and
 * gradient are NaN then). */
int nmsa_loss_cos_emb_fwd_grad_supported(int dtype, int D, int H, int W, int L);
size_t nmsa_loss_cos_emb_fwd_grad_workspace_bytes(int B, int D, int H, int W, int L);
int nmsa_loss_cos_emb_fwd_grad(const void* pred, int dtype, const int32_t* indices, const float* lut,
                               int B, int D, int H, int W, int L, const float* expected_gscale,
                               double* loss_sum, int64_t* n_rows, void* grad_pred, int32_t* status,
                               void* workspace, size_t workspace_bytes, nmsa_stream_t stream);
int nmsa_loss_cos_emb_bwd_unless(const void* pred, int dtype, const int32_t* indices, const float* lut,
                                 int B, int D, int H, int W, int L, const float* grad_scale,
                                 void* grad_pred, const float* computed_for, int32_t* counters,
                                 void* workspace, size_t workspace_bytes, nmsa_stream_t stream);
/* 1 when the (L, D, H*W) combination runs the LDS-LUT kernel that can keep `dots` */
int nmsa_loss_cos_emb_can_keep_dots(int D, int H, int W, int L);

/* ---------------------------------------------------------------------------
 * a7 / a9  the forms of the same loss classes that no task helper calls (csrc/losses_forms.hip)
 * nmsa_loss_elementwise_none_*  MSELoss / L1Loss with reduction='none' (loss/mse.py:21-41 else
 *     branch, loss/l1.py:21-41): out[i] = (pred[i] - target[i])^2 or |pred[i] - target[i]| over n
 *     contiguous elements, op-math in fp32, rounded once to out_dtype (NMSA_F32 or `dtype` — the
 *     type promotion of pred and target); kind 0 = MSE, 1 = L1.  _bwd: grad_pred[i] =
 *     upstream[i] * d out[i] / d pred[i], upstream in upstream_dtype (NMSA_F32 or `dtype`).
 *     The reference's 2-D [N, C] rows with 'sum' / 'mean' need no entry point of their own:
 *     sum_n mean_c f = (1 / C) * nmsa_loss_masked_fwd over the n * C elements as one plane.
 * nmsa_loss_cos_rows_*  CosineEmbeddingLoss on rows with labels (loss/cos_emb.py:21-56 with
 *     `target_similarity`; torch.nn.CosineEmbeddingLoss, margin 0): input [n_rows, D] in `dtype`,
 *     target f32 [n_rows, D], labels f32 [n_rows] (+1 similar: 1 - cos; -1 dissimilar:
 *     max(0, cos - margin); anything else: 0) or NULL (all +1); loss_rows f32 [n_rows] — the
 *     'none' reduction; 'sum' / 'mean' are the sum of that vector.  _bwd: grad_input =
 *     upstream[row] * d loss_rows[row] / d input (upstream f32 [n_rows], or ONE f32 scalar when
 *     upstream_is_scalar != 0). */
int nmsa_loss_elementwise_none_fwd(const void* pred, int dtype, const float* target, int64_t n,
                                   int kind, int out_dtype, void* out, nmsa_stream_t stream);
int nmsa_loss_elementwise_none_bwd(const void* pred, int dtype, const float* target, int64_t n,
                                   int kind, int upstream_dtype, const void* upstream,
                                   void* grad_pred, nmsa_stream_t stream);
int nmsa_loss_cos_rows_fwd(const void* input, int dtype, const float* target, const float* labels,
                           int64_t n_rows, int D, float margin, float* loss_rows,
                           nmsa_stream_t stream);
int nmsa_loss_cos_rows_bwd(const void* input, int dtype, const float* target, const float* labels,
                           int64_t n_rows, int D, float margin, const float* upstream,
                           int upstream_is_scalar, void* grad_input, nmsa_stream_t stream);
/* VonMisesLossBiternion with reduction='none' (loss/vonmises.py:27-51): input [n_rows, 2] in
 * `dtype`, target f32 [n_rows, 2] -> loss_rows f32 [n_rows] = 1 - exp(kappa (x.y - 1));
 * _bwd: grad_input = upstream[row] * d loss_rows[row] / d input, upstream f32 [n_rows] */
int nmsa_loss_vonmises_rows_fwd(const void* input, int dtype, const float* target, int64_t n_rows,
                                float kappa, float* loss_rows, nmsa_stream_t stream);
int nmsa_loss_vonmises_rows_bwd(const void* input, int dtype, const float* target, int64_t n_rows,
                                float kappa, const float* upstream, void* grad_input,
                                nmsa_stream_t stream);

/* ---------------------------------------------------------------------------
 * a10  the losses of a task helper in one call
 *      InstanceTaskHelper._compute_losses  task_helper/instance.py:92-269
 *      SemanticTaskHelper._compute_losses  task_helper/semantic.py:57-90
 *      TaskHelperBase.accumulate_losses    task_helper/base.py:161-182
 * Every (loss, supervision scale) pair is an ITEM; the items whose sums the caller adds and
 * divides by their summed element counts (accumulate_losses) form a TOTAL.  One call
 *   counts the labels / mask bytes of all items (launch 1, 1 B/px); the launch's last workgroup
 *     forms per total the divisor n and the EXPECTED upstream gradient w / n of its loss sums
 *     (w lives in the total's spec record and is learned from the backward passes: loss
 *     weights, AMP scale, ... need no hint from the caller)
 *   computes all forward sums and writes all gradients for that expectation (launch 2: block
 *     ranges per item; cross entropies above 48 classes run k_ce_split as launches of their own)
 *   reduces the block partials per item in a fixed order and forms per item loss_sum / count
 *     and per total sum(loss sums) / divisor (launch 3; the reductions of accumulate_losses, in
 *     item order, float32)
 * A call without any gradient buffer (validation) skips the count and takes the divisors from
 * the finalized counts (2 launches).
 * nmsa_multitask_loss_bwd_unless is ONE launch (+ one per wide cross entropy / cosine item): a
 * small grid that walks the block list; every workgroup turns the upstream gradients of the
 * three outputs into one upstream scale per item and compares it with the expectation,
 * workgroup 0 also updates w; only the items that differ are recomputed and the grid is gone at
 * once when every gradient stands.  A NaN expectation ("the forward
 * pass wrote no gradient") is never confirmed — not even by an upstream gradient that is the same
 * NaN: gradient-only launches always write, so a NaN upstream comes out as NaN gradients.
 * `spec` (and `counters`) may be NULL there: a recompute nobody predicted (a second backward
 * through a retained graph) leaves the caller's record and the tally alone.
 * `workspace`: the forward call's buffer, contents dead by then (used as the granule exchange of
 * cosine items with D > 512; may be NULL without such items).
 *   items        HOST array; pointers inside are device pointers.  kind NMSA_LOSS_*; CE: pred =
 *                logits [B,C,H,W], mask = labels u8 [B,H,W] (0 = void), weights f32 [C] or NULL,
 *                param = label smoothing; MSE / L1 / FOCAL: pred [B,C,H,W] (C = 1 for [B,H,W]),
 *                target f32, mask u8 [B,H,W] or NULL; VONMISES: pred / target [B,2,H,W], mask,
 *                param = kappa; COS_EMB (loss/cos_emb.py:21-56 with the LUT gather of
 *                task_helper/dense_visual_embedding.py:110-171): pred [B,D,H,W] with C = D, target =
 *                LUT f32 [B,L,D] with L in `reserved`, mask = indices i32 [B,H,W] (0 = no target) —
 *                shapes nmsa_loss_cos_emb_fwd_grad_supported accepts, else NMSA_ERR_UNSUPPORTED.
 *                grad = gradient buffer shaped like pred, or NULL (forward only).
 *                clamp_count: the item's count enters its loss and its total as max(count, 1)
 *                (task_helper/instance.py:206-211)
 *   spec         i32 [n_totals + 1][8] device, persistent, owned by the caller (zero it, then store
 *                the fp32 bits of the initial upstream weight, usually 1.0f, at [t][2], t < n_totals):
 *                [0] backward passes that found their gradient written [1] ... that recomputed;
 *                the LAST row is scratch of the calls (the tickets their launches draw to find
 *                their last workgroup): zero it once, every call leaves it zero again.  One
 *                record set serves one stream at a time.
 *   expect       f32 [n_totals][2] device, out: expected upstream gradient (NaN: none) and the
 *                divisor as float; must be handed to nmsa_multitask_loss_bwd_unless unchanged
 *   loss_sums    f64 [n_items], counts i64 [n_items], aux f64 [n_items] or NULL (CE: sum of the
 *                label weights, ce.py:57-68) — device
 *   out_f32      f32 [2 n_items + n_totals] device or NULL: the loss sums as float32, then
 *                item_loss[i] = float32(sum_i) / float32(count_i), then total_loss[t] =
 *                (sum of the float32 sums of its items, in item order) / divisor_t
 *   grad_sums / grad_item_losses / grad_total_losses   f32 [n_items] / [n_items] / [n_totals]
 *                device or NULL: upstream gradients of the three parts of out_f32; item i's
 *                upstream scale is grad_sums[i] + grad_item_losses[i] / count_i +
 *                grad_total_losses[total_i] / divisor_t
 *   grad_scales  f32 [n_items] device, scratch: receives those upstream scales
 *   counters     i32 [2] device or NULL: per total and backward pass, [0] += 1 when the expectation
 *                held, [1] += 1 when it did not (a tally over all callers; the spec records keep
 *                their own)
 * ------------------------------------------------------------------------- */
enum { NMSA_LOSS_CE = 0, NMSA_LOSS_MSE = 1, NMSA_LOSS_L1 = 2, NMSA_LOSS_FOCAL = 3, NMSA_LOSS_VONMISES = 4,
       NMSA_LOSS_COS_EMB = 5 };
#define NMSA_MULTI_MAX_ITEMS 16
#define NMSA_MULTI_MAX_TOTALS 8
typedef struct nmsa_loss_item {
    int32_t kind, dtype, B, C, H, W;
    int32_t total;        /* index of the total this item belongs to */
    int32_t clamp_count;  /* 1: max(count, 1) is the item's divisor and enters the total; 2: the item's divisor only */
    float param;          /* label smoothing | kappa */
    int32_t reserved;     /* COS_EMB: L, the LUT rows per image; otherwise 0 */
    const void* pred;
    const void* target;
    const void* mask;     /* labels (CE) or mask bytes; NULL = every pixel */
    const float* weights;
    void* grad;
} nmsa_loss_item;
size_t nmsa_multitask_loss_workspace_bytes(const nmsa_loss_item* items_host, int n_items);
int nmsa_multitask_loss_fwd_grad(const nmsa_loss_item* items_host, int n_items, int n_totals,
                                 int32_t* spec, float* expect, double* loss_sums, int64_t* counts,
                                 double* aux, float* out_f32, int32_t* status, void* workspace,
                                 size_t workspace_bytes, nmsa_stream_t stream);
int nmsa_multitask_loss_bwd_unless(const nmsa_loss_item* items_host, int n_items, int n_totals,
                                   const float* grad_sums, const float* grad_item_losses,
                                   const float* grad_total_losses, const int64_t* counts,
                                   const float* expect, int32_t* spec, float* grad_scales,
                                   int32_t* counters, void* workspace, size_t workspace_bytes,
                                   nmsa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NMSA_H */
