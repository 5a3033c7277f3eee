/*
 * nmsa_oracle.c — TEST INFRASTRUCTURE ONLY (the checker, never the product).
 *
 * Plain-C, single-threaded CPU restatement of the reference hot path of
 * TUI-NICR/nicr-multitask-scene-analysis (v0.3.0).  Every function cites the
 * reference file:line (relative to /root/reference/src/nicr_mt_scene_analysis)
 * whose algorithm it restates.  It is pinned against golden vectors produced
 * by running the reference's own Python in the build container
 * (oracle/gen_golden.py -> tests/golden/ *.npz) and against the known-answer
 * tables of the reference's tests (tests/test_metrics.py:76-446).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (the timed CPU
 * baseline and, behind the timed regions, its checks of GPU results) may load this
 * library.  The product path (nicr_mt_scene_analysis_amd) never does.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: every float op below is
 * a single IEEE-754 rounding unless written as fmaf()).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_ARG (-1)
#define ORC_ERR_RANGE (-2)
#define ORC_ERR_CAPACITY (-3)

int orc_version(void) { return 1; }

/* ------------------------------------------------------------------------- */
/* a1  SemanticPostprocessing._postprocess_inference                          */
/*     model/postprocessing/semantic.py:52-53  softmax(dim=1) -> max(dim=1)    */
/* idx  : first index attaining the maximum PROBABILITY (torch.max tie rule). */
/*        softmax is monotone, so that is the first maximum of the logits —    */
/*        except where a LOWER class < 1.5 * 2^-24 below the maximum gets      */
/*        the maximum's fp32 probability.  That is decided with the arithmetic */
/*        of ATen's CPU softmax over a non-last dim (vec_softmax: per pixel,   */
/*        sequentially over the classes): e_c = Sleef expf_u10(x_c - max) in   */
/*        its FMA form, S = ((e_0 + e_1) + ...) in fp32, p_c = e_c / S.  The   */
/*        reference-run fixture tests/golden/argmax_ties.npz pins it (all 1680 */
/*        near-tie columns incl. the gaps in (2^-25, 2^-23], two whole maps).  */
/*        Non-finite maximum (NaN anywhere, +inf, or all -inf) makes every     */
/*        softmax output NaN in the reference -> torch.max returns index 0.    */
/* score: 1 / sum_c exp(x_c - max)  (fp64 accumulate, rounded once).           */
/* ------------------------------------------------------------------------- */
static float aten_vec_expf(float d)          /* Sleef_expf16_u10 as bundled with torch (FMA build) */
{
    if (!(d >= -104.0f)) return 0.0f;
    const volatile float dq = d * 1.442695040888963407359924681001892137426645954152985934135449406931f;
    const float q = rintf(dq);
    float s = fmaf(q, -0.693145751953125f, d);
    s = fmaf(q, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = fmaf(u, s, 0.00139304355252534151077271f);
    u = fmaf(u, s, 0.00833336077630519866943359f);
    u = fmaf(u, s, 0.0416664853692054748535156f);
    u = fmaf(u, s, 0.166666671633720397949219f);
    u = fmaf(u, s, 0.5f);
    const volatile float s2 = s * s;
    const volatile float t = fmaf(s2, u, s);
    const volatile float r = t + 1.0f;
    const int qi = (int)q, qh = qi >> 1;
    const volatile float r1 = r * ldexpf(1.0f, qh);
    return r1 * ldexpf(1.0f, qi - qh);
}

int orc_semantic_argmax(const float* logits, int B, int C, int H, int W,
                        int64_t* idx, float* score)
{
    const int64_t P = (int64_t)H * W;
    for (int b = 0; b < B; ++b) {
        const float* lb = logits + (int64_t)b * C * P;
        for (int64_t p = 0; p < P; ++p) {
            float m = lb[p];
            int am = 0;
            int bad = isnan(m);
            for (int c = 1; c < C; ++c) {
                float v = lb[(int64_t)c * P + p];
                if (isnan(v)) bad = 1;
                if (v > m) { m = v; am = c; }
            }
            if (bad || isinf(m)) {
                if (idx) idx[b * P + p] = 0;
                if (score) score[b * P + p] = NAN;
                continue;
            }
            double s = 0.0;
            for (int c = 0; c < C; ++c)
                s += exp((double)lb[(int64_t)c * P + p] - (double)m);
            int cand = 0;
            for (int c = 0; c < am && !cand; ++c) {
                const volatile float d = lb[(int64_t)c * P + p] - m;      /* fp32, as ATen */
                if (d >= -0x1p-23f) cand = 1;
            }
            if (cand) {                       /* a lower class may share the maximum's probability */
                volatile float S = 0.0f;
                for (int c = 0; c < C; ++c) {
                    const volatile float d = lb[(int64_t)c * P + p] - m;
                    S = S + aten_vec_expf(d);
                }
                const volatile float pm = 1.0f / S;
                for (int c = 0; c < am; ++c) {
                    const volatile float d = lb[(int64_t)c * P + p] - m;
                    const volatile float pc = aten_vec_expf(d) / S;
                    if (pc == pm) { am = c; break; }
                }
            }
            if (idx) idx[b * P + p] = am;
            if (score) score[b * P + p] = (float)(1.0 / s);
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* a2  InstancePostprocessing._get_instance_centers                           */
/*     model/postprocessing/instance.py:79-168                                */
/* ------------------------------------------------------------------------- */
static int cmp_float_desc(const void* a, const void* b)
{
    float fa = *(const float*)a, fb = *(const float*)b;
    return (fa < fb) - (fa > fb);
}

/* k-th largest (1-based) of v[0..n) — what torch.topk(...)[..., -1] returns
 * (instance.py:133-134,147).  Values are NaN-free here (instance.py:129 turns
 * every NaN into -1).  Counting sort over the distinct "interesting" values:
 * everything equal to -1 is counted, the rest is sorted. */
static float kth_largest(const float* v, int64_t n, int k, float* scratch)
{
    int64_t m = 0, n_le = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (v[i] > -1.0f) scratch[m++] = v[i];
        else if (v[i] == -1.0f) n_le++;
    }
    if (m >= k) {
        qsort(scratch, (size_t)m, sizeof(float), cmp_float_desc);
        return scratch[k - 1];
    }
    if (m + n_le >= k) return -1.0f;
    /* values below -1 exist (threshold < -1): full sort, literal fallback */
    memcpy(scratch, v, (size_t)n * sizeof(float));
    qsort(scratch, (size_t)n, sizeof(float), cmp_float_desc);
    return scratch[k - 1];
}

int orc_center_nms_topk(const float* center, const uint8_t* fg,
                        int B, int H, int W,
                        float threshold, int ksize, int topk, int apply_fg,
                        int max_centers,
                        int32_t* centers_yx, int32_t* n_centers, float* scores,
                        uint8_t* center_mask)
{
    if (ksize < 1 || (ksize % 2) != 1 || topk < 1) return ORC_ERR_ARG;
    if (apply_fg && !fg) return ORC_ERR_ARG;
    const int64_t P = (int64_t)H * W;
    if (P < topk) return ORC_ERR_ARG;                 /* torch.topk would raise */
    const int pad = (ksize - 1) / 2;
    float* heat = (float*)malloc((size_t)P * sizeof(float));
    float* scratch = (float*)malloc((size_t)P * sizeof(float));
    if (!heat || !scratch) { free(heat); free(scratch); return ORC_ERR_ARG; }
    int rc = ORC_OK;

    for (int b = 0; b < B; ++b) {
        const float* src = center + (int64_t)b * P;
        /* F.threshold(x, thr, -1): y = x if x > thr else -1   (instance.py:86-88);
         * NaN > thr is false in ATen's `x <= thr ? value : x` form -> NaN is kept */
        for (int64_t i = 0; i < P; ++i) {
            float x = src[i];
            heat[i] = (x <= threshold) ? -1.0f : x;
        }
        /* max_pool2d(k, stride 1, return_indices) + zero pad + index test +
         * equality test (instance.py:97-129) evaluated per pixel on the
         * thresholded map (the tests only read `heat`, writes go to scratch) */
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                const int64_t self = (int64_t)y * W + x;
                float pooled; int64_t pidx;
                if (y < pad || y >= H - pad || x < pad || x >= W - pad) {
                    pooled = 0.0f; pidx = 0;          /* F.pad zeros (104-109) */
                } else {
                    /* ATen max_pool2d: scan window row-major, take when
                     * (val > max) || isnan(val) -> first maximum, last NaN */
                    pooled = -INFINITY; pidx = (int64_t)(y - pad) * W + (x - pad);
                    for (int dy = -pad; dy <= pad; ++dy)
                        for (int dx = -pad; dx <= pad; ++dx) {
                            const int64_t q = (int64_t)(y + dy) * W + (x + dx);
                            const float v = heat[q];
                            if (v > pooled || isnan(v)) { pooled = v; pidx = q; }
                        }
                }
                float h = heat[self];
                if (pidx != self) h = -1.0f;          /* instance.py:125-127 */
                if (h != pooled) h = -1.0f;           /* instance.py:129     */
                scratch[self] = h;
            }
        }
        memcpy(heat, scratch, (size_t)P * sizeof(float));

        /* topk before the optional foreground mask (instance.py:133 vs 142) */
        float kth = kth_largest(heat, P, topk, scratch);
        if (kth < 0.0f) kth = 0.0f;                   /* clamp_(min=0) :149   */

        if (apply_fg) {
            const uint8_t* fgb = fg + (int64_t)b * P;
            for (int64_t i = 0; i < P; ++i) if (!fgb[i]) heat[i] = -1.0f;
        }
        /* (heat >= kth).nonzero().int() -> raster order (instance.py:152-166) */
        int32_t n = 0;
        for (int64_t i = 0; i < P; ++i) {
            const int keep = heat[i] >= kth;
            if (center_mask) center_mask[(int64_t)b * P + i] = (uint8_t)keep;
            if (keep) {
                if (n < max_centers) {
                    centers_yx[((int64_t)b * max_centers + n) * 2 + 0] = (int32_t)(i / W);
                    centers_yx[((int64_t)b * max_centers + n) * 2 + 1] = (int32_t)(i % W);
                    /* meta 'score' = raw heatmap at the center (instance.py:265) */
                    if (scores) scores[(int64_t)b * max_centers + n] = src[i];
                }
                n++;
            }
        }
        n_centers[b] = n;
        if (n > max_centers) rc = ORC_ERR_CAPACITY;
    }
    free(heat); free(scratch);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* a3  InstancePostprocessing._get_instance_segmentation                      */
/*     model/postprocessing/instance.py:171-268                               */
/* offsets are de-normalised by the caller in the reference (panoptic.py:105-  */
/* 111 / instance.py:361-365: off[:,0]*h, off[:,1]*w, one fp32 rounding); here  */
/* scale_y/scale_x carry that multiply (1.0f when normalized_offset=False —     */
/* an exact no-op).                                                            */
/* distance: torch.norm(int32 centers - fp32 loc, dim=-1) == ATen's            */
/* norm-2 last-dim reduction  acc = 0 + dy*dy; acc = fma(dx,dx,acc); sqrt(acc)  */
/* (verified bit-for-bit against torch 2.10 CPU on 12.8 M pairs, see           */
/* oracle/gen_golden.py::check_norm_formula).                                  */
/* ------------------------------------------------------------------------- */
int orc_group_offsets(const float* offset, const uint8_t* fg,
                      const int32_t* centers_yx, const int32_t* n_centers,
                      int B, int H, int W, int max_centers,
                      float scale_y, float scale_x,
                      int use_thr, float dist_thr,
                      uint8_t* inst, int32_t* area /* [B,256] */)
{
    const int64_t P = (int64_t)H * W;
    memset(inst, 0, (size_t)B * P);
    if (area) memset(area, 0, (size_t)B * 256 * sizeof(int32_t));
    for (int b = 0; b < B; ++b) {
        const int n = n_centers[b];
        if (n == 0) continue;                                  /* :214-215 */
        if (n > max_centers) return ORC_ERR_CAPACITY;
        const int32_t* cyx = centers_yx + (int64_t)b * max_centers * 2;
        const float* offy = offset + (int64_t)b * 2 * P;
        const float* offx = offy + P;
        const uint8_t* fgb = fg + (int64_t)b * P;
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                const int64_t p = (int64_t)y * W + x;
                if (!fgb[p]) continue;
                const float oy = offy[p] * scale_y;            /* panoptic.py:108 */
                const float ox = offx[p] * scale_x;            /* panoptic.py:109 */
                const float ly = (float)y + oy;                /* instance.py:194 */
                const float lx = (float)x + ox;
                /* torch.min(distance, dim=0) (instance.py:235): ATen scans with
                 * `if (!(value >= min))` and stops at the first NaN */
                float best = 0.0f; int besti = 0;
                for (int i = 0; i < n; ++i) {
                    const float dy = (float)cyx[2 * i + 0] - ly;   /* :231 */
                    const float dx = (float)cyx[2 * i + 1] - lx;
                    const float d = sqrtf(fmaf(dx, dx, dy * dy));
                    if (i == 0) { best = d; besti = 0; if (isnan(d)) break; continue; }
                    if (!(d >= best)) {
                        best = d; besti = i;
                        if (isnan(d)) break;
                    }
                }
                uint8_t id = (uint8_t)((besti + 1) & 0xFF);    /* :236 uint8 wrap */
                if (use_thr && best > dist_thr) id = 0;        /* :246-247 */
                inst[(int64_t)b * P + p] = id;                 /* :250 */
                if (area) area[b * 256 + id]++;                /* :253 */
            }
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* a5  deeplab_merge_semantic_and_instance  utils/panoptic_merge.py:172-225   */
/*     (and its numpy twin :110-169 — identical arithmetic)                   */
/* sem: class per pixel (0 = void), ins: instance id per pixel (>= 0),         */
/* thing_seg: bool.  Output pan (int64) and the id dict as two parallel arrays */
/* in insertion order (= ascending instance id).                              */
/* ------------------------------------------------------------------------- */
static int cmp_i64(const void* a, const void* b)
{
    int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
    return (x > y) - (x < y);
}

static int in_list(int64_t v, const int64_t* list, int n)
{
    for (int i = 0; i < n; ++i) if (list[i] == v) return 1;
    return 0;
}

int orc_deeplab_merge(const int64_t* sem, const int64_t* ins, const uint8_t* thing_seg,
                      int B, int H, int W,
                      int64_t max_instances_per_category,
                      const int64_t* thing_ids, int n_thing_ids,
                      int64_t void_label,
                      int64_t* pan,
                      int id_capacity,
                      int64_t* id_pan /* [B,cap] */, int64_t* id_ins /* [B,cap] */,
                      int32_t* n_ids /* [B] */)
{
    const int64_t P = (int64_t)H * W;
    int64_t* sorted = (int64_t*)malloc((size_t)P * sizeof(int64_t));
    int64_t* votes = NULL; int64_t n_class_alloc = 0;
    int rc = ORC_OK;
    for (int b = 0; b < B; ++b) {
        const int64_t* s = sem + b * P;
        const int64_t* in = ins + b * P;
        const uint8_t* th = thing_seg + b * P;
        int64_t* pn = pan + b * P;
        int64_t max_class = 0;
        for (int64_t p = 0; p < P; ++p) {
            pn[p] = void_label;                                 /* :181 */
            if (s[p] < 0) { rc = ORC_ERR_RANGE; goto done; }
            if (s[p] > max_class) max_class = s[p];
        }
        if (max_class + 1 > n_class_alloc) {
            free(votes);
            n_class_alloc = max_class + 1;
            votes = (int64_t*)malloc((size_t)n_class_alloc * sizeof(int64_t));
        }
        int64_t* counter = (int64_t*)calloc((size_t)n_class_alloc, sizeof(int64_t));
        /* torch.unique(ins_seg) -> ascending (:192) */
        memcpy(sorted, in, (size_t)P * sizeof(int64_t));
        qsort(sorted, (size_t)P, sizeof(int64_t), cmp_i64);
        int32_t nid = 0;
        for (int64_t u = 0; u < P; ++u) {
            if (u > 0 && sorted[u] == sorted[u - 1]) continue;
            const int64_t ins_id = sorted[u];
            if (ins_id == 0) continue;                          /* :195-196 */
            /* thing_mask = (ins == id) & (ins > 0) & thing_seg   (:182,:198) */
            memset(votes, 0, (size_t)n_class_alloc * sizeof(int64_t));
            int64_t cnt = 0;
            for (int64_t p = 0; p < P; ++p)
                if (in[p] == ins_id && in[p] > 0 && th[p]) { votes[s[p]]++; cnt++; }
            if (cnt == 0) continue;                             /* :199-200 */
            /* torch.mode: most frequent, smallest value on ties (:201) */
            int64_t cls = 0, bestc = -1;
            for (int64_t c = 0; c < n_class_alloc; ++c)
                if (votes[c] > bestc) { bestc = votes[c]; cls = c; }
            if (cls == 0) continue;                             /* :203-204 */
            const int64_t k = ++counter[cls];                   /* :206-207 */
            const int64_t pid = cls * max_instances_per_category + k;   /* :208 */
            if (nid < id_capacity) {
                id_pan[(int64_t)b * id_capacity + nid] = pid;   /* :209 */
                id_ins[(int64_t)b * id_capacity + nid] = ins_id;
            } else rc = ORC_ERR_CAPACITY;
            nid++;
            for (int64_t p = 0; p < P; ++p)
                if (in[p] == ins_id && in[p] > 0 && th[p]) pn[p] = pid;   /* :210 */
        }
        n_ids[b] = nid;
        /* stuff paste (:213-223) */
        for (int64_t p = 0; p < P; ++p) {
            const int64_t c = s[p];
            if (c == 0) continue;
            if (in_list(c, thing_ids, n_thing_ids)) continue;
            if (in[p] == 0) pn[p] = c * max_instances_per_category;
        }
        free(counter);
    }
done:
    free(sorted); free(votes);
    return rc;
}

/* a5'  naive_merge_semantic_and_instance_np   utils/panoptic_merge.py:43-107  */
int orc_naive_merge(const int64_t* sem, const int64_t* ins,
                    int B, int H, int W,
                    int64_t max_instances_per_category,
                    const int64_t* thing_ids, int n_thing_ids,
                    int64_t void_label,
                    int64_t* pan,
                    int id_capacity, int64_t* id_pan, int64_t* id_ins, int32_t* n_ids)
{
    const int64_t P = (int64_t)H * W;
    int64_t* sorted = (int64_t*)malloc((size_t)P * sizeof(int64_t));
    int rc = ORC_OK;
    for (int b = 0; b < B; ++b) {
        const int64_t* s = sem + b * P;
        const int64_t* in = ins + b * P;
        int64_t* pn = pan + b * P;
        int64_t max_class = 0;
        for (int64_t p = 0; p < P; ++p) {
            pn[p] = void_label;                                   /* :56 */
            if (s[p] < 0) { free(sorted); return ORC_ERR_RANGE; }
            if (s[p] > max_class) max_class = s[p];
        }
        int64_t* counter = (int64_t*)calloc((size_t)max_class + 1, sizeof(int64_t));
        uint8_t* present = (uint8_t*)malloc((size_t)max_class + 1);
        memcpy(sorted, in, (size_t)P * sizeof(int64_t));
        qsort(sorted, (size_t)P, sizeof(int64_t), cmp_i64);
        int32_t nid = 0;
        for (int64_t u = 0; u < P; ++u) {
            if (u > 0 && sorted[u] == sorted[u - 1]) continue;
            const int64_t ins_id = sorted[u];
            if (ins_id == 0) continue;                            /* :69-70 */
            memset(present, 0, (size_t)max_class + 1);
            for (int64_t p = 0; p < P; ++p) if (in[p] == ins_id) present[s[p]] = 1;
            /* every semantic label inside the instance, ascending (:76-92) */
            for (int64_t c = 1; c <= max_class; ++c) {
                if (!present[c]) continue;
                const int64_t k = ++counter[c];
                const int64_t pid = c * max_instances_per_category + k;
                if (nid < id_capacity) {
                    id_pan[(int64_t)b * id_capacity + nid] = pid;
                    id_ins[(int64_t)b * id_capacity + nid] = ins_id;
                } else rc = ORC_ERR_CAPACITY;
                nid++;
                for (int64_t p = 0; p < P; ++p)
                    if (in[p] == ins_id && s[p] == c) pn[p] = pid;
            }
        }
        n_ids[b] = nid;
        for (int64_t p = 0; p < P; ++p) {                          /* :95-105 */
            const int64_t c = s[p];
            if (c == 0) continue;
            if (in_list(c, thing_ids, n_thing_ids)) continue;
            if (in[p] == 0) pn[p] = c * max_instances_per_category;
        }
        free(counter); free(present);
    }
    free(sorted);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* next-1  InstancePostprocessing._get_instance_orientation                   */
/*     model/postprocessing/instance.py:271-319: per instance id (>0) inside   */
/*     the mask, sum the two biternion channels, angle = atan2(sum_1, sum_0).  */
/* Output: angle[b,id] (NaN where the id is absent), present[b,id].            */
/* ------------------------------------------------------------------------- */
int orc_instance_orientation(const float* orientation /* [B,2,H,W] */,
                             const uint8_t* inst, const uint8_t* mask /* or NULL */,
                             int B, int H, int W,
                             float* angle /* [B,256] */, uint8_t* present /* [B,256] */)
{
    const int64_t P = (int64_t)H * W;
    for (int b = 0; b < B; ++b) {
        double s0[256] = {0}, s1[256] = {0};
        uint8_t pr[256] = {0};
        const float* o0 = orientation + (int64_t)b * 2 * P;
        const float* o1 = o0 + P;
        for (int64_t p = 0; p < P; ++p) {
            if (mask && !mask[b * P + p]) continue;
            const uint8_t id = inst[b * P + p];
            if (!id) continue;
            s0[id] += o0[p]; s1[id] += o1[p]; pr[id] = 1;
        }
        for (int i = 0; i < 256; ++i) {
            present[b * 256 + i] = pr[i];
            angle[b * 256 + i] = pr[i] ? (float)atan2((double)(float)s1[i], (double)(float)s0[i]) : NAN;
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* a11  MeanIntersectionOverUnion.update   metric/miou.py:44-56               */
/*      confmat[t, p] += 1  ( bincount(t*n + p, minlength=n*n) )              */
/* ------------------------------------------------------------------------- */
int orc_confmat_update(const int64_t* preds, const int64_t* target, int64_t n_px,
                       int n_classes, int64_t* confmat)
{
    for (int64_t i = 0; i < n_px; ++i) {
        const int64_t t = target[i], p = preds[i];
        /* bincount needs non-negative input and the reshape to (n,n) needs
         * every bin < n*n */
        const int64_t bin = t * n_classes + p;
        if (bin < 0 || bin >= (int64_t)n_classes * n_classes) return ORC_ERR_RANGE;
        confmat[bin]++;
    }
    return ORC_OK;
}

/* metric/miou.py:58-94  compute(): returns miou, fills ious[n] (NaN = ignored) */
float orc_miou_compute(const int64_t* confmat, int n, int ignore_first_class, float* ious)
{
    const int s = ignore_first_class ? 1 : 0;
    float acc = 0.0f; int cnt = 0;
    /* torch.mean over float32 values: reference sums fp32; order effects are
     * below the stated 1e-5 tolerance, fp64 accumulate here */
    double dacc = 0.0;
    for (int c = 0; c < n; ++c) if (ious) ious[c] = NAN;
    for (int c = s; c < n; ++c) {
        int64_t sum_pred = 0, sum_gt = 0;
        for (int r = 0; r < n; ++r) sum_pred += confmat[(int64_t)r * n + c];
        for (int k = 0; k < n; ++k) sum_gt += confmat[(int64_t)c * n + k];
        float tp = (float)confmat[(int64_t)c * n + c];
        float fsp = (float)sum_pred, fsg = (float)sum_gt;
        if (ignore_first_class) fsp -= (float)confmat[c];       /* confmat[0, c] :68 */
        if (fsg == 0.0f) continue;                              /* :71-74 */
        float iou = tp / (fsp + fsg - tp);                      /* :77-79 */
        if (ious) ious[c] = iou;
        dacc += iou; cnt++;
    }
    (void)acc;
    return cnt ? (float)(dacc / cnt) : NAN;
}

/* ------------------------------------------------------------------------- */
/* a12  compare_and_accumulate   metric/pq.py:60-179                          */
/* Per image.  Results are ADDED to iou/tp/fn/fp [num_categories] (float64).   */
/* matches: (gt_segment_id, pred_segment_id) pairs in ascending intersection-  */
/* id order (the dict iteration order of the reference).                       */
/* ------------------------------------------------------------------------- */
typedef struct { int64_t id; int64_t cnt; } orc_idcnt;

static int64_t unique_counts(const int64_t* v, int64_t n, int64_t* scratch, orc_idcnt* out)
{
    memcpy(scratch, v, (size_t)n * sizeof(int64_t));
    qsort(scratch, (size_t)n, sizeof(int64_t), cmp_i64);
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (m > 0 && out[m - 1].id == scratch[i]) out[m - 1].cnt++;
        else { out[m].id = scratch[i]; out[m].cnt = 1; m++; }
    }
    return m;
}

static int64_t lookup(const orc_idcnt* t, int64_t m, int64_t id)
{
    int64_t lo = 0, hi = m - 1;
    while (lo <= hi) {
        int64_t mid = (lo + hi) / 2;
        if (t[mid].id == id) return t[mid].cnt;
        if (t[mid].id < id) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

/* Python floor division / modulo for int64 */
static int64_t floordiv(int64_t a, int64_t b)
{
    int64_t q = a / b, r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) q--;
    return q;
}
static int64_t floormod(int64_t a, int64_t b)
{
    int64_t r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) r += b;
    return r;
}

int orc_pq_compare_and_accumulate(const int64_t* pred, const int64_t* target, int64_t n_px,
                                  int num_categories, int64_t ignored_label,
                                  int64_t max_instances_per_category, int64_t offset,
                                  int64_t void_segment_id,
                                  double* iou_per_class, double* tp_per_class,
                                  double* fn_per_class, double* fp_per_class,
                                  int match_capacity, int64_t* matches /* [cap,2] */,
                                  int32_t* n_matches)
{
    int64_t* scratch = (int64_t*)malloc((size_t)n_px * sizeof(int64_t));
    int64_t* iid = (int64_t*)malloc((size_t)n_px * sizeof(int64_t));
    orc_idcnt* ta = (orc_idcnt*)malloc((size_t)n_px * sizeof(orc_idcnt));
    orc_idcnt* pa = (orc_idcnt*)malloc((size_t)n_px * sizeof(orc_idcnt));
    orc_idcnt* ia = (orc_idcnt*)malloc((size_t)n_px * sizeof(orc_idcnt));
    int rc = ORC_OK;
    /* pq.py:83-84 */
    const int64_t nt = unique_counts(target, n_px, scratch, ta);
    const int64_t np_ = unique_counts(pred, n_px, scratch, pa);
    /* pq.py:104 (int64 wrap-around arithmetic like torch) */
    for (int64_t i = 0; i < n_px; ++i)
        iid[i] = (int64_t)((uint64_t)target[i] * (uint64_t)offset + (uint64_t)pred[i]);
    const int64_t ni = unique_counts(iid, n_px, scratch, ia);      /* :109 */

    uint8_t* gt_matched = (uint8_t*)calloc((size_t)nt, 1);
    uint8_t* pred_matched = (uint8_t*)calloc((size_t)np_, 1);
    int32_t nm = 0;

    for (int64_t e = 0; e < ni; ++e) {                              /* :119 */
        const int64_t intersection_id = ia[e].id;
        if (intersection_id == void_segment_id) continue;           /* :120-121 */
        const int64_t gt_id = floordiv(intersection_id, offset);    /* :123 */
        const int64_t pr_id = floormod(intersection_id, offset);    /* :124 */
        const int64_t gt_cat = floordiv(gt_id, max_instances_per_category);
        const int64_t pr_cat = floordiv(pr_id, max_instances_per_category);
        if (gt_cat != pr_cat) continue;                             /* :128-129 */
        /* prediction_void_overlap (:35-44) */
        int64_t r = lookup(ia, ni, (int64_t)((uint64_t)void_segment_id * (uint64_t)offset + (uint64_t)pr_id));
        if (r < 0) r = 0;
        const int64_t tsa = lookup(ta, nt, gt_id);                  /* KeyError in ref if absent */
        const int64_t psa = lookup(pa, np_, pr_id);
        if (tsa < 0 || psa < 0) { rc = ORC_ERR_RANGE; goto done; }
        const int64_t uni = tsa + psa - ia[e].cnt - r;              /* :143 */
        const double iou = (double)ia[e].cnt / (double)uni;         /* :145 */
        if (iou > 0.5) {                                            /* :147 */
            if (gt_cat < 0 || gt_cat >= num_categories) { rc = ORC_ERR_RANGE; goto done; }
            tp_per_class[gt_cat] += 1.0;
            iou_per_class[gt_cat] += iou;
            for (int64_t k = 0; k < nt; ++k) if (ta[k].id == gt_id) gt_matched[k] = 1;
            for (int64_t k = 0; k < np_; ++k) if (pa[k].id == pr_id) pred_matched[k] = 1;
            if (matches) {
                if (nm < match_capacity) { matches[2 * nm] = gt_id; matches[2 * nm + 1] = pr_id; }
                else rc = ORC_ERR_CAPACITY;
            }
            nm++;
        }
    }
    /* false negatives (:155-163) */
    for (int64_t k = 0; k < nt; ++k) {
        if (gt_matched[k]) continue;
        const int64_t cat = floordiv(ta[k].id, max_instances_per_category);
        if (cat == ignored_label) continue;
        if (cat < 0 || cat >= num_categories) { rc = ORC_ERR_RANGE; goto done; }
        fn_per_class[cat] += 1.0;
    }
    /* false positives (:165-177); ignored segments: gt ids whose category is the
     * ignored label (:89-93) */
    for (int64_t k = 0; k < np_; ++k) {
        if (pred_matched[k]) continue;
        int64_t pio = 0;
        for (int64_t g = 0; g < nt; ++g) {
            if (floordiv(ta[g].id, max_instances_per_category) != ignored_label) continue;
            int64_t c = lookup(ia, ni, (int64_t)((uint64_t)ta[g].id * (uint64_t)offset + (uint64_t)pa[k].id));
            if (c > 0) pio += c;
        }
        if ((double)pio / (double)pa[k].cnt > 0.5) continue;
        const int64_t cat = floordiv(pa[k].id, max_instances_per_category);
        if (cat < 0 || cat >= num_categories) { rc = ORC_ERR_RANGE; goto done; }
        fp_per_class[cat] += 1.0;
    }
    if (n_matches) *n_matches = nm;
done:
    free(scratch); free(iid); free(ta); free(pa); free(ia); free(gt_matched); free(pred_matched);
    return rc;
}

/* ------------------------------------------------------------------------- */
/* a6  CrossEntropyLossSemantic._compute_loss   loss/ce.py:40-68              */
/* target: uint8 labels, 0 = void (shifted by -1 -> ignore_index -1).          */
/* torch.nn.CrossEntropyLoss(weight, reduction='sum', label_smoothing=ls):     */
/*   per px:  (1-ls) * w[t] * (-logp[t])  +  (ls/C) * sum_c w[c] * (-logp[c])   */
/* returns sum over non-void px (fp64), n_elements; optional grad wrt logits   */
/* (d loss_sum / d logits, fp32).                                             */
/* weighted_reduction (ESANet, :57-68): loss_sum / sum_c n_c * w_c.            */
/* ------------------------------------------------------------------------- */
int orc_loss_ce(const float* logits, const uint8_t* target, const float* weights /* or NULL */,
                int B, int C, int H, int W, float label_smoothing,
                double* loss_sum, int64_t* n_elements, double* weighted_divisor,
                float* grad /* or NULL, [B,C,H,W] */)
{
    const int64_t P = (int64_t)H * W;
    double total = 0.0, wdiv = 0.0; int64_t n = 0;
    double* logp = (double*)malloc((size_t)C * sizeof(double));
    double wsum_all = 0.0;
    for (int c = 0; c < C; ++c) wsum_all += weights ? weights[c] : 1.0;
    for (int b = 0; b < B; ++b) {
        const float* lb = logits + (int64_t)b * C * P;
        for (int64_t p = 0; p < P; ++p) {
            const int t = (int)target[b * P + p] - 1;              /* ce.py:46 */
            if (t < 0) {
                if (grad) for (int c = 0; c < C; ++c) grad[((int64_t)b * C + c) * P + p] = 0.0f;
                continue;
            }
            if (t >= C) { free(logp); return ORC_ERR_RANGE; }
            double m = -INFINITY;
            for (int c = 0; c < C; ++c) { double v = lb[(int64_t)c * P + p]; if (v > m) m = v; }
            double s = 0.0;
            for (int c = 0; c < C; ++c) s += exp((double)lb[(int64_t)c * P + p] - m);
            const double lse = m + log(s);
            double smooth = 0.0;
            for (int c = 0; c < C; ++c) {
                logp[c] = (double)lb[(int64_t)c * P + p] - lse;
                smooth += (weights ? weights[c] : 1.0) * (-logp[c]);
            }
            const double wt = weights ? weights[t] : 1.0;
            total += (1.0 - label_smoothing) * wt * (-logp[t]) + (label_smoothing / C) * smooth;
            wdiv += wt;
            n++;
            if (grad) {
                /* d/dx_j of  a*(-logp_t) + sum_c b_c*(-logp_c)  =  (a + sum b) p_j - a [j==t] - b_j */
                const double a = (1.0 - label_smoothing) * wt;
                const double bsum = (label_smoothing / C) * wsum_all;
                for (int c = 0; c < C; ++c) {
                    const double pj = exp(logp[c]);
                    const double bj = (label_smoothing / C) * (weights ? weights[c] : 1.0);
                    grad[((int64_t)b * C + c) * P + p] =
                        (float)((a + bsum) * pj - (c == t ? a : 0.0) - bj);
                }
            }
        }
    }
    free(logp);
    *loss_sum = total; *n_elements = n;
    if (weighted_divisor) *weighted_divisor = wdiv;
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* a7  MSELoss / L1Loss ._compute_loss (reduction='sum')  loss/mse.py:21-41,   */
/*     loss/l1.py:21-41, with the task-helper masking of                       */
/*     task_helper/instance.py:129-139 (center: pred*mask, n = sum(mask)) and   */
/*     :154-167 (offset: pred*fg broadcast over the 2 channels, channel mean,   */
/*     n = sum(fg)).   kind: 0 = MSE, 1 = L1.   C = 1 (3-D input, no channel    */
/*     mean) or C > 1 (4-D, mean over C then sum).                             */
/* grad = d loss_sum / d pred (through the mask multiply).                     */
/* ------------------------------------------------------------------------- */
int orc_loss_masked_elementwise(const float* pred, const float* target, const uint8_t* mask /* [B,H,W] or NULL */,
                                int B, int C, int H, int W, int kind,
                                double* loss_sum, int64_t* n_mask, float* grad)
{
    const int64_t P = (int64_t)H * W;
    double total = 0.0; int64_t n = 0;
    for (int b = 0; b < B; ++b)
        for (int64_t p = 0; p < P; ++p) {
            const int mk = mask ? (mask[b * P + p] != 0) : 1;
            n += mk;
            double acc = 0.0;
            for (int c = 0; c < C; ++c) {
                const int64_t i = ((int64_t)b * C + c) * P + p;
                const float x = pred[i] * (float)mk;              /* pred*mask */
                const float d = x - target[i];
                double l, g;
                if (kind == 0) { l = (double)d * d; g = 2.0 * d; }
                else { l = fabs((double)d); g = (d > 0) - (d < 0); }
                acc += l;
                if (grad) grad[i] = (float)(g * mk / C);
            }
            total += acc / C;
        }
    *loss_sum = total; *n_mask = n;
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* a8  VonMisesLossBiternion._compute_loss   loss/vonmises.py:27-51           */
/*     with the boolean gather of task_helper/instance.py:186-216: rows are    */
/*     the pixels where mask is set (pred/target in [B,2,H,W] layout here).    */
/*     loss = sum_rows 1 - exp(kappa * (x0*y0 + x1*y1 - 1)),  n = #rows        */
/* ------------------------------------------------------------------------- */
int orc_loss_vonmises(const float* pred, const float* target, const uint8_t* mask,
                      int B, int H, int W, float kappa,
                      double* loss_sum, int64_t* n_rows, float* grad /* [B,2,H,W] or NULL */)
{
    const int64_t P = (int64_t)H * W;
    double total = 0.0; int64_t n = 0;
    for (int b = 0; b < B; ++b)
        for (int64_t p = 0; p < P; ++p) {
            const int64_t i0 = ((int64_t)b * 2) * P + p, i1 = i0 + P;
            if (mask && !mask[b * P + p]) {
                if (grad) { grad[i0] = 0.0f; grad[i1] = 0.0f; }
                continue;
            }
            const double dot = (double)pred[i0] * target[i0] + (double)pred[i1] * target[i1];
            const double e = exp((double)kappa * (dot - 1.0));
            total += 1.0 - e; n++;
            if (grad) {
                grad[i0] = (float)(-e * kappa * target[i0]);
                grad[i1] = (float)(-e * kappa * target[i1]);
            }
        }
    *loss_sum = total; *n_rows = n;
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* a9  CosineEmbeddingLoss._compute_loss   loss/cos_emb.py:21-56              */
/*     torch.nn.CosineEmbeddingLoss(reduction='none') with label +1:           */
/*       1 - x.y / sqrt((|x|^2 + eps) * (|y|^2 + eps)),  eps = 1e-12 (ATen      */
/*       cosine_embedding_loss: EPSILON = 1e-12 added to the squared norms)    */
/*     rows gathered as in task_helper/dense_visual_embedding.py:110-171:      */
/*     pred [B,D,H,W] at px where indices != 0; target = lut[b][indices-1].    */
/* ------------------------------------------------------------------------- */
int orc_loss_cosine_embedding(const float* pred /* [B,D,H,W] */, const int32_t* indices /* [B,H,W] */,
                              const float* lut /* [B,L,D] */, int B, int D, int H, int W, int L,
                              double* loss_sum, int64_t* n_rows, float* grad /* [B,D,H,W] or NULL */)
{
    const int64_t P = (int64_t)H * W;
    const double EPS = 1e-12;
    double total = 0.0; int64_t n = 0;
    for (int b = 0; b < B; ++b)
        for (int64_t p = 0; p < P; ++p) {
            const int32_t ix = indices[b * P + p];
            if (ix == 0) {
                if (grad) for (int d = 0; d < D; ++d) grad[((int64_t)b * D + d) * P + p] = 0.0f;
                continue;
            }
            if (ix < 0 || ix > L) return ORC_ERR_RANGE;
            const float* y = lut + ((int64_t)b * L + (ix - 1)) * D;
            double xy = 0, xx = 0, yy = 0;
            for (int d = 0; d < D; ++d) {
                const double xv = pred[((int64_t)b * D + d) * P + p];
                xy += xv * y[d]; xx += xv * xv; yy += (double)y[d] * y[d];
            }
            const double den = sqrt((xx + EPS) * (yy + EPS));
            total += 1.0 - xy / den; n++;
            if (grad) {
                for (int d = 0; d < D; ++d) {
                    const double xv = pred[((int64_t)b * D + d) * P + p];
                    /* d/dx (-xy/den) = -y/den + xy * x / ((xx+eps) * den) */
                    grad[((int64_t)b * D + d) * P + p] =
                        (float)(-(double)y[d] / den + xy * xv / ((xx + EPS) * den));
                }
            }
        }
    *loss_sum = total; *n_rows = n;
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* f2  DensePostprocessingBase._crop_to_valid_region_and_resize_prediction     */
/*     model/postprocessing/dense_base.py:15-58                                */
/* crop [y0:y0+h, x0:x0+w] of every [Hs,Ws] plane, then F.interpolate to       */
/* (Ho,Wo).  The index / weight arithmetic restates ATen's CPU kernels         */
/* (aten/src/ATen/native/UpSample.h: nearest_idx, area_pixel_compute_source_   */
/* index, compute_source_index_and_lambda; cpu/UpSampleKernel.cpp generic      */
/* 2-d linear kernel) and is pinned bit-for-bit by tests/golden/fullres_*.npz. */
/* ------------------------------------------------------------------------- */
static int nearest_src(float scale, int dst, int in)
{
    int i = (int)floorf((float)dst * scale);
    return i < in - 1 ? i : in - 1;
}

static void bilinear_src(float scale, int dst, int in, int* i0, int* i1, float* w0, float* w1)
{
    float s = fmaf(scale, (float)dst + 0.5f, -0.5f);
    if (s < 0.f) s = 0.f;
    *i0 = (int)s < in - 1 ? (int)s : in - 1;
    *i1 = *i0 + 1 < in - 1 ? *i0 + 1 : in - 1;
    float l = s - (float)*i0;
    if (l < 0.f) l = 0.f;
    if (l > 1.f) l = 1.f;
    *w1 = l;
    *w0 = 1.0f - l;
}

/* elem_bytes: 1, 2, 4, 8 (integer maps) or -4 (float32 maps).  Integer maps wider  */
/* than 2 bytes go through float32 like the reference (dense_base.py:38-40, :52).    */
int orc_resize_nearest(const void* src, int elem_bytes, int planes, int Hs, int Ws,
                       int y0, int x0, int h, int w, int Ho, int Wo, void* dst)
{
    if (!src || !dst || planes <= 0 || h <= 0 || w <= 0 || Ho <= 0 || Wo <= 0) return ORC_ERR_ARG;
    if (y0 < 0 || x0 < 0 || y0 + h > Hs || x0 + w > Ws) return ORC_ERR_ARG;
    const float sy = (float)h / (float)Ho, sx = (float)w / (float)Wo;
    for (int p = 0; p < planes; ++p)
        for (int y = 0; y < Ho; ++y) {
            const size_t srow = ((size_t)p * Hs + y0 + nearest_src(sy, y, h)) * Ws + x0;
            const size_t drow = ((size_t)p * Ho + y) * Wo;
            for (int x = 0; x < Wo; ++x) {
                const size_t s = srow + nearest_src(sx, x, w), d = drow + x;
                switch (elem_bytes) {
                    case 1: ((uint8_t*)dst)[d] = ((const uint8_t*)src)[s]; break;
                    case 2: ((int16_t*)dst)[d] = ((const int16_t*)src)[s]; break;
                    case 4: ((int32_t*)dst)[d] = (int32_t)(float)((const int32_t*)src)[s]; break;
                    case 8: ((int64_t*)dst)[d] = (int64_t)(float)((const int64_t*)src)[s]; break;
                    case -4: ((float*)dst)[d] = ((const float*)src)[s]; break;
                    default: return ORC_ERR_ARG;
                }
            }
        }
    return ORC_OK;
}

int orc_resize_bilinear(const float* src, int planes, int Hs, int Ws,
                        int y0, int x0, int h, int w, int Ho, int Wo, float* dst)
{
    if (!src || !dst || planes <= 0 || h <= 0 || w <= 0 || Ho <= 0 || Wo <= 0) return ORC_ERR_ARG;
    if (y0 < 0 || x0 < 0 || y0 + h > Hs || x0 + w > Ws) return ORC_ERR_ARG;
    const float sy = (float)h / (float)Ho, sx = (float)w / (float)Wo;
    for (int p = 0; p < planes; ++p)
        for (int y = 0; y < Ho; ++y) {
            int iy0, iy1;
            float wy0, wy1;
            bilinear_src(sy, y, h, &iy0, &iy1, &wy0, &wy1);
            const float* r0 = src + ((size_t)p * Hs + y0 + iy0) * Ws + x0;
            const float* r1 = src + ((size_t)p * Hs + y0 + iy1) * Ws + x0;
            float* d = dst + ((size_t)p * Ho + y) * Wo;
            for (int x = 0; x < Wo; ++x) {
                int ix0, ix1;
                float wx0, wx1;
                bilinear_src(sx, x, w, &ix0, &ix1, &wx0, &wx1);
                const float t0 = fmaf(r0[ix0], wx0, r0[ix1] * wx1);
                const float t1 = fmaf(r1[ix0], wx0, r1[ix1] * wx1);
                d[x] = fmaf(t0, wy0, t1 * wy1);
            }
        }
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* f3  PanopticPostprocessing compute_scores   model/postprocessing/panoptic.py:171-239 */
/* semantic score = softmax(logits)[panoptic class - 1] (0 on void, :176-195);            */
/* per (pan_id -> ins_id) of the id dict, in dict order: mask = (pan == pan_id);          */
/* instance score -> mask (:208-209); mean semantic score over mask (:213-215);           */
/* panoptic score = mean * instance score (f32 product) -> mask (:225-227).               */
/* ------------------------------------------------------------------------- */
int orc_panoptic_scores(const float* logits, const int64_t* pan_sem, const int64_t* pan,
                        const int64_t* ids_pan, const int64_t* ids_ins, const int32_t* n_ids,
                        int cap, const float* inst_score_by_id /* [B,256] */,
                        int B, int C, int H, int W,
                        float* sem_score, float* inst_score, float* pan_score,
                        float* mean_by_id /* [B,256], NaN where unused */)
{
    const int64_t P = (int64_t)H * W;
    for (int b = 0; b < B; ++b) {
        const float* x = logits + (size_t)b * C * P;
        for (int64_t p = 0; p < P; ++p) {
            const size_t o = (size_t)b * P + p;
            const int64_t k = pan_sem[o];
            float s = 0.f;
            if (k > 0 && k <= C) {
                float m = -INFINITY;
                for (int c = 0; c < C; ++c) if (x[(size_t)c * P + p] > m) m = x[(size_t)c * P + p];
                double se = 0.0;
                int bad = 0;
                for (int c = 0; c < C; ++c) {
                    const float v = x[(size_t)c * P + p];
                    if (v != v) bad = 1;
                    se += exp((double)v - (double)m);
                }
                if (bad || !(m > -INFINITY) || m == INFINITY) s = NAN;
                else s = (float)(exp((double)x[(size_t)(k - 1) * P + p] - (double)m) / se);
            }
            sem_score[o] = s;
            inst_score[o] = 0.f;
            pan_score[o] = s;
        }
        for (int i = 0; i < 256; ++i) mean_by_id[(size_t)b * 256 + i] = NAN;
        for (int j = 0; j < n_ids[b]; ++j) {
            const int64_t pid = ids_pan[(size_t)b * cap + j], iid = ids_ins[(size_t)b * cap + j];
            if (iid < 0 || iid > 255) return ORC_ERR_RANGE;
            const float is = inst_score_by_id[(size_t)b * 256 + iid];
            double sum = 0.0;
            int64_t n = 0;
            for (int64_t p = 0; p < P; ++p)
                if (pan[(size_t)b * P + p] == pid) { sum += sem_score[(size_t)b * P + p]; ++n; }
            const float mean = n ? (float)(sum / (double)n) : NAN;
            const float prod = mean * is;
            mean_by_id[(size_t)b * 256 + iid] = mean;
            for (int64_t p = 0; p < P; ++p)
                if (pan[(size_t)b * P + p] == pid) {
                    inst_score[(size_t)b * P + p] = is;
                    pan_score[(size_t)b * P + p] = prod;
                }
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* f4  InstanceTargetGenerator._preprocess   data/preprocessing/instance.py:157-286 */
/* one image; sem u8 [H,W], ins i32 [H,W] (uint16 ids).  Returns ORC_ERR_RANGE  */
/* for ids outside [0,65535] / labels outside [0,n_classes).                    */
/* encoded_ids / skipped_ids: ascending (np.unique order), capacity `cap`.      */
/* ------------------------------------------------------------------------- */
int orc_instance_targets(const uint8_t* sem, const int32_t* ins, int H, int W, int n_classes,
                         const uint8_t* is_thing /* or NULL */, const uint8_t* is_stuff /* or NULL */,
                         int sigma, int normalized,
                         float* center, float* offset_f32 /* [2,H,W] or NULL */,
                         int16_t* offset_i16 /* [2,H,W] or NULL */,
                         uint8_t* fg, uint8_t* center_mask,
                         int32_t* encoded_ids, int32_t* n_encoded,
                         int32_t* skipped_ids, int32_t* n_skipped, int cap)
{
    const int64_t P = (int64_t)H * W;
    uint8_t* present = (uint8_t*)calloc(65536, 1);
    int64_t* hist = (int64_t*)malloc((size_t)n_classes * sizeof(int64_t));
    int16_t* off = (int16_t*)calloc((size_t)2 * P, sizeof(int16_t));
    int rc = ORC_OK;
    for (int64_t p = 0; p < P; ++p) {
        if (ins[p] < 0 || ins[p] > 65535 || sem[p] >= n_classes) { rc = ORC_ERR_RANGE; goto done; }
        present[ins[p]] = 1;
        center[p] = 0.f;
        fg[p] = 0;
    }
    *n_encoded = 0;
    *n_skipped = 0;
    const int r = 3 * sigma + 1;
    for (int id = 1; id < 65536; ++id) {                       /* np.unique: ascending (:191) */
        if (!present[id]) continue;
        memset(hist, 0, (size_t)n_classes * sizeof(int64_t));
        int64_t n = 0, sy = 0, sx = 0;
        for (int64_t p = 0; p < P; ++p)
            if (ins[p] == id) { ++hist[sem[p]]; ++n; sy += p / W; sx += p % W; }
        int cls = 0;
        for (int c = 1; c < n_classes; ++c) if (hist[c] > hist[cls]) cls = c;   /* bincount.argmax (:207-209) */
        if (is_thing && !is_thing[cls]) {                      /* :210-213 */
            if (*n_skipped >= cap) { rc = ORC_ERR_CAPACITY; goto done; }
            skipped_ids[(*n_skipped)++] = id;
            continue;
        }
        if (*n_encoded >= cap) { rc = ORC_ERR_CAPACITY; goto done; }
        encoded_ids[(*n_encoded)++] = id;
        const int cy = (int)((double)sy / (double)n), cx = (int)((double)sx / (double)n);   /* :221-222 */
        for (int y = cy - r; y <= cy + r; ++y) {               /* patch (:223-240) */
            if (y < 0 || y >= H) continue;
            for (int x = cx - r; x <= cx + r; ++x) {
                if (x < 0 || x >= W) continue;
                const double d2 = (double)((x - cx) * (x - cx) + (y - cy) * (y - cy));
                const float g = (float)exp(-d2 / (double)(2 * sigma * sigma));
                if (g > center[(int64_t)y * W + x]) center[(int64_t)y * W + x] = g;
            }
        }
        for (int64_t p = 0; p < P; ++p)
            if (ins[p] == id) {
                fg[p] = 1;
                off[p] = (int16_t)(cy - (int)(p / W));          /* :243-247 */
                off[P + p] = (int16_t)(cx - (int)(p % W));
            }
    }
    for (int64_t p = 0; p < P; ++p) {
        if (offset_i16) { offset_i16[p] = off[p]; offset_i16[P + p] = off[P + p]; }
        if (offset_f32) {
            if (normalized) {                                   /* :249-253 */
                offset_f32[p] = (float)off[p] / (float)H;
                offset_f32[P + p] = (float)off[P + p] / (float)W;
            } else {
                offset_f32[p] = (float)off[p];
                offset_f32[P + p] = (float)off[P + p];
            }
        }
        if (center_mask) center_mask[p] = fg[p] || (is_stuff && is_stuff[sem[p]]);   /* :271-277 */
    }
done:
    free(present); free(hist); free(off);
    return rc;
}

/* f4  DenseVisualEmbeddingTargetGenerator._process_scale   dense_visual_embedding.py:22-45 */
int orc_dve_indices(const int64_t* pan, int64_t n_px, const int64_t* keys, int n_keys, int32_t* indices)
{
    for (int64_t p = 0; p < n_px; ++p) {
        int32_t idx = 0;
        for (int k = 0; k < n_keys; ++k) if (keys[k] == pan[p]) idx = k + 1;    /* later key overwrites */
        indices[p] = idx;
    }
    return ORC_OK;
}
