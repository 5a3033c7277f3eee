// targets.hip — ground-truth target generation on gfx950.
//
// The reference builds these per SAMPLE in numpy inside the dataloader workers; here they are
// built per BATCH on the device from the label maps in their on-wire dtypes (semantic uint8,
// instance uint16 -> int32, data/preprocessing/torch.py:60-66):
//
//   nmsa_instance_targets   InstanceTargetGenerator._preprocess
//                           data/preprocessing/instance.py:157-286
//       per instance id (ascending, np.unique): majority semantic class (bincount.argmax),
//       skipped when that class is not a thing; center = int(mean(y)), int(mean(x));
//       center heat-map = max over the encoded instances of a (6s+3)^2 Gaussian patch;
//       offset = (cy - y, cx - x) on the instance's pixels (int16, or / (H, W) in float32);
//       foreground; center mask = foreground | stuff pixels
//   nmsa_panoptic_targets   PanopticTargetGenerator._preprocess -> naive_merge_semantic_
//                           and_instance_np   data/preprocessing/panoptic.py:48-85,
//                           utils/panoptic_merge.py:43-107
//       every (instance id, semantic class != void) pair present in the image is a segment:
//       panoptic id = class * max_instances + (number of instance ids <= this one that contain
//       the class); stuff classes paste class * max_instances where there is no instance
//   nmsa_dve_targets        DenseVisualEmbeddingTargetGenerator._preprocess
//                           data/preprocessing/dense_visual_embedding.py:22-93
//       LUT rows = normalise(embedding - diff_factor * image_embedding); indices = 1 + position
//       of the pixel's panoptic id in the key list (0 = none; the LAST duplicate key wins)
//
// Instance ids 0..65535 are ranked per image with the presence-bitmap scheme of id_rank.hpp
// (ascending id order = np.unique order, which the running counters depend on); all
// per-instance statistics are exact integers (class histograms, sum of y / x, counts), so the
// centers, offsets, ids and masks are bit-identical to the reference; the heat-map values come
// from a caller-provided table indexed by the integer squared distance.
#include "nmsa_common.hpp"
#include "id_rank.hpp"

namespace nmsa {
namespace {

constexpr int TG_ST_OVERFLOW = 1;        // more distinct instance ids than max_instances
constexpr int TG_ST_ID_RANGE = 32;       // instance id outside [0, 65535]
constexpr int TG_ST_CLASS_RANGE = 64;    // semantic label outside [0, n_classes)
constexpr int TG_ST_PAIR_OVERFLOW = 128; // more (instance, class) segments than the id table holds

struct TgView {
    unsigned long long* sum_y;   // [cap]
    unsigned long long* sum_x;   // [cap]
    uint32_t* bitmap;            // [MW_WORDS]
    uint32_t* prefix;            // [MW_WORDS]
    int32_t* id_of_dense;        // [cap]
    int32_t* center_yx;          // [cap * 2]   (valid where enc)
    int32_t* enc;                // [cap]       1 = encoded instance
    int32_t* enc_list;           // [cap * 2]   compacted (cy, cx) of the encoded instances
    int32_t* counters;           // [4]         n_dense, n_encoded
    uint32_t* votes;             // [cap * NC]  class histogram per instance (then: ranks)
};

__host__ __device__ inline size_t tg_image_bytes(int cap, int NC)
{
    size_t n = (size_t)cap * 8 * 2 + (size_t)MW_WORDS * 4 * 2 + (size_t)cap * 4 * (1 + 2 + 1 + 2) + 16 +
               (size_t)cap * NC * 4;
    return (n + 15) & ~(size_t)15;
}

__device__ __forceinline__ TgView tg_view(unsigned char* ws, int b, int cap, int NC)
{
    unsigned char* base = ws + (size_t)b * tg_image_bytes(cap, NC);
    TgView v;
    v.sum_y = (unsigned long long*)base;
    v.sum_x = v.sum_y + cap;
    v.bitmap = (uint32_t*)(v.sum_x + cap);
    v.prefix = v.bitmap + MW_WORDS;
    v.id_of_dense = (int32_t*)(v.prefix + MW_WORDS);
    v.center_yx = v.id_of_dense + cap;
    v.enc = v.center_yx + 2 * cap;
    v.enc_list = v.enc + cap;
    v.counters = v.enc_list + 2 * cap;
    v.votes = (uint32_t*)(v.counters + 4);
    return v;
}

__device__ __forceinline__ int wave_reduce_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// ---- presence of every non-zero instance id ------------------------------------------------
__global__ __launch_bounds__(256) void k_tg_presence(
    const void* __restrict__ ins, int ins_dtype, int P, int cap, int NC,
    unsigned char* __restrict__ ws, int* __restrict__ status)
{
    const int b = blockIdx.y;
    TgView v = tg_view(ws, b, cap, NC);
    const int stride = gridDim.x * blockDim.x;
    const int trips = (P + stride - 1) / stride;
    bool bad = false;
    for (int k = 0; k < trips; ++k) {
        const int p = (k * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        int key = -1;
        if (p < P) {
            const int64_t i = mw_load(ins, ins_dtype, (size_t)b * P + p);
            if (i < 0 || i > MW_MAX_ID) bad = true;
            else if (i > 0) key = (int)i;
        }
        wave_aggregate_add(key, [&](int id, uint32_t) { atomicOr(&v.bitmap[id >> 5], 1u << (id & 31)); });
    }
    if (bad) atomicOr(status, TG_ST_ID_RANGE);
}

// ---- rank: one 1024-thread workgroup per image ---------------------------------------------------
__global__ __launch_bounds__(1024) void k_tg_rank(unsigned char* __restrict__ ws, int cap, int NC,
                                                  int* __restrict__ status)
{
    __shared__ int scratch[32];
    const int b = blockIdx.x, t = threadIdx.x;
    TgView v = tg_view(ws, b, cap, NC);
    const uint32_t w0 = v.bitmap[2 * t], w1 = v.bitmap[2 * t + 1];
    const int c = __popc(w0) + __popc(w1);
    int total;
    const int excl = mw_block_scan(c, scratch, &total) - c;
    v.prefix[2 * t] = excl;
    v.prefix[2 * t + 1] = excl + __popc(w0);
    int at = excl;
    for (int half = 0; half < 2; ++half) {
        uint32_t m = half ? w1 : w0;
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1;
            if (at < cap) v.id_of_dense[at] = (2 * t + half) * 32 + bit;
            ++at;
        }
    }
    if (t == 0) {
        v.counters[0] = min(total, cap);
        if (total > cap) atomicOr(status, TG_ST_OVERFLOW);
    }
}

// ---- per-instance statistics: class histogram, sum of y, sum of x --------------------------------
template <bool WITH_MOMENTS>
__global__ __launch_bounds__(256) void k_tg_stats(
    const void* __restrict__ sem, int sem_dtype, const void* __restrict__ ins, int ins_dtype,
    int P, int W, int cap, int NC, unsigned char* __restrict__ ws, int* __restrict__ status)
{
    const int b = blockIdx.y;
    TgView v = tg_view(ws, b, cap, NC);
    const int stride = gridDim.x * blockDim.x;
    const int trips = (P + stride - 1) / stride;
    bool bad = false;
    for (int k = 0; k < trips; ++k) {
        const int p = (k * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        int d = -1, vote_key = -1, y = 0, x = 0;
        if (p < P) {
            const size_t o = (size_t)b * P + p;
            const int64_t i = mw_load(ins, ins_dtype, o);
            if (i > 0 && i <= MW_MAX_ID) {
                const int dd = id_rank_dense(v.bitmap, v.prefix, (int)i);
                if (dd < cap) {
                    d = dd;
                    const int64_t s = mw_load(sem, sem_dtype, o);
                    if (s < 0 || s >= NC) bad = true;
                    else vote_key = dd * NC + (int)s;
                    y = p / W;
                    x = p - y * W;
                }
            }
        }
        wave_aggregate_add(vote_key, [&](int kk, uint32_t cnt) { atomicAdd(&v.votes[kk], cnt); });
        if (WITH_MOMENTS) {
            unsigned long long todo = __ballot(d >= 0);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const int kd = __shfl(d, leader);
                const bool mine = (d == kd);
                const unsigned long long same = __ballot(mine) & todo;
                const int sy = wave_reduce_sum_i(mine ? y : 0);
                const int sx = wave_reduce_sum_i(mine ? x : 0);
                if (lane_id() == 0) {       // __shfl_down reductions land in lane 0
                    atomicAdd(&v.sum_y[kd], (unsigned long long)sy);
                    atomicAdd(&v.sum_x[kd], (unsigned long long)sx);
                }
                todo &= ~same;
            }
        }
    }
    if (bad) atomicOr(status, TG_ST_CLASS_RANGE);
}

// ---- decide: majority class, thing filter, center; ordered lists (one WG per image) --------------
__global__ __launch_bounds__(1024) void k_tg_decide(
    unsigned char* __restrict__ ws, int cap, int NC, const uint8_t* __restrict__ is_thing_class,
    int32_t* __restrict__ encoded_ids, int32_t* __restrict__ n_encoded,
    int32_t* __restrict__ skipped_ids, int32_t* __restrict__ n_skipped)
{
    __shared__ int scratch[32];
    const int b = blockIdx.x, t = threadIdx.x;
    TgView v = tg_view(ws, b, cap, NC);
    const int n_dense = v.counters[0];
    const int per = cap / 1024;                      // cap is a multiple of 1024
    int enc[4], present[4];
    int n_enc = 0, n_skip = 0;
    for (int j = 0; j < per; ++j) {
        const int slot = t * per + j;
        enc[j] = 0;
        present[j] = slot < n_dense;
        if (!present[j]) { v.enc[slot] = 0; continue; }
        const uint32_t* row = v.votes + (size_t)slot * NC;
        uint32_t total = 0;
        int64_t best = -1;
        int c_best = 0;
        for (int c = 0; c < NC; ++c) {
            const uint32_t x = row[c];
            total += x;
            if ((int64_t)x > best) { best = x; c_best = c; }      // np.bincount(..).argmax()
        }
        const bool thing = is_thing_class ? (is_thing_class[c_best] != 0) : true;
        enc[j] = thing && total > 0;
        v.enc[slot] = enc[j];
        if (enc[j]) {
            // int(np.mean(rows)), int(np.mean(cols)): exact integer floor (instance.py:210-211)
            v.center_yx[2 * slot] = (int)(v.sum_y[slot] / total);
            v.center_yx[2 * slot + 1] = (int)(v.sum_x[slot] / total);
        }
        n_enc += enc[j];
        n_skip += !enc[j];
    }
    int tot_enc, tot_skip;
    int pe = mw_block_scan(n_enc, scratch, &tot_enc) - n_enc;
    int ps = mw_block_scan(n_skip, scratch, &tot_skip) - n_skip;
    for (int j = 0; j < per; ++j) {
        const int slot = t * per + j;
        if (!present[j]) continue;
        if (enc[j]) {
            v.enc_list[2 * pe] = v.center_yx[2 * slot];
            v.enc_list[2 * pe + 1] = v.center_yx[2 * slot + 1];
            if (encoded_ids) encoded_ids[(size_t)b * cap + pe] = v.id_of_dense[slot];
            ++pe;
        } else {
            if (skipped_ids) skipped_ids[(size_t)b * cap + ps] = v.id_of_dense[slot];
            ++ps;
        }
    }
    if (t == 0) {
        v.counters[1] = tot_enc;
        if (n_encoded) n_encoded[b] = tot_enc;
        if (n_skipped) n_skipped[b] = tot_skip;
    }
}

// ---- paint: heat-map, offsets, foreground, center mask -------------------------------------------
constexpr int TGP_THREADS = 256;
constexpr int TGP_PX = 1024;                 // consecutive pixels per workgroup
constexpr int TGP_LUT_LDS = 4096;            // heat-map table entries kept in LDS (sigma <= 14)

template <bool NORMALIZED>
__global__ __launch_bounds__(TGP_THREADS) void k_tg_paint(
    const void* __restrict__ sem, int sem_dtype, const void* __restrict__ ins, int ins_dtype,
    const uint8_t* __restrict__ is_stuff_class, const float* __restrict__ gauss_lut, int lut_n,
    int radius, int H, int W, int cap, int NC, unsigned char* __restrict__ ws,
    float* __restrict__ center, void* __restrict__ offset, uint8_t* __restrict__ foreground,
    uint8_t* __restrict__ center_mask)
{
    extern __shared__ int tg_lds[];              // [cap * 2] candidate centers, then the table
    __shared__ int s_n;
    const int b = blockIdx.y;
    const int P = H * W;
    TgView v = tg_view(ws, b, cap, NC);
    const int n_enc = v.counters[1];
    int* s_cand = tg_lds;
    float* s_lut = (float*)(tg_lds + 2 * cap);
    const bool lut_in_lds = lut_n <= TGP_LUT_LDS;

    const int p_begin = blockIdx.x * TGP_PX;
    const int p_end = min(p_begin + TGP_PX, P);
    const int y_lo = p_begin / W - radius, y_hi = (p_end - 1) / W + radius;
    if (threadIdx.x == 0) s_n = 0;
    if (lut_in_lds) for (int i = threadIdx.x; i < lut_n; i += TGP_THREADS) s_lut[i] = gauss_lut[i];
    __syncthreads();
    // centers whose patch can reach this workgroup's rows (unordered: max is order-free)
    for (int i0 = 0; i0 < n_enc; i0 += TGP_THREADS) {
        const int i = i0 + threadIdx.x;
        int cy = 0, cx = 0;
        bool keep = false;
        if (i < n_enc) {
            cy = v.enc_list[2 * i];
            cx = v.enc_list[2 * i + 1];
            keep = cy >= y_lo && cy <= y_hi;
        }
        const unsigned long long m = __ballot(keep);
        int base = 0;
        if (lane_id() == 0 && m) base = atomicAdd(&s_n, __popcll(m));
        base = __shfl(base, 0);
        if (keep) {
            const int at = base + __popcll(m & ((1ull << lane_id()) - 1ull));
            s_cand[2 * at] = cy;
            s_cand[2 * at + 1] = cx;
        }
    }
    __syncthreads();
    const int n_cand = s_n;
    const float* lut = lut_in_lds ? s_lut : gauss_lut;

    for (int p = p_begin + threadIdx.x; p < p_end; p += TGP_THREADS) {
        const int y = p / W, x = p - y * W;
        const size_t o = (size_t)b * P + p;
        // heat-map: max over the patches that cover this pixel (instance.py:213-229)
        float hm = 0.f;
        for (int i = 0; i < n_cand; ++i) {
            const int dy = y - s_cand[2 * i], dx = x - s_cand[2 * i + 1];
            if (abs(dy) <= radius && abs(dx) <= radius) hm = fmaxf(hm, lut[dy * dy + dx * dx]);
        }
        center[o] = hm;
        // offsets / foreground (instance.py:201-206,232-237)
        const int64_t id = mw_load(ins, ins_dtype, o);
        bool fg = false;
        int oy = 0, ox = 0;
        if (id > 0 && id <= MW_MAX_ID) {
            const int d = id_rank_dense(v.bitmap, v.prefix, (int)id);
            if (d < cap && v.enc[d]) {
                fg = true;
                oy = (int)(int16_t)(v.center_yx[2 * d] - y);        // int16 image (instance.py:180)
                ox = (int)(int16_t)(v.center_yx[2 * d + 1] - x);
            }
        }
        const size_t oo = (size_t)b * 2 * P + p;
        if (NORMALIZED) {
            ((float*)offset)[oo] = __fdiv_rn((float)oy, (float)H);   // instance.py:239-243
            ((float*)offset)[oo + P] = __fdiv_rn((float)ox, (float)W);
        } else {
            ((int16_t*)offset)[oo] = (int16_t)oy;
            ((int16_t*)offset)[oo + P] = (int16_t)ox;
        }
        foreground[o] = fg;
        if (center_mask) {
            bool cm = fg;
            if (is_stuff_class) {
                const int64_t s = mw_load(sem, sem_dtype, o);
                if (s >= 0 && s < NC && is_stuff_class[s]) cm = true;   // instance.py:263-269
            }
            center_mask[o] = cm;
        }
    }
}

// ---- naive merge: ranks of the (instance, class) segments ------------------------------------------
// One workgroup per image, thread c owns class c: walks the instances in ascending id order and
// replaces every non-zero histogram entry by the running per-class counter
// (class_id_tracker, panoptic_merge.py:76-79).  Then the id dict in the reference's insertion
// order (instance ascending, class ascending).
__global__ __launch_bounds__(1024) void k_tg_naive_ranks(
    unsigned char* __restrict__ ws, int cap, int NC, int pair_cap, int64_t max_inst,
    int64_t* __restrict__ ids_pan, int64_t* __restrict__ ids_ins, int32_t* __restrict__ n_ids,
    int* __restrict__ status)
{
    __shared__ int scratch[32];
    const int b = blockIdx.x, t = threadIdx.x;
    TgView v = tg_view(ws, b, cap, NC);
    const int n_dense = v.counters[0];
    for (int c = t; c < NC; c += 1024) {
        uint32_t run = 0;
        for (int d = 0; d < n_dense; ++d) {
            uint32_t* cell = &v.votes[(size_t)d * NC + c];
            if (c == 0) { *cell = 0; continue; }              // void is ignored (:73-74)
            if (*cell) *cell = ++run;
        }
    }
    __syncthreads();
    const int per = cap / 1024;
    int cnt = 0;
    for (int j = 0; j < per; ++j) {
        const int slot = t * per + j;
        if (slot >= n_dense) continue;
        for (int c = 1; c < NC; ++c) cnt += v.votes[(size_t)slot * NC + c] != 0;
    }
    int total;
    int pos = mw_block_scan(cnt, scratch, &total) - cnt;
    for (int j = 0; j < per; ++j) {
        const int slot = t * per + j;
        if (slot >= n_dense) continue;
        for (int c = 1; c < NC; ++c) {
            const uint32_t r = v.votes[(size_t)slot * NC + c];
            if (!r) continue;
            if (pos < pair_cap) {
                ids_pan[(size_t)b * pair_cap + pos] = (int64_t)c * max_inst + r;
                ids_ins[(size_t)b * pair_cap + pos] = v.id_of_dense[slot];
            }
            ++pos;
        }
    }
    if (t == 0) {
        n_ids[b] = min(total, pair_cap);
        if (total > pair_cap) atomicOr(status, TG_ST_PAIR_OVERFLOW);
    }
}

__global__ __launch_bounds__(256) void k_tg_naive_paint(
    const void* __restrict__ sem, int sem_dtype, const void* __restrict__ ins, int ins_dtype,
    const uint8_t* __restrict__ is_thing_class, int P, int cap, int NC, int64_t max_inst,
    int64_t void_label, unsigned char* __restrict__ ws, int64_t* __restrict__ pan)
{
    const int b = blockIdx.y;
    TgView v = tg_view(ws, b, cap, NC);
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const size_t o = (size_t)b * P + p;
        const int64_t i = mw_load(ins, ins_dtype, o);
        const int64_t s = mw_load(sem, sem_dtype, o);
        int64_t r = void_label;
        if (s > 0 && s < NC) {
            if (i > 0 && i <= MW_MAX_ID) {
                const int d = id_rank_dense(v.bitmap, v.prefix, (int)i);
                if (d < cap) r = s * max_inst + v.votes[(size_t)d * NC + s];
            } else if (i == 0 && !(is_thing_class && is_thing_class[s])) {
                r = s * max_inst;                          // stuff paste (:90-101)
            }
        }
        pan[o] = r;
    }
}

// ---- dense visual embedding targets -----------------------------------------------------------
__global__ __launch_bounds__(256) void k_dve_indices(
    const int64_t* __restrict__ pan, const int64_t* __restrict__ keys, const int32_t* __restrict__ n_keys,
    int K, int P, int32_t* __restrict__ indices)
{
    extern __shared__ int64_t s_keys[];
    const int b = blockIdx.y;
    const int n = min(n_keys[b], K);
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_keys[i] = keys[(size_t)b * K + i];
    __syncthreads();
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const int64_t id = pan[(size_t)b * P + p];
        int idx = 0;
        for (int i = 0; i < n; ++i) idx = (s_keys[i] == id) ? i + 1 : idx;   // last match wins
        indices[(size_t)b * P + p] = idx;
    }
}

// lut[b,k,:] = normalise(emb[b,k,:] - diff * image_emb[b,:]), one wave per row, float32
// arithmetic in the reference's order (sub of a rounded product, then / sqrt(sum of squares))
__global__ __launch_bounds__(64) void k_dve_lut(
    const float* __restrict__ emb, const float* __restrict__ image_emb, float diff, int K, int D,
    float* __restrict__ lut)
{
    const int row = blockIdx.x;                 // b * K + k
    const int b = row / K;
    const float* e = emb + (size_t)row * D;
    const float* g = image_emb + (size_t)b * D;
    float ss = 0.f;
    for (int i = lane_id(); i < D; i += 64) {
        const float v = __fsub_rn(e[i], __fmul_rn(diff, g[i]));
        ss = __fmaf_rn(v, v, ss);
    }
    ss = wave_reduce_sum(ss);
    ss = __shfl(ss, 0);
    const float nrm = __fsqrt_rn(ss);
    for (int i = lane_id(); i < D; i += 64)
        lut[(size_t)row * D + i] = __fdiv_rn(__fsub_rn(e[i], __fmul_rn(diff, g[i])), nrm);
}

// InstanceClearStuffIDs._preprocess (data/preprocessing/instance.py:46-93): id 0 on stuff pixels
template <typename T>
__global__ __launch_bounds__(256) void k_clear_stuff(
    const void* __restrict__ sem, int sem_dtype, T* __restrict__ ins, const uint8_t* __restrict__ is_stuff,
    int NC, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int64_t s = mw_load(sem, sem_dtype, i);
        if (s >= 0 && s < NC && is_stuff[s]) ins[i] = 0;
    }
}

int tg_cap(int max_instances) { return ((max_instances + 1023) / 1024) * 1024; }

int tg_common(const void* sem, int sem_dtype, const void* ins, int ins_dtype, int B, int NC, int P,
              int W, int cap, bool moments, unsigned char* ws, size_t need, int32_t* status,
              hipStream_t stream)
{
    int rc = check_hip(hipMemsetAsync(ws, 0, need, stream));
    if (rc) return rc;
    int gx = (P + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_tg_presence, dim3(gx, B), dim3(256), 0, stream, ins, ins_dtype, P, cap, NC, ws,
                       status);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(k_tg_rank, dim3(B), dim3(1024), 0, stream, ws, cap, NC, status);
    if ((rc = check_launch())) return rc;
    if (moments)
        hipLaunchKernelGGL(k_tg_stats<true>, dim3(gx, B), dim3(256), 0, stream, sem, sem_dtype, ins,
                           ins_dtype, P, W, cap, NC, ws, status);
    else
        hipLaunchKernelGGL(k_tg_stats<false>, dim3(gx, B), dim3(256), 0, stream, sem, sem_dtype, ins,
                           ins_dtype, P, W, cap, NC, ws, status);
    return check_launch();
}

bool tg_bad_dtype(int d) { return d < NMSA_U8 || d > NMSA_I64; }

}  // namespace
}  // namespace nmsa

using namespace nmsa;

extern "C" size_t nmsa_targets_workspace_bytes(int B, int n_classes, int max_instances)
{
    if (B <= 0 || n_classes <= 0 || max_instances <= 0 || max_instances > 4096) return 0;
    return (size_t)B * tg_image_bytes(tg_cap(max_instances), n_classes);
}

extern "C" int nmsa_instance_targets(const void* semantic, int sem_dtype, const void* instance,
                                     int ins_dtype, const uint8_t* is_thing_class,
                                     const uint8_t* is_stuff_class, int B, int n_classes, int H, int W,
                                     int sigma, const float* gauss_lut, int normalized_offset,
                                     int max_instances,
                                     float* center, void* offset, uint8_t* foreground,
                                     uint8_t* center_mask,
                                     int32_t* encoded_ids, int32_t* n_encoded,
                                     int32_t* skipped_ids, int32_t* n_skipped,
                                     int32_t* status, void* workspace, size_t workspace_bytes,
                                     nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!semantic || !instance || !gauss_lut || !center || !offset || !foreground || !status || !workspace)
        return NMSA_ERR_ARG;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || H > 32767 || W > 32767 ||
        (int64_t)H * W > ((int64_t)1 << 30))
        return NMSA_ERR_ARG;
    if (n_classes <= 0 || n_classes > 65536 || sigma <= 0 || sigma > 64) return NMSA_ERR_ARG;
    if (max_instances <= 0 || max_instances > 4096) return NMSA_ERR_ARG;
    if (tg_bad_dtype(sem_dtype) || tg_bad_dtype(ins_dtype)) return NMSA_ERR_ARG;
    const int cap = tg_cap(max_instances);
    const size_t need = nmsa_targets_workspace_bytes(B, n_classes, max_instances);
    if (workspace_bytes < need) return NMSA_ERR_WORKSPACE;
    if ((uintptr_t)workspace % 8) return NMSA_ERR_ARG;
    unsigned char* ws = (unsigned char*)workspace;
    const int P = H * W;
    int rc = tg_common(semantic, sem_dtype, instance, ins_dtype, B, n_classes, P, W, cap, true, ws, need,
                       status, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tg_decide, dim3(B), dim3(1024), 0, stream, ws, cap, n_classes, is_thing_class,
                       encoded_ids, n_encoded, skipped_ids, n_skipped);
    if ((rc = check_launch())) return rc;
    const int radius = 3 * sigma + 1;
    const int lut_n = 2 * radius * radius + 1;
    const size_t lds = (size_t)cap * 2 * sizeof(int) + (lut_n <= TGP_LUT_LDS ? (size_t)lut_n * 4 : 0);
    dim3 grid((P + TGP_PX - 1) / TGP_PX, B);
    if (normalized_offset)
        hipLaunchKernelGGL(k_tg_paint<true>, grid, dim3(TGP_THREADS), lds, stream, semantic, sem_dtype,
                           instance, ins_dtype, is_stuff_class, gauss_lut, lut_n, radius, H, W, cap,
                           n_classes, ws, center, offset, foreground, center_mask);
    else
        hipLaunchKernelGGL(k_tg_paint<false>, grid, dim3(TGP_THREADS), lds, stream, semantic, sem_dtype,
                           instance, ins_dtype, is_stuff_class, gauss_lut, lut_n, radius, H, W, cap,
                           n_classes, ws, center, offset, foreground, center_mask);
    return check_launch();
}

extern "C" int nmsa_panoptic_targets(const void* semantic, int sem_dtype, const void* instance,
                                     int ins_dtype, const uint8_t* is_thing_class,
                                     int B, int n_classes, int H, int W,
                                     int64_t max_instances_per_category, int64_t void_label,
                                     int max_instances, int max_segments,
                                     int64_t* panoptic, int64_t* ids_pan, int64_t* ids_ins, int32_t* n_ids,
                                     int32_t* status, void* workspace, size_t workspace_bytes,
                                     nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!semantic || !instance || !panoptic || !ids_pan || !ids_ins || !n_ids || !status || !workspace)
        return NMSA_ERR_ARG;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || (int64_t)H * W > ((int64_t)1 << 30)) return NMSA_ERR_ARG;
    if (n_classes <= 0 || n_classes > 65536 || max_instances_per_category <= 0 || void_label < 0)
        return NMSA_ERR_ARG;
    if (max_instances <= 0 || max_instances > 4096 || max_segments <= 0) return NMSA_ERR_ARG;
    if (tg_bad_dtype(sem_dtype) || tg_bad_dtype(ins_dtype)) return NMSA_ERR_ARG;
    const int cap = tg_cap(max_instances);
    const size_t need = nmsa_targets_workspace_bytes(B, n_classes, max_instances);
    if (workspace_bytes < need) return NMSA_ERR_WORKSPACE;
    if ((uintptr_t)workspace % 8) return NMSA_ERR_ARG;
    unsigned char* ws = (unsigned char*)workspace;
    const int P = H * W;
    int rc = tg_common(semantic, sem_dtype, instance, ins_dtype, B, n_classes, P, W, cap, false, ws, need,
                       status, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tg_naive_ranks, dim3(B), dim3(1024), 0, stream, ws, cap, n_classes, max_segments,
                       max_instances_per_category, ids_pan, ids_ins, n_ids, status);
    if ((rc = check_launch())) return rc;
    int gx = (P + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_tg_naive_paint, dim3(gx, B), dim3(256), 0, stream, semantic, sem_dtype, instance,
                       ins_dtype, is_thing_class, P, cap, n_classes, max_instances_per_category, void_label,
                       ws, panoptic);
    return check_launch();
}

extern "C" int nmsa_dve_targets(const int64_t* panoptic, const int64_t* keys, const int32_t* n_keys,
                                const float* embeddings, const float* image_embedding, float diff_factor,
                                int B, int K, int D, int H, int W,
                                float* lut, int32_t* indices, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!panoptic || !keys || !n_keys || !indices) return NMSA_ERR_ARG;
    if (B <= 0 || B > 65535 || K <= 0 || K > 4096 || H <= 0 || W <= 0 ||
        (int64_t)H * W > ((int64_t)1 << 30))
        return NMSA_ERR_ARG;
    const int P = H * W;
    int gx = (P + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_dve_indices, dim3(gx, B), dim3(256), (size_t)K * sizeof(int64_t), stream,
                       panoptic, keys, n_keys, K, P, indices);
    int rc = check_launch();
    if (rc) return rc;
    if (lut) {
        if (!embeddings || !image_embedding || D <= 0) return NMSA_ERR_ARG;
        hipLaunchKernelGGL(k_dve_lut, dim3((unsigned)(B * K)), dim3(64), 0, stream, embeddings,
                           image_embedding, diff_factor, K, D, lut);
        rc = check_launch();
    }
    return rc;
}

extern "C" int nmsa_instance_clear_stuff(const void* semantic, int sem_dtype, void* instance, int ins_dtype,
                                         const uint8_t* is_stuff_class, int n_classes, int64_t n_px,
                                         nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!semantic || !instance || !is_stuff_class || n_classes <= 0 || n_px <= 0) return NMSA_ERR_ARG;
    if (tg_bad_dtype(sem_dtype) || tg_bad_dtype(ins_dtype)) return NMSA_ERR_ARG;
    long long blocks = (n_px + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const dim3 grid((unsigned)blocks), block(256);
    switch (ins_dtype) {
        case NMSA_U8: hipLaunchKernelGGL(k_clear_stuff<uint8_t>, grid, block, 0, stream, semantic, sem_dtype, (uint8_t*)instance, is_stuff_class, n_classes, (size_t)n_px); break;
        case NMSA_I16: hipLaunchKernelGGL(k_clear_stuff<int16_t>, grid, block, 0, stream, semantic, sem_dtype, (int16_t*)instance, is_stuff_class, n_classes, (size_t)n_px); break;
        case NMSA_I32: hipLaunchKernelGGL(k_clear_stuff<int32_t>, grid, block, 0, stream, semantic, sem_dtype, (int32_t*)instance, is_stuff_class, n_classes, (size_t)n_px); break;
        default: hipLaunchKernelGGL(k_clear_stuff<int64_t>, grid, block, 0, stream, semantic, sem_dtype, (int64_t*)instance, is_stuff_class, n_classes, (size_t)n_px); break;
    }
    return check_launch();
}
