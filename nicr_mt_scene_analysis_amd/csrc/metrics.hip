// metrics.hip — dense mIoU confusion-matrix and panoptic-quality accumulators.
//
// Replaces the ATen / Python chains of
//   MeanIntersectionOverUnion.update      (reference metric/miou.py:44-56)
//   compare_and_accumulate                (reference metric/pq.py:60-179)
// on the device, so that predictions never travel to the host for validation.
//
// k_confmat      : 17 B/px stream (pred i64 + target i64|u8), LDS-privatised
//                  n x n histogram, one int64 atomic per non-empty bin per block.
// k_pq_count     : the three `torch.unique(return_counts=True)` of pq.py:83-109
//                  as per-image open-addressing hash tables (target ids, pred ids,
//                  target*offset+pred ids).  Wave-level aggregation (__ballot /
//                  __shfl over (target, pred) pairs) + LDS-privatised tables keep
//                  the global atomic traffic at O(#segments) per block.
// k_pq_match     : one workgroup per image: compact + bitonic-sort the
//                  intersection table (ascending id = the reference's dict order),
//                  TP / FN / FP decisions with exact integer areas and the IoU in
//                  fp64, per-class sums accumulated in that order -> the fp64
//                  state is bit-identical to the reference's.
// k_pq_accumulate: state += per-image results, in image order (deterministic).
#include "nmsa_common.hpp"

namespace nmsa {

// status bits written by the kernels (checked by the host wrapper)
constexpr int ST_TABLE_OVERFLOW = 1;     // more segments than the hash tables hold
constexpr int ST_CATEGORY_RANGE = 2;     // category outside [0, num_categories)
constexpr int ST_MISSING_KEY = 4;        // inconsistent ids (reference: KeyError)
constexpr int ST_VALUE_RANGE = 8;        // confmat: bin outside [0, n*n)
constexpr int ST_SENTINEL_KEY = 16;      // an id equals INT64_MIN

__device__ __forceinline__ int64_t load_int_m(const void* p, int dtype, size_t i)
{
    switch (dtype) {
        case NMSA_U8: return ((const uint8_t*)p)[i];
        case NMSA_I16: return ((const int16_t*)p)[i];
        case NMSA_I32: return ((const int32_t*)p)[i];
        default: return ((const int64_t*)p)[i];
    }
}

// Python-style floor division / modulo (pq.py uses `//` and `%` on Python ints)
__device__ __forceinline__ int64_t floordiv64(int64_t a, int64_t b)
{
    int64_t q = a / b;
    const int64_t r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) --q;
    return q;
}
__device__ __forceinline__ int64_t floormod64(int64_t a, int64_t b)
{
    int64_t r = a % b;
    if (r != 0 && ((r < 0) != (b < 0))) r += b;
    return r;
}

// =================================================================================
// a11: confusion matrix
// mode 0: confmat[t, p]++ for every element            (PanopticTaskHelper: with void)
// mode 1: skip t == 0, then confmat[t-1, p]++           (SemanticTaskHelper masking,
//                                                        task_helper/semantic.py:124-128)
// pred_div: preds are raw / pred_div (panoptic id // max_instances, panoptic.py:123)
// =================================================================================
constexpr int CM_LDS_BINS = 24 * 1024;     // 96 KB of u32 (n <= 156)

__global__ __launch_bounds__(256) void k_confmat(
    const void* __restrict__ preds, int pred_dtype, int64_t pred_div,
    const void* __restrict__ target, int target_dtype,
    int64_t n_px, int n, int mode, int use_lds,
    unsigned long long* __restrict__ confmat, int* __restrict__ status)
{
    extern __shared__ uint32_t cm_hist[];
    const int nbins = n * n;
    if (use_lds) {
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) cm_hist[i] = 0;
        __syncthreads();
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t trips = (n_px + stride - 1) / stride;
    bool bad = false;
    for (int64_t k = 0; k < trips; ++k) {
        const int64_t i = (k * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        int key = -1;
        if (i < n_px) {
            int64_t t = load_int_m(target, target_dtype, (size_t)i);
            int64_t p = load_int_m(preds, pred_dtype, (size_t)i);
            if (p < 0) bad = true;                          // bincount rejects negatives
            if (pred_div != 1) p = p / pred_div;           // torch `//` on non-negative ids
            bool skip = false;
            if (mode == 1) { skip = (t == 0); t -= 1; }
            if (!skip) {
                const int64_t bin = t * n + p;              // miou.py:50
                if (t < 0 || p < 0 || bin >= nbins || bin < 0) bad = true;
                else key = (int)bin;
            }
        }
        // coherent label maps: whole wave often hits one bin -> one atomic
        const int first = __shfl(key, __ffsll((long long)__ballot(key >= 0)) - 1);
        const unsigned long long act = __ballot(key >= 0);
        if (act && __ballot(key == first) == act) {
            if (lane_id() == __ffsll((long long)act) - 1) {
                if (use_lds) atomicAdd(&cm_hist[first], (uint32_t)__popcll(act));
                else atomicAdd(&confmat[first], (unsigned long long)__popcll(act));
            }
        } else if (key >= 0) {
            if (use_lds) atomicAdd(&cm_hist[key], 1u);
            else atomicAdd(&confmat[key], 1ull);
        }
    }
    if (bad) atomicOr(status, ST_VALUE_RANGE);
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) {
            const uint32_t v = cm_hist[i];
            if (v) atomicAdd(&confmat[i], (unsigned long long)v);
        }
    }
}

// =================================================================================
// a12: hash tables
// =================================================================================
constexpr int64_t KEY_EMPTY = INT64_MIN;
constexpr int PQ_T_CAP = 2048;      // distinct target ids / image
constexpr int PQ_P_CAP = 2048;      // distinct predicted ids / image
constexpr int PQ_I_CAP = 4096;      // distinct (target, pred) intersections / image
constexpr int PQ_LT = 256, PQ_LP = 256, PQ_LI = 1024;   // LDS-privatised tables per block

__device__ __forceinline__ uint32_t hash64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return (uint32_t)k;
}

// insert-or-add; returns false when no slot was found within `max_probe` probes
__device__ __forceinline__ bool table_add(int64_t* keys, uint32_t* cnts, uint32_t mask,
                                          int64_t key, uint32_t n, int max_probe)
{
    uint32_t slot = hash64((uint64_t)key) & mask;
    for (int probe = 0; probe < max_probe; ++probe) {
        const int64_t cur = *(volatile int64_t*)&keys[slot];
        if (cur == key) { atomicAdd(&cnts[slot], n); return true; }
        if (cur == KEY_EMPTY) {
            const int64_t prev = (int64_t)atomicCAS((unsigned long long*)&keys[slot],
                                                    (unsigned long long)KEY_EMPTY,
                                                    (unsigned long long)key);
            if (prev == KEY_EMPTY || prev == key) { atomicAdd(&cnts[slot], n); return true; }
        }
        slot = (slot + 1) & mask;
    }
    return false;
}

// lookup in a quiescent table: slot index or -1
__device__ __forceinline__ int table_find(const int64_t* keys, uint32_t mask, int64_t key)
{
    uint32_t slot = hash64((uint64_t)key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const int64_t cur = keys[slot];
        if (cur == key) return (int)slot;
        if (cur == KEY_EMPTY) return -1;
        slot = (slot + 1) & mask;
    }
    return -1;
}

struct PqTables {            // per-image views into the workspace
    int64_t* keyT; uint32_t* cntT;
    int64_t* keyP; uint32_t* cntP;
    int64_t* keyI; uint32_t* cntI;
};

__host__ __device__ inline size_t pq_image_bytes()
{
    return (size_t)(PQ_T_CAP + PQ_P_CAP + PQ_I_CAP) * (sizeof(int64_t) + sizeof(uint32_t));
}

__device__ __forceinline__ PqTables pq_tables(unsigned char* ws, int b)
{
    unsigned char* base = ws + (size_t)b * pq_image_bytes();
    PqTables t;
    t.keyT = (int64_t*)base;
    t.keyP = t.keyT + PQ_T_CAP;
    t.keyI = t.keyP + PQ_P_CAP;
    t.cntT = (uint32_t*)(t.keyI + PQ_I_CAP);
    t.cntP = t.cntT + PQ_T_CAP;
    t.cntI = t.cntP + PQ_P_CAP;
    return t;
}

__global__ __launch_bounds__(256) void k_pq_init(unsigned char* __restrict__ ws, int B)
{
    const int b = blockIdx.y;
    PqTables t = pq_tables(ws, b);
    const int total = PQ_T_CAP + PQ_P_CAP + PQ_I_CAP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        t.keyT[i] = KEY_EMPTY;        // the three key arrays are contiguous
        t.cntT[i] = 0;                // ... and so are the three count arrays
    }
}

__global__ __launch_bounds__(256) void k_pq_count(
    const int64_t* __restrict__ pred, const int64_t* __restrict__ target,
    int P, int64_t offset, int px_per_block,
    unsigned char* __restrict__ ws, int* __restrict__ status)
{
    __shared__ int64_t lkT[PQ_LT], lkP[PQ_LP], lkI[PQ_LI];
    __shared__ uint32_t lcT[PQ_LT], lcP[PQ_LP], lcI[PQ_LI];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < PQ_LI; i += blockDim.x) {
        lkI[i] = KEY_EMPTY; lcI[i] = 0;
        if (i < PQ_LT) { lkT[i] = KEY_EMPTY; lcT[i] = 0; lkP[i] = KEY_EMPTY; lcP[i] = 0; }
    }
    __syncthreads();
    PqTables g = pq_tables(ws, b);
    const int64_t* pr = pred + (size_t)b * P;
    const int64_t* tg = target + (size_t)b * P;
    const int start = blockIdx.x * px_per_block;
    const int end = min(start + px_per_block, P);
    int st = 0;

    auto add3 = [&](int64_t t, int64_t p, uint32_t cnt) {
        // intersection id with torch's int64 wrap-around arithmetic (pq.py:104)
        const int64_t iid = (int64_t)((uint64_t)t * (uint64_t)offset + (uint64_t)p);
        if (t == KEY_EMPTY || p == KEY_EMPTY || iid == KEY_EMPTY) { st |= ST_SENTINEL_KEY; return; }
        if (!table_add(lkT, lcT, PQ_LT - 1, t, cnt, 16) &&
            !table_add(g.keyT, g.cntT, PQ_T_CAP - 1, t, cnt, PQ_T_CAP)) st |= ST_TABLE_OVERFLOW;
        if (!table_add(lkP, lcP, PQ_LP - 1, p, cnt, 16) &&
            !table_add(g.keyP, g.cntP, PQ_P_CAP - 1, p, cnt, PQ_P_CAP)) st |= ST_TABLE_OVERFLOW;
        if (!table_add(lkI, lcI, PQ_LI - 1, iid, cnt, 32) &&
            !table_add(g.keyI, g.cntI, PQ_I_CAP - 1, iid, cnt, PQ_I_CAP)) st |= ST_TABLE_OVERFLOW;
    };

    const int span = end - start;
    const int trips = (span + (int)blockDim.x - 1) / (int)blockDim.x;
    for (int k = 0; k < trips; ++k) {
        const int i = start + k * blockDim.x + threadIdx.x;
        const bool valid = i < end;
        const int64_t t = valid ? tg[i] : 0;
        const int64_t p = valid ? pr[i] : 0;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int64_t lt = __shfl(t, leader);
            const int64_t lp = __shfl(p, leader);
            const unsigned long long same = __ballot(valid && t == lt && p == lp) & todo;
            if (lane_id() == leader) add3(lt, lp, (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }
    __syncthreads();
    // flush the block-private tables
    for (int i = threadIdx.x; i < PQ_LI; i += blockDim.x) {
        if (lkI[i] != KEY_EMPTY &&
            !table_add(g.keyI, g.cntI, PQ_I_CAP - 1, lkI[i], lcI[i], PQ_I_CAP)) st |= ST_TABLE_OVERFLOW;
        if (i < PQ_LT) {
            if (lkT[i] != KEY_EMPTY &&
                !table_add(g.keyT, g.cntT, PQ_T_CAP - 1, lkT[i], lcT[i], PQ_T_CAP)) st |= ST_TABLE_OVERFLOW;
            if (lkP[i] != KEY_EMPTY &&
                !table_add(g.keyP, g.cntP, PQ_P_CAP - 1, lkP[i], lcP[i], PQ_P_CAP)) st |= ST_TABLE_OVERFLOW;
        }
    }
    if (st) atomicOr(status, st);
}

// ---- block helpers ------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan_i(int v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane_id() >= o) v += t;
    }
    return v;
}

__device__ __forceinline__ int block_incl_scan_i(int v, int* scratch, int* total)
{
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int incl = wave_incl_scan_i(v);
    __syncthreads();
    if (lane_id() == 63) scratch[w] = incl;
    __syncthreads();
    if (w == 0) {
        int s = (lane_id() < nw) ? scratch[lane_id()] : 0;
        s = wave_incl_scan_i(s);
        if (lane_id() < nw) scratch[lane_id()] = s;
    }
    __syncthreads();
    *total = scratch[nw - 1];
    return incl + ((w == 0) ? 0 : scratch[w - 1]);
}

constexpr int PQ_MATCH_THREADS = 1024;
constexpr int PQ_MAX_CATEGORIES = 1024;

__global__ __launch_bounds__(PQ_MATCH_THREADS) void k_pq_match(
    unsigned char* __restrict__ ws, int num_categories, int64_t ignored_label,
    int64_t max_inst, int64_t offset, int64_t void_segment_id,
    double* __restrict__ img_state /* [B,4,num_categories] */,
    int64_t* __restrict__ matches /* [B,match_cap,2] or null */, int match_cap,
    int32_t* __restrict__ n_matches, int* __restrict__ status)
{
    __shared__ int64_t sKey[PQ_I_CAP];
    __shared__ uint32_t sCnt[PQ_I_CAP];
    __shared__ double sIou[PQ_I_CAP];
    __shared__ int16_t sCat[PQ_I_CAP];          // category of a TP entry, -1 otherwise
    __shared__ uint8_t fT[PQ_T_CAP], fP[PQ_P_CAP];
    __shared__ int fnI[PQ_MAX_CATEGORIES], fpI[PQ_MAX_CATEGORIES];
    __shared__ int64_t ignKeys[64];
    __shared__ int nIgn;
    __shared__ int scratch[32];

    const int b = blockIdx.x, tid = threadIdx.x;
    PqTables g = pq_tables(ws, b);
    int st = 0;

    for (int i = tid; i < PQ_T_CAP; i += PQ_MATCH_THREADS) { fT[i] = 0; fP[i] = 0; }
    for (int i = tid; i < num_categories; i += PQ_MATCH_THREADS) { fnI[i] = 0; fpI[i] = 0; }
    if (tid == 0) nIgn = 0;

    // ---- 1. compact the intersection table --------------------------------------------
    const int per = PQ_I_CAP / PQ_MATCH_THREADS;      // 4 slots / thread
    int mine = 0;
    for (int j = 0; j < per; ++j) mine += (g.keyI[tid * per + j] != KEY_EMPTY);
    int nI;
    int pos = block_incl_scan_i(mine, scratch, &nI) - mine;
    for (int j = 0; j < per; ++j) {
        const int s = tid * per + j;
        if (g.keyI[s] != KEY_EMPTY) { sKey[pos] = g.keyI[s]; sCnt[pos] = g.cntI[s]; ++pos; }
    }
    int n2 = 1;
    while (n2 < nI) n2 <<= 1;
    for (int i = nI + tid; i < n2; i += PQ_MATCH_THREADS) { sKey[i] = INT64_MAX; sCnt[i] = 0; }
    __syncthreads();

    // ---- 2. bitonic sort, ascending id (= reference dict iteration order) ---------------
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += PQ_MATCH_THREADS) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool up = (i & k) == 0;
                    const int64_t a = sKey[i], c = sKey[ixj];
                    if ((a > c) == up) {
                        sKey[i] = c; sKey[ixj] = a;
                        const uint32_t ca = sCnt[i]; sCnt[i] = sCnt[ixj]; sCnt[ixj] = ca;
                    }
                }
            }
            __syncthreads();
        }
    }

    // ignored segments: target ids whose category is the ignored label (pq.py:89-93)
    for (int s = tid; s < PQ_T_CAP; s += PQ_MATCH_THREADS) {
        const int64_t k = g.keyT[s];
        if (k != KEY_EMPTY && floordiv64(k, max_inst) == ignored_label) {
            const int at = atomicAdd(&nIgn, 1);
            if (at < 64) ignKeys[at] = k;
        }
    }

    // ---- 3. TP decision per intersection (pq.py:119-153) ----------------------------------
    for (int e = tid; e < nI; e += PQ_MATCH_THREADS) {
        int16_t cat = -1;
        double iou = 0.0;
        const int64_t iid = sKey[e];
        if (iid != void_segment_id) {
            const int64_t gt = floordiv64(iid, offset);
            const int64_t pr = floormod64(iid, offset);
            const int64_t gcat = floordiv64(gt, max_inst);
            const int64_t pcat = floordiv64(pr, max_inst);
            if (gcat == pcat) {
                // prediction_void_overlap (pq.py:35-44): binary search in the sorted list
                const int64_t vid = (int64_t)((uint64_t)void_segment_id * (uint64_t)offset + (uint64_t)pr);
                int64_t r = 0;
                {
                    int lo = 0, hi = nI - 1;
                    while (lo <= hi) {
                        const int mid = (lo + hi) >> 1;
                        if (sKey[mid] == vid) { r = sCnt[mid]; break; }
                        if (sKey[mid] < vid) lo = mid + 1; else hi = mid - 1;
                    }
                }
                const int sT = table_find(g.keyT, PQ_T_CAP - 1, gt);
                const int sP = table_find(g.keyP, PQ_P_CAP - 1, pr);
                if (sT < 0 || sP < 0) st |= ST_MISSING_KEY;
                else {
                    const int64_t ia = sCnt[e];
                    const int64_t uni = (int64_t)g.cntT[sT] + (int64_t)g.cntP[sP] - ia - r;   // :143
                    iou = (double)ia / (double)uni;                                             // :145
                    if (iou > 0.5) {
                        if (gcat < 0 || gcat >= num_categories) st |= ST_CATEGORY_RANGE;
                        else { cat = (int16_t)gcat; fT[sT] = 1; fP[sP] = 1; }
                    }
                }
            }
        }
        sCat[e] = cat;
        sIou[e] = iou;
    }
    __syncthreads();

    // ---- 4. per-class TP / IoU sums in ascending-id order (bit-exact fp64) -----------------
    double* out = img_state + (size_t)b * 4 * num_categories;
    for (int c = tid; c < num_categories; c += PQ_MATCH_THREADS) {
        double iou = 0.0, tp = 0.0;
        for (int e = 0; e < nI; ++e)
            if (sCat[e] == c) { tp += 1.0; iou += sIou[e]; }
        out[0 * num_categories + c] = iou;
        out[1 * num_categories + c] = tp;
    }

    // ---- 5. false negatives (pq.py:155-163) -----------------------------------------------
    for (int s = tid; s < PQ_T_CAP; s += PQ_MATCH_THREADS) {
        const int64_t k = g.keyT[s];
        if (k == KEY_EMPTY || fT[s]) continue;
        const int64_t cat = floordiv64(k, max_inst);
        if (cat == ignored_label) continue;
        if (cat < 0 || cat >= num_categories) { st |= ST_CATEGORY_RANGE; continue; }
        atomicAdd(&fnI[cat], 1);
    }
    // ---- 6. false positives (pq.py:165-177) -------------------------------------------------
    const int n_ign = nIgn;
    if (n_ign > 64) st |= ST_TABLE_OVERFLOW;
    for (int s = tid; s < PQ_P_CAP; s += PQ_MATCH_THREADS) {
        const int64_t k = g.keyP[s];
        if (k == KEY_EMPTY || fP[s]) continue;
        int64_t pio = 0;                                  // prediction_ignored_overlap :47-57
        for (int q = 0; q < min(n_ign, 64); ++q) {
            const int64_t id = (int64_t)((uint64_t)ignKeys[q] * (uint64_t)offset + (uint64_t)k);
            int lo = 0, hi = nI - 1;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1;
                if (sKey[mid] == id) { pio += sCnt[mid]; break; }
                if (sKey[mid] < id) lo = mid + 1; else hi = mid - 1;
            }
        }
        if ((double)pio / (double)g.cntP[s] > 0.5) continue;
        const int64_t cat = floordiv64(k, max_inst);
        if (cat < 0 || cat >= num_categories) { st |= ST_CATEGORY_RANGE; continue; }
        atomicAdd(&fpI[cat], 1);
    }
    __syncthreads();
    for (int c = tid; c < num_categories; c += PQ_MATCH_THREADS) {
        out[2 * num_categories + c] = (double)fnI[c];
        out[3 * num_categories + c] = (double)fpI[c];
    }

    // ---- 7. matched (gt, pred) pairs in id order (for the orientation MAE) -------------------
    {
        const int chunk = (nI + PQ_MATCH_THREADS - 1) / PQ_MATCH_THREADS;
        const int e0 = min(tid * chunk, nI), e1 = min(e0 + chunk, nI);
        int m = 0;
        for (int e = e0; e < e1; ++e) m += (sCat[e] >= 0);
        int total;
        int at = block_incl_scan_i(m, scratch, &total) - m;
        if (matches) {
            for (int e = e0; e < e1; ++e) {
                if (sCat[e] < 0) continue;
                if (at < match_cap) {
                    matches[((size_t)b * match_cap + at) * 2 + 0] = floordiv64(sKey[e], offset);
                    matches[((size_t)b * match_cap + at) * 2 + 1] = floormod64(sKey[e], offset);
                }
                ++at;
            }
        }
        if (tid == 0 && n_matches) n_matches[b] = total;
    }
    if (st) atomicOr(status, st);
}

__global__ void k_pq_accumulate(const double* __restrict__ img_state, int B, int num_categories,
                                double* __restrict__ iou, double* __restrict__ tp,
                                double* __restrict__ fn, double* __restrict__ fp)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= num_categories) return;
    double a0 = iou[c], a1 = tp[c], a2 = fn[c], a3 = fp[c];
    for (int b = 0; b < B; ++b) {                       // image order, like pq.py:291-296
        const double* s = img_state + (size_t)b * 4 * num_categories;
        a0 += s[0 * num_categories + c];
        a1 += s[1 * num_categories + c];
        a2 += s[2 * num_categories + c];
        a3 += s[3 * num_categories + c];
    }
    iou[c] = a0; tp[c] = a1; fn[c] = a2; fp[c] = a3;
}

}  // namespace nmsa

using namespace nmsa;

extern "C" int nmsa_confmat_update(const void* preds, int pred_dtype, int64_t pred_div,
                                   const void* target, int target_dtype,
                                   int64_t n_px, int n_classes, int mode,
                                   int64_t* confmat, int32_t* status, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!preds || !target || !confmat || !status) return NMSA_ERR_ARG;
    if (n_px < 0 || n_classes <= 0 || n_classes > 46340 || pred_div <= 0) return NMSA_ERR_ARG;
    if (pred_dtype < NMSA_U8 || pred_dtype > NMSA_I64 || target_dtype < NMSA_U8 || target_dtype > NMSA_I64)
        return NMSA_ERR_ARG;
    if (mode != 0 && mode != 1) return NMSA_ERR_ARG;
    if (n_px == 0) return NMSA_OK;
    const int nbins = n_classes * n_classes;
    const int use_lds = nbins <= CM_LDS_BINS;
    const size_t lds = use_lds ? (size_t)nbins * 4 : 0;
    // enough blocks to fill the chip, few enough that the per-block flush stays small
    int64_t blocks = (n_px + 256 * 16 - 1) / (256 * 16);
    const int64_t cap = (lds > 40 * 1024) ? 256 : 2048;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_confmat, dim3((unsigned)blocks), dim3(256), lds, stream, preds, pred_dtype,
                       pred_div, target, target_dtype, n_px, n_classes, mode, use_lds,
                       (unsigned long long*)confmat, status);
    return check_launch();
}

extern "C" size_t nmsa_pq_workspace_bytes(int B, int num_categories)
{
    if (B <= 0 || num_categories <= 0) return 0;
    return (size_t)B * pq_image_bytes() + (size_t)B * 4 * num_categories * sizeof(double);
}

extern "C" int nmsa_pq_update(const int64_t* pred, const int64_t* target, int B, int H, int W,
                              int num_categories, int64_t ignored_label,
                              int64_t max_instances_per_category, int64_t offset,
                              int64_t void_segment_id,
                              double* iou_per_class, double* tp_per_class,
                              double* fn_per_class, double* fp_per_class,
                              int64_t* matches, int match_capacity, int32_t* n_matches,
                              int32_t* status, void* workspace, size_t workspace_bytes,
                              nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !target || !iou_per_class || !tp_per_class || !fn_per_class || !fp_per_class ||
        !status || !workspace)
        return NMSA_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || (int64_t)H * W > ((int64_t)1 << 30) || B > 65535) return NMSA_ERR_ARG;
    if (num_categories <= 0 || num_categories > PQ_MAX_CATEGORIES) return NMSA_ERR_ARG;
    if (max_instances_per_category <= 0 || offset <= 0) return NMSA_ERR_ARG;
    if (matches && (match_capacity <= 0 || !n_matches)) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_pq_workspace_bytes(B, num_categories)) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    unsigned char* ws = (unsigned char*)workspace;
    double* img_state = (double*)(ws + (size_t)B * pq_image_bytes());

    hipLaunchKernelGGL(k_pq_init, dim3(8, B), dim3(256), 0, stream, ws, B);
    int rc = check_launch();
    if (rc) return rc;
    const int px_per_block = 4096;
    hipLaunchKernelGGL(k_pq_count, dim3((P + px_per_block - 1) / px_per_block, B), dim3(256), 0, stream,
                       pred, target, P, offset, px_per_block, ws, status);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(k_pq_match, dim3(B), dim3(PQ_MATCH_THREADS), 0, stream, ws, num_categories,
                       ignored_label, max_instances_per_category, offset, void_segment_id,
                       img_state, matches, match_capacity, n_matches, status);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(k_pq_accumulate, dim3((num_categories + 63) / 64), dim3(64), 0, stream,
                       img_state, B, num_categories, iou_per_class, tp_per_class, fn_per_class,
                       fp_per_class);
    return check_launch();
}
