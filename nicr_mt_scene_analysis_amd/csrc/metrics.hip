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
#include <stdlib.h>
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
constexpr int CM_UNROLL = 4;               // 2-element loads in flight per lane and map

// two consecutive elements starting at i (i even) as int64; VEC = one 2/4/8/16-B load
template <int DT, bool VEC>
__device__ __forceinline__ void load2(const void* p, int64_t i, int64_t n, int64_t& a, int64_t& b)
{
    if (VEC && i + 2 <= n) {
        if (DT == NMSA_U8) {
            const uint16_t v = *(const uint16_t*)((const uint8_t*)p + i);
            a = v & 0xFF; b = v >> 8;
        } else if (DT == NMSA_I16) {
            const uint32_t v = *(const uint32_t*)((const int16_t*)p + i);
            a = (int16_t)(v & 0xFFFF); b = (int16_t)(v >> 16);
        } else if (DT == NMSA_I32) {
            const int2 v = *(const int2*)((const int32_t*)p + i);
            a = v.x; b = v.y;
        } else {
            const longlong2 v = *(const longlong2*)((const int64_t*)p + i);
            a = v.x; b = v.y;
        }
    } else {
        a = (i < n) ? load_int_m(p, DT, (size_t)i) : 0;
        b = (i + 1 < n) ? load_int_m(p, DT, (size_t)(i + 1)) : 0;
    }
}

// Lanes of a wave hold consecutive element pairs (coalesced loads: every wave-instruction
// reads one contiguous run).  Label maps are coherent along a row, so the wave is cut into
// RUNS of equal bins and every run head issues ONE LDS atomic with the run length; with
// incoherent labels every lane is a head and the atomics simply run in parallel.
template <int PD, int TD, bool VEC>
__global__ __launch_bounds__(1024) void k_confmat(
    const void* __restrict__ preds, int64_t pred_div, const void* __restrict__ target,
    int64_t n_px, int n, int mode, int use_lds,
    unsigned long long* __restrict__ confmat, uint32_t* __restrict__ slab,
    int* __restrict__ status)
{
    extern __shared__ uint32_t cm_hist[];
    const int nbins = n * n;
    if (use_lds) {
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) cm_hist[i] = 0;
        __syncthreads();
    }
    bool bad = false;
    auto bin_of = [&](int64_t t, int64_t p, bool valid) -> int {
        if (!valid) return -1;
        if (p < 0) bad = true;                              // bincount rejects negatives
        if (pred_div != 1) p = p / pred_div;               // torch `//` on non-negative ids
        if (mode == 1) { if (t == 0) return -1; t -= 1; }
        const int64_t bin = t * n + p;                      // miou.py:50
        if (t < 0 || p < 0 || bin >= nbins || bin < 0) { bad = true; return -1; }
        return (int)bin;
    };
    auto wave_runs = [&](int key, uint32_t weight) {
        const int prev = __shfl_up(key, 1);
        const bool head = key >= 0 && (lane_id() == 0 || prev != key);
        const unsigned long long heads = __ballot(head);
        const unsigned long long gaps = __ballot(key < 0);
        if (head) {
            const int l = lane_id();
            const unsigned long long stop = (heads | gaps) & ~((2ull << l) - 1ull);
            const int nxt = stop ? (__ffsll((long long)stop) - 1) : 64;
            const uint32_t len = weight * (uint32_t)(nxt - l);
            if (use_lds) atomicAdd(&cm_hist[key], len);
            else atomicAdd(&confmat[key], (unsigned long long)len);
        }
    };

    const int64_t tile = (int64_t)blockDim.x * 2 * CM_UNROLL;
    const int64_t n_tiles = (n_px + tile - 1) / tile;
    for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        int64_t ta[CM_UNROLL], tb[CM_UNROLL], pa[CM_UNROLL], pb[CM_UNROLL];
#pragma unroll
        for (int u = 0; u < CM_UNROLL; ++u) {
            const int64_t i = tl * tile + ((int64_t)u * blockDim.x + threadIdx.x) * 2;
            load2<TD, VEC>(target, i, n_px, ta[u], tb[u]);
            load2<PD, VEC>(preds, i, n_px, pa[u], pb[u]);
        }
#pragma unroll
        for (int u = 0; u < CM_UNROLL; ++u) {
            const int64_t i = tl * tile + ((int64_t)u * blockDim.x + threadIdx.x) * 2;
            const int k0 = bin_of(ta[u], pa[u], i < n_px);
            const int k1 = bin_of(tb[u], pb[u], i + 1 < n_px);
            if (__all(k0 == k1)) wave_runs(k0, 2u);
            else { wave_runs(k0, 1u); wave_runs(k1, 1u); }
        }
    }
    if (bad) atomicOr(status, ST_VALUE_RANGE);
    if (use_lds) {
        // no atomics: every block stores its private histogram as one coalesced slab;
        // k_confmat_reduce sums the slabs (with random labels every bin of every block
        // is non-zero, and thousands of same-address atomics would dominate the kernel)
        __syncthreads();
        uint32_t* mine = slab + (size_t)blockIdx.x * nbins;
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) mine[i] = cm_hist[i];
    }
}

// Both maps uint8 and no division of the prediction (SemanticTaskHelper's mIoU on the class map's
// uint8 twin, task_helper/semantic.py:124-128): 16 pixels per lane in ONE 16-byte load per map,
// 32-bit bins, the runs of equal bins inside a lane merged in registers before they touch LDS, a
// wave on a single bin sends one lane.  (The general kernel spends 64-bit arithmetic, a division
// test and two ballots per pixel pair on inputs that need none of it: 27.8 us for 20 MB.)
__global__ __launch_bounds__(256) void k_confmat_u8(
    const uint8_t* __restrict__ preds, const uint8_t* __restrict__ target, int64_t n_px, int n, int mode,
    uint32_t* __restrict__ slab, int* __restrict__ status)
{
    extern __shared__ uint32_t cm_hist[];
    const uint32_t nbins = (uint32_t)(n * n);
    for (uint32_t i = threadIdx.x; i < nbins; i += blockDim.x) cm_hist[i] = 0;
    __syncthreads();
    bool bad = false;
    // miou.py:50 (bin = t * n + p); mode 1: void targets skipped, classes shifted down
    auto bin_of = [&](uint32_t t, uint32_t p) -> uint32_t {
        if (mode == 1) { if (t == 0) return 0xFFFFFFFFu; t -= 1; }
        const uint32_t bin = t * (uint32_t)n + p;
        if (bin >= nbins) { bad = true; return 0xFFFFFFFFu; }
        return bin;
    };
    const int64_t tile = (int64_t)blockDim.x * 16;
    const int64_t n_tiles = (n_px + tile - 1) / tile;
    for (int64_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
        const int64_t i = tl * tile + (int64_t)threadIdx.x * 16;
        if (i + 16 <= n_px) {
            const uint4 tv = *(const uint4*)(target + i), pv = *(const uint4*)(preds + i);
            const uint32_t tw[4] = {tv.x, tv.y, tv.z, tv.w}, pw[4] = {pv.x, pv.y, pv.z, pv.w};
            const bool flat = tv.x == (tv.x & 0xFFu) * 0x01010101u && tv.x == tv.y && tv.x == tv.z && tv.x == tv.w &&
                              pv.x == (pv.x & 0xFFu) * 0x01010101u && pv.x == pv.y && pv.x == pv.z && pv.x == pv.w;
            const uint32_t b0 = bin_of(tv.x & 0xFFu, pv.x & 0xFFu);
            const uint32_t bf = (uint32_t)__builtin_amdgcn_readfirstlane((int)b0);
            if (__all(flat && b0 == bf)) {                            // the wave sits on one bin
                // (the lanes in here are those whose 16 pixels exist: a prefix of the wave in the
                // image's last tile — lane 0 is one of them and speaks for all that are)
                const uint32_t lanes = (uint32_t)__popcll(__ballot(true));
                if (lane_id() == 0 && bf != 0xFFFFFFFFu) atomicAdd(&cm_hist[bf], 16u * lanes);
                continue;
            }
            uint32_t prev = 0xFFFFFFFFu, run = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t bin = bin_of((tw[j >> 2] >> (8 * (j & 3))) & 0xFFu, (pw[j >> 2] >> (8 * (j & 3))) & 0xFFu);
                if (bin != prev) {
                    if (prev != 0xFFFFFFFFu) atomicAdd(&cm_hist[prev], run);
                    prev = bin; run = 0;
                }
                ++run;
            }
            if (prev != 0xFFFFFFFFu) atomicAdd(&cm_hist[prev], run);
        } else {
            for (int64_t k = i; k < n_px && k < i + 16; ++k) {
                const uint32_t bin = bin_of(target[k], preds[k]);
                if (bin != 0xFFFFFFFFu) atomicAdd(&cm_hist[bin], 1u);
            }
        }
    }
    if (bad) atomicOr(status, ST_VALUE_RANGE);
    __syncthreads();
    uint32_t* mine = slab + (size_t)blockIdx.x * nbins;
    for (uint32_t i = threadIdx.x; i < nbins; i += blockDim.x) mine[i] = cm_hist[i];
}

constexpr int CM_REDUCE_GROUPS = 16;

__global__ __launch_bounds__(256) void k_confmat_reduce(
    const uint32_t* __restrict__ slab, int n_slabs, int nbins,
    unsigned long long* __restrict__ confmat)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbins) return;
    unsigned long long acc = 0;
    int k = blockIdx.y;
    for (; k + 7 * CM_REDUCE_GROUPS < n_slabs; k += 8 * CM_REDUCE_GROUPS) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = slab[(size_t)(k + u * CM_REDUCE_GROUPS) * nbins + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; k < n_slabs; k += CM_REDUCE_GROUPS) acc += slab[(size_t)k * nbins + i];
    if (acc) atomicAdd(&confmat[i], acc);
}

// =================================================================================
// a12: panoptic quality
// The hot kernel keeps ONE table per image: counts per intersection id
// target*offset+pred (pq.py:104-109).  Target / prediction segment areas
// (pq.py:83-84) are the marginals of that table and are rebuilt per image in
// k_pq_match — valid whenever every id decodes uniquely (0 <= pred < offset,
// target >= 0); otherwise the reference itself fails (KeyError) or silently
// mixes segments, and ST_MISSING_KEY is raised here.
// =================================================================================
constexpr int64_t KEY_EMPTY = INT64_MIN;
constexpr int PQ_T_CAP = 2048;      // distinct target ids / image      (LDS, k_pq_match)
constexpr int PQ_P_CAP = 2048;      // distinct predicted ids / image   (LDS, k_pq_match)
constexpr int PQ_I_CAP_MIN = 4096;  // slots of the per-image intersection table: a power of two
constexpr int PQ_I_CAP_MAX = 131072; //   chosen from the image size (pq_i_cap), kept <= half full
constexpr int PQ_LI = 1024;         // LDS-privatised intersection table per block

// Slots of the per-image (target, pred) intersection table.  Distinct intersections grow with
// the total segment boundary length, i.e. with the image size: one slot per 24 px, rounded up
// to a power of two (640x480 -> 16384, 1024x768 -> 32768; blobby synthetic maps with 150
// classes hold ~8500 intersections at 1024x768, real label maps a few hundred).  More than
// cap/2 distinct intersections in one image raise ST_TABLE_OVERFLOW.
__host__ __device__ inline int pq_i_cap(int64_t P)
{
    int cap = PQ_I_CAP_MIN;
    while (cap < PQ_I_CAP_MAX && (int64_t)cap * 24 < P) cap <<= 1;
    return cap;
}

// cheap id hash: one 32-bit multiply per half (64-bit multiplies are quarter rate
// and dominated the first version of k_pq_count)
__device__ __forceinline__ uint32_t hash_id(int64_t k)
{
    const uint32_t lo = (uint32_t)k, hi = (uint32_t)((uint64_t)k >> 32);
    uint32_t h = lo * 0x9E3779B1u;
    h ^= (hi + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= h >> 15;
    return h;
}

// insert-or-add; returns false when no slot was found within `max_probe` probes
// (`#pragma unroll 1`: with the probe counts constant at the call sites the compiler unrolled these
// loops completely — k_pq_count was 37 900 instructions with 5 400 SGPR spills to lane registers,
// now 5 700 and 14; the hot path, a hit in the home slot, never enters them)
__device__ __forceinline__ bool table_add(int64_t* keys, uint32_t* cnts, uint32_t mask,
                                          int64_t key, uint32_t n, int max_probe)
{
    uint32_t slot = hash_id(key) & mask;
#pragma unroll 1
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

constexpr int PQ_LIST_STRIDE = 64;      // ints between the images' list counters: same-line atomics serialise (~12 ns each)

// the same on the image's GLOBAL table: the lane that claims an empty slot also appends the slot
// to the image's entry list, so that k_pq_match starts from the few hundred intersections of the
// image instead of compacting all `cap` slots (16384 at 640x480: 196 KB of dependent L2 reads)
__device__ __forceinline__ bool table_add_listed(int64_t* keys, uint32_t* cnts, uint32_t mask,
                                                 int64_t key, uint32_t n, int max_probe,
                                                 int* list_n, uint32_t* list_slots, int list_cap)
{
    uint32_t slot = hash_id(key) & mask;
#pragma unroll 1
    for (int probe = 0; probe < max_probe; ++probe) {
        const int64_t cur = *(volatile int64_t*)&keys[slot];
        if (cur == key) { atomicAdd(&cnts[slot], n); return true; }
        if (cur == KEY_EMPTY) {
            const int64_t prev = (int64_t)atomicCAS((unsigned long long*)&keys[slot],
                                                    (unsigned long long)KEY_EMPTY,
                                                    (unsigned long long)key);
            if (prev == KEY_EMPTY) {
                const int at = atomicAdd(list_n, 1);
                if (at < list_cap) list_slots[at] = slot;
            }
            if (prev == KEY_EMPTY || prev == key) { atomicAdd(&cnts[slot], n); return true; }
        }
        slot = (slot + 1) & mask;
    }
    return false;
}

// lookup in a quiescent table: slot index or -1
__device__ __forceinline__ int table_find(const int64_t* keys, uint32_t mask, int64_t key)
{
    uint32_t slot = hash_id(key) & mask;
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const int64_t cur = keys[slot];
        if (cur == key) return (int)slot;
        if (cur == KEY_EMPTY) return -1;
        slot = (slot + 1) & mask;
    }
    return -1;
}

// per image: table keys i64[cap] | table counts u32[cap] | the compacted entry list of
// k_pq_match: keys i64[cap/2] | counts u32[cap/2] | slots u32[cap/2]
__host__ __device__ inline size_t pq_image_bytes(int cap)
{
    return (size_t)cap * (sizeof(int64_t) + sizeof(uint32_t)) +
           (size_t)(cap / 2) * (sizeof(int64_t) + 2 * sizeof(uint32_t));
}
__device__ __forceinline__ int64_t* pq_keys(unsigned char* ws, int b, int cap)
{
    return (int64_t*)(ws + (size_t)b * pq_image_bytes(cap));
}
__device__ __forceinline__ uint32_t* pq_cnts(unsigned char* ws, int b, int cap)
{
    return (uint32_t*)(pq_keys(ws, b, cap) + cap);
}
__device__ __forceinline__ int64_t* pq_list_keys(unsigned char* ws, int b, int cap)
{
    return (int64_t*)(pq_cnts(ws, b, cap) + cap);
}

__global__ __launch_bounds__(256) void k_pq_init(unsigned char* __restrict__ ws, int cap,
                                                 int* __restrict__ list_n)
{
    const int b = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) list_n[b * PQ_LIST_STRIDE] = 0;
    int64_t* k = pq_keys(ws, b, cap);
    uint32_t* c = pq_cnts(ws, b, cap);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x) {
        k[i] = KEY_EMPTY;
        c[i] = 0;
    }
}

constexpr int PQ_UNROLL = 4;      // 16-B loads in flight per lane and map (2 px each)

typedef long long i64x2_t __attribute__((ext_vector_type(2)));
// both maps are read exactly once: streaming (non-temporal) 16-B loads
__device__ __forceinline__ longlong2 ld_i64x2_stream(const int64_t* p)
{
    const i64x2_t v = __builtin_nontemporal_load((const i64x2_t*)p);
    return make_longlong2(v.x, v.y);
}

// WITH_CM: the same pass over the prediction also feeds the mIoU confusion matrix of
// task_helper/panoptic.py:123-126 (confmat[target_sem, pred // pred_div]++): the i64 prediction
// map is read ONCE for both metrics.  Per-block LDS histogram -> slab, summed by
// k_confmat_reduce exactly like the stand-alone k_confmat.
// POW2: `offset` and the class divisor are powers of two (the reference's 256^3 and 2^16,
// task_helper/panoptic.py:57-63): shifts, and every per-pixel range test as branch-free selects —
// the early returns of the generic form are eight divergent branch diamonds per load, with an
// inlined 64-bit division in each.
// The prediction as the PARTS the merge painted its int64 map from — semantic class u8, instance id
// u8 and the per-image table pan_of_inst (k_paint2's rule, panoptic.hip): 2 B/px instead of 8, read
// by k_pq_count_parts below.  The map itself is still written for the API.
struct PqPredParts {
    const uint8_t* sem;                // [B,P] class index 0..C-1
    const uint8_t* inst;               // [B,P] instance id 0..255
    const int64_t* pan_of_inst;        // [B,256]
    const uint8_t* is_thing;           // [C]
    int C;
    int64_t max_inst, void_label;
};

// Diagnosis build only (-DNMSA_PQ_STAMPS, tools/diag_pq_stamps.py): thread 0 of every workgroup
// leaves the 100 MHz wall clock at the phase boundaries of k_pq_count.
#ifdef NMSA_PQ_STAMPS
__device__ unsigned long long g_pq_stamps[4096 * 8];
#define PQ_STAMP(i) do { if (threadIdx.x == 0) { const int wg_ = blockIdx.y * gridDim.x + blockIdx.x;          \
        if (wg_ < 4096) g_pq_stamps[wg_ * 8 + (i)] = wall_clock64(); } } while (0)
#else
#define PQ_STAMP(i) do { } while (0)
#endif

template <bool WITH_CM, bool POW2>
__global__ __launch_bounds__(256) void k_pq_count(
    const int64_t* __restrict__ pred, const int64_t* __restrict__ target,
    int P, int64_t offset, int px_per_block,
    unsigned char* __restrict__ ws, int cap, int* __restrict__ status,
    const uint8_t* __restrict__ target_sem, int cm_n, int64_t cm_div, int cm_shift,
    uint32_t* __restrict__ cm_slab, int* __restrict__ cm_status, int ablate,
    int* __restrict__ list_n_all)
{
    __shared__ int64_t lkI[PQ_LI];
    __shared__ uint32_t lcI[PQ_LI];
    extern __shared__ uint32_t cm_hist_pq[];
    const int b = blockIdx.y;
    PQ_STAMP(0);
    const int cm_bins = WITH_CM ? cm_n * cm_n : 0;
    for (int i = threadIdx.x; i < PQ_LI; i += blockDim.x) { lkI[i] = KEY_EMPTY; lcI[i] = 0; }
    if (WITH_CM) for (int i = threadIdx.x; i < cm_bins; i += blockDim.x) cm_hist_pq[i] = 0;
    __syncthreads();
    PQ_STAMP(1);
    const uint8_t* ts = WITH_CM ? target_sem + (size_t)b * P : nullptr;
    bool cm_bad = false;
    auto cm_key = [&](int64_t t, int64_t p, bool valid) -> int {
        if (POW2) {
            // a negative prediction keeps its sign through the arithmetic shift: ONE unsigned compare
            // covers "negative" (bincount rejects it) and "class out of range"; t is a u8 label
            const int64_t pc = p >> cm_shift;
            const bool bad = valid && ((uint64_t)pc >= (uint64_t)cm_n || t >= cm_n);
            cm_bad = cm_bad || bad;
            return (valid && !bad && !(ablate & 1)) ? (int)t * cm_n + (int)pc : -1;       // miou.py:50
        }
        if (!valid || (ablate & 1)) return -1;
        if (p < 0) { cm_bad = true; return -1; }            // bincount rejects negatives
        const int64_t pc = cm_shift >= 0 ? (p >> cm_shift) : (p / cm_div);
        if (t >= cm_n || pc >= cm_n) { cm_bad = true; return -1; }
        return (int)(t * cm_n + pc);                        // miou.py:50
    };
    auto cm_runs = [&](int key, uint32_t weight) {
        const int prev = __shfl_up(key, 1);
        const bool head = key >= 0 && (lane_id() == 0 || prev != key);
        const unsigned long long heads = __ballot(head);
        const unsigned long long gaps = __ballot(key < 0);
        if (head) {
            const int l = lane_id();
            const unsigned long long stop = (heads | gaps) & ~((2ull << l) - 1ull);
            const int nxt = stop ? (__ffsll((long long)stop) - 1) : 64;
            atomicAdd(&cm_hist_pq[key], weight * (uint32_t)(nxt - l));
        }
    };
    int64_t* gk = pq_keys(ws, b, cap);
    uint32_t* gc = pq_cnts(ws, b, cap);
    int* list_n = list_n_all + b * PQ_LIST_STRIDE;     // one counter per image, each on a line of its own
    uint32_t* list_slots = (uint32_t*)(pq_list_keys(ws, b, cap) + cap / 2) + cap / 2;
    const int list_cap = cap / 2;
    const int64_t* pr = pred + (size_t)b * P;
    const int64_t* tg = target + (size_t)b * P;
    const int start = blockIdx.x * px_per_block;
    const int end = min(start + px_per_block, P);
    int st = 0;

    // called by MANY lanes at once (one per run of equal pixels)
    auto add = [&](int64_t iid, uint32_t cnt) {
        if (ablate & 2) return;
        if (iid == KEY_EMPTY) { st |= ST_SENTINEL_KEY; return; }
        // fast path: key already in its home slot -> one LDS read, one LDS atomic
        const uint32_t sI = hash_id(iid) & (PQ_LI - 1);
        if (lkI[sI] == iid) { atomicAdd(&lcI[sI], cnt); return; }
        if (!table_add(lkI, lcI, PQ_LI - 1, iid, cnt, 32) &&
            !table_add_listed(gk, gc, cap - 1, iid, cnt, 256, list_n, list_slots, list_cap)) st |= ST_TABLE_OVERFLOW;
    };
    // intersection id with torch's int64 wrap-around arithmetic (pq.py:104); ids that would
    // not decode uniquely (pq.py:83-109 then fails or mixes segments) raise ST_MISSING_KEY
    // (offset is a power of two in practice — 256^3, task_helper/panoptic.py:57-63 — then the
    // 64-bit multiply is a shift; same wrap-around either way)
    const int off_shift = ((offset & (offset - 1)) == 0) ? (63 - __clzll((long long)offset)) : -1;
    auto iid_of = [&](int64_t t, int64_t p, bool valid) -> int64_t {
        if (POW2) {
            st |= (valid && ((t | p) < 0 || p >= offset)) ? ST_MISSING_KEY : 0;
            return (int64_t)(((uint64_t)t << off_shift) + (uint64_t)p);
        }
        if (valid && (t < 0 || p < 0 || p >= offset)) st |= ST_MISSING_KEY;
        if (off_shift >= 0) return (int64_t)(((uint64_t)t << off_shift) + (uint64_t)p);
        return (int64_t)((uint64_t)t * (uint64_t)offset + (uint64_t)p);
    };
    // One (target, pred) pair per lane, lanes = consecutive pixels.  Neighbouring pixels
    // mostly share the pair, so the wave is cut into RUNS of equal intersection ids: every run
    // head inserts its run length — all heads in parallel, no leader-serial loop.
    auto wave_runs = [&](bool valid, int64_t iid, uint32_t weight) {
        const int64_t prev = __shfl_up(iid, 1);
        const bool pvalid = __shfl_up((int)valid, 1) != 0;
        const bool head = valid && (lane_id() == 0 || !pvalid || prev != iid);
        const unsigned long long heads = __ballot(head);
        const unsigned long long vmask = __ballot(valid);
        if (head) {
            const int l = lane_id();
            const unsigned long long stop = (heads | ~vmask) & ~((2ull << l) - 1ull);
            const int nxt = stop ? (__ffsll((long long)stop) - 1) : 64;
            add(iid, weight * (uint32_t)(nxt - l));
        }
    };

    // the image plane is consumed as 16-B (2 px) loads when the rows allow it
    const bool vec = ((P & 1) == 0) && ((start & 1) == 0) &&
                     ((((uintptr_t)pr | (uintptr_t)tg) & 15) == 0) &&
                     (!WITH_CM || (((uintptr_t)ts) & 1) == 0);
    if (vec) {
        const int tile = blockDim.x * 2 * PQ_UNROLL;            // px per block iteration
        // (the NEXT tile's loads in flight while this one is counted — register double buffering —
        // was measured twice, before and after the kernel's code shrank: 72.6 vs 68.9 us per update)
        for (int base = start; base < end; base += tile) {
            longlong2 tv[PQ_UNROLL], pv[PQ_UNROLL];
            uint32_t sv[PQ_UNROLL];
            bool ok[PQ_UNROLL];
#pragma unroll
            for (int u = 0; u < PQ_UNROLL; ++u) {
                const int i = base + (u * blockDim.x + threadIdx.x) * 2;
                ok[u] = i < end;                                // end is even on this path
                tv[u] = ok[u] ? ld_i64x2_stream(tg + i) : make_longlong2(0, 0);
                pv[u] = ok[u] ? ld_i64x2_stream(pr + i) : make_longlong2(0, 0);
                sv[u] = (WITH_CM && ok[u]) ? (uint32_t)*(const uint16_t*)(ts + i) : 0u;
            }
            if (WITH_CM) {
#pragma unroll
                for (int u = 0; u < PQ_UNROLL; ++u) {
                    const int k0 = cm_key(sv[u] & 0xFFu, pv[u].x, ok[u]);
                    const int k1 = cm_key(sv[u] >> 8, pv[u].y, ok[u]);
                    // lanes whose two pixels agree form the runs (one LDS atomic per run); the
                    // others — segment boundaries, or every lane when the maps are incoherent —
                    // add their two pixels themselves: ONE run pass per load either way
                    const bool same = k0 == k1;
                    const int kf = __builtin_amdgcn_readfirstlane(k0);
                    if (__all(same && k0 == kf)) {                    // the wave sits on one bin
                        if (lane_id() == 0 && kf >= 0) atomicAdd(&cm_hist_pq[kf], 128u);
                    } else {
                        if (k0 >= 0) atomicAdd(&cm_hist_pq[k0], same ? 2u : 1u);
                        if (!same && k1 >= 0) atomicAdd(&cm_hist_pq[k1], 1u);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < PQ_UNROLL; ++u) {
                const int64_t i0 = iid_of(tv[u].x, pv[u].x, ok[u]);
                const int64_t i1 = iid_of(tv[u].y, pv[u].y, ok[u]);
                const bool same = i0 == i1;
                // Insertion per LANE, not per run: cutting the wave into runs (shuffles, two
                // ballots, per-lane 64-bit mask arithmetic) cost more issue slots than the LDS
                // atomics it saved (47 -> 41 us on the bench's 30-60 px segments).  Only a wave
                // that sits on ONE pair — the inside of a large segment, where 64 lanes would
                // hit one LDS address — sends a single lane with the whole weight.
                const int64_t f0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(i0 >> 32)) << 32) |
                                             (uint32_t)__builtin_amdgcn_readfirstlane((int)i0));
                if (__all(ok[u] && same && i0 == f0)) {
                    if (lane_id() == 0) add(f0, 128u);
                } else {
                    if (ok[u]) add(i0, same ? 2u : 1u);
                    if (ok[u] && !same) add(i1, 1u);
                }
            }
            if (base == start) PQ_STAMP(2);
        }
    } else {
        const int span = end - start;
        const int trips = (span + (int)blockDim.x - 1) / (int)blockDim.x;
        for (int k = 0; k < trips; ++k) {
            const int i = start + k * blockDim.x + threadIdx.x;
            const bool valid = i < end;
            const int64_t pi = valid ? pr[i] : 0;
            wave_runs(valid, iid_of(valid ? tg[i] : 0, pi, valid), 1u);
            if (WITH_CM) cm_runs(cm_key(valid ? ts[i] : 0, pi, valid), 1u);
        }
    }
    PQ_STAMP(3);
    __syncthreads();
    PQ_STAMP(4);
    if (WITH_CM) {
        uint32_t* mine = cm_slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * cm_bins;
        if (!(ablate & 4)) for (int i = threadIdx.x; i < cm_bins; i += blockDim.x) mine[i] = cm_hist_pq[i];
        if (cm_bad) atomicOr(cm_status, ST_VALUE_RANGE);
    }
    // flush the block-private table
    if (!(ablate & 8))
    for (int i = threadIdx.x; i < PQ_LI; i += blockDim.x)
        if (lkI[i] != KEY_EMPTY && !table_add_listed(gk, gc, cap - 1, lkI[i], lcI[i], 256, list_n, list_slots, list_cap))
            st |= ST_TABLE_OVERFLOW;
    if (st) atomicOr(status, st);
    PQ_STAMP(5);
}

// ---- the same count for a prediction given as its PARTS (PqPredParts), compact keys ----------
// k_pq_count's time is VALU issue (a wave64 instruction holds a 16-lane SIMD for four cycles:
// 2430 per wave = 19 us of its 46), most of it 64-bit: forming t * offset + p, hashing and
// comparing it, the class of p for the confusion matrix, range tests — per pixel.  With the parts
// a pixel's prediction is one of 512 CODES (its instance id, else 256 + its class), so the block
// counts (target id, code) pairs and decodes each DISTINCT pair once, at the flush: the
// intersection id comes from a 512-entry LDS table there, and the confusion-matrix class of a
// code from another one per pixel.  Per pixel that leaves a 24-bit multiplicative hash
// (v_mul_u32_u24 is full rate, a 32-bit multiply is not), one 8-byte LDS read, one compare, one
// LDS atomic, and for the matrix one 2-byte LDS read, one multiply-add, one LDS atomic.  Targets
// that do not fit 32 bits (negative ids included) go straight to the decode, pixel by pixel —
// exact, slow, and not what label maps hold.  Full tiles run without per-lane bounds; the ragged
// rest one pixel per lane.
constexpr unsigned long long PQP_EMPTY = ~0ull;
constexpr uint32_t PQP_BAD_CLASS = 0x8000u;          // > any n * n (n <= PQ_CM_MAX_CLASSES = 64)

__device__ __forceinline__ uint32_t pqp_slot(uint32_t tlo, uint32_t code)
{
    return ((uint32_t)__umul24(tlo, 0x9E3779u) + (uint32_t)__umul24(code, 0x85EBCBu)) >> 22;     // PQ_LI = 1024 slots
}

__global__ __launch_bounds__(256) void k_pq_count_parts(
    const int64_t* __restrict__ target, int P, int64_t offset, int px_per_block,
    unsigned char* __restrict__ ws, int cap, int* __restrict__ status,
    const uint8_t* __restrict__ target_sem, int cm_n, int64_t cm_div, int cm_shift,
    uint32_t* __restrict__ cm_slab, int* __restrict__ cm_status,
    int* __restrict__ list_n_all, PqPredParts parts)
{
    static_assert(PQ_LI == 1024, "pqp_slot keeps the top ten bits of the hash");
    __shared__ unsigned long long lk[PQ_LI];        // code << 32 | target id
    __shared__ uint32_t lc[PQ_LI];
    __shared__ int64_t s_pan[512];                  // code -> painted panoptic id (k_paint2's rule)
    __shared__ uint16_t s_cls[512];                 // code -> confusion-matrix class of that id
    extern __shared__ uint32_t cm_hist_pq[];
    const int b = blockIdx.y;
    PQ_STAMP(0);
    const uint32_t cm_bins = (uint32_t)(cm_n * cm_n);
    const bool with_cm = cm_n > 0;                  // (block-uniform: the PQ-only form passes 0 classes)
    {
        const int t = threadIdx.x;
        const int64_t pi = parts.pan_of_inst[(size_t)b * 256 + t];
        const int64_t ps = (t < parts.C && !parts.is_thing[t]) ? (int64_t)(t + 1) * parts.max_inst : parts.void_label;
        auto cls_of = [&](int64_t p) -> uint16_t {          // miou.py:50 on the painted id
            if (p < 0) return (uint16_t)PQP_BAD_CLASS;      // bincount rejects negatives
            const int64_t pc = cm_shift >= 0 ? (p >> cm_shift) : (p / cm_div);
            return pc >= cm_n ? (uint16_t)PQP_BAD_CLASS : (uint16_t)pc;
        };
        s_pan[t] = pi;       s_cls[t] = cls_of(pi);
        s_pan[256 + t] = ps; s_cls[256 + t] = cls_of(ps);
    }
    for (int i = threadIdx.x; i < PQ_LI; i += blockDim.x) { lk[i] = PQP_EMPTY; lc[i] = 0; }
    for (int i = threadIdx.x; i < (int)cm_bins; i += blockDim.x) cm_hist_pq[i] = 0;
    __syncthreads();
    PQ_STAMP(1);
    int64_t* gk = pq_keys(ws, b, cap);
    uint32_t* gc = pq_cnts(ws, b, cap);
    int* list_n = list_n_all + b * PQ_LIST_STRIDE;
    uint32_t* list_slots = (uint32_t*)(pq_list_keys(ws, b, cap) + cap / 2) + cap / 2;
    const int list_cap = cap / 2;
    const uint8_t* psem = parts.sem + (size_t)b * P;
    const uint8_t* pins = parts.inst + (size_t)b * P;
    const uint8_t* ts = with_cm ? target_sem + (size_t)b * P : nullptr;
    const int64_t* tg = target + (size_t)b * P;
    const int start = blockIdx.x * px_per_block;
    const int end = min(start + px_per_block, P);
    const int off_shift = ((offset & (offset - 1)) == 0) ? (63 - __clzll((long long)offset)) : -1;
    int st = 0;
    bool cm_bad = false;

    // one distinct (target id, code) with its pixel count: the decode
    auto decode = [&](int64_t t, uint32_t code, uint32_t cnt) {
        const int64_t p = s_pan[code];
        if (t < 0 || p < 0 || p >= offset) st |= ST_MISSING_KEY;
        // torch's int64 wrap-around arithmetic (pq.py:104)
        const int64_t iid = off_shift >= 0 ? (int64_t)(((uint64_t)t << off_shift) + (uint64_t)p)
                                           : (int64_t)((uint64_t)t * (uint64_t)offset + (uint64_t)p);
        if (iid == KEY_EMPTY) { st |= ST_SENTINEL_KEY; return; }
        if (!table_add_listed(gk, gc, cap - 1, iid, cnt, 256, list_n, list_slots, list_cap)) st |= ST_TABLE_OVERFLOW;
    };
    // off the fast path: a key not (yet) in its home slot, or a target id beyond 32 bits
    auto slow = [&](int64_t t, uint32_t code, uint32_t cnt) {
        bool placed = false;
        if (((uint64_t)t >> 32) == 0) {
            const unsigned long long key = ((unsigned long long)code << 32) | (uint32_t)t;
            uint32_t slot = pqp_slot((uint32_t)t, code);
#pragma unroll 1
            for (int probe = 0; probe < 32 && !placed; ++probe) {
                const unsigned long long cur = __hip_atomic_load(&lk[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (cur == key) placed = true;
                else if (cur == PQP_EMPTY) {
                    const unsigned long long prev = atomicCAS(&lk[slot], PQP_EMPTY, key);
                    placed = prev == PQP_EMPTY || prev == key;
                }
                if (placed) atomicAdd(&lc[slot], cnt);
                else slot = (slot + 1) & (PQ_LI - 1);
            }
        }
        if (!placed) decode(t, code, cnt);
    };
    auto code_of = [](uint32_t sm, uint32_t in) -> uint32_t { return in ? in : (256u | sm); };
    // confusion matrix (miou.py:50) of a lane's two pixels: bin = target class * n + class of the
    // painted id; a bad class of either side lands beyond the bins
    auto cm_pair = [&](uint32_t ts0, uint32_t ts1, uint32_t cls0, uint32_t cls1) {
        const uint32_t k0 = ts0 * (uint32_t)cm_n + cls0, k1 = ts1 * (uint32_t)cm_n + cls1;
        const bool in0 = k0 < cm_bins, in1 = k1 < cm_bins;
        cm_bad = cm_bad || !in0 || !in1;
        const bool ksame = in0 && k0 == k1;
        const uint32_t kf = (uint32_t)__builtin_amdgcn_readfirstlane((int)k0);
        if (__all(ksame && k0 == kf)) {                           // the wave sits on one bin
            if (lane_id() == 0) atomicAdd(&cm_hist_pq[kf], 128u);
        } else {
            if (in0) atomicAdd(&cm_hist_pq[k0], ksame ? 2u : 1u);
            if (in1 && !ksame) atomicAdd(&cm_hist_pq[k1], 1u);
        }
    };

    const bool vec = ((P & 1) == 0) && ((start & 1) == 0) && ((((uintptr_t)tg) & 15) == 0) &&
                     (((((uintptr_t)psem | (uintptr_t)pins | (uintptr_t)ts)) & 1) == 0);
    int base = start;
    if (vec) {
        const int tile = blockDim.x * 2 * PQ_UNROLL;
        auto step = [&](int64_t t0, int64_t t1, uint32_t c0, uint32_t c1, uint32_t ts0, uint32_t ts1) {
            if (with_cm) cm_pair(ts0, ts1, s_cls[c0], s_cls[c1]);
            const uint32_t l0 = (uint32_t)t0, l1 = (uint32_t)t1;
            const bool big = (((uint64_t)t0 | (uint64_t)t1) >> 32) != 0;
            const bool same = l0 == l1 && c0 == c1;
            const uint32_t fl = (uint32_t)__builtin_amdgcn_readfirstlane((int)l0);
            const uint32_t fc = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
            const bool uni = __all(same && !big && l0 == fl && c0 == fc);
            bool act0 = uni ? lane_id() == 0 : true;
            bool act1 = !uni && (big || !same);
            const uint32_t cnt0 = uni ? 128u : ((same && !big) ? 2u : 1u);
            if (act0 && !big) {
                const uint32_t slot = pqp_slot(l0, c0);
                const unsigned long long key = ((unsigned long long)c0 << 32) | l0, cur = lk[slot];
                if (cur == key) { atomicAdd(&lc[slot], cnt0); act0 = false; }
                else if (cur == PQP_EMPTY) {
                    const unsigned long long prev = atomicCAS(&lk[slot], PQP_EMPTY, key);
                    if (prev == PQP_EMPTY || prev == key) { atomicAdd(&lc[slot], cnt0); act0 = false; }
                }
            }
            if (act1 && !big) {
                const uint32_t slot = pqp_slot(l1, c1);
                const unsigned long long key = ((unsigned long long)c1 << 32) | l1, cur = lk[slot];
                if (cur == key) { atomicAdd(&lc[slot], 1u); act1 = false; }
                else if (cur == PQP_EMPTY) {
                    const unsigned long long prev = atomicCAS(&lk[slot], PQP_EMPTY, key);
                    if (prev == PQP_EMPTY || prev == key) { atomicAdd(&lc[slot], 1u); act1 = false; }
                }
            }
            if (__any(act0 || act1)) {
#pragma unroll 1
                for (int k = 0; k < 2; ++k)
                    if (k ? act1 : act0) slow(k ? t1 : t0, k ? c1 : c0, k ? 1u : cnt0);
            }
        };
        for (; base + tile <= end; base += tile) {
            longlong2 tv[PQ_UNROLL];
            uint32_t ls[PQ_UNROLL], li[PQ_UNROLL], lt[PQ_UNROLL];
#pragma unroll
            for (int u = 0; u < PQ_UNROLL; ++u) {
                const int i = base + (u * blockDim.x + threadIdx.x) * 2;
                tv[u] = ld_i64x2_stream(tg + i);
                ls[u] = *(const uint16_t*)(psem + i);
                li[u] = *(const uint16_t*)(pins + i);
                lt[u] = with_cm ? (uint32_t)*(const uint16_t*)(ts + i) : 0u;
            }
#pragma unroll
            for (int u = 0; u < PQ_UNROLL; ++u)
                step(tv[u].x, tv[u].y, code_of(ls[u] & 0xFFu, li[u] & 0xFFu), code_of(ls[u] >> 8, li[u] >> 8),
                     lt[u] & 0xFFu, lt[u] >> 8);
            if (base == start) PQ_STAMP(2);
        }
        for (; base + (int)blockDim.x * 2 <= end; base += blockDim.x * 2) {      // single steps up to the last full one
            const int i = base + threadIdx.x * 2;
            const longlong2 t = ld_i64x2_stream(tg + i);
            const uint32_t s2 = *(const uint16_t*)(psem + i), i2 = *(const uint16_t*)(pins + i);
            const uint32_t t2 = with_cm ? (uint32_t)*(const uint16_t*)(ts + i) : 0u;
            step(t.x, t.y, code_of(s2 & 0xFFu, i2 & 0xFFu), code_of(s2 >> 8, i2 >> 8), t2 & 0xFFu, t2 >> 8);
        }
    }
    for (int i0 = base; i0 < end; i0 += blockDim.x) {          // the ragged rest, one pixel per lane
        const int i = i0 + threadIdx.x;
        if (i < end) {
            const uint32_t code = code_of(psem[i], pins[i]);
            if (with_cm) {
                const uint32_t k = (uint32_t)ts[i] * (uint32_t)cm_n + s_cls[code];
                if (k < cm_bins) atomicAdd(&cm_hist_pq[k], 1u); else cm_bad = true;
            }
            slow(tg[i], code, 1u);
        }
    }
    PQ_STAMP(3);
    __syncthreads();
    PQ_STAMP(4);
    if (with_cm) {
        uint32_t* mine = cm_slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * cm_bins;
        for (int i = threadIdx.x; i < (int)cm_bins; i += blockDim.x) mine[i] = cm_hist_pq[i];
        if (cm_bad) atomicOr(cm_status, ST_VALUE_RANGE);
    }
    // decode the block's distinct pairs into the image's table
    for (int i = threadIdx.x; i < PQ_LI; i += blockDim.x) {
        const unsigned long long key = lk[i];
        if (key != PQP_EMPTY) decode((int64_t)(uint32_t)key, (uint32_t)(key >> 32), lc[i]);
    }
    if (st) atomicOr(status, st);
    PQ_STAMP(5);
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
constexpr int PQ_TP_CAP = 1024;             // matched pairs per image
constexpr int PQ_E_LDS = 2048;              // intersections per image whose list entry stays in LDS

// One workgroup per image.  The image's intersection table stays where k_pq_count built it
// (global memory, L2-resident: 12 B x cap); its non-empty slots are first compacted into an
// entry list, every later step walks that list and looks other intersections up in the table
// with O(1) probes (load factor <= 1/2).  Only the MATCHED pairs (a few dozen) are put in
// ascending-id order, which is all the reference's fp64 summation order depends on.
__global__ __launch_bounds__(PQ_MATCH_THREADS) void k_pq_match(
    unsigned char* __restrict__ ws, int cap, int num_categories, int64_t ignored_label,
    int64_t max_inst, int64_t offset, int64_t void_segment_id,
    double* __restrict__ img_state /* [B,4,num_categories] */,
    int64_t* __restrict__ matches /* [B,match_cap,2] or null */, int match_cap,
    int32_t* __restrict__ n_matches, int* __restrict__ status, int* __restrict__ list_n_all)
{
    __shared__ int64_t kT[PQ_T_CAP], kP[PQ_P_CAP];     // segment-area tables (marginals)
    __shared__ uint32_t cT[PQ_T_CAP], cP[PQ_P_CAP];
    __shared__ uint8_t fT[PQ_T_CAP], fP[PQ_P_CAP];     // matched flags
    __shared__ int64_t tpKey[PQ_TP_CAP], tpKeyS[PQ_TP_CAP];   // TP list: unordered / sorted
    __shared__ double tpIou[PQ_TP_CAP], tpIouS[PQ_TP_CAP];
    __shared__ int16_t tpCat[PQ_TP_CAP], tpCatS[PQ_TP_CAP];
    __shared__ int fnI[PQ_MAX_CATEGORIES], fpI[PQ_MAX_CATEGORIES];
    __shared__ int64_t ignKeys[64];
    __shared__ int nIgn, nTPs, nEnt;
    // the first PQ_E_LDS entries of the image's intersection list live in LDS (a few hundred on
    // label maps of real scenes); the list in the workspace takes what is beyond
    __shared__ int64_t sK[PQ_E_LDS];
    __shared__ uint32_t sC[PQ_E_LDS], sS[PQ_E_LDS];

    const int b = blockIdx.x, tid = threadIdx.x;
    // offset and max_inst are powers of two in the reference's use (256^3, 2^16): floor division
    // and floor modulo are then an arithmetic shift and a mask — a 64-bit integer division is
    // ~150 instructions, and an entry went through ten of them in a row
    const int off_sh = (offset > 0 && (offset & (offset - 1)) == 0) ? (63 - __clzll((long long)offset)) : -1;
    const int inst_sh = (max_inst > 0 && (max_inst & (max_inst - 1)) == 0) ? (63 - __clzll((long long)max_inst)) : -1;
    auto div_off = [&](int64_t a) -> int64_t { return off_sh >= 0 ? (a >> off_sh) : floordiv64(a, offset); };
    auto mod_off = [&](int64_t a) -> int64_t { return off_sh >= 0 ? (a & (offset - 1)) : floormod64(a, offset); };
    auto div_inst = [&](int64_t a) -> int64_t { return inst_sh >= 0 ? (a >> inst_sh) : floordiv64(a, max_inst); };
    int64_t* gk = pq_keys(ws, b, cap);
    uint32_t* gc = pq_cnts(ws, b, cap);
    int64_t* eK = pq_list_keys(ws, b, cap);            // entry list: key, count, table slot
    uint32_t* eC = (uint32_t*)(eK + cap / 2);
    uint32_t* eS = eC + cap / 2;
    const int ecap = cap / 2;
    int st = 0;

    for (int i = tid; i < PQ_T_CAP; i += PQ_MATCH_THREADS) {
        fT[i] = 0; fP[i] = 0; kT[i] = KEY_EMPTY; kP[i] = KEY_EMPTY; cT[i] = 0; cP[i] = 0;
    }
    for (int i = tid; i < num_categories; i += PQ_MATCH_THREADS) { fnI[i] = 0; fpI[i] = 0; }
    if (tid == 0) { nIgn = 0; nTPs = 0; nEnt = 0; }
    __syncthreads();

    // ---- 0. the image's intersections: k_pq_count listed every slot it claimed -----------------
    const int nListed = list_n_all[b * PQ_LIST_STRIDE];
    if (tid == 0) nEnt = nListed;
    for (int e = tid; e < min(nListed, ecap); e += PQ_MATCH_THREADS) {
        const uint32_t slot = eS[e];
        if (e < PQ_E_LDS) { sS[e] = slot; sK[e] = gk[slot]; sC[e] = gc[slot]; }
        else { eK[e] = gk[slot]; eC[e] = gc[slot]; }
    }
    __syncthreads();
    if (nEnt > ecap) st |= ST_TABLE_OVERFLOW;          // table more than half full
    const int nE = min(nEnt, ecap);
    __threadfence_block();
    auto entry_key = [&](int e) -> int64_t { return e < PQ_E_LDS ? sK[e] : eK[e]; };
    auto entry_count = [&](int e) -> uint32_t { return e < PQ_E_LDS ? sC[e] : eC[e]; };
    auto entry_slot = [&](int e) -> uint32_t { return e < PQ_E_LDS ? sS[e] : eS[e]; };

    // ---- 1. segment areas = marginals of the intersection table (pq.py:83-84) ------------
    for (int e = tid; e < nE; e += PQ_MATCH_THREADS) {
        const int64_t iid = entry_key(e);
        const uint32_t cnt_e = entry_count(e);
        const int64_t gt = div_off(iid), pr = mod_off(iid);
        if (!table_add(kT, cT, PQ_T_CAP - 1, gt, cnt_e, PQ_T_CAP)) st |= ST_TABLE_OVERFLOW;
        if (!table_add(kP, cP, PQ_P_CAP - 1, pr, cnt_e, PQ_P_CAP)) st |= ST_TABLE_OVERFLOW;
    }
    __syncthreads();
    // ignored segments: target ids whose category is the ignored label (pq.py:89-93)
    for (int s = tid; s < PQ_T_CAP; s += PQ_MATCH_THREADS) {
        const int64_t k = kT[s];
        if (k != KEY_EMPTY && div_inst(k) == ignored_label) {
            const int at = atomicAdd(&nIgn, 1);
            if (at < 64) ignKeys[at] = k;
        }
    }

    // ---- 2. TP decision per intersection (pq.py:119-153), unordered TP list --------------
    for (int e = tid; e < nE; e += PQ_MATCH_THREADS) {
        const int64_t iid = entry_key(e);
        if (iid == void_segment_id) continue;                               // :120-121
        const int64_t gt = div_off(iid), pr = mod_off(iid);
        const int64_t gcat = div_inst(gt), pcat = div_inst(pr);
        if (gcat != pcat) continue;                                         // :128-129
        // prediction_void_overlap (pq.py:35-44)
        const int64_t vid = (int64_t)((uint64_t)void_segment_id * (uint64_t)offset + (uint64_t)pr);
        const int sV = table_find(gk, cap - 1, vid);
        const int64_t r = sV >= 0 ? (int64_t)gc[sV] : 0;
        const int sT = table_find(kT, PQ_T_CAP - 1, gt);
        const int sP = table_find(kP, PQ_P_CAP - 1, pr);
        if (sT < 0 || sP < 0) { st |= ST_MISSING_KEY; continue; }
        const int64_t ia = entry_count(e);
        const int64_t uni = (int64_t)cT[sT] + (int64_t)cP[sP] - ia - r;     // :143
        const double iou = (double)ia / (double)uni;                        // :145
        if (iou > 0.5) {                                                    // :147
            if (gcat < 0 || gcat >= num_categories) { st |= ST_CATEGORY_RANGE; continue; }
            fT[sT] = 1; fP[sP] = 1;
            const int at = atomicAdd(&nTPs, 1);
            if (at < PQ_TP_CAP) { tpKey[at] = iid; tpIou[at] = iou; tpCat[at] = (int16_t)gcat; }
        }
    }
    __syncthreads();
    const int nTP = nTPs;
    if (nTP > PQ_TP_CAP) st |= ST_TABLE_OVERFLOW;
    const int nTPc = min(nTP, PQ_TP_CAP);

    // ---- 3. rank sort of the TP list by id (= reference dict iteration order) --------------
    for (int i = tid; i < nTPc; i += PQ_MATCH_THREADS) {
        const int64_t k = tpKey[i];
        int rank = 0;
        for (int j = 0; j < nTPc; ++j) rank += (tpKey[j] < k);              // ids are distinct
        tpKeyS[rank] = k; tpIouS[rank] = tpIou[i]; tpCatS[rank] = tpCat[i];
    }
    __syncthreads();

    // ---- 4. per-class TP / IoU sums in ascending-id order (bit-exact fp64) ------------------
    double* out = img_state + (size_t)b * 4 * num_categories;
    for (int c = tid; c < num_categories; c += PQ_MATCH_THREADS) {
        double iou = 0.0, tp = 0.0;
        for (int e = 0; e < nTPc; ++e)
            if (tpCatS[e] == c) { tp += 1.0; iou += tpIouS[e]; }
        out[0 * num_categories + c] = iou;
        out[1 * num_categories + c] = tp;
    }

    // ---- 5. false negatives (pq.py:155-163) -----------------------------------------------
    for (int s = tid; s < PQ_T_CAP; s += PQ_MATCH_THREADS) {
        const int64_t k = kT[s];
        if (k == KEY_EMPTY || fT[s]) continue;
        const int64_t cat = div_inst(k);
        if (cat == ignored_label) continue;
        if (cat < 0 || cat >= num_categories) { st |= ST_CATEGORY_RANGE; continue; }
        atomicAdd(&fnI[cat], 1);
    }
    // ---- 6. false positives (pq.py:165-177) -------------------------------------------------
    const int n_ign = nIgn;
    if (n_ign > 64) st |= ST_TABLE_OVERFLOW;
    for (int s = tid; s < PQ_P_CAP; s += PQ_MATCH_THREADS) {
        const int64_t k = kP[s];
        if (k == KEY_EMPTY || fP[s]) continue;
        int64_t pio = 0;                                  // prediction_ignored_overlap :47-57
        for (int q = 0; q < min(n_ign, 64); ++q) {
            const int sI = table_find(gk, cap - 1,
                                      (int64_t)((uint64_t)ignKeys[q] * (uint64_t)offset + (uint64_t)k));
            if (sI >= 0) pio += gc[sI];
        }
        if ((double)pio / (double)cP[s] > 0.5) continue;
        const int64_t cat = div_inst(k);
        if (cat < 0 || cat >= num_categories) { st |= ST_CATEGORY_RANGE; continue; }
        atomicAdd(&fpI[cat], 1);
    }
    __syncthreads();
    for (int c = tid; c < num_categories; c += PQ_MATCH_THREADS) {
        out[2 * num_categories + c] = (double)fnI[c];
        out[3 * num_categories + c] = (double)fpI[c];
    }

    // ---- 7. matched (gt, pred) pairs in id order (for the orientation MAE) -------------------
    if (matches) {
        for (int i = tid; i < nTPc && i < match_cap; i += PQ_MATCH_THREADS) {
            matches[((size_t)b * match_cap + i) * 2 + 0] = div_off(tpKeyS[i]);
            matches[((size_t)b * match_cap + i) * 2 + 1] = mod_off(tpKeyS[i]);
        }
    }
    if (tid == 0 && n_matches) n_matches[b] = nTP;
    if (st) atomicOr(status, st);
    // leave the image's table empty for the next update (no memset per step): the lookups above
    // are done once every thread has passed this barrier
    __syncthreads();
    if (nEnt > ecap) {                                 // overflowed: the list is incomplete
        for (int i = tid; i < cap; i += PQ_MATCH_THREADS) { gk[i] = KEY_EMPTY; gc[i] = 0; }
    } else {
        for (int e = tid; e < nE; e += PQ_MATCH_THREADS) { const uint32_t sl = entry_slot(e); gk[sl] = KEY_EMPTY; gc[sl] = 0; }
    }
    if (tid == 0) list_n_all[b * PQ_LIST_STRIDE] = 0;  // the list is consumed: clean for the next update
}

// (the workgroups behind the first `num_categories` sum the confusion-matrix slabs of
// k_pq_count<true> — k_confmat_reduce's arithmetic, independent of the matching — so the chain of
// an update is count -> match -> this: one launch less to wait for wave slots)
__global__ __launch_bounds__(256) void k_pq_accumulate(
    const double* __restrict__ img_state, int B, int num_categories,
    double* __restrict__ iou, double* __restrict__ tp,
    double* __restrict__ fn, double* __restrict__ fp,
    const uint32_t* __restrict__ cm_slab, int cm_slabs, int cm_bins,
    unsigned long long* __restrict__ confmat)
{
    if ((int)blockIdx.x >= num_categories) {
        const int rb = blockIdx.x - num_categories;
        const int i = (rb / CM_REDUCE_GROUPS) * 256 + threadIdx.x;
        if (i >= cm_bins) return;
        unsigned long long acc = 0;
        int k = rb % CM_REDUCE_GROUPS;
        for (; k + 7 * CM_REDUCE_GROUPS < cm_slabs; k += 8 * CM_REDUCE_GROUPS) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = cm_slab[(size_t)(k + u * CM_REDUCE_GROUPS) * cm_bins + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; k < cm_slabs; k += CM_REDUCE_GROUPS) acc += cm_slab[(size_t)k * cm_bins + i];
        if (acc) atomicAdd(&confmat[i], acc);
        return;
    }
    if (threadIdx.x >= 64) return;                     // one wave per category (no workgroup barrier below)
    __shared__ double buf[4][64];
    const int c = blockIdx.x, l = threadIdx.x;
    double a0 = iou[c], a1 = tp[c], a2 = fn[c], a3 = fp[c];
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + l;
        if (b < B) {
            const double* s = img_state + (size_t)b * 4 * num_categories;
#pragma unroll
            for (int k = 0; k < 4; ++k) buf[k][l] = s[k * num_categories + c];
        }
        __builtin_amdgcn_wave_barrier(); __threadfence_block();
        if (l == 0) {
            const int nb = min(64, B - b0);
            for (int j = 0; j < nb; ++j) {              // image order, like pq.py:291-296
                a0 += buf[0][j]; a1 += buf[1][j]; a2 += buf[2][j]; a3 += buf[3][j];
            }
        }
        __builtin_amdgcn_wave_barrier(); __threadfence_block();
    }
    if (l == 0) { iou[c] = a0; tp[c] = a1; fn[c] = a2; fp[c] = a3; }
}

}  // namespace nmsa

using namespace nmsa;

namespace {
constexpr int CM_MAX_BLOCKS = 1024;
}

extern "C" size_t nmsa_confmat_workspace_bytes(int n_classes)
{
    if (n_classes <= 0 || n_classes > 46340) return 0;
    const int64_t nbins = (int64_t)n_classes * n_classes;
    if (nbins > CM_LDS_BINS) return 16;                 // global-atomic path: no slabs
    return (size_t)CM_MAX_BLOCKS * nbins * sizeof(uint32_t);
}

extern "C" int nmsa_confmat_update(const void* preds, int pred_dtype, int64_t pred_div,
                                   const void* target, int target_dtype,
                                   int64_t n_px, int n_classes, int mode,
                                   int64_t* confmat, int32_t* status,
                                   void* workspace, size_t workspace_bytes,
                                   nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!preds || !target || !confmat || !status || !workspace) return NMSA_ERR_ARG;
    if (n_px < 0 || n_classes <= 0 || n_classes > 46340 || pred_div <= 0) return NMSA_ERR_ARG;
    if (pred_dtype < NMSA_U8 || pred_dtype > NMSA_I64 || target_dtype < NMSA_U8 || target_dtype > NMSA_I64)
        return NMSA_ERR_ARG;
    if (mode != 0 && mode != 1) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_confmat_workspace_bytes(n_classes)) return NMSA_ERR_WORKSPACE;
    if (n_px == 0) return NMSA_OK;
    const int nbins = n_classes * n_classes;
    const int use_lds = nbins <= CM_LDS_BINS;
    const size_t lds = use_lds ? (size_t)nbins * 4 : 0;
    const int64_t n_chunks = (n_px + 2 * CM_UNROLL - 1) / (2 * CM_UNROLL);     // per-thread work items
    // >= 4 workgroups per CU for latency hiding; a big LDS histogram allows only one
    static const int blocks_env = getenv("NMSA_CM_BLOCKS") ? atoi(getenv("NMSA_CM_BLOCKS")) : 0;
    const int one_per_cu = device_geometry().cus < CM_MAX_BLOCKS ? device_geometry().cus : CM_MAX_BLOCKS;
    int64_t blocks = (lds > 40 * 1024) ? one_per_cu : (blocks_env > 0 && blocks_env <= CM_MAX_BLOCKS ? blocks_env : CM_MAX_BLOCKS);
    const int64_t need = (n_chunks + 256 - 1) / 256;
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    const bool vec = (((uintptr_t)preds | (uintptr_t)target) & 15) == 0;
    // a big LDS histogram allows one workgroup per CU: 16 waves share it instead of 4
    // (n = 151: 110 -> measured below; 256 threads elsewhere)
    static const int big_threads = getenv("NMSA_CM_BIG_THREADS") ? atoi(getenv("NMSA_CM_BIG_THREADS")) : 1024;
    dim3 grid((unsigned)blocks), block(lds > 40 * 1024 ? big_threads : 256);
    unsigned long long* cm = (unsigned long long*)confmat;
    uint32_t* slab = (uint32_t*)workspace;
    static const bool no_u8_path = getenv("NMSA_CM_NO_U8") != nullptr;            // A/B knob
    if (pred_dtype == NMSA_U8 && target_dtype == NMSA_U8 && pred_div == 1 && use_lds && lds <= 40 * 1024 && vec &&
        !no_u8_path) {
        int64_t b8 = (n_px + 256 * 16 - 1) / (256 * 16);
        if (b8 > CM_MAX_BLOCKS) b8 = CM_MAX_BLOCKS;
        hipLaunchKernelGGL(k_confmat_u8, dim3((unsigned)b8), dim3(256), lds, stream, (const uint8_t*)preds,
                           (const uint8_t*)target, n_px, n_classes, mode, slab, status);
        int rc8 = check_launch();
        if (rc8) return rc8;
        hipLaunchKernelGGL(k_confmat_reduce, dim3((nbins + 255) / 256, CM_REDUCE_GROUPS), dim3(256), 0,
                           stream, (const uint32_t*)workspace, (int)b8, nbins, cm);
        return check_launch();
    }
#define NMSA_CM_LAUNCH(PD, TD)                                                                   \
    do {                                                                                         \
        if (vec) hipLaunchKernelGGL((k_confmat<PD, TD, true>), grid, block, lds, stream, preds,  \
                                    pred_div, target, n_px, n_classes, mode, use_lds, cm, slab,  \
                                    status);                                                     \
        else hipLaunchKernelGGL((k_confmat<PD, TD, false>), grid, block, lds, stream, preds,     \
                                pred_div, target, n_px, n_classes, mode, use_lds, cm, slab,      \
                                status);                                                         \
    } while (0)
#define NMSA_CM_TD(PD)                                                      \
    switch (target_dtype) {                                                 \
        case NMSA_U8: NMSA_CM_LAUNCH(PD, NMSA_U8); break;                   \
        case NMSA_I16: NMSA_CM_LAUNCH(PD, NMSA_I16); break;                 \
        case NMSA_I32: NMSA_CM_LAUNCH(PD, NMSA_I32); break;                 \
        default: NMSA_CM_LAUNCH(PD, NMSA_I64); break;                       \
    }
    switch (pred_dtype) {
        case NMSA_U8: NMSA_CM_TD(NMSA_U8); break;
        case NMSA_I16: NMSA_CM_TD(NMSA_I16); break;
        case NMSA_I32: NMSA_CM_TD(NMSA_I32); break;
        default: NMSA_CM_TD(NMSA_I64); break;
    }
#undef NMSA_CM_TD
#undef NMSA_CM_LAUNCH
    int rc = check_launch();
    if (rc || !use_lds) return rc;
    hipLaunchKernelGGL(k_confmat_reduce, dim3((nbins + 255) / 256, CM_REDUCE_GROUPS), dim3(256), 0,
                       stream, (const uint32_t*)workspace, (int)blocks, nbins, cm);
    return check_launch();
}

extern "C" size_t nmsa_pq_workspace_bytes(int B, int H, int W, int num_categories)
{
    if (B <= 0 || H <= 0 || W <= 0 || num_categories <= 0) return 0;
    return (size_t)B * pq_image_bytes(pq_i_cap((int64_t)H * W)) +
           (size_t)B * 4 * num_categories * sizeof(double) + (size_t)B * PQ_LIST_STRIDE * sizeof(int) + 128;   // + list counters
}

namespace {

constexpr int PQ_CM_MAX_CLASSES = 64;        // fused confusion matrix: n*n u32 in LDS next to the PQ table

int pq_px_per_block()
{
    static const int px_per_block_env = getenv("NMSA_PQ_PXB") ? atoi(getenv("NMSA_PQ_PXB")) : 0;
    return px_per_block_env > 0 ? (px_per_block_env & ~1) : 8192;     // tuning knob
}

int pq_update_impl(const int64_t* pred, const int64_t* target, int B, int H, int W,
                   int num_categories, int64_t ignored_label,
                   int64_t max_instances_per_category, int64_t offset, int64_t void_segment_id,
                   double* iou_per_class, double* tp_per_class, double* fn_per_class,
                   double* fp_per_class, int64_t* matches, int match_capacity, int32_t* n_matches,
                   int32_t* status, void* workspace, size_t workspace_bytes, int workspace_is_clean,
                   const uint8_t* target_sem, int cm_n, int64_t cm_div, int64_t* confmat,
                   int32_t* cm_status, void* cm_workspace, hipStream_t stream,
                   const PqPredParts* parts = nullptr)
{
    if ((!pred && !parts) || !target || !iou_per_class || !tp_per_class || !fn_per_class || !fp_per_class ||
        !status || !workspace)
        return NMSA_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || (int64_t)H * W > ((int64_t)1 << 30) || B > 65535) return NMSA_ERR_ARG;
    if (num_categories <= 0 || num_categories > PQ_MAX_CATEGORIES) return NMSA_ERR_ARG;
    if (max_instances_per_category <= 0 || offset <= 0) return NMSA_ERR_ARG;
    if (matches && (match_capacity <= 0 || !n_matches)) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_pq_workspace_bytes(B, H, W, num_categories)) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    const int cap = pq_i_cap(P);
    unsigned char* ws = (unsigned char*)workspace;
    double* img_state = (double*)(ws + (size_t)B * pq_image_bytes(cap));
    int* list_n = (int*)(((uintptr_t)(img_state + (size_t)B * 4 * num_categories) + 127) & ~(uintptr_t)127);

    int rc;
    if (!workspace_is_clean) {
        hipLaunchKernelGGL(k_pq_init, dim3(8, B), dim3(256), 0, stream, ws, cap, list_n);
        if ((rc = check_launch())) return rc;
    }
    const int px_per_block = pq_px_per_block();
    static const int ablate = getenv("NMSA_PQ_ABLATE") ? atoi(getenv("NMSA_PQ_ABLATE")) : 0;   // diagnostics
    const dim3 grid((P + px_per_block - 1) / px_per_block, B);
    const bool off_pow2 = (offset & (offset - 1)) == 0;
    if (target_sem) {
        int shift = -1;
        if ((cm_div & (cm_div - 1)) == 0) shift = __builtin_ctzll((unsigned long long)cm_div);
        const bool p2 = off_pow2 && shift >= 0;
        if (parts) {
            hipLaunchKernelGGL(k_pq_count_parts, grid, dim3(256), (size_t)cm_n * cm_n * sizeof(uint32_t), stream,
                               target, P, offset, px_per_block, ws, cap, status, target_sem, cm_n, cm_div, shift,
                               (uint32_t*)cm_workspace, cm_status, list_n, *parts);
        } else {
            auto kern = p2 ? k_pq_count<true, true> : k_pq_count<true, false>;
            hipLaunchKernelGGL(kern, grid, dim3(256), (size_t)cm_n * cm_n * sizeof(uint32_t),
                               stream, pred, target, P, offset, px_per_block, ws, cap, status, target_sem, cm_n,
                               cm_div, shift, (uint32_t*)cm_workspace, cm_status, ablate, list_n);
        }
    } else if (parts) {                             // PQ only (more classes than the fused matrix holds)
        hipLaunchKernelGGL(k_pq_count_parts, grid, dim3(256), 0, stream,
                           target, P, offset, px_per_block, ws, cap, status, (const uint8_t*)nullptr, 0, (int64_t)1, -1,
                           (uint32_t*)nullptr, (int*)nullptr, list_n, *parts);
    } else {
        auto kern = off_pow2 ? k_pq_count<false, true> : k_pq_count<false, false>;
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, stream,
                           pred, target, P, offset,
                           px_per_block, ws, cap, status, (const uint8_t*)nullptr, 0, (int64_t)1, -1,
                           (uint32_t*)nullptr, (int*)nullptr, ablate, list_n);
    }
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(k_pq_match, dim3(B), dim3(PQ_MATCH_THREADS), 0, stream, ws, cap, num_categories,
                       ignored_label, max_instances_per_category, offset, void_segment_id,
                       img_state, matches, match_capacity, n_matches, status, list_n);
    rc = check_launch();
    if (rc) return rc;
    const int nbins = target_sem ? cm_n * cm_n : 0;
    const int extra = target_sem ? ((nbins + 255) / 256) * CM_REDUCE_GROUPS : 0;
    hipLaunchKernelGGL(k_pq_accumulate, dim3(num_categories + extra), dim3(256), 0, stream,
                       img_state, B, num_categories, iou_per_class, tp_per_class, fn_per_class,
                       fp_per_class, (const uint32_t*)cm_workspace, (int)(grid.x * grid.y), nbins,
                       (unsigned long long*)confmat);
    return check_launch();
}

}  // namespace

extern "C" int nmsa_pq_update(const int64_t* pred, const int64_t* target, int B, int H, int W,
                              int num_categories, int64_t ignored_label,
                              int64_t max_instances_per_category, int64_t offset,
                              int64_t void_segment_id,
                              double* iou_per_class, double* tp_per_class,
                              double* fn_per_class, double* fp_per_class,
                              int64_t* matches, int match_capacity, int32_t* n_matches,
                              int32_t* status, void* workspace, size_t workspace_bytes,
                              int workspace_is_clean, nmsa_stream_t stream_)
{
    return pq_update_impl(pred, target, B, H, W, num_categories, ignored_label,
                          max_instances_per_category, offset, void_segment_id, iou_per_class,
                          tp_per_class, fn_per_class, fp_per_class, matches, match_capacity, n_matches,
                          status, workspace, workspace_bytes, workspace_is_clean,
                          nullptr, 0, 1, nullptr, nullptr, nullptr, (hipStream_t)stream_);
}

extern "C" size_t nmsa_pq_confmat_workspace_bytes(int B, int H, int W, int n_classes)
{
    if (B <= 0 || H <= 0 || W <= 0 || n_classes <= 0 || n_classes > PQ_CM_MAX_CLASSES) return 0;
    const int px_per_block = pq_px_per_block();
    const size_t blocks = (size_t)(((int64_t)H * W + px_per_block - 1) / px_per_block) * B;
    return blocks * (size_t)n_classes * n_classes * sizeof(uint32_t);
}

extern "C" int nmsa_pq_update_with_confmat(
    const int64_t* pred, const int64_t* target, const uint8_t* target_semantic,
    int B, int H, int W,
    int num_categories, int64_t ignored_label, int64_t max_instances_per_category, int64_t offset,
    int64_t void_segment_id,
    double* iou_per_class, double* tp_per_class, double* fn_per_class, double* fp_per_class,
    int64_t* matches, int match_capacity, int32_t* n_matches,
    int32_t* status, void* workspace, size_t workspace_bytes, int workspace_is_clean,
    int confmat_classes, int64_t pred_div, int64_t* confmat, int32_t* confmat_status,
    void* confmat_workspace, size_t confmat_workspace_bytes, nmsa_stream_t stream_)
{
    if (!target_semantic || !confmat || !confmat_status || !confmat_workspace) return NMSA_ERR_ARG;
    if (confmat_classes <= 0 || confmat_classes > PQ_CM_MAX_CLASSES || pred_div <= 0) return NMSA_ERR_ARG;
    const size_t need = nmsa_pq_confmat_workspace_bytes(B, H, W, confmat_classes);
    if (need == 0) return NMSA_ERR_ARG;
    if (confmat_workspace_bytes < need) return NMSA_ERR_WORKSPACE;
    return pq_update_impl(pred, target, B, H, W, num_categories, ignored_label,
                          max_instances_per_category, offset, void_segment_id, iou_per_class,
                          tp_per_class, fn_per_class, fp_per_class, matches, match_capacity, n_matches,
                          status, workspace, workspace_bytes, workspace_is_clean,
                          target_semantic, confmat_classes, pred_div, confmat, confmat_status,
                          confmat_workspace, (hipStream_t)stream_);
}

// the same with the prediction given as the PARTS the merge painted it from (PqPredParts): the
// pixel's predicted panoptic id is pan_of_inst[b][inst] where inst != 0, else (sem + 1) *
// max_instances_per_category for a stuff class, else void_label — nmsa_panoptic_paint's rule — so
// the result is bit-identical to nmsa_pq_update_with_confmat on the painted map, which is not read
extern "C" int nmsa_pq_update_with_confmat_parts(
    const uint8_t* pred_semantic, const uint8_t* pred_instance, const int64_t* pan_of_inst,
    const uint8_t* is_thing_class, int n_sem_classes, int64_t void_label,
    const int64_t* target, const uint8_t* target_semantic,
    int B, int H, int W,
    int num_categories, int64_t ignored_label, int64_t max_instances_per_category, int64_t offset,
    int64_t void_segment_id,
    double* iou_per_class, double* tp_per_class, double* fn_per_class, double* fp_per_class,
    int32_t* status, void* workspace, size_t workspace_bytes, int workspace_is_clean,
    int confmat_classes, int64_t pred_div, int64_t* confmat, int32_t* confmat_status,
    void* confmat_workspace, size_t confmat_workspace_bytes, nmsa_stream_t stream_)
{
    if (!pred_semantic || !pred_instance || !pan_of_inst || !is_thing_class) return NMSA_ERR_ARG;
    if (n_sem_classes <= 0 || n_sem_classes > 255 || void_label < 0) return NMSA_ERR_ARG;
    // confmat_classes == 0 with NULL target_semantic / confmat / confmat_status: the PQ update alone
    const bool pq_only = confmat_classes == 0 && !target_semantic && !confmat && !confmat_status;
    if (!pq_only) {
        if (!target_semantic || !confmat || !confmat_status || !confmat_workspace) return NMSA_ERR_ARG;
        if (confmat_classes <= 0 || confmat_classes > PQ_CM_MAX_CLASSES || pred_div <= 0) return NMSA_ERR_ARG;
        const size_t need = nmsa_pq_confmat_workspace_bytes(B, H, W, confmat_classes);
        if (need == 0) return NMSA_ERR_ARG;
        if (confmat_workspace_bytes < need) return NMSA_ERR_WORKSPACE;
    }
    const PqPredParts parts{pred_semantic, pred_instance, pan_of_inst, is_thing_class, n_sem_classes,
                            max_instances_per_category, void_label};
    return pq_update_impl(nullptr, target, B, H, W, num_categories, ignored_label,
                          max_instances_per_category, offset, void_segment_id, iou_per_class,
                          tp_per_class, fn_per_class, fp_per_class, nullptr, 0, nullptr,
                          status, workspace, workspace_bytes, workspace_is_clean,
                          target_semantic, confmat_classes, pred_div, confmat, confmat_status,
                          confmat_workspace, (hipStream_t)stream_, &parts);
}

#ifdef NMSA_PQ_STAMPS
extern "C" int nmsa_debug_pq_stamps(unsigned long long* host_dst, int n_words, int clear)
{
    if (clear) {
        void* p = nullptr;
        if (nmsa::check_hip(hipGetSymbolAddress(&p, HIP_SYMBOL(nmsa::g_pq_stamps)))) return NMSA_ERR_LAUNCH;
        return nmsa::check_hip(hipMemset(p, 0, sizeof(unsigned long long) * 4096 * 8));
    }
    return nmsa::check_hip(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(nmsa::g_pq_stamps),
                                               sizeof(unsigned long long) * (size_t)n_words));
}
#endif
