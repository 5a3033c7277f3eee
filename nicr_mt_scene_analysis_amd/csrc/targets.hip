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

// ---- labels of 4 consecutive pixels ---------------------------------------------------------
// FAST = the on-wire dtypes (semantic uint8, instance int32) with 16-byte aligned rows of 4:
// one 16-B and one 4-B load per thread.  Everything else takes the generic per-element loader.
template <bool FAST>
__device__ __forceinline__ void load_ins4(const void* ins, int ins_dtype, size_t o, int nvalid,
                                          int64_t out[4])
{
    if (FAST) {
        const int4 v = *(const int4*)((const int32_t*)ins + o);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = (j < nvalid) ? mw_load(ins, ins_dtype, o + j) : 0;
    }
}

template <bool FAST>
__device__ __forceinline__ void load_sem4(const void* sem, int sem_dtype, size_t o, int nvalid,
                                          int64_t out[4])
{
    if (FAST) {
        const uchar4 v = *(const uchar4*)((const uint8_t*)sem + o);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = (j < nvalid) ? mw_load(sem, sem_dtype, o + j) : 0;
    }
}

// wave_aggregate_add with a wave-uniform weight per contributing lane
template <typename AddFn>
__device__ __forceinline__ void wave_aggregate_add_w(int key, uint32_t weight, AddFn add)
{
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader);
        const unsigned long long same = __ballot(key == k) & todo;
        if (lane_id() == leader) add(k, (uint32_t)__popcll(same) * weight);
        todo &= ~same;
    }
}

constexpr int TG_PX_PER_BLOCK = 256 * 4;

// A 1024-pixel workgroup meets only a handful of instances, but a big instance is met by
// hundreds of workgroups: accumulate in small LDS hash tables (linear probing, 4 tries, global
// atomic as the fallback) and flush each used slot with ONE global atomic per workgroup.
constexpr int TG_H1 = 64;       // slots keyed by dense instance id: sum of y, sum of x
constexpr int TG_H2 = 256;      // slots keyed by (dense id, class): votes

struct TgLdsTables {
    int k1[TG_H1];
    unsigned long long sy[TG_H1], sx[TG_H1];
    int k2[TG_H2];
    uint32_t cnt[TG_H2];
};

// ---- presence of every non-zero instance id ------------------------------------------------
template <bool FAST>
__global__ __launch_bounds__(256) void k_tg_presence(
    const void* __restrict__ ins, int ins_dtype, int P, int cap, int NC,
    unsigned char* __restrict__ ws, int* __restrict__ status)
{
    __shared__ int s_ids[TG_H1];
    const int b = blockIdx.y;
    TgView v = tg_view(ws, b, cap, NC);
    if (threadIdx.x < TG_H1) s_ids[threadIdx.x] = -1;
    __syncthreads();
    bool bad = false;
    for (int p0 = (blockIdx.x * 256 + threadIdx.x) * 4; p0 - (int)threadIdx.x * 4 < P;
         p0 += gridDim.x * TG_PX_PER_BLOCK) {
        int key[4] = {-1, -1, -1, -1};
        if (p0 < P) {
            int64_t id[4];
            load_ins4<FAST>(ins, ins_dtype, (size_t)b * P + p0, min(4, P - p0), id);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (id[j] < 0 || id[j] > MW_MAX_ID) bad = true;
                else if (id[j] > 0) key[j] = (int)id[j];
            }
        }
        // lanes whose 4 pixels agree (instance maps are coherent: almost all) are cut into runs,
        // one atomic per run head; a boundary lane sets its own few bits
        const bool same4 = key[0] == key[1] && key[1] == key[2] && key[2] == key[3];
        int rl, rlast;
        const int k4 = same4 ? key[0] : -1;
        // big instances would hammer one bitmap word with thousands of same-address atomics:
        // collect the workgroup's ids in a small LDS set, one global atomic per id at the end
        auto set_bit = [&](int id) {
            if (lds_hash_slot(s_ids, TG_H1, id) < 0) atomicOr(&v.bitmap[id >> 5], 1u << (id & 31));
        };
        if (wave_run_head(k4, rl, rlast)) set_bit(k4);
        if (!same4) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (key[j] >= 0 && (j == 0 || key[j] != key[j - 1])) set_bit(key[j]);
        }
    }
    __syncthreads();
    if (threadIdx.x < TG_H1 && s_ids[threadIdx.x] >= 0) {
        const int id = s_ids[threadIdx.x];
        atomicOr(&v.bitmap[id >> 5], 1u << (id & 31));
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
template <bool WITH_MOMENTS, bool FAST>
__global__ __launch_bounds__(256) void k_tg_stats(
    const void* __restrict__ sem, int sem_dtype, const void* __restrict__ ins, int ins_dtype,
    int P, int W, int cap, int NC, unsigned char* __restrict__ ws, int* __restrict__ status)
{
    __shared__ TgLdsTables T;
    const int b = blockIdx.y;
    TgView v = tg_view(ws, b, cap, NC);
    for (int i = threadIdx.x; i < TG_H2; i += 256) {
        T.k2[i] = -1; T.cnt[i] = 0;
        if (i < TG_H1) { T.k1[i] = -1; T.sy[i] = 0; T.sx[i] = 0; }
    }
    __syncthreads();
    auto add_vote = [&](int key, uint32_t n) {
        const int slot = lds_hash_slot(T.k2, TG_H2, key);
        if (slot >= 0) atomicAdd(&T.cnt[slot], n);
        else atomicAdd(&v.votes[key], n);
    };
    auto add_moments = [&](int dd, unsigned long long ay, unsigned long long ax) {
        const int slot = lds_hash_slot(T.k1, TG_H1, dd);
        if (slot >= 0) { atomicAdd(&T.sy[slot], ay); atomicAdd(&T.sx[slot], ax); }
        else { atomicAdd(&v.sum_y[dd], ay); atomicAdd(&v.sum_x[dd], ax); }
    };
    bool bad = false;
    for (int p0 = (blockIdx.x * 256 + threadIdx.x) * 4; p0 - (int)threadIdx.x * 4 < P;
         p0 += gridDim.x * TG_PX_PER_BLOCK) {
        int d[4] = {-1, -1, -1, -1}, vote_key[4] = {-1, -1, -1, -1};
        int y0 = 0, x0 = 0;
        if (p0 < P) {
            const int nvalid = min(4, P - p0);
            int64_t id[4], sm[4];
            load_ins4<FAST>(ins, ins_dtype, (size_t)b * P + p0, nvalid, id);
            load_sem4<FAST>(sem, sem_dtype, (size_t)b * P + p0, nvalid, sm);
            y0 = p0 / W;
            x0 = p0 - y0 * W;
            int dd_prev = -1;
            int64_t id_prev = -1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (id[j] <= 0 || id[j] > MW_MAX_ID) continue;
                const int dd = (id[j] == id_prev) ? dd_prev : id_rank_dense(v.bitmap, v.prefix, (int)id[j]);
                id_prev = id[j];
                dd_prev = dd;
                if (dd >= cap) continue;
                d[j] = dd;
                if (sm[j] < 0 || sm[j] >= NC) bad = true;
                else vote_key[j] = dd * NC + (int)sm[j];
            }
        }
        // class votes: lanes whose 4 pixels agree (the usual case) share ONE wave-aggregated
        // round of weight 4; a boundary lane adds its own pixels
        const bool votes_same = vote_key[0] == vote_key[1] && vote_key[1] == vote_key[2] &&
                                vote_key[2] == vote_key[3];
        int rl, rlast;
        const int vk = votes_same ? vote_key[0] : -1;
        if (wave_run_head(vk, rl, rlast)) add_vote(vk, 4u * (uint32_t)rl);
        if (!votes_same) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (vote_key[j] >= 0) add_vote(vote_key[j], 1u);
        }
        if (WITH_MOMENTS) {
            // same split for the coordinate sums: a lane with one instance in one row contributes
            // (4y, 4x + 6); the sum over a run is a difference of two wave prefix sums
            // (runs are cut at row changes — the key carries the row — so a run of n lanes is
            // 4n consecutive pixels of ONE row starting at x0: sum y = 4 n y0, sum x =
            // n (4 x0 + 6) + 16 n (n - 1) / 2: closed forms at the run head, no wave scans)
            const bool lane_uniform = d[0] == d[1] && d[1] == d[2] && d[2] == d[3] && x0 + 3 < W;
            const int dj = lane_uniform ? d[0] : -1;
            const int rk = dj >= 0 ? (dj | (y0 << 12)) : -1;          // cap <= 4096, y0 < 32768
            if (wave_run_head(rk, rl, rlast)) {
                const unsigned long long n = (unsigned long long)rl;
                add_moments(dj, 4ull * n * (unsigned long long)y0,
                            n * (unsigned long long)(4 * x0 + 6) + 8ull * n * (n - 1ull));
            }
            if (!lane_uniform) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (d[j] < 0) continue;
                    const int pj = x0 + j;                     // < W + 3: at most one row further
                    const bool wrap = pj >= W;                 // (W < 4: the generic division below)
                    if (W >= 4) add_moments(d[j], (unsigned long long)(y0 + (wrap ? 1 : 0)),
                                            (unsigned long long)(wrap ? pj - W : pj));
                    else add_moments(d[j], (unsigned long long)(y0 + pj / W), (unsigned long long)(pj % W));
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TG_H2; i += 256) {
        if (T.k2[i] >= 0 && T.cnt[i]) atomicAdd(&v.votes[T.k2[i]], T.cnt[i]);
        if (WITH_MOMENTS && i < TG_H1 && T.k1[i] >= 0) {
            atomicAdd(&v.sum_y[T.k1[i]], T.sy[i]);
            atomicAdd(&v.sum_x[T.k1[i]], T.sx[i]);
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
    // majority class per present instance: one WAVE per slot, lane = class (a few dozen instances
    // per image: one thread per slot would leave 1000 threads idle behind 41 dependent loads)
    __shared__ int s_enc[4096];
    for (int slot = t >> 6; slot < n_dense; slot += 16) {
        const uint32_t* row = v.votes + (size_t)slot * NC;
        uint32_t total = 0;
        long long best = -1;                          // (count << 32) | ~class: max = larger count, lower class
        for (int c = lane_id(); c < NC; c += 64) {
            const uint32_t x = row[c];
            total += x;
            const long long key = ((long long)x << 32) | (uint32_t)(0x7fffffff - c);
            best = key > best ? key : best;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            total += __shfl_down(total, o);
            const long long other = __shfl_down(best, o);
            best = other > best ? other : best;
        }
        if (lane_id() == 0) {
            const int c_best = 0x7fffffff - (int)(uint32_t)(best & 0xffffffffll);   // np.bincount(..).argmax()
            const bool thing = is_thing_class ? (is_thing_class[c_best] != 0) : true;
            const int e = thing && total > 0;
            s_enc[slot] = e;
            if (e) {
                // int(np.mean(rows)), int(np.mean(cols)): exact integer floor (instance.py:210-211)
                v.center_yx[2 * slot] = (int)(v.sum_y[slot] / total);
                v.center_yx[2 * slot + 1] = (int)(v.sum_x[slot] / total);
            }
        }
    }
    __syncthreads();
    int enc[4], present[4];
    int n_enc = 0, n_skip = 0;
    for (int j = 0; j < per; ++j) {
        const int slot = t * per + j;
        present[j] = slot < n_dense;
        enc[j] = present[j] ? s_enc[slot] : 0;
        v.enc[slot] = enc[j];
        n_enc += enc[j];
        n_skip += present[j] && !enc[j];
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

// ---- ONE launch for presence + rank + statistics + decide (round 5) -------------------------------
// Rounds 2-4 ran presence -> rank -> statistics -> decide (-> naive ranks) as four launches behind
// a memset: each waits for the one before and each is a chain of latencies, not bytes (48 us of
// kernels + gaps for 49 MB of labels at B = 32, 640 x 480), and the one-workgroup-per-image steps
// leave 7/8 of the chip idle.  k_tg_scan is ONE launch in front of the paint kernel:
//   scan   `wpi` workgroups per image, as many as the device holds at once (ONE round of resident
//          workgroups: a workgroup's life is a chain of memory latencies, so rounds are what cost):
//          each walks a contiguous range of its image — the next round's labels are requested
//          before the current one is worked on — and counts votes and coordinate sums per
//          (instance id, class) in ONE LDS table keyed by the RAW id (no dense rank needed while
//          scanning: the two dependent bitmap / prefix loads per lane of k_tg_stats are gone, one
//          table probe serves the vote and both sums), flushed ONCE into per-image accumulators
//          indexed by the slot the id gets in a per-image open-addressing table in global memory
//          (`hkeys`, key 0 = empty, atomicCAS insert).  Everything handed to the image's last
//          workgroup goes through device-scope atomics (performed at the memory side: no release
//          fence — a fence per workgroup cost 26 us in k_multi_count — only a drain of the
//          workgroup's own atomics before its ticket).
//   tail   the workgroup that draws an image's LAST ticket does what k_tg_rank + k_tg_decide
//          (+ k_tg_naive_ranks) did: presence bitmap and prefix from the table's keys (in LDS),
//          dense ranks in ascending id order (= np.unique order), majority class / thing filter /
//          center, the ordered lists — or the per-class running ranks of the naive merge.
// The paint kernels are unchanged.  (Measured on the way, docs/measurement_log.md: the same scan
// with 150 small workgroups per image 64 us — 23 us of it the two-table LDS work, 16 us flush
// chains, 10 us tail; a persistent launch that also paints, work pulled from queues: 159 us — every
// pull, flush, ticket and sc1 table load is a ~2 us round trip to the memory side, in series per
// workgroup, with 4 workgroups per CU to hide them.)
constexpr int TGF_MAX_NC = 16384;                        // (id, class) as one 30-bit key
struct TgHash {
    int32_t* hkeys;              // [HT] raw instance id, 0 = empty
    unsigned long long* hsum_y;  // [HT]
    unsigned long long* hsum_x;  // [HT]
    uint32_t* hvotes;            // [HT * NC]
    uint32_t* ticket;            // [0] workgroups of the image that are done
};

__host__ __device__ inline int tg_hash_slots(int cap)
{
    int ht = 1024;
    while (ht < cap) ht <<= 1;
    return ht;
}

// behind the images' hash tables: [0] status bits of the running call, [1] tails that are done
constexpr size_t TG_GLOBAL_BYTES = 256;

__host__ __device__ inline size_t tg_hash_bytes(int cap, int NC)
{
    const size_t ht = (size_t)tg_hash_slots(cap);
    return (ht * 4 + ht * 8 * 2 + ht * NC * 4 + 256 + 255) & ~(size_t)255;       // (the ticket: a line of its own)
}


__device__ __forceinline__ TgHash tg_hash_view(unsigned char* hs, int b, int cap, int NC)
{
    const size_t ht = (size_t)tg_hash_slots(cap);
    unsigned char* base = hs + (size_t)b * tg_hash_bytes(cap, NC);
    TgHash h;
    h.hsum_y = (unsigned long long*)base;
    h.hsum_x = h.hsum_y + ht;
    h.hkeys = (int32_t*)(h.hsum_x + ht);
    h.hvotes = (uint32_t*)(h.hkeys + ht);
    h.ticket = h.hvotes + ht * NC;
    return h;
}

// slot of `id` (> 0) in the image's table, inserting it when new; -1: the table is full
__device__ __forceinline__ int tg_hash_insert(int32_t* hkeys, int HT, int id)
{
    int slot = (int)(((uint32_t)id * 2654435761u) >> 12) & (HT - 1);
    for (int t = 0; t < HT; ++t) {
        const int old = atomicCAS(&hkeys[slot], 0, id);
        if (old == 0 || old == id) return slot;
        slot = (slot + 1) & (HT - 1);
    }
    return -1;
}

// bytes another workgroup of the same launch wrote / will read: write-through stores, L1-bypassing loads
template <typename T>
__device__ __forceinline__ void st_shared(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ T ld_shared(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// ---- the scan -------------------------------------------------------------------------------------
struct TgScanTab { int key[TG_H2]; uint32_t cnt[TG_H2], sy[TG_H2], sx[TG_H2]; };
constexpr int TGS_AHEAD = 2;                             // rounds of labels in flight per thread

// Diagnosis build only (-DNMSA_TG_STAMPS, tools/diag_f4_stamps.py): thread 0 of every workgroup of
// k_tg_scan stamps the 100 MHz wall clock at its phase boundaries into a buffer of the code object
// that nothing else reads; the product build has no stamp.
#ifdef NMSA_TG_STAMPS
__device__ unsigned long long g_tg_stamps[8192 * 8];
#define TG_STAMP(i) do { if (threadIdx.x == 0) { const int wg_ = blockIdx.y * gridDim.x + blockIdx.x;          \
        if (wg_ < 8192) g_tg_stamps[wg_ * 8 + (i)] = wall_clock64(); } } while (0)
#else
#define TG_STAMP(i) do { } while (0)
#endif

// ---- tail: what k_tg_rank + k_tg_decide (WITH_MOMENTS) or k_tg_rank + k_tg_naive_ranks did, by
// the workgroup (256 threads) that drew the image's last ticket.  What the NEXT launch reads goes
// out as plain stores (write-through sc1 stores left no copy behind and the paint kernels took
// 9 us longer to find their tables); what this workgroup needs again stays in LDS:
// scratch [32] | bitmap [2048] | prefix [2048] | slot of dense u16 [cap] | enc u8 [cap] (or, naive
// merge: the dict base of every instance i32 [cap]); once the
// ranks are known, bitmap + prefix are dead and hold the packed centers (WITH_MOMENTS) or the
// u16 rank cells of the naive merge (up to 8192 cells; beyond: the rows in global memory).
constexpr int TGS_CELLS_LDS = 2 * MW_WORDS * 2;          // u16 cells in the bitmap + prefix region

template <bool WITH_MOMENTS, int KPT>                  // KPT: table slots per thread (HT <= 256 KPT)
__device__ __forceinline__ void tg_rank_decide_tail(
    TgView v, TgHash h, int b, int cap, int HT, int NC, uint32_t* __restrict__ lds,
    const uint8_t* __restrict__ is_thing_class, int32_t* __restrict__ encoded_ids,
    int32_t* __restrict__ n_encoded, int32_t* __restrict__ skipped_ids, int32_t* __restrict__ n_skipped,
    int pair_cap, int64_t max_inst, int64_t* __restrict__ ids_pan, int64_t* __restrict__ ids_ins,
    int32_t* __restrict__ n_ids, int* __restrict__ status /* the call's internal word */,
    uint32_t* __restrict__ gblock, int n_images, int* __restrict__ user_status)
{
    int* scratch = (int*)lds;                            // [32]
    uint32_t* s_bitmap = lds + 32;
    uint32_t* s_prefix = s_bitmap + MW_WORDS;
    uint16_t* s_slot = (uint16_t*)(s_prefix + MW_WORDS);
    uint8_t* s_enc = (uint8_t*)(s_slot + cap);
    const int t = threadIdx.x;
    for (int i = t; i < MW_WORDS; i += 256) s_bitmap[i] = 0u;
    __syncthreads();
    // the keys of the table: every thread keeps its share in registers (HT / 256 = 4 for up to
    // 1024 instances — the usual table —, 16 for the largest)
    int key[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        const int sl = t + k * 256;
        key[k] = sl < HT ? ld_shared(&h.hkeys[sl]) : 0;
    }
#pragma unroll
    for (int k = 0; k < KPT; ++k) if (key[k] > 0) atomicOr(&s_bitmap[key[k] >> 5], 1u << (key[k] & 31));
    __syncthreads();
    // exclusive popcount prefix over the 2048 words (8 consecutive words per thread)
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    uint32_t w8[8], p8[8];
    int c = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { w8[k] = s_bitmap[8 * t + k]; c += __popc(w8[k]); }
    int total;
    int run = mw_block_scan(c, scratch, &total) - c;
#pragma unroll
    for (int k = 0; k < 8; ++k) { p8[k] = (uint32_t)run; s_prefix[8 * t + k] = p8[k]; run += __popc(w8[k]); }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u32x4_t wv, pv;
        wv.x = w8[4 * q]; wv.y = w8[4 * q + 1]; wv.z = w8[4 * q + 2]; wv.w = w8[4 * q + 3];
        pv.x = p8[4 * q]; pv.y = p8[4 * q + 1]; pv.z = p8[4 * q + 2]; pv.w = p8[4 * q + 3];
        *(u32x4_t*)(v.bitmap + 8 * t + 4 * q) = wv;
        *(u32x4_t*)(v.prefix + 8 * t + 4 * q) = pv;
    }
    __syncthreads();
    const int n_dense = min(total, cap);
    if (t == 0) {
        v.counters[0] = n_dense;
        if (total > cap) atomicOr(status, TG_ST_OVERFLOW);
    }
    // dense rank of every key: ascending id order = np.unique order
    int rank_of_key[KPT];
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        rank_of_key[k] = -1;
        if (key[k] <= 0) continue;
        const int d = (int)s_prefix[key[k] >> 5] + __popc(s_bitmap[key[k] >> 5] & ((1u << (key[k] & 31)) - 1u));
        if (d < cap) { s_slot[d] = (uint16_t)(t + k * 256); v.id_of_dense[d] = key[k]; rank_of_key[k] = d; }
    }
    __syncthreads();                                     // (bitmap + prefix in LDS are dead from here)
    TG_STAMP(5);
    // the ids by dense rank, kept in the dead prefix words (up to 2048 ranks; beyond: the table)
    int* s_id = (int*)s_prefix;
#pragma unroll
    for (int k = 0; k < KPT; ++k) if (rank_of_key[k] >= 0 && rank_of_key[k] < MW_WORDS) s_id[rank_of_key[k]] = key[k];
    auto id_of = [&](int d) -> int { return d < MW_WORDS ? s_id[d] : ld_shared(&h.hkeys[s_slot[d]]); };
    bool handed_over = false;
    if (WITH_MOMENTS) {
        // every status bit of this image is known (the overflow bit above was this tail's last):
        // the batch's last tail hands the call's status to the caller — requested here, so that the
        // atomic's round trip hides behind the histogram loads below
        uint32_t my_turn = 0;
        if (t == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (my atomicOr above has been performed)
            my_turn = atomicAdd(&gblock[1], 1u);                 // (looked at behind the loop below)
        }
        handed_over = true;
        // [cap]: (cy << 16) | cx — in the dead bitmap words; with more than 2048 dense slots the
        // centers beyond go behind the ids (both regions are 2048 words)
        uint32_t* s_center = s_bitmap;                   // [min(cap, 2048)]
        // per instance: a group of 16 lanes per dense slot (lane = class, strided), four slots per
        // group in flight: with up to 64 instances every histogram load of the image is issued at
        // once (the loop is a chain of memory latencies, not work)
        const int grp = t >> 4, l16 = t & 15;
        for (int d0 = grp; d0 < n_dense; d0 += 64) {
            int slot[4];
            uint32_t tot[4] = {0u, 0u, 0u, 0u};
            long long best[4] = {-1, -1, -1, -1};     // (count << 32) | ~class: max = larger count, lower class
#pragma unroll
            for (int u = 0; u < 4; ++u) slot[u] = (d0 + 16 * u < n_dense) ? (int)s_slot[d0 + 16 * u] : -1;
            // (lane u of the group will finish slot u: its coordinate sums are requested with the
            // histogram loads, not after their reduction)
            int my_slot = -1;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (l16 == u) my_slot = slot[u];
            unsigned long long my_sy = 0, my_sx = 0;
            if (l16 < 4 && my_slot >= 0) { my_sy = ld_shared(&h.hsum_y[my_slot]); my_sx = ld_shared(&h.hsum_x[my_slot]); }
            for (int cc = l16; cc < NC; cc += 16) {
                uint32_t x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    x[u] = slot[u] >= 0 ? ld_shared(&h.hvotes[(size_t)slot[u] * NC + cc]) : 0u;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (slot[u] < 0) continue;
                    tot[u] += x[u];
                    const long long kk = ((long long)x[u] << 32) | (uint32_t)(0x7fffffff - cc);
                    best[u] = kk > best[u] ? kk : best[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    tot[u] += __shfl_down(tot[u], o, 16);
                    const long long other = __shfl_down(best[u], o, 16);
                    best[u] = other > best[u] ? other : best[u];
                }
            }
            // lane 0 of the group holds the four results; lanes 0..3 take one each
            uint32_t tu = 0; long long bu = -1; int su = -1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t t0 = __shfl(tot[u], 0, 16);
                const long long b0 = __shfl(best[u], 0, 16);
                if (l16 == u) { tu = t0; bu = b0; su = slot[u]; }
            }
            if (l16 < 4 && su >= 0) {
                const int d = d0 + 16 * l16;
                const int c_best = 0x7fffffff - (int)(uint32_t)(bu & 0xffffffffll);   // np.bincount(..).argmax()
                const bool thing = is_thing_class ? (is_thing_class[c_best] != 0) : true;
                const int e = thing && tu > 0;
                s_enc[d] = (uint8_t)e;
                v.enc[d] = e;
                if (e) {
                    // int(np.mean(rows)), int(np.mean(cols)): exact integer floor (instance.py:210-211)
                    const int cy = (int)(my_sy / tu), cx = (int)(my_sx / tu);
                    st_shared(&v.center_yx[2 * d], cy);
                    st_shared(&v.center_yx[2 * d + 1], cx);
                    if (d < MW_WORDS) s_center[d] = ((uint32_t)cy << 16) | (uint32_t)cx;       // H, W < 32768
                }
            }
        }
        if (t == 0 && my_turn == (uint32_t)(n_images - 1)) {
            *user_status = (int)atomicExch(&gblock[0], 0u);
            gblock[1] = 0u;
        }
        __syncthreads();
        TG_STAMP(6);
        // the ordered lists: encoded / skipped ids ascending, the encoded centers compacted
        const int per = cap / 256;                       // cap is a multiple of 1024
        int n_enc = 0, n_skip = 0;
        for (int j = 0; j < per; ++j) {
            const int d = t * per + j;
            if (d >= n_dense) break;
            n_enc += s_enc[d];
            n_skip += !s_enc[d];
        }
        // ONE block scan for both lists: (encoded << 16) | skipped (each at most cap <= 4096)
        int tot_both;
        const int both = mw_block_scan((n_enc << 16) | n_skip, scratch, &tot_both) - ((n_enc << 16) | n_skip);
        int pe = both >> 16, ps = both & 0xffff;
        const int tot_enc = tot_both >> 16, tot_skip = tot_both & 0xffff;
        for (int j = 0; j < per; ++j) {
            const int d = t * per + j;
            if (d >= n_dense) break;
            const int id = id_of(d);
            if (s_enc[d]) {
                if (d < MW_WORDS) {
                    v.enc_list[2 * pe] = (int)(s_center[d] >> 16);
                    v.enc_list[2 * pe + 1] = (int)(s_center[d] & 0xffffu);
                } else {                                  // (more than 2048 instances: through L2)
                    v.enc_list[2 * pe] = ld_shared(&v.center_yx[2 * d]);
                    v.enc_list[2 * pe + 1] = ld_shared(&v.center_yx[2 * d + 1]);
                }
                if (encoded_ids) encoded_ids[(size_t)b * cap + pe] = id;
                ++pe;
            } else {
                if (skipped_ids) skipped_ids[(size_t)b * cap + ps] = id;
                ++ps;
            }
        }
        if (t == 0) {
            v.counters[1] = tot_enc;
            if (n_encoded) n_encoded[b] = tot_enc;
            if (n_skipped) n_skipped[b] = tot_skip;
        }
    } else {
        // naive merge (k_tg_naive_ranks): class c's running counter over the instances in ascending
        // id order replaces every non-zero histogram entry (class_id_tracker, panoptic_merge.py:
        // 76-79); the ranks go to the DENSE rows the paint reads.  The cells (present or not, then
        // the rank) live in LDS as u16 when the image's rows fit
        const int n_cells = n_dense * NC;
        const bool in_lds = n_cells <= TGS_CELLS_LDS;
        uint16_t* s_cell = (uint16_t*)s_bitmap;
        if (in_lds) {
            for (int i0 = t; i0 < n_cells; i0 += 4 * 256) {
                uint32_t x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * 256;
                    const int d = i / NC, cc = i - d * NC;
                    x[u] = i < n_cells ? ld_shared(&h.hvotes[(size_t)s_slot[d] * NC + cc]) : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (i0 + u * 256 < n_cells) s_cell[i0 + u * 256] = x[u] != 0;
            }
            __syncthreads();
            // a WAVE per class, lanes = instances in ascending order: the running counter of a
            // class is a prefix count of its non-zero cells (ballot + popcount, 64 instances a step)
            for (int cc = t >> 6; cc < NC; cc += 4) {
                uint32_t carry = 0;
                for (int d0 = 0; d0 < n_dense; d0 += 64) {
                    const int d = d0 + lane_id();
                    const bool nz = cc != 0 && d < n_dense && s_cell[d * NC + cc] != 0;   // void is ignored (:73-74)
                    const unsigned long long m = __ballot(nz);
                    if (d < n_dense) s_cell[d * NC + cc] = nz ? (uint16_t)(carry + __popcll(m & ((2ull << lane_id()) - 1ull))) : 0;
                    carry += (uint32_t)__popcll(m);
                }
            }
            __syncthreads();
            for (int i = t; i < n_cells; i += 256) v.votes[i] = s_cell[i];
        } else {
            for (int cc = t; cc < NC; cc += 256) {
                uint32_t runc = 0;
                for (int d = 0; d < n_dense; ++d) {
                    const uint32_t x = ld_shared(&h.hvotes[(size_t)s_slot[d] * NC + cc]);
                    const uint32_t r = (cc != 0 && x) ? ++runc : 0u;
                    v.votes[(size_t)d * NC + cc] = r;
                    st_shared(&h.hvotes[(size_t)s_slot[d] * NC + cc], r);   // (read back below, past the L1)
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        auto rank_of = [&](int d, int cc) -> uint32_t {
            return in_lds ? (uint32_t)s_cell[d * NC + cc] : ld_shared(&h.hvotes[(size_t)s_slot[d] * NC + cc]);
        };
        // the id dict in the reference's insertion order (instance ascending, class ascending): the
        // segments of an instance are counted by a WAVE (lanes = classes), an exclusive scan over
        // the instances gives every row its base, then every non-zero cell knows its place
        int* s_base = (int*)(s_slot + cap);              // [cap] (the naive merge has no enc bytes)
        for (int d = t >> 6; d < n_dense; d += 4) {
            int n = 0;
            for (int c0 = 0; c0 < NC; c0 += 64) {
                const int cc = c0 + lane_id();
                n += __popcll(__ballot(cc >= 1 && cc < NC && rank_of(d, cc) != 0));
            }
            if (lane_id() == 0) s_base[d] = n;
        }
        __syncthreads();
        const int per = cap / 256;
        int cnt = 0;
        for (int j = 0; j < per; ++j) {
            const int d = t * per + j;
            if (d >= n_dense) break;
            cnt += s_base[d];
        }
        int totp;
        int pos = mw_block_scan(cnt, scratch, &totp) - cnt;
        for (int j = 0; j < per; ++j) {
            const int d = t * per + j;
            if (d >= n_dense) break;
            const int n = s_base[d];
            s_base[d] = pos;
            pos += n;
        }
        __syncthreads();
        for (int d = t >> 6; d < n_dense; d += 4) {
            const int id = id_of(d);
            int at = s_base[d];
            for (int c0 = 0; c0 < NC; c0 += 64) {
                const int cc = c0 + lane_id();
                const uint32_t r = (cc >= 1 && cc < NC) ? rank_of(d, cc) : 0u;
                const unsigned long long m = __ballot(r != 0);
                const int mine = at + __popcll(m & ((1ull << lane_id()) - 1ull));
                if (r && mine < pair_cap) {
                    ids_pan[(size_t)b * pair_cap + mine] = (int64_t)cc * max_inst + r;
                    ids_ins[(size_t)b * pair_cap + mine] = id;
                }
                at += __popcll(m);
            }
        }
        if (t == 0) {
            n_ids[b] = min(totp, pair_cap);
            if (totp > pair_cap) atomicOr(status, TG_ST_PAIR_OVERFLOW);
        }
    }
    TG_STAMP(7);
    // ---- leave the image's hash tables as they were found: all zero.  The next call on this
    // workspace then needs no memset (`workspace_is_clean`).  Every slot that holds a key is
    // cleaned by the thread that loaded it, whatever the status bits say.
    __syncthreads();                                     // (every read of the accumulators is done)
#pragma unroll
    for (int k = 0; k < KPT; ++k) {
        if (key[k] <= 0) continue;
        const int sl = t + k * 256;
        h.hkeys[sl] = 0;
        h.hsum_y[sl] = 0ull;
        h.hsum_x[sl] = 0ull;
    }
    if (n_dense == total) {                              // the usual case: rows by all threads
        const int n_cells = n_dense * NC;
        for (int i = t; i < n_cells; i += 256) {
            const int d = i / NC;
            h.hvotes[(size_t)s_slot[d] * NC + (i - d * NC)] = 0u;
        }
    } else {                                             // more ids than dense slots: by the key's owner
#pragma unroll 1
        for (int k = 0; k < KPT; ++k) {
            if (key[k] <= 0) continue;
            uint32_t* row = h.hvotes + (size_t)(t + k * 256) * NC;
            for (int cc = 0; cc < NC; ++cc) row[cc] = 0u;
        }
    }
    if (t == 0) h.ticket[0] = 0u;
    if (handed_over) return;
    // the call's status goes to the caller's word with the LAST tail of the batch (set, not OR-ed)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) {
        if (atomicAdd(&gblock[1], 1u) == (uint32_t)(n_images - 1)) {
            *user_status = (int)atomicExch(&gblock[0], 0u);
            gblock[1] = 0u;
        }
    }
}

// The on-wire layout only (uint8 semantic, int32 ids, 16-byte aligned rows of 4 pixels, W % 4 == 0,
// NC <= 16384): everything else keeps the launches of rounds 2-4.
template <bool WITH_MOMENTS, int KPT>
__global__ __launch_bounds__(256) void k_tg_scan(
    const uint8_t* __restrict__ sem, const int32_t* __restrict__ ins, int P, int W, int cap, int NC,
    int px_per_wg, unsigned char* __restrict__ ws, unsigned char* __restrict__ hs,
    const uint8_t* __restrict__ is_thing_class, int32_t* __restrict__ encoded_ids,
    int32_t* __restrict__ n_encoded, int32_t* __restrict__ skipped_ids, int32_t* __restrict__ n_skipped,
    int pair_cap, int64_t max_inst, int64_t* __restrict__ ids_pan, int64_t* __restrict__ ids_ins,
    int32_t* __restrict__ n_ids, int* __restrict__ user_status)
{
    // all LDS is dynamic (no static variable in front of it: the base stays 16-byte aligned):
    // [ control word (16 B) | the scan table, then the tail's arrays ]
    extern __shared__ __attribute__((aligned(16))) uint32_t tgs_lds[];
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned char u8x4_t __attribute__((ext_vector_type(4)));
    volatile uint32_t* ctl = tgs_lds;
    TgScanTab& T = *(TgScanTab*)(tgs_lds + 4);
    const int b = blockIdx.y;
    const int HT = tg_hash_slots(cap);
    TgHash h = tg_hash_view(hs, b, cap, NC);
    // the call's status bits collect in the workspace (zero between calls) and reach the caller's
    // word with the last tail of the batch
    uint32_t* gblock = (uint32_t*)(hs + (size_t)gridDim.y * tg_hash_bytes(cap, NC));
    int* status = (int*)gblock;
    const int p_begin = blockIdx.x * px_per_wg, p_end = min(P, p_begin + px_per_wg);   // px_per_wg % 1024 == 0
    const int n_rounds = (p_end - p_begin + 1023) / 1024;
    TG_STAMP(0);
    // the labels of the next TGS_AHEAD rounds are on their way while a round is worked on
    i32x4_t id4[TGS_AHEAD];
    u8x4_t sm4[TGS_AHEAD];
    auto request = [&](int r, i32x4_t& idv, u8x4_t& smv) {
        const int p = p_begin + (r * 256 + (int)threadIdx.x) * 4;      // P % 4 == 0: whole groups
        idv = i32x4_t{0, 0, 0, 0};
        smv = u8x4_t{0, 0, 0, 0};
        if (r < n_rounds && p < p_end) {
            // (plain loads: the paint kernel reads the labels again — with non-temporal loads here it
            // found none of them in the Infinity Cache and took 9 us longer)
            idv = *(const i32x4_t*)(ins + (size_t)b * P + p);
            smv = *(const u8x4_t*)(sem + (size_t)b * P + p);
        }
    };
#pragma unroll
    for (int k = 0; k < TGS_AHEAD; ++k) request(k, id4[k], sm4[k]);
    T.key[threadIdx.x] = -1; T.cnt[threadIdx.x] = 0; T.sy[threadIdx.x] = 0; T.sx[threadIdx.x] = 0;
    __syncthreads();
    int st = 0;
    auto global_add = [&](int key, uint32_t n, unsigned long long ay, unsigned long long ax) {
        const int g = tg_hash_insert(h.hkeys, HT, key >> 14);
        if (g < 0) { st |= TG_ST_OVERFLOW; return; }
        if (n) atomicAdd(&h.hvotes[(size_t)g * NC + (key & (TGF_MAX_NC - 1))], n);
        if (WITH_MOMENTS && n) { atomicAdd(&h.hsum_y[g], ay); atomicAdd(&h.hsum_x[g], ax); }
    };
    auto add = [&](int key, uint32_t n, uint32_t ay, uint32_t ax) {
        const int slot = lds_hash_slot(T.key, TG_H2, key);
        if (slot < 0) { global_add(key, n, ay, ax); return; }
        atomicAdd(&T.cnt[slot], n);
        if (WITH_MOMENTS) { atomicAdd(&T.sy[slot], ay); atomicAdd(&T.sx[slot], ax); }
    };
    // (32-bit sums in LDS: a workgroup's range is at most 2^17 px (the host caps it), x and y < 2^15)
    for (int r0 = 0; r0 < n_rounds; r0 += TGS_AHEAD) {
#pragma unroll
        for (int k = 0; k < TGS_AHEAD; ++k) {
            const int r = r0 + k;
            const i32x4_t idv = id4[k];
            const u8x4_t smv = sm4[k];
            request(r + TGS_AHEAD, id4[k], sm4[k]);
            if (r >= n_rounds) continue;                          // (uniform)
            const int p = p_begin + (r * 256 + (int)threadIdx.x) * 4;
            const int id[4] = {idv.x, idv.y, idv.z, idv.w};
            const int cl[4] = {smv.x, smv.y, smv.z, smv.w};
            int key[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                key[j] = -1;
                if ((unsigned)id[j] > (unsigned)MW_MAX_ID) st |= TG_ST_ID_RANGE;
                else if (id[j] > 0) {
                    if (cl[j] < NC) key[j] = (id[j] << 14) | cl[j];
                    else { st |= TG_ST_CLASS_RANGE; global_add(id[j] << 14, 0u, 0u, 0u); }   // present, no vote
                }
            }
            if (!__any(key[0] >= 0 || key[1] >= 0 || key[2] >= 0 || key[3] >= 0)) continue;   // a wave of background
            const int y0 = p / W, x0 = p - y0 * W;            // W % 4 == 0: a group never leaves its row
            // lanes whose 4 pixels agree (the usual case) are cut into runs of lanes — at row changes
            // too: the run key carries the row's parity — and the run head adds the closed forms for
            // its 4 n pixels: n lanes from x0 in row y0 sum to y = 4 n y0, x = n (4 x0 + 6) + 8 n (n - 1)
            const bool same = key[0] == key[1] && key[1] == key[2] && key[2] == key[3];
            const int rk = (same && key[0] >= 0) ? ((key[0] << 1) | (y0 & 1)) : -1;
            int rl, rlast;
            if (wave_run_head(rk, rl, rlast)) {
                const uint32_t n = (uint32_t)rl;
                add(key[0], 4u * n, 4u * n * (uint32_t)y0, n * (uint32_t)(4 * x0 + 6) + 8u * n * (n - 1u));
            }
            if (!same) {
#pragma unroll
                for (int j = 0; j < 4; ++j) if (key[j] >= 0) add(key[j], 1u, (uint32_t)y0, (uint32_t)(x0 + j));
            }
        }
    }
    __syncthreads();
    TG_STAMP(1);
    if (T.key[threadIdx.x] >= 0)
        global_add(T.key[threadIdx.x], T.cnt[threadIdx.x], T.sy[threadIdx.x], T.sx[threadIdx.x]);
    if (st) atomicOr(status, st);
    // every atomic of this workgroup has been performed before its ticket is drawn
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TG_STAMP(2);
    if (threadIdx.x == 0) {
        const bool last = atomicAdd(&h.ticket[0], 1u) == gridDim.x - 1;
        ctl[0] = last ? 1u : 0u;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    TG_STAMP(3);
    if (!ctl[0]) return;
    tg_rank_decide_tail<WITH_MOMENTS, KPT>(tg_view(ws, b, cap, NC), h, b, cap, HT, NC, tgs_lds + 4, is_thing_class,
                                      encoded_ids, n_encoded, skipped_ids, n_skipped, pair_cap, max_inst,
                                      ids_pan, ids_ins, n_ids, status, gblock, (int)gridDim.y, user_status);
    TG_STAMP(4);
}

// ---- paint: heat-map, offsets, foreground, center mask -------------------------------------------
constexpr int TGP_THREADS = 256;
constexpr int TGP_PX = 1024;                 // consecutive pixels per workgroup
constexpr int TGP_LUT_LDS = 4096;            // heat-map table entries kept in LDS (sigma <= 14)

template <bool NORMALIZED, bool FAST>
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

    const int p0 = p_begin + threadIdx.x * 4;             // 4 consecutive pixels per thread
    if (p0 >= p_end) return;
    const int nvalid = min(4, p_end - p0);
    const size_t o = (size_t)b * P + p0;
    int yy[4], xx[4];
    {
        int y = p0 / W, x = p0 - y * W;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            yy[j] = y; xx[j] = x;
            if (++x == W) { x = 0; ++y; }
        }
    }
    // heat-map: max over the patches that cover the pixel (instance.py:213-229)
    float hm[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < n_cand; ++i) {
        const int cy = s_cand[2 * i], cx = s_cand[2 * i + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int dy = yy[j] - cy, dx = xx[j] - cx;
            if (abs(dy) <= radius && abs(dx) <= radius) hm[j] = fmaxf(hm[j], lut[dy * dy + dx * dx]);
        }
    }
    // offsets / foreground (instance.py:201-206,232-237)
    int64_t id[4], sm[4] = {0, 0, 0, 0};
    load_ins4<FAST>(ins, ins_dtype, o, nvalid, id);
    if (center_mask && is_stuff_class) load_sem4<FAST>(sem, sem_dtype, o, nvalid, sm);
    uint8_t fg[4], cm[4];
    int oy[4], ox[4];
    int64_t id_prev = -1;
    int d_prev = -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        fg[j] = 0; oy[j] = 0; ox[j] = 0;
        if (id[j] > 0 && id[j] <= MW_MAX_ID) {
            const int d = (id[j] == id_prev) ? d_prev : id_rank_dense(v.bitmap, v.prefix, (int)id[j]);
            id_prev = id[j];
            d_prev = d;
            if (d < cap && v.enc[d]) {
                fg[j] = 1;
                oy[j] = (int)(int16_t)(v.center_yx[2 * d] - yy[j]);       // int16 image (instance.py:180)
                ox[j] = (int)(int16_t)(v.center_yx[2 * d + 1] - xx[j]);
            }
        }
        cm[j] = fg[j];
        if (center_mask && is_stuff_class && sm[j] >= 0 && sm[j] < NC && is_stuff_class[sm[j]])
            cm[j] = 1;                                                     // instance.py:263-269
    }
    const size_t oo = (size_t)b * 2 * P + p0;
    if (FAST) {
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        typedef short i16x4_t __attribute__((ext_vector_type(4)));
        typedef unsigned char u8x4_t __attribute__((ext_vector_type(4)));
        f32x4_t h4; h4.x = hm[0]; h4.y = hm[1]; h4.z = hm[2]; h4.w = hm[3];
        *(f32x4_t*)(center + o) = h4;
        if (NORMALIZED) {
            f32x4_t a, c;                                                  // instance.py:239-243
            a.x = __fdiv_rn((float)oy[0], (float)H); a.y = __fdiv_rn((float)oy[1], (float)H);
            a.z = __fdiv_rn((float)oy[2], (float)H); a.w = __fdiv_rn((float)oy[3], (float)H);
            c.x = __fdiv_rn((float)ox[0], (float)W); c.y = __fdiv_rn((float)ox[1], (float)W);
            c.z = __fdiv_rn((float)ox[2], (float)W); c.w = __fdiv_rn((float)ox[3], (float)W);
            *(f32x4_t*)((float*)offset + oo) = a;
            *(f32x4_t*)((float*)offset + oo + P) = c;
        } else {
            i16x4_t a, c;
            a.x = (short)oy[0]; a.y = (short)oy[1]; a.z = (short)oy[2]; a.w = (short)oy[3];
            c.x = (short)ox[0]; c.y = (short)ox[1]; c.z = (short)ox[2]; c.w = (short)ox[3];
            *(i16x4_t*)((int16_t*)offset + oo) = a;
            *(i16x4_t*)((int16_t*)offset + oo + P) = c;
        }
        u8x4_t f4; f4.x = fg[0]; f4.y = fg[1]; f4.z = fg[2]; f4.w = fg[3];
        *(u8x4_t*)(foreground + o) = f4;
        if (center_mask) {
            u8x4_t c4; c4.x = cm[0]; c4.y = cm[1]; c4.z = cm[2]; c4.w = cm[3];
            *(u8x4_t*)(center_mask + o) = c4;
        }
    } else {
        for (int j = 0; j < nvalid; ++j) {
            center[o + j] = hm[j];
            if (NORMALIZED) {
                ((float*)offset)[oo + j] = __fdiv_rn((float)oy[j], (float)H);
                ((float*)offset)[oo + P + j] = __fdiv_rn((float)ox[j], (float)W);
            } else {
                ((int16_t*)offset)[oo + j] = (int16_t)oy[j];
                ((int16_t*)offset)[oo + P + j] = (int16_t)ox[j];
            }
            foreground[o + j] = fg[j];
            if (center_mask) center_mask[o + j] = cm[j];
        }
    }
}

// ---- paint on 2-d pixel tiles (the fast layout: uint8 / int32 labels, W % 4 == 0) ---------------
// A workgroup covers a 128 x 8 pixel TILE instead of 1024 consecutive pixels (1.6 rows of a
// 640-px image): the centers whose patch reaches the tile are culled in BOTH directions (a
// (6 sigma + 3)^2 patch reaches 3 % of the tiles, 11 % of the row segments), more than half of
// the tiles meet no patch at all and skip the heat-map table altogether; the label loads are
// issued first and the dependent rank / center lookups run while the candidates are collected.
constexpr int TGT_W = 128, TGT_ROWS = 8;     // rows per pass of the 256 threads; a tile is G passes high

template <bool NORMALIZED, int G>
__global__ __launch_bounds__(TGP_THREADS) void k_tg_paint_tile(
    const uint8_t* __restrict__ sem, const int32_t* __restrict__ ins,
    const uint8_t* __restrict__ is_stuff_class, const float* __restrict__ gauss_lut, int lut_n,
    int radius, int H, int W, int tiles_x, int cap, int NC, unsigned char* __restrict__ ws,
    float* __restrict__ center, void* __restrict__ offset, uint8_t* __restrict__ foreground,
    uint8_t* __restrict__ center_mask)
{
    extern __shared__ int tg_lds[];              // [cap * 2] candidate centers, then the table
    __shared__ int s_n;
    constexpr int TGT_H = TGT_ROWS * G;
    const int b = blockIdx.y;
    const int P = H * W;
    TgView v = tg_view(ws, b, cap, NC);
    int* s_cand = tg_lds;
    float* s_lut = (float*)(tg_lds + 2 * cap);
    const bool lut_in_lds = lut_n <= TGP_LUT_LDS;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int x = tx * TGT_W + (threadIdx.x & 31) * 4;
    // a thread owns G groups of 4 pixels, TGT_ROWS rows apart: their label loads and the dependent
    // rank / center lookups are all in flight together (the kernel is a chain of latencies)
    int y[G];
    bool inside[G];
    size_t o[G];
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned char u8x4_t __attribute__((ext_vector_type(4)));
    i32x4_t id4[G];
    u8x4_t sm4[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        y[g] = ty * TGT_H + g * TGT_ROWS + (threadIdx.x >> 5);
        inside[g] = y[g] < H && x < W;                        // W % 4 == 0: whole groups of 4
        o[g] = (size_t)b * P + (size_t)y[g] * W + x;
        id4[g] = i32x4_t{0, 0, 0, 0};
        sm4[g] = u8x4_t{0, 0, 0, 0};
        if (inside[g]) {
            id4[g] = *(const i32x4_t*)(ins + o[g]);
            if (center_mask && is_stuff_class) sm4[g] = *(const u8x4_t*)(sem + o[g]);
        }
    }
    const int n_enc = v.counters[1];
    const int y_lo = ty * TGT_H - radius, y_hi = ty * TGT_H + TGT_H - 1 + radius;
    const int x_lo = tx * TGT_W - radius, x_hi = tx * TGT_W + TGT_W - 1 + radius;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_enc; i0 += TGP_THREADS) {
        const int i = i0 + threadIdx.x;
        int cy = 0, cx = 0;
        bool keep = false;
        if (i < n_enc) {
            cy = v.enc_list[2 * i];
            cx = v.enc_list[2 * i + 1];
            keep = cy >= y_lo && cy <= y_hi && cx >= x_lo && cx <= x_hi;
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
    // the dense rank of every group's first instance id while the candidates settle
    int d_first[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int i0 = id4[g].x;
        d_first[g] = (i0 > 0 && i0 <= MW_MAX_ID) ? id_rank_dense(v.bitmap, v.prefix, i0) : cap;
    }
    __syncthreads();
    const int n_cand = s_n;
    if (n_cand > 0 && lut_in_lds) {                           // (uniform over the workgroup)
        for (int i = threadIdx.x; i < lut_n; i += TGP_THREADS) s_lut[i] = gauss_lut[i];
        __syncthreads();
    }
    const float* lut = lut_in_lds ? s_lut : gauss_lut;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (!inside[g]) continue;
        // heat-map: max over the patches that cover the pixel (instance.py:213-229)
        float hm[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < n_cand; ++i) {
            const int dy = y[g] - s_cand[2 * i], dx0 = x - s_cand[2 * i + 1];
            if (abs(dy) > radius) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dx = dx0 + j;
                if (abs(dx) <= radius) hm[j] = fmaxf(hm[j], lut[dy * dy + dx * dx]);
            }
        }
        // offsets / foreground (instance.py:201-206,232-237)
        const int id[4] = {id4[g].x, id4[g].y, id4[g].z, id4[g].w};
        const int sm[4] = {sm4[g].x, sm4[g].y, sm4[g].z, sm4[g].w};
        uint8_t fg[4], cm[4];
        int oy[4], ox[4];
        int id_prev = -1, cy_prev = 0, cx_prev = 0;
        bool enc_prev = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            fg[j] = 0; oy[j] = 0; ox[j] = 0;
            if (id[j] > 0 && id[j] <= MW_MAX_ID) {
                if (id[j] != id_prev) {
                    const int d = (j == 0) ? d_first[g] : id_rank_dense(v.bitmap, v.prefix, id[j]);
                    enc_prev = d < cap && v.enc[d];
                    if (enc_prev) { cy_prev = v.center_yx[2 * d]; cx_prev = v.center_yx[2 * d + 1]; }
                    id_prev = id[j];
                }
                if (enc_prev) {
                    fg[j] = 1;
                    oy[j] = (int)(int16_t)(cy_prev - y[g]);                // int16 image (instance.py:180)
                    ox[j] = (int)(int16_t)(cx_prev - (x + j));
                }
            }
            cm[j] = fg[j];
            if (center_mask && is_stuff_class && sm[j] < NC && is_stuff_class[sm[j]])
                cm[j] = 1;                                                 // instance.py:263-269
        }
        const size_t oo = (size_t)b * 2 * P + (size_t)y[g] * W + x;
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        typedef short i16x4_t __attribute__((ext_vector_type(4)));
        f32x4_t h4; h4.x = hm[0]; h4.y = hm[1]; h4.z = hm[2]; h4.w = hm[3];
        *(f32x4_t*)(center + o[g]) = h4;
        if (NORMALIZED) {
            f32x4_t a, c;                                                  // instance.py:239-243
            a.x = __fdiv_rn((float)oy[0], (float)H); a.y = __fdiv_rn((float)oy[1], (float)H);
            a.z = __fdiv_rn((float)oy[2], (float)H); a.w = __fdiv_rn((float)oy[3], (float)H);
            c.x = __fdiv_rn((float)ox[0], (float)W); c.y = __fdiv_rn((float)ox[1], (float)W);
            c.z = __fdiv_rn((float)ox[2], (float)W); c.w = __fdiv_rn((float)ox[3], (float)W);
            *(f32x4_t*)((float*)offset + oo) = a;
            *(f32x4_t*)((float*)offset + oo + P) = c;
        } else {
            i16x4_t a, c;
            a.x = (short)oy[0]; a.y = (short)oy[1]; a.z = (short)oy[2]; a.w = (short)oy[3];
            c.x = (short)ox[0]; c.y = (short)ox[1]; c.z = (short)ox[2]; c.w = (short)ox[3];
            *(i16x4_t*)((int16_t*)offset + oo) = a;
            *(i16x4_t*)((int16_t*)offset + oo + P) = c;
        }
        u8x4_t f4; f4.x = fg[0]; f4.y = fg[1]; f4.z = fg[2]; f4.w = fg[3];
        *(u8x4_t*)(foreground + o[g]) = f4;
        if (center_mask) {
            u8x4_t c4; c4.x = cm[0]; c4.y = cm[1]; c4.z = cm[2]; c4.w = cm[3];
            *(u8x4_t*)(center_mask + o[g]) = c4;
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
    // the histogram rows in use usually fit the LDS (a few dozen instances x NC classes): stage
    // them with coalesced loads, run the per-class counters there, write the ranks back — instead
    // of n_dense dependent read-modify-writes per class in global memory
    constexpr int NR_LDS = 12288;
    __shared__ uint32_t s_tab[NR_LDS];
    const int n_cells = n_dense * NC;
    const bool in_lds = n_cells <= NR_LDS;
    uint32_t* tab = in_lds ? s_tab : v.votes;
    if (in_lds) {
        for (int i = t; i < n_cells; i += 1024) s_tab[i] = v.votes[i];
        __syncthreads();
    }
    for (int c = t; c < NC; c += 1024) {
        uint32_t run = 0;
        for (int d = 0; d < n_dense; ++d) {
            uint32_t* cell = &tab[(size_t)d * NC + c];
            if (c == 0) { *cell = 0; continue; }              // void is ignored (:73-74)
            if (*cell) *cell = ++run;
        }
    }
    __syncthreads();
    if (in_lds) for (int i = t; i < n_cells; i += 1024) v.votes[i] = s_tab[i];
    const int per = cap / 1024;
    int cnt = 0;
    for (int j = 0; j < per; ++j) {
        const int slot = t * per + j;
        if (slot >= n_dense) continue;
        for (int c = 1; c < NC; ++c) cnt += tab[(size_t)slot * NC + c] != 0;
    }
    int total;
    int pos = mw_block_scan(cnt, scratch, &total) - cnt;
    for (int j = 0; j < per; ++j) {
        const int slot = t * per + j;
        if (slot >= n_dense) continue;
        for (int c = 1; c < NC; ++c) {
            const uint32_t r = tab[(size_t)slot * NC + c];
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

template <bool FAST>
__global__ __launch_bounds__(256) void k_tg_naive_paint(
    const void* __restrict__ sem, int sem_dtype, const void* __restrict__ ins, int ins_dtype,
    const uint8_t* __restrict__ is_thing_class, int P, int cap, int NC, int64_t max_inst,
    int64_t void_label, unsigned char* __restrict__ ws, int64_t* __restrict__ pan)
{
    const int b = blockIdx.y;
    TgView v = tg_view(ws, b, cap, NC);
    for (int p0 = (blockIdx.x * 256 + threadIdx.x) * 4; p0 < P; p0 += gridDim.x * TG_PX_PER_BLOCK) {
        const int nvalid = min(4, P - p0);
        const size_t o = (size_t)b * P + p0;
        int64_t id[4], sm[4], r[4];
        load_ins4<FAST>(ins, ins_dtype, o, nvalid, id);
        load_sem4<FAST>(sem, sem_dtype, o, nvalid, sm);
        int64_t id_prev = -1;
        int d_prev = -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = void_label;
            const int64_t i = id[j], c = sm[j];
            if (c > 0 && c < NC) {
                if (i > 0 && i <= MW_MAX_ID) {
                    const int d = (i == id_prev) ? d_prev : id_rank_dense(v.bitmap, v.prefix, (int)i);
                    id_prev = i;
                    d_prev = d;
                    if (d < cap) r[j] = c * max_inst + v.votes[(size_t)d * NC + c];
                } else if (i == 0 && !(is_thing_class && is_thing_class[c])) {
                    r[j] = c * max_inst;                       // stuff paste (:90-101)
                }
            }
        }
        if (FAST) {
            typedef long i64x4_t __attribute__((ext_vector_type(4)));
            i64x4_t q; q.x = r[0]; q.y = r[1]; q.z = r[2]; q.w = r[3];
            *(i64x4_t*)(pan + o) = q;
        } else {
            for (int j = 0; j < nvalid; ++j) pan[o + j] = r[j];
        }
    }
}

// ---- dense visual embedding targets -----------------------------------------------------------
// The keys are searched by the WAVE, not per pixel: for every distinct panoptic id among the
// wave's pixels (1-3 in practice) lane l compares key l, one ballot finds the last match.
__global__ __launch_bounds__(256) void k_dve_indices(
    const int64_t* __restrict__ pan, const int64_t* __restrict__ keys, const int32_t* __restrict__ n_keys,
    int K, int P, int32_t* __restrict__ indices)
{
    extern __shared__ int64_t s_keys[];
    const int b = blockIdx.y;
    const int n = min(n_keys[b], K);
    for (int i = threadIdx.x; i < n; i += blockDim.x) s_keys[i] = keys[(size_t)b * K + i];
    __syncthreads();
    const bool vec = (P % 4 == 0) && ((uintptr_t)pan % 32 == 0) && ((uintptr_t)indices % 16 == 0);
    for (int p0 = (blockIdx.x * 256 + threadIdx.x) * 4; p0 - (int)threadIdx.x * 4 < P;
         p0 += gridDim.x * TG_PX_PER_BLOCK) {
        const int nvalid = max(0, min(4, P - p0));
        const size_t o = (size_t)b * P + p0;
        int64_t id[4] = {0, 0, 0, 0};
        if (nvalid == 4 && vec) {
            typedef long i64x4_t __attribute__((ext_vector_type(4)));
            const i64x4_t q = *(const i64x4_t*)(pan + o);
            id[0] = q.x; id[1] = q.y; id[2] = q.z; id[3] = q.w;
        } else {
            for (int j = 0; j < nvalid; ++j) id[j] = pan[o + j];
        }
        int idx[4] = {0, 0, 0, 0};
        for (int j = 0; j < 4; ++j) {
            const bool repeat = j > 0 && id[j] == id[j - 1];
            if (repeat) idx[j] = idx[j - 1];
            if (__all(repeat || j >= nvalid)) continue;
            unsigned long long todo = __ballot(!repeat && j < nvalid);
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                const long long want = __shfl((long long)id[j], leader);
                int found = 0;
                for (int c0 = 0; c0 < n; c0 += 64) {
                    const int i = c0 + lane_id();
                    const unsigned long long m = __ballot(i < n && s_keys[i] == want);
                    if (m) found = c0 + 64 - __clzll((long long)m);       // last match wins
                }
                const bool mine = !repeat && j < nvalid && id[j] == want;
                if (mine) idx[j] = found;
                todo &= ~__ballot(mine);
            }
        }
        if (nvalid == 4 && vec) {
            *(int4*)(indices + o) = make_int4(idx[0], idx[1], idx[2], idx[3]);
        } else {
            for (int j = 0; j < nvalid; ++j) indices[o + j] = idx[j];
        }
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

// ---- orientation sums for sparse (ground-truth) instance ids ------------------------------------
// InstancePostprocessing._get_instance_orientation (model/postprocessing/instance.py:271-319)
// with dataset instance maps (uint16 ids): ids are ranked like above, the biternion sums are
// taken per dense slot (fp64, LDS-privatised per workgroup).
__global__ __launch_bounds__(256) void k_ow_presence(
    const void* __restrict__ ins, int ins_dtype, const uint8_t* __restrict__ mask, int P, int cap,
    unsigned char* __restrict__ ws, int* __restrict__ status)
{
    __shared__ int s_ids[TG_H1];
    const int b = blockIdx.y;
    TgView v = tg_view(ws, b, cap, 1);
    if (threadIdx.x < TG_H1) s_ids[threadIdx.x] = -1;
    __syncthreads();
    bool bad = false;
    const int stride = gridDim.x * blockDim.x;
    const int trips = (P + stride - 1) / stride;
    for (int k = 0; k < trips; ++k) {
        const int p = (k * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        int key = -1;
        if (p < P) {
            const size_t o = (size_t)b * P + p;
            const int64_t i = mw_load(ins, ins_dtype, o);
            if (i < 0 || i > MW_MAX_ID) bad = true;
            else if (i > 0 && (!mask || mask[o])) key = (int)i;
        }
        int rl, rlast;
        if (wave_run_head(key, rl, rlast) && lds_hash_slot(s_ids, TG_H1, key) < 0)
            atomicOr(&v.bitmap[key >> 5], 1u << (key & 31));
    }
    __syncthreads();
    if (threadIdx.x < TG_H1 && s_ids[threadIdx.x] >= 0) {
        const int id = s_ids[threadIdx.x];
        atomicOr(&v.bitmap[id >> 5], 1u << (id & 31));
    }
    if (bad) atomicOr(status, TG_ST_ID_RANGE);
}

__global__ __launch_bounds__(256) void k_ow_sums(
    const float* __restrict__ orientation, const void* __restrict__ ins, int ins_dtype,
    const uint8_t* __restrict__ mask, int P, int cap, unsigned char* __restrict__ ws,
    double* __restrict__ sums, int32_t* __restrict__ count)
{
    __shared__ int s_key[TG_H1];
    __shared__ double s_sum[TG_H1 * 2];
    __shared__ int s_cnt[TG_H1];
    const int b = blockIdx.y;
    TgView v = tg_view(ws, b, cap, 1);
    if (threadIdx.x < TG_H1) {
        s_key[threadIdx.x] = -1; s_cnt[threadIdx.x] = 0;
        s_sum[2 * threadIdx.x] = 0.0; s_sum[2 * threadIdx.x + 1] = 0.0;
    }
    __syncthreads();
    const float* o0 = orientation + (size_t)b * 2 * P;
    const float* o1 = o0 + P;
    double* gs = sums + (size_t)b * cap * 2;
    int32_t* gc = count + (size_t)b * cap;
    const int stride = gridDim.x * blockDim.x;
    const int trips = (P + stride - 1) / stride;
    for (int k = 0; k < trips; ++k) {
        const int p = (k * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        int d = -1;
        double v0 = 0.0, v1 = 0.0;
        if (p < P) {
            const size_t o = (size_t)b * P + p;
            const int64_t i = mw_load(ins, ins_dtype, o);
            if (i > 0 && i <= MW_MAX_ID && (!mask || mask[o])) {
                const int dd = id_rank_dense(v.bitmap, v.prefix, (int)i);
                if (dd < cap) { d = dd; v0 = (double)o0[p]; v1 = (double)o1[p]; }
            }
        }
        // equal ids come in runs of lanes: the run head resolves the LDS slot once and passes it
        // down its run; every lane then adds its fp64 values with LDS atomics (same-address lanes
        // serialise in the LDS unit — cheaper than fp64 wave reductions per distinct id)
        int run_len, run_last;
        const bool head = wave_run_head(d, run_len, run_last);
        int slot = -1;
        if (head) slot = lds_hash_slot(s_key, TG_H1, d);
        // lane l belongs to the run whose head is the last head at or before l
        const unsigned long long heads = __ballot(head);
        const unsigned long long upto = heads & ((2ull << lane_id()) - 1ull);
        const int my_head = upto ? 63 - __clzll((long long)upto) : 0;
        slot = __shfl(slot, my_head);
        if (d >= 0) {
            if (slot >= 0) {
                atomicAdd(&s_sum[2 * slot], v0); atomicAdd(&s_sum[2 * slot + 1], v1);
            } else {
                atomicAdd(&gs[2 * d], v0); atomicAdd(&gs[2 * d + 1], v1);
            }
            if (head) {
                if (slot >= 0) atomicAdd(&s_cnt[slot], run_len);
                else atomicAdd(&gc[d], run_len);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < TG_H1 && s_key[threadIdx.x] >= 0) {
        const int kd = s_key[threadIdx.x];
        atomicAdd(&gs[2 * kd], s_sum[2 * threadIdx.x]);
        atomicAdd(&gs[2 * kd + 1], s_sum[2 * threadIdx.x + 1]);
        atomicAdd(&gc[kd], s_cnt[threadIdx.x]);
    }
}

__global__ __launch_bounds__(256) void k_ow_export(unsigned char* __restrict__ ws, int cap,
                                                   int32_t* __restrict__ ids, int32_t* __restrict__ n_ids)
{
    const int b = blockIdx.x;
    TgView v = tg_view(ws, b, cap, 1);
    const int n = v.counters[0];
    for (int i = threadIdx.x; i < n; i += blockDim.x) ids[(size_t)b * cap + i] = v.id_of_dense[i];
    if (threadIdx.x == 0) n_ids[b] = n;
}

int tg_cap(int max_instances) { return ((max_instances + 1023) / 1024) * 1024; }

// the vectorised label loaders apply to the on-wire dtypes with 4-pixel aligned images
bool tg_fast(const void* sem, int sem_dtype, const void* ins, int ins_dtype, int P)
{
    return sem_dtype == NMSA_U8 && ins_dtype == NMSA_I32 && P % 4 == 0 &&
           (uintptr_t)sem % 4 == 0 && (uintptr_t)ins % 16 == 0;
}

int tg_grid_x(int P)
{
    // 1024-px chunks per workgroup: 2 halve the table set-up / flush per pixel (instance targets
    // 108 -> 99 us at B=32 640x480; 4 the same, 8 slower)
    static const int iters = getenv("NMSA_TG_ITERS") ? atoi(getenv("NMSA_TG_ITERS")) : 2;
    int gx = (P + TG_PX_PER_BLOCK * iters - 1) / (TG_PX_PER_BLOCK * iters);
    return gx > 1024 ? 1024 : gx;
}

// the one-launch front end takes the on-wire layout; NMSA_TG_FUSED=0 (read per call: same-process
// A/B): never
bool tg_scan_ok(const void* sem, int sem_dtype, const void* ins, int ins_dtype, int P, int W, int NC)
{
    const char* e = getenv("NMSA_TG_FUSED");
    if (e && atoi(e) == 0) return false;
    return tg_fast(sem, sem_dtype, ins, ins_dtype, P) && W % 4 == 0 && NC <= TGF_MAX_NC;
}

size_t tg_scan_lds_bytes(int cap, bool moments)
{
    // scratch | bitmap + prefix | slot u16 [cap] | enc u8 [cap] (instance targets) or base i32 [cap] (naive merge)
    const size_t tail = 32 * 4 + (size_t)2 * MW_WORDS * 4 + (size_t)cap * 2 + (size_t)cap * (moments ? 1 : 4);
    return 16 + (((tail > sizeof(TgScanTab) ? tail : sizeof(TgScanTab)) + 15) & ~(size_t)15);
}

// pixels per workgroup: ONE round of resident workgroups over the batch (8 workgroups of 256
// threads per CU, fewer when the LDS footprint says so), whole rounds of 1024 px, at most 2^17 px
int tg_scan_px_per_wg(int B, int P, size_t lds, int resident_per_cu)
{
    const DeviceGeometry g = device_geometry();
    long long per_cu = (long long)(g.lds_per_cu / lds);
    if (per_cu > 8) per_cu = 8;
    if (resident_per_cu > 0 && per_cu > resident_per_cu) per_cu = resident_per_cu;   // (registers: the occupancy query)
    if (per_cu < 1) per_cu = 1;
    const char* e = getenv("NMSA_TG_WGS_PER_CU");       // (per call: tuning)
    if (e && atoi(e) > 0) per_cu = atoi(e);
    long long wpi = (long long)g.cus * per_cu / B;     // workgroups per image
    if (wpi < 1) wpi = 1;
    long long px = ((P + wpi - 1) / wpi + 1023) / 1024 * 1024;
    if (px > (1 << 17)) px = 1 << 17;
    if (px < 1024) px = 1024;
    return (int)px;
}

int tg_common(const void* sem, int sem_dtype, const void* ins, int ins_dtype, int B, int NC, int P,
              int W, int cap, bool moments, unsigned char* ws, size_t need, int32_t* status,
              hipStream_t stream, const uint8_t* is_thing_class = nullptr, int32_t* encoded_ids = nullptr,
              int32_t* n_encoded = nullptr, int32_t* skipped_ids = nullptr, int32_t* n_skipped = nullptr,
              int pair_cap = 0, int64_t max_inst = 0, int64_t* ids_pan = nullptr, int64_t* ids_ins = nullptr,
              int32_t* n_ids = nullptr, bool* did_tail = nullptr, int workspace_is_clean = 0)
{
    if (did_tail) *did_tail = false;
    if (tg_scan_ok(sem, sem_dtype, ins, ins_dtype, P, W, NC) && (uintptr_t)ws % 16 == 0) {
        // ONE launch behind a memset of the hash tables (k_tg_scan: scan + rank + decide / naive ranks)
        unsigned char* hs = ws + (size_t)B * tg_image_bytes(cap, NC);
        // a workspace the previous call left behind is all zero again (the tails clean what the
        // scan dirtied): no memset then.  The status word is SET by the call's last tail.
        const size_t hbytes = (size_t)B * tg_hash_bytes(cap, NC) + TG_GLOBAL_BYTES;
        int rc = NMSA_OK;
        if (!workspace_is_clean && (rc = check_hip(hipMemsetAsync(hs, 0, hbytes, stream)))) return rc;
        const size_t lds = tg_scan_lds_bytes(cap, moments);
        const bool small = tg_hash_slots(cap) <= 1024;
        // ONE round of resident workgroups: what the registers and the LDS of this instantiation admit
        int resident = 0;
        {
            hipError_t e = hipErrorUnknown;
            if (moments) e = small ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, k_tg_scan<true, 4>, 256, lds)
                                   : hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, k_tg_scan<true, 16>, 256, lds);
            else e = small ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, k_tg_scan<false, 4>, 256, lds)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, k_tg_scan<false, 16>, 256, lds);
            if (e != hipSuccess) { resident = 0; (void)hipGetLastError(); }
            // (the occupancy query reads one workgroup per CU high for SGPR-heavy 256-thread kernels —
            // 106 SGPRs here: 6 admitted, 7 reported, MI355X_MICROARCH.md "Residency"; a workgroup that
            // is not resident starts 12 us late and the whole launch ends that much later)
            if (resident > 2) resident -= 1;
        }
        const int px = tg_scan_px_per_wg(B, P, lds, resident);
        const dim3 grid((P + px - 1) / px, B);
#define NMSA_LAUNCH_SCAN(M, K) do { rc = allow_dynamic_lds(k_tg_scan<M, K>, lds); if (rc) return rc;            \
        hipLaunchKernelGGL((k_tg_scan<M, K>), grid, dim3(256), lds, stream, (const uint8_t*)sem, (const int32_t*)ins, \
                           P, W, cap, NC, px, ws, hs, is_thing_class, encoded_ids, n_encoded, skipped_ids,       \
                           n_skipped, pair_cap, max_inst, ids_pan, ids_ins, n_ids, status); } while (0)
        if (moments) { if (small) NMSA_LAUNCH_SCAN(true, 4); else NMSA_LAUNCH_SCAN(true, 16); }
        else { if (small) NMSA_LAUNCH_SCAN(false, 4); else NMSA_LAUNCH_SCAN(false, 16); }
#undef NMSA_LAUNCH_SCAN
        if (did_tail) *did_tail = true;
        return check_launch();
    }
    int rc = check_hip(hipMemsetAsync(ws, 0, (size_t)B * tg_image_bytes(cap, NC), stream));
    if (rc) return rc;
    const int gx = tg_grid_x(P);
    const bool fast = tg_fast(sem, sem_dtype, ins, ins_dtype, P);
    if (fast)
        hipLaunchKernelGGL(k_tg_presence<true>, dim3(gx, B), dim3(256), 0, stream, ins, ins_dtype, P, cap,
                           NC, ws, status);
    else
        hipLaunchKernelGGL(k_tg_presence<false>, dim3(gx, B), dim3(256), 0, stream, ins, ins_dtype, P, cap,
                           NC, ws, status);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(k_tg_rank, dim3(B), dim3(1024), 0, stream, ws, cap, NC, status);
    if ((rc = check_launch())) return rc;
#define NMSA_LAUNCH_STATS(M, F)                                                                     \
    hipLaunchKernelGGL((k_tg_stats<M, F>), dim3(gx, B), dim3(256), 0, stream, sem, sem_dtype, ins,  \
                       ins_dtype, P, W, cap, NC, ws, status)
    if (moments) { if (fast) NMSA_LAUNCH_STATS(true, true); else NMSA_LAUNCH_STATS(true, false); }
    else { if (fast) NMSA_LAUNCH_STATS(false, true); else NMSA_LAUNCH_STATS(false, false); }
#undef NMSA_LAUNCH_STATS
    return check_launch();
}

bool tg_bad_dtype(int d) { return d < NMSA_U8 || d > NMSA_I64; }

}  // namespace
}  // namespace nmsa

using namespace nmsa;

extern "C" size_t nmsa_targets_workspace_bytes(int B, int n_classes, int max_instances)
{
    if (B <= 0 || n_classes <= 0 || max_instances <= 0 || max_instances > 4096) return 0;
    // [ dense per-image tables (TgView) | per-image hash tables of the one-launch front end (TgHash) ]
    const int cap = tg_cap(max_instances);
    return (size_t)B * (tg_image_bytes(cap, n_classes) + tg_hash_bytes(cap, n_classes)) + TG_GLOBAL_BYTES;
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
                                     int workspace_is_clean, nmsa_stream_t stream_)
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
    bool decided = false;
    int rc = tg_common(semantic, sem_dtype, instance, ins_dtype, B, n_classes, P, W, cap, true, ws, need,
                       status, stream, is_thing_class, encoded_ids, n_encoded, skipped_ids, n_skipped, 0, 0,
                       nullptr, nullptr, nullptr, &decided, workspace_is_clean);
    if (rc) return rc;
    if (!decided) {
        hipLaunchKernelGGL(k_tg_decide, dim3(B), dim3(1024), 0, stream, ws, cap, n_classes, is_thing_class,
                           encoded_ids, n_encoded, skipped_ids, n_skipped);
        if ((rc = check_launch())) return rc;
    }
    const int radius = 3 * sigma + 1;
    const int lut_n = 2 * radius * radius + 1;
    const size_t lds = (size_t)cap * 2 * sizeof(int) + (lut_n <= TGP_LUT_LDS ? (size_t)lut_n * 4 : 0);
    dim3 grid((P + TGP_PX - 1) / TGP_PX, B);
    const bool fast = tg_fast(semantic, sem_dtype, instance, ins_dtype, P) &&
                      (uintptr_t)center % 16 == 0 && (uintptr_t)offset % 16 == 0 &&
                      (uintptr_t)foreground % 4 == 0 && (uintptr_t)center_mask % 4 == 0;
#define NMSA_LAUNCH_TGP(N, F)                                                                          \
    hipLaunchKernelGGL((k_tg_paint<N, F>), grid, dim3(TGP_THREADS), lds, stream, semantic, sem_dtype,  \
                       instance, ins_dtype, is_stuff_class, gauss_lut, lut_n, radius, H, W, cap,       \
                       n_classes, ws, center, offset, foreground, center_mask)
    static const int tiled = getenv("NMSA_TG_PAINT_TILED") ? atoi(getenv("NMSA_TG_PAINT_TILED")) : 2;
    if (fast && tiled && W % 4 == 0) {
        // 128 x (8 G) pixel tiles, G = NMSA_TG_PAINT_TILED groups of 4 pixels per thread
        const int G = tiled >= 4 ? 4 : tiled >= 2 ? 2 : 1;
        const int tiles_x = (W + TGT_W - 1) / TGT_W, tiles_y = (H + TGT_ROWS * G - 1) / (TGT_ROWS * G);
        const dim3 tgrid(tiles_x * tiles_y, B);
        const size_t tlds = (size_t)cap * 2 * sizeof(int) + (lut_n <= TGP_LUT_LDS ? (size_t)lut_n * 4 : 0);
#define NMSA_LAUNCH_TGT(N, GG)                                                                          \
        hipLaunchKernelGGL((k_tg_paint_tile<N, GG>), tgrid, dim3(TGP_THREADS), tlds, stream,            \
                           (const uint8_t*)semantic, (const int32_t*)instance, is_stuff_class, gauss_lut, \
                           lut_n, radius, H, W, tiles_x, cap, n_classes, ws, center, offset, foreground,  \
                           center_mask)
        if (normalized_offset) { if (G == 4) NMSA_LAUNCH_TGT(true, 4); else if (G == 2) NMSA_LAUNCH_TGT(true, 2); else NMSA_LAUNCH_TGT(true, 1); }
        else { if (G == 4) NMSA_LAUNCH_TGT(false, 4); else if (G == 2) NMSA_LAUNCH_TGT(false, 2); else NMSA_LAUNCH_TGT(false, 1); }
#undef NMSA_LAUNCH_TGT
        return check_launch();
    }
    if (normalized_offset) { if (fast) NMSA_LAUNCH_TGP(true, true); else NMSA_LAUNCH_TGP(true, false); }
    else { if (fast) NMSA_LAUNCH_TGP(false, true); else NMSA_LAUNCH_TGP(false, false); }
#undef NMSA_LAUNCH_TGP
    return check_launch();
}

extern "C" int nmsa_panoptic_targets(const void* semantic, int sem_dtype, const void* instance,
                                     int ins_dtype, const uint8_t* is_thing_class,
                                     int B, int n_classes, int H, int W,
                                     int64_t max_instances_per_category, int64_t void_label,
                                     int max_instances, int max_segments,
                                     int64_t* panoptic, int64_t* ids_pan, int64_t* ids_ins, int32_t* n_ids,
                                     int32_t* status, void* workspace, size_t workspace_bytes,
                                     int workspace_is_clean, nmsa_stream_t stream_)
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
    bool ranked = false;
    int rc = tg_common(semantic, sem_dtype, instance, ins_dtype, B, n_classes, P, W, cap, false, ws, need,
                       status, stream, nullptr, nullptr, nullptr, nullptr, nullptr, max_segments,
                       max_instances_per_category, ids_pan, ids_ins, n_ids, &ranked, workspace_is_clean);
    if (rc) return rc;
    if (!ranked) {
        hipLaunchKernelGGL(k_tg_naive_ranks, dim3(B), dim3(1024), 0, stream, ws, cap, n_classes, max_segments,
                           max_instances_per_category, ids_pan, ids_ins, n_ids, status);
        if ((rc = check_launch())) return rc;
    }
    const int gx = tg_grid_x(P);
    if (tg_fast(semantic, sem_dtype, instance, ins_dtype, P) && (uintptr_t)panoptic % 32 == 0)
        hipLaunchKernelGGL(k_tg_naive_paint<true>, dim3(gx, B), dim3(256), 0, stream, semantic, sem_dtype,
                           instance, ins_dtype, is_thing_class, P, cap, n_classes,
                           max_instances_per_category, void_label, ws, panoptic);
    else
        hipLaunchKernelGGL(k_tg_naive_paint<false>, dim3(gx, B), dim3(256), 0, stream, semantic, sem_dtype,
                           instance, ins_dtype, is_thing_class, P, cap, n_classes,
                           max_instances_per_category, void_label, ws, panoptic);
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
    const int gx = tg_grid_x(P);
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

extern "C" int nmsa_instance_orientation_wide(const float* orientation, const void* instance,
                                              int ins_dtype, const uint8_t* mask,
                                              int B, int H, int W, int max_instances,
                                              int32_t* ids, int32_t* n_ids, double* sums, int32_t* count,
                                              int32_t* status, void* workspace, size_t workspace_bytes,
                                              nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!orientation || !instance || !ids || !n_ids || !sums || !count || !status || !workspace)
        return NMSA_ERR_ARG;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || (int64_t)H * W > ((int64_t)1 << 30)) return NMSA_ERR_ARG;
    if (max_instances <= 0 || max_instances > 4096 || tg_bad_dtype(ins_dtype)) return NMSA_ERR_ARG;
    const int cap = tg_cap(max_instances);
    const size_t need = nmsa_targets_workspace_bytes(B, 1, max_instances);
    if (workspace_bytes < need) return NMSA_ERR_WORKSPACE;
    if ((uintptr_t)workspace % 8) return NMSA_ERR_ARG;
    unsigned char* ws = (unsigned char*)workspace;
    const int P = H * W;
    int rc = check_hip(hipMemsetAsync(ws, 0, need, stream));
    if (rc) return rc;
    if ((rc = check_hip(hipMemsetAsync(sums, 0, (size_t)B * cap * 2 * sizeof(double), stream)))) return rc;
    if ((rc = check_hip(hipMemsetAsync(count, 0, (size_t)B * cap * sizeof(int32_t), stream)))) return rc;
    int gx = (P + 4095) / 4096;
    hipLaunchKernelGGL(k_ow_presence, dim3(gx, B), dim3(256), 0, stream, instance, ins_dtype, mask, P, cap,
                       ws, status);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(k_tg_rank, dim3(B), dim3(1024), 0, stream, ws, cap, 1, status);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(k_ow_sums, dim3(gx, B), dim3(256), 0, stream, orientation, instance, ins_dtype,
                       mask, P, cap, ws, sums, count);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(k_ow_export, dim3(B), dim3(256), 0, stream, ws, cap, ids, n_ids);
    return check_launch();
}

#ifdef NMSA_TG_STAMPS
extern "C" int nmsa_debug_tg_stamps(unsigned long long* host_dst, int n_words)
{
    return check_hip(hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(nmsa::g_tg_stamps),
                                         (size_t)n_words * sizeof(unsigned long long)));
}
extern "C" int nmsa_debug_tg_stamps_clear(void)
{
    void* p = nullptr;
    if (check_hip(hipGetSymbolAddress(&p, HIP_SYMBOL(nmsa::g_tg_stamps)))) return NMSA_ERR_LAUNCH;
    return check_hip(hipMemset(p, 0, sizeof(unsigned long long) * 8192 * 8));
}
#endif
