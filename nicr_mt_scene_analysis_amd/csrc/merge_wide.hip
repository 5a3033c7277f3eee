// merge_wide.hip — deeplab merge for instance ids beyond uint8 (ground-truth side).
//
// reference: deeplab_merge_semantic_and_instance  utils/panoptic_merge.py:172-225 called
// with dataset instance maps (uint16 ids stored as int32, e.g. Hypersim with > 256
// instances per image — task_helper/instance.py:61, tests/test_merge.py:49-58).
//
// The prediction path uses uint8 ids and the direct-indexed kernels of panoptic.hip.
// Here ids 0..65535 are first ranked per image (presence bitmap -> prefix popcounts:
// ascending id order is preserved, which the running per-class counter of the merge
// depends on), then the same vote / assign / paint scheme runs on the dense ranks.
#include "nmsa_common.hpp"
#include "id_rank.hpp"

namespace nmsa {

struct MwView {
    uint32_t* bitmap;      // [MW_WORDS]
    uint32_t* prefix;      // [MW_WORDS] exclusive popcount prefix
    int32_t* id_of_dense;  // [cap]
    int64_t* pan_of_dense; // [cap]
    uint32_t* votes;       // [cap * NC]
};

__host__ __device__ inline size_t mw_image_bytes(int cap, int NC)
{
    size_t n = (size_t)MW_WORDS * 4 * 2 + (size_t)cap * 4 + (size_t)cap * 8 + (size_t)cap * NC * 4;
    return (n + 15) & ~(size_t)15;
}

__device__ __forceinline__ MwView mw_view(unsigned char* ws, int b, int cap, int NC)
{
    unsigned char* base = ws + (size_t)b * mw_image_bytes(cap, NC);
    MwView v;
    v.pan_of_dense = (int64_t*)base;
    v.bitmap = (uint32_t*)(v.pan_of_dense + cap);
    v.prefix = v.bitmap + MW_WORDS;
    v.id_of_dense = (int32_t*)(v.prefix + MW_WORDS);
    v.votes = (uint32_t*)(v.id_of_dense + cap);
    return v;
}

// thing pixel with a rankable id?  (is_thing = (ins > 0) & thing_seg, panoptic_merge.py:182)
__device__ __forceinline__ bool mw_is_thing(int64_t i, uint8_t th) { return i > 0 && th; }

__global__ __launch_bounds__(256) void k_mw_presence(
    const void* __restrict__ ins, int ins_dtype, const uint8_t* __restrict__ thing_seg,
    int P, int cap, int NC, unsigned char* __restrict__ ws, int* __restrict__ status)
{
    __shared__ int s_ids[64];               // the workgroup's ids (lds_hash_slot), flushed once
    const int b = blockIdx.y;
    MwView v = mw_view(ws, b, cap, NC);
    if (threadIdx.x < 64) s_ids[threadIdx.x] = -1;
    __syncthreads();
    const int stride = gridDim.x * blockDim.x;
    const int trips = (P + stride - 1) / stride;
    bool bad = false;
    for (int k = 0; k < trips; ++k) {
        const int p = (k * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        int key = -1;
        if (p < P) {
            const size_t o = (size_t)b * P + p;
            const int64_t i = mw_load(ins, ins_dtype, o);
            if (mw_is_thing(i, thing_seg[o])) {
                if (i > MW_MAX_ID) bad = true;
                else key = (int)i;
            }
        }
        int run_len, run_last;              // one insert per run of equal ids, all heads in parallel
        if (wave_run_head(key, run_len, run_last) && lds_hash_slot(s_ids, 64, key) < 0)
            atomicOr(&v.bitmap[key >> 5], 1u << (key & 31));
    }
    __syncthreads();
    if (threadIdx.x < 64 && s_ids[threadIdx.x] >= 0) {
        const int id = s_ids[threadIdx.x];
        atomicOr(&v.bitmap[id >> 5], 1u << (id & 31));
    }
    if (bad) atomicOr(status, MW_ST_ID_RANGE);
}

// one 1024-thread workgroup per image: rank = number of smaller present ids
__global__ __launch_bounds__(1024) void k_mw_rank(unsigned char* __restrict__ ws, int cap, int NC,
                                                  int* __restrict__ status)
{
    __shared__ int scratch[32];
    const int b = blockIdx.x, t = threadIdx.x;
    MwView v = mw_view(ws, b, cap, NC);
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
    if (t == 0 && total > cap) atomicOr(status, MW_ST_OVERFLOW);
}

__device__ __forceinline__ int mw_dense(const MwView& v, int id)
{
    const uint32_t w = v.bitmap[id >> 5];
    return (int)v.prefix[id >> 5] + __popc(w & ((1u << (id & 31)) - 1u));
}

__global__ __launch_bounds__(256) void k_mw_votes(
    const void* __restrict__ sem, int sem_dtype, const void* __restrict__ ins, int ins_dtype,
    const uint8_t* __restrict__ thing_seg, int P, int cap, int NC, unsigned char* __restrict__ ws)
{
    __shared__ int s_key[256];
    __shared__ uint32_t s_cnt[256];
    const int b = blockIdx.y;
    MwView v = mw_view(ws, b, cap, NC);
    s_key[threadIdx.x] = -1;
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int stride = gridDim.x * blockDim.x;
    const int trips = (P + stride - 1) / stride;
    for (int k = 0; k < trips; ++k) {
        const int p = (k * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        int key = -1;
        if (p < P) {
            const size_t o = (size_t)b * P + p;
            const int64_t i = mw_load(ins, ins_dtype, o);
            const int64_t s = mw_load(sem, sem_dtype, o);
            if (mw_is_thing(i, thing_seg[o]) && i <= MW_MAX_ID && s >= 0 && s < NC) {
                const int d = mw_dense(v, (int)i);
                if (d < cap) key = d * NC + (int)s;
            }
        }
        int run_len, run_last;              // the head of a run of equal keys adds the run length
        if (wave_run_head(key, run_len, run_last)) {
            const int slot = lds_hash_slot(s_key, 256, key);
            if (slot >= 0) atomicAdd(&s_cnt[slot], (uint32_t)run_len);
            else atomicAdd(&v.votes[key], (uint32_t)run_len);
        }
    }
    __syncthreads();
    if (s_key[threadIdx.x] >= 0 && s_cnt[threadIdx.x])
        atomicAdd(&v.votes[s_key[threadIdx.x]], s_cnt[threadIdx.x]);
}

// mode (smallest class on ties) + running per-class counter in ascending id order.  Any cap up
// to 65536 (every id the maps can hold): the class of every slot is parked in pan_of_dense between
// the two walks (not in per-thread arrays), the classes of the valid segments sit in LDS as u16.
__global__ __launch_bounds__(1024) void k_mw_assign(
    unsigned char* __restrict__ ws, int cap, int NC, int64_t max_inst, int64_t void_label,
    int64_t* __restrict__ ids_pan, int64_t* __restrict__ ids_ins, int32_t* __restrict__ n_ids)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t s_vcls[];   // [cap] classes of the valid segments, ascending id
    __shared__ int scratch[32];
    const int b = blockIdx.x, t = threadIdx.x;
    MwView v = mw_view(ws, b, cap, NC);
    const int per = cap / 1024;
    int nvalid = 0;
    for (int j = 0; j < per; ++j) {
        const int slot = t * per + j;
        const uint32_t* row = v.votes + (size_t)slot * NC;
        uint32_t total = 0;
        int64_t bestc = -1;
        int c_best = 0;
        for (int c = 0; c < NC; ++c) {
            const uint32_t x = row[c];
            total += x;
            if ((int64_t)x > bestc) { bestc = x; c_best = c; }
        }
        const bool valid = (total > 0) && (c_best != 0);     // empty mask / void majority are skipped
        v.pan_of_dense[slot] = valid ? (int64_t)c_best : -1;  // (this thread reads it back below)
        nvalid += valid;
    }
    int total_valid;
    int pos = mw_block_scan(nvalid, scratch, &total_valid) - nvalid;
    const int pos0 = pos;
    for (int j = 0; j < per; ++j) {
        const int64_t c = v.pan_of_dense[t * per + j];
        if (c >= 0) s_vcls[pos++] = (uint16_t)c;
    }
    __syncthreads();
    pos = pos0;
    for (int j = 0; j < per; ++j) {
        const int slot = t * per + j;
        const int64_t c = v.pan_of_dense[slot];
        int64_t pid = void_label;
        if (c >= 0) {
            int rank = 1;
            for (int k = 0; k < pos; ++k) rank += (s_vcls[k] == (uint16_t)c);
            pid = c * max_inst + rank;
            ids_pan[(size_t)b * cap + pos] = pid;
            ids_ins[(size_t)b * cap + pos] = v.id_of_dense[slot];
            ++pos;
        }
        v.pan_of_dense[slot] = pid;
    }
    if (t == 0) n_ids[b] = total_valid;
}

__global__ __launch_bounds__(256) void k_mw_paint(
    const void* __restrict__ sem, int sem_dtype, const void* __restrict__ ins, int ins_dtype,
    const uint8_t* __restrict__ thing_seg, const uint8_t* __restrict__ is_thing_class,
    int P, int cap, int NC, int64_t max_inst, int64_t void_label,
    unsigned char* __restrict__ ws, int64_t* __restrict__ pan)
{
    const int b = blockIdx.y;
    MwView v = mw_view(ws, b, cap, NC);
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const size_t o = (size_t)b * P + p;
        const int64_t i = mw_load(ins, ins_dtype, o);
        const int64_t s = mw_load(sem, sem_dtype, o);
        int64_t r = void_label;
        if (i != 0) {
            if (mw_is_thing(i, thing_seg[o]) && i <= MW_MAX_ID) {
                const int d = mw_dense(v, (int)i);
                if (d < cap) r = v.pan_of_dense[d];
            }
        } else if (s > 0 && s < NC && !is_thing_class[s]) {
            r = s * max_inst;                              // stuff paste (panoptic_merge.py:222-223)
        }
        pan[o] = r;
    }
}

}  // namespace nmsa

using namespace nmsa;

extern "C" size_t nmsa_panoptic_merge_wide_workspace_bytes(int B, int n_classes, int max_segments)
{
    if (B <= 0 || n_classes <= 0 || max_segments <= 0) return 0;
    const int cap = ((max_segments + 1023) / 1024) * 1024;
    if (cap > 65536 || n_classes > 65535) return 0;
    return (size_t)B * mw_image_bytes(cap, n_classes);
}

extern "C" int nmsa_panoptic_merge_wide(const void* sem, int sem_dtype, const void* ins, int ins_dtype,
                                        const uint8_t* thing_seg, const uint8_t* is_thing_class,
                                        int B, int n_classes, int H, int W,
                                        int64_t max_instances_per_category, int64_t void_label,
                                        int max_segments,
                                        int64_t* pan, int64_t* ids_pan, int64_t* ids_ins, int32_t* n_ids,
                                        int32_t* status, void* workspace, size_t workspace_bytes,
                                        nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!sem || !ins || !thing_seg || !is_thing_class || !pan || !ids_pan || !ids_ins || !n_ids ||
        !status || !workspace)
        return NMSA_ERR_ARG;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || (int64_t)H * W > ((int64_t)1 << 30)) return NMSA_ERR_ARG;
    if (n_classes <= 0 || n_classes > 65535 || max_segments <= 0 || max_segments > 65536) return NMSA_ERR_ARG;
    if (sem_dtype < NMSA_U8 || sem_dtype > NMSA_I64 || ins_dtype < NMSA_U8 || ins_dtype > NMSA_I64)
        return NMSA_ERR_ARG;
    const int cap = ((max_segments + 1023) / 1024) * 1024;        // ids_* are [B, cap]
    const size_t need = nmsa_panoptic_merge_wide_workspace_bytes(B, n_classes, max_segments);
    if (workspace_bytes < need) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    unsigned char* ws = (unsigned char*)workspace;
    int rc = check_hip(hipMemsetAsync(ws, 0, need, stream));
    if (rc) return rc;
    int gx = (P + 255) / 256;
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_mw_presence, dim3(gx, B), dim3(256), 0, stream, ins, ins_dtype, thing_seg, P,
                       cap, n_classes, ws, status);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(k_mw_rank, dim3(B), dim3(1024), 0, stream, ws, cap, n_classes, status);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(k_mw_votes, dim3(gx, B), dim3(256), 0, stream, sem, sem_dtype, ins, ins_dtype,
                       thing_seg, P, cap, n_classes, ws);
    if ((rc = check_launch())) return rc;
    if ((rc = allow_dynamic_lds(k_mw_assign, (size_t)cap * sizeof(uint16_t)))) return rc;    // (128 KB at 65536 ids)
    hipLaunchKernelGGL(k_mw_assign, dim3(B), dim3(1024), (size_t)cap * sizeof(uint16_t), stream, ws, cap,
                       n_classes, max_instances_per_category, void_label, ids_pan, ids_ins, n_ids);
    if ((rc = check_launch())) return rc;
    hipLaunchKernelGGL(k_mw_paint, dim3(gx, B), dim3(256), 0, stream, sem, sem_dtype, ins, ins_dtype,
                       thing_seg, is_thing_class, P, cap, n_classes, max_instances_per_category,
                       void_label, ws, pan);
    return check_launch();
}
