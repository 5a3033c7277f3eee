// scores.hip — the `compute_scores` branch of PanopticPostprocessing on gfx950
// (reference model/postprocessing/panoptic.py:171-239).
//
// The reference materialises softmax [B,C,H,W], gathers the probability of the panoptic
// class per pixel (take_along_dim), then loops in Python over every (image, instance):
// mask = (panoptic == pan_id); instance score -> mask; mean semantic score over mask;
// product -> mask.  Here:
//   k_scores_semantic   per pixel: p = softmax(logits)[panoptic class] WITHOUT the softmax
//                       tensor: the fused pass already produced the argmax class and its
//                       probability 1/sum(exp(x - max)), so
//                       p = exp(x[class] - x[argmax]) * p_max   (2 gathered logits / pixel);
//                       per-instance sum (fp64) and count of p over the instance's painted
//                       pixels: LDS-privatised [256] tables, one global atomic per used id
//   k_scores_paint      mean = sum / count per instance (LDS table), then per pixel
//                       instance score = center score, panoptic score = mean * center score
//                       on painted pixels, else 0 / the semantic score
// "painted" = the pixel carries an instance id whose panoptic id is the pixel's panoptic id
// (what `panoptic_seg == pan_id` selects in the reference).
#include "nmsa_common.hpp"

namespace nmsa {
namespace {

constexpr int SC_THREADS = 256;
constexpr int SC_PX_PER_THREAD = 4;
constexpr int SC_PX_PER_BLOCK = SC_THREADS * SC_PX_PER_THREAD * 4;     // 4096 px per block

template <int DTYPE>
__device__ __forceinline__ float ld_logit(const void* p, size_t i)
{
    if (DTYPE == NMSA_F32) return ((const float*)p)[i];
    const uint16_t h = ((const uint16_t*)p)[i];
    return (DTYPE == NMSA_BF16) ? bf16_to_f32(h) : f16_to_f32(h);
}

__device__ __forceinline__ int64_t class_of(int64_t pan, int64_t mipc, int shift)
{
    return shift >= 0 ? (pan >> shift) : (pan / mipc);
}

// 4 consecutive pixels per thread; VEC = 4-pixel aligned images (wide loads / stores)
template <int DTYPE, bool VEC>
__global__ __launch_bounds__(SC_THREADS) void k_scores_semantic(
    const void* __restrict__ logits, const uint8_t* __restrict__ sem_idx,
    const float* __restrict__ sem_prob, const uint8_t* __restrict__ inst,
    const int64_t* __restrict__ pan, const int64_t* __restrict__ pan_of_inst,
    int C, int P, int64_t mipc, int shift,
    float* __restrict__ out_sem_score, double* __restrict__ sums, uint32_t* __restrict__ counts)
{
    __shared__ double s_sum[256];
    __shared__ uint32_t s_cnt[256];
    __shared__ int64_t s_pan[256];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 256; i += SC_THREADS) {
        s_sum[i] = 0.0;
        s_cnt[i] = 0u;
        s_pan[i] = pan_of_inst[(size_t)b * 256 + i];
    }
    __syncthreads();
    const size_t img = (size_t)b * P;
    const size_t img_logits = (size_t)b * C * P;
    const int begin = blockIdx.x * SC_PX_PER_BLOCK;
    const int end = min(begin + SC_PX_PER_BLOCK, P);
    for (int p0 = begin + threadIdx.x * 4; p0 < end; p0 += SC_THREADS * 4) {
        const int nvalid = min(4, end - p0);
        int64_t pn[4] = {0, 0, 0, 0};
        int am[4] = {0, 0, 0, 0}, id[4] = {0, 0, 0, 0};
        float pm[4] = {0.f, 0.f, 0.f, 0.f}, s[4];
        if (VEC) {
            const longlong2 a = *(const longlong2*)(pan + img + p0), c = *(const longlong2*)(pan + img + p0 + 2);
            pn[0] = a.x; pn[1] = a.y; pn[2] = c.x; pn[3] = c.y;
            const uchar4 m4 = *(const uchar4*)(sem_idx + img + p0), i4 = *(const uchar4*)(inst + img + p0);
            am[0] = m4.x; am[1] = m4.y; am[2] = m4.z; am[3] = m4.w;
            id[0] = i4.x; id[1] = i4.y; id[2] = i4.z; id[3] = i4.w;
            const float4 f = *(const float4*)(sem_prob + img + p0);
            pm[0] = f.x; pm[1] = f.y; pm[2] = f.z; pm[3] = f.w;
        } else {
            for (int j = 0; j < nvalid; ++j) {
                pn[j] = pan[img + p0 + j]; am[j] = sem_idx[img + p0 + j];
                id[j] = inst[img + p0 + j]; pm[j] = sem_prob[img + p0 + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t k = class_of(pn[j], mipc, shift);     // 0 = void, else class + 1
            s[j] = 0.f;
            if (j < nvalid && k > 0 && k <= C) {
                const int c = (int)k - 1;
                if (c == am[j]) {
                    s[j] = pm[j];
                } else {
                    const float xc = ld_logit<DTYPE>(logits, img_logits + (size_t)c * P + p0 + j);
                    const float xm = ld_logit<DTYPE>(logits, img_logits + (size_t)am[j] * P + p0 + j);
                    s[j] = __expf(xc - xm) * pm[j];
                }
            }
        }
        if (VEC) *(float4*)(out_sem_score + img + p0) = make_float4(s[0], s[1], s[2], s[3]);
        else for (int j = 0; j < nvalid; ++j) out_sem_score[img + p0 + j] = s[j];
        // per-instance sum / count of the painted pixels: one LDS atomic per lane when its 4
        // pixels belong to the same painted instance
        bool painted[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) painted[j] = j < nvalid && id[j] > 0 && s_pan[id[j]] == pn[j];
        if (painted[0] && painted[1] && painted[2] && painted[3] && id[0] == id[1] && id[1] == id[2] &&
            id[2] == id[3]) {
            atomicAdd(&s_sum[id[0]], (double)s[0] + (double)s[1] + (double)s[2] + (double)s[3]);
            atomicAdd(&s_cnt[id[0]], 4u);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (painted[j]) { atomicAdd(&s_sum[id[j]], (double)s[j]); atomicAdd(&s_cnt[id[j]], 1u); }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += SC_THREADS) {
        if (s_cnt[i]) {
            atomicAdd(&sums[(size_t)b * 256 + i], s_sum[i]);
            atomicAdd(&counts[(size_t)b * 256 + i], s_cnt[i]);
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(SC_THREADS) void k_scores_paint(
    const float* __restrict__ sem_score, const uint8_t* __restrict__ inst,
    const int64_t* __restrict__ pan, const int64_t* __restrict__ pan_of_inst,
    const float* __restrict__ inst_score_tab, const double* __restrict__ sums,
    const uint32_t* __restrict__ counts, int P,
    float* __restrict__ out_inst_score, float* __restrict__ out_pan_score,
    float* __restrict__ mean_sem)
{
    __shared__ float s_inst[256];
    __shared__ float s_prod[256];
    __shared__ int64_t s_pan[256];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 256; i += SC_THREADS) {
        const uint32_t n = counts[(size_t)b * 256 + i];
        // torch.mean of an empty selection is NaN (never read: no pixel is painted with it)
        const float mean = n ? (float)(sums[(size_t)b * 256 + i] / (double)n)
                             : __int_as_float(0x7fc00000);
        const float is = inst_score_tab[(size_t)b * 256 + i];
        s_inst[i] = is;
        s_prod[i] = __fmul_rn(mean, is);                  // panoptic.py:225 (f32 product)
        s_pan[i] = pan_of_inst[(size_t)b * 256 + i];
        if (blockIdx.x == 0 && mean_sem) mean_sem[(size_t)b * 256 + i] = mean;
    }
    __syncthreads();
    const size_t img = (size_t)b * P;
    const int begin = blockIdx.x * SC_PX_PER_BLOCK;
    const int end = min(begin + SC_PX_PER_BLOCK, P);
    for (int p0 = begin + threadIdx.x * 4; p0 < end; p0 += SC_THREADS * 4) {
        const int nvalid = min(4, end - p0);
        int64_t pn[4] = {0, 0, 0, 0};
        int id[4] = {0, 0, 0, 0};
        float ss[4] = {0.f, 0.f, 0.f, 0.f}, oi[4], op[4];
        if (VEC) {
            const longlong2 a = *(const longlong2*)(pan + img + p0), c = *(const longlong2*)(pan + img + p0 + 2);
            pn[0] = a.x; pn[1] = a.y; pn[2] = c.x; pn[3] = c.y;
            const uchar4 i4 = *(const uchar4*)(inst + img + p0);
            id[0] = i4.x; id[1] = i4.y; id[2] = i4.z; id[3] = i4.w;
            const float4 f = *(const float4*)(sem_score + img + p0);
            ss[0] = f.x; ss[1] = f.y; ss[2] = f.z; ss[3] = f.w;
        } else {
            for (int j = 0; j < nvalid; ++j) {
                pn[j] = pan[img + p0 + j]; id[j] = inst[img + p0 + j]; ss[j] = sem_score[img + p0 + j];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool painted = id[j] > 0 && s_pan[id[j]] == pn[j];
            oi[j] = painted ? s_inst[id[j]] : 0.f;
            op[j] = painted ? s_prod[id[j]] : ss[j];
        }
        if (VEC) {
            *(float4*)(out_inst_score + img + p0) = make_float4(oi[0], oi[1], oi[2], oi[3]);
            *(float4*)(out_pan_score + img + p0) = make_float4(op[0], op[1], op[2], op[3]);
        } else {
            for (int j = 0; j < nvalid; ++j) { out_inst_score[img + p0 + j] = oi[j]; out_pan_score[img + p0 + j] = op[j]; }
        }
    }
}

template <int DTYPE>
int launch_scores(const void* logits, const uint8_t* sem_idx, const float* sem_prob,
                  const uint8_t* inst, const int64_t* pan, const int64_t* pan_of_inst,
                  const float* inst_score_tab, int B, int C, int P, int64_t mipc,
                  float* out_sem, float* out_inst, float* out_pan, float* mean_sem,
                  double* sums, uint32_t* counts, hipStream_t stream)
{
    int shift = -1;
    if (mipc > 0 && (mipc & (mipc - 1)) == 0) shift = __builtin_ctzll((unsigned long long)mipc);
    int rc = check_hip(hipMemsetAsync(sums, 0, (size_t)B * 256 * sizeof(double), stream));
    if (rc) return rc;
    rc = check_hip(hipMemsetAsync(counts, 0, (size_t)B * 256 * sizeof(uint32_t), stream));
    if (rc) return rc;
    dim3 grid((P + SC_PX_PER_BLOCK - 1) / SC_PX_PER_BLOCK, B), block(SC_THREADS);
    const bool vec = P % 4 == 0 && (uintptr_t)pan % 16 == 0 && (uintptr_t)sem_idx % 4 == 0 &&
                     (uintptr_t)inst % 4 == 0 && (uintptr_t)sem_prob % 16 == 0 &&
                     ((uintptr_t)out_sem | (uintptr_t)out_inst | (uintptr_t)out_pan) % 16 == 0;
    if (vec) hipLaunchKernelGGL((k_scores_semantic<DTYPE, true>), grid, block, 0, stream, logits, sem_idx,
                                sem_prob, inst, pan, pan_of_inst, C, P, mipc, shift, out_sem, sums, counts);
    else hipLaunchKernelGGL((k_scores_semantic<DTYPE, false>), grid, block, 0, stream, logits, sem_idx,
                            sem_prob, inst, pan, pan_of_inst, C, P, mipc, shift, out_sem, sums, counts);
    rc = check_launch();
    if (rc) return rc;
    if (vec) hipLaunchKernelGGL(k_scores_paint<true>, grid, block, 0, stream, out_sem, inst, pan, pan_of_inst,
                                inst_score_tab, sums, counts, P, out_inst, out_pan, mean_sem);
    else hipLaunchKernelGGL(k_scores_paint<false>, grid, block, 0, stream, out_sem, inst, pan, pan_of_inst,
                            inst_score_tab, sums, counts, P, out_inst, out_pan, mean_sem);
    return check_launch();
}

}  // namespace
}  // namespace nmsa

using namespace nmsa;

extern "C" size_t nmsa_panoptic_scores_workspace_bytes(int B)
{
    return B > 0 ? (size_t)B * 256 * (sizeof(double) + sizeof(uint32_t)) : 0;
}

extern "C" int nmsa_panoptic_scores(const void* logits, int logits_dtype,
                                    const uint8_t* sem_idx, const float* sem_prob,
                                    const uint8_t* inst, const int64_t* pan,
                                    const int64_t* pan_of_inst, const float* inst_score_tab,
                                    int B, int C, int H, int W, int64_t max_instances_per_category,
                                    float* out_semantic_score, float* out_instance_score,
                                    float* out_panoptic_score, float* mean_semantic_score,
                                    void* workspace, size_t workspace_bytes, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!logits || !sem_idx || !sem_prob || !inst || !pan || !pan_of_inst || !inst_score_tab ||
        !out_semantic_score || !out_instance_score || !out_panoptic_score || !workspace)
        return NMSA_ERR_ARG;
    if (B <= 0 || B > 65535 || C <= 0 || C > 256 || H <= 0 || W <= 0 ||
        (int64_t)H * W > ((int64_t)1 << 30) || max_instances_per_category <= 0)
        return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_panoptic_scores_workspace_bytes(B)) return NMSA_ERR_WORKSPACE;
    if ((uintptr_t)workspace % 8) return NMSA_ERR_ARG;
    double* sums = (double*)workspace;
    uint32_t* counts = (uint32_t*)(sums + (size_t)B * 256);
    const int P = H * W;
    switch (logits_dtype) {
        case NMSA_F32:
            return launch_scores<NMSA_F32>(logits, sem_idx, sem_prob, inst, pan, pan_of_inst,
                                           inst_score_tab, B, C, P, max_instances_per_category,
                                           out_semantic_score, out_instance_score, out_panoptic_score,
                                           mean_semantic_score, sums, counts, stream);
        case NMSA_BF16:
            return launch_scores<NMSA_BF16>(logits, sem_idx, sem_prob, inst, pan, pan_of_inst,
                                            inst_score_tab, B, C, P, max_instances_per_category,
                                            out_semantic_score, out_instance_score, out_panoptic_score,
                                            mean_semantic_score, sums, counts, stream);
        case NMSA_F16:
            return launch_scores<NMSA_F16>(logits, sem_idx, sem_prob, inst, pan, pan_of_inst,
                                           inst_score_tab, B, C, P, max_instances_per_category,
                                           out_semantic_score, out_instance_score, out_panoptic_score,
                                           mean_semantic_score, sums, counts, stream);
        default: return NMSA_ERR_ARG;
    }
}
