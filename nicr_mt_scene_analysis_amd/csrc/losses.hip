// losses.hip — per-pixel multi-task losses, forward and backward, for gfx950.
//
// Replaces the ATen chains of the reference's loss classes together with the
// masking conventions of its task helpers (so that no `pred*mask`, boolean
// gather or permute copy is materialised):
//   CrossEntropyLossSemantic._compute_loss   loss/ce.py:40-68
//   MSELoss / L1Loss ._compute_loss          loss/mse.py:21-41, loss/l1.py:21-41
//       + masking of task_helper/instance.py:129-139 (center), :154-167 (offset)
//   VonMisesLossBiternion._compute_loss      loss/vonmises.py:27-51
//       + boolean gather of task_helper/instance.py:186-216
//   CosineEmbeddingLoss._compute_loss        loss/cos_emb.py:21-56
//       + LUT gather of task_helper/dense_visual_embedding.py:110-171
//
// Every kernel is a memory-bound stream: predictions are read once (f32, bf16 or
// f16; 4 px per lane), arithmetic and accumulation are fp32 per pixel, block
// partials are fp64 and are summed in a FIXED order by k_loss_finalize, so results
// are run-to-run deterministic.  Backward kernels recompute from the inputs and
// scale by the upstream gradient read from a device scalar (no host sync).
#include <stdlib.h>
#include "loss_bodies.hpp"

namespace nmsa {

// one workgroup sums the block partials in a FIXED order (thread t takes partials t, t + 1024,
// ...; then a tree over the threads): deterministic.  1024 threads with 4 loads in flight each —
// the 9600 partials of a B=64 640x480 cross entropy took 12 us with 256 serial threads.
constexpr int FIN_THREADS = 1024;
__global__ __launch_bounds__(FIN_THREADS) void k_loss_finalize(
    const LossPartial* __restrict__ partials, int n, double* __restrict__ out_sum,
    double* __restrict__ out_aux, long long* __restrict__ out_count)
{
    __shared__ double s_sum[FIN_THREADS], s_aux[FIN_THREADS];
    __shared__ long long s_cnt[FIN_THREADS];
    double a = 0, b = 0; long long c = 0;
    int i = threadIdx.x;
    for (; i + 3 * FIN_THREADS < n; i += 4 * FIN_THREADS) {
        LossPartial p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = partials[i + u * FIN_THREADS];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a += p[u].sum; b += p[u].aux; c += p[u].count; }
    }
    for (; i < n; i += FIN_THREADS) { a += partials[i].sum; b += partials[i].aux; c += partials[i].count; }
    s_sum[threadIdx.x] = a; s_aux[threadIdx.x] = b; s_cnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = FIN_THREADS / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
            s_aux[threadIdx.x] += s_aux[threadIdx.x + o];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (out_sum) *out_sum = s_sum[0];
        if (out_aux) *out_aux = s_aux[0];
        if (out_count) *out_count = s_cnt[0];
    }
}

template <int DTYPE, int PXT, bool SMOOTH, int U>
__global__ __launch_bounds__(LOSS_THREADS) void k_ce_fwd(
    const void* __restrict__ logits, const uint8_t* __restrict__ target,
    const float* __restrict__ weights, int C, int P, float ls, int vec,
    LossPartial* __restrict__ partials, int* __restrict__ status, float* __restrict__ lse2_out, int wide)
{
    extern __shared__ float s_w[];
    ce_fwd_body<DTYPE, PXT, SMOOTH, U>(logits, target, weights, C, P, ls, vec,
                                       partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x, status,
                                       lse2_out, s_w, blockIdx.x, gridDim.x, blockIdx.y, wide);
}

// d loss_sum / d logits, times the upstream gradient *gscale  (ce.py via autograd)
//   grad_j = g * [ (a + bsum) p_j - a [j == t] - b_j ],  a = (1-ls) w_t, b_j = (ls/C) w_j
// pass 1: max / sum (as forward); pass 2 re-reads the tile (L2) and writes
// g*((a+bsum) p_j - b_j); the "- a" at the target class is one read-modify-write per px.
// LOSS: the same two walks also produce the forward sum (block partials as k_ce_fwd) — forward +
// gradient for an expected upstream scale in ONE launch for class counts whose column does not
// fit the registers of k_ce_fused (C > 48): the second walk re-reads the tile through the
// Infinity Cache at ~40 % of a first read (tools/diag_two_pass.py) instead of a second kernel
// reading it from HBM plus a log-sum-exp round trip.
template <int DTYPE, int PXT, bool SMOOTH, int UB, bool LOSS = false>
__global__ __launch_bounds__(LOSS_THREADS) void k_ce_bwd(
    const void* __restrict__ logits, const uint8_t* __restrict__ target,
    const float* __restrict__ weights, int C, int P, float ls, int vec,
    const float* __restrict__ gscale, void* __restrict__ grad, const float* __restrict__ lse2,
    const float* __restrict__ computed_for, int* __restrict__ counters,
    LossPartial* __restrict__ partials = nullptr, int* __restrict__ status = nullptr, int wide = 0)
{
    extern __shared__ float s_w[];
    if (!LOSS && grad_already_computed(gscale, computed_for, counters)) return;
    double acc = 0.0, accw = 0.0;
    long long cnt = 0;
    bool bad = false;
    for (int c = threadIdx.x; c < C; c += LOSS_THREADS) s_w[c] = weights ? weights[c] : 1.0f;
    __syncthreads();
    float wsum = 0.f;
    if (SMOOTH) for (int c = 0; c < C; ++c) wsum += s_w[c];
    const float g = *gscale;
    const int b = blockIdx.y;
    const size_t img = (size_t)b * C * P;
    constexpr int U = (PXT == 4) ? 8 : 4;              // first pass (no saved log-sum-exp)
    for (int p0 = (blockIdx.x * LOSS_THREADS + threadIdx.x) * PXT; p0 < P;
         p0 += gridDim.x * LOSS_THREADS * PXT) {
        const int nvalid = min(PXT, P - p0);
        float m[PXT], s[PXT], swx[PXT], xts[PXT];
        int t[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j)
            t[j] = (j < nvalid) ? ce_label(target, wide, (size_t)b * P + p0 + j) - 1 : -1;
        float ag[PXT], abg[PXT], k0[PXT];
        if (lse2) {
            // the forward pass left -log2(sum exp) per pixel: no first pass over the logits
#pragma unroll
            for (int j = 0; j < PXT; ++j) k0[j] = 0.f;
            const float* q = lse2 + (size_t)b * P + p0;
            if (vec && nvalid == PXT) {
#pragma unroll
                for (int j = 0; j < PXT; j += 4) {
                    const float4 f = *(const float4*)(q + j);
                    k0[j] = f.x; k0[j + 1] = f.y; k0[j + 2] = f.z; k0[j + 3] = f.w;
                }
            } else {
                for (int j = 0; j < nvalid; ++j) k0[j] = q[j];
            }
        } else {
            // pass 1 with regular loads (the tile may still be in L2 / MALL for pass 2)
            ce_scan<DTYPE, PXT, U, LOSS && SMOOTH, LOSS, false>(logits, img, P, p0, nvalid, vec, C,
                                                                s_w, t, m, s, swx, xts);
#pragma unroll
            for (int j = 0; j < PXT; ++j) k0[j] = -(fmaf(m[j], LOG2E, __log2f(s[j])));
            if (LOSS) {
                float part = 0.f, partw = 0.f;
#pragma unroll
                for (int j = 0; j < PXT; ++j) {
                    if (t[j] < 0) continue;                                 // void: ignore_index
                    if (t[j] >= C) { bad = true; continue; }
                    const float lse = fmaf(__log2f(s[j]), LN2, m[j]);
                    const float wt = s_w[t[j]];
                    float l = (1.0f - ls) * wt * (lse - xts[j]);
                    if (SMOOTH) l += (ls / C) * (lse * wsum - swx[j]);
                    part += l;
                    partw += wt;
                    ++cnt;
                }
                acc += part; accw += partw;
            }
        }
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const bool on = t[j] >= 0 && t[j] < C;
            const float a = on ? (1.0f - ls) * s_w[t[j]] : 0.f;
            ag[j] = g * a;
            abg[j] = on ? g * (a + (SMOOTH ? (ls / C) * wsum : 0.f)) : 0.f;   // p = 2^(x log2e + k0)
        }
        auto plane = [&](const float v[PXT], int c) {
            float o[PXT];
            const float bjg = SMOOTH ? g * (ls / C) * s_w[c] : 0.f;
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const float pj = __builtin_amdgcn_exp2f(fmaf(v[j], LOG2E, k0[j]));
                float r = fmaf(abg[j], pj, (SMOOTH && abg[j] != 0.f) ? -bjg : 0.f);
                r -= (t[j] == c) ? ag[j] : 0.f;
                o[j] = r;
            }
            stpx<DTYPE, PXT, GRAD_NT>(grad, img + (size_t)c * P + p0, nvalid, vec, o);
        };
        // UB plane loads in flight, then UB plane stores
        int c = 0;
        for (; c + UB <= C; c += UB) {
            float v[UB][PXT];
#pragma unroll
            for (int u = 0; u < UB; ++u)
                ldpx<DTYPE, PXT, true>(logits, img + (size_t)(c + u) * P + p0, nvalid, vec, v[u]);
#pragma unroll
            for (int u = 0; u < UB; ++u) plane(v[u], c + u);
        }
        for (; c < C; ++c) {
            float v[PXT];
            ldpx<DTYPE, PXT, true>(logits, img + (size_t)c * P + p0, nvalid, vec, v);
            plane(v, c);
        }
    }
    if (LOSS) {
        if (bad) atomicOr(status, 8);
        block_partial(acc, accw, cnt, partials);
    }
}

template <int DTYPE, int NG, bool SMOOTH, bool LOSS = true>
__global__ __launch_bounds__(LOSS_THREADS) void k_ce_fused(
    const void* __restrict__ logits, const uint8_t* __restrict__ target,
    const float* __restrict__ weights, int C, int P, float ls, int vec,
    const float* __restrict__ expected_gscale, void* __restrict__ grad,
    LossPartial* __restrict__ partials, int* __restrict__ status,
    const float* __restrict__ computed_for = nullptr, int* __restrict__ counters = nullptr)
{
    extern __shared__ float s_w[];
    if (!LOSS && grad_already_computed(expected_gscale, computed_for, counters)) return;
    ce_fused_body<DTYPE, NG, SMOOTH, LOSS ? 0 : 2>(logits, target, weights, C, P, ls, vec,
                                           grad ? *expected_gscale : __int_as_float(0x7fc00000), grad,
                                           partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x, status,
                                           s_w, blockIdx.x, blockIdx.y);
}

// number of bytes v with lo <= v <= hi (labels 1..C, mask bytes != 0): the element count a loss
// is going to be divided by, known BEFORE the loss kernel runs (1 B/px).  One partial per
// workgroup, summed by k_count_finalize (thousands of atomics on one address cost 10 ns each).
constexpr int COUNT_MAX_BLOCKS = 512;

__global__ __launch_bounds__(LOSS_THREADS) void k_count_u8(
    const uint8_t* __restrict__ v, long long n, int lo, int hi, int vec,
    long long* __restrict__ partials)
{
    __shared__ long long s_cnt[LOSS_THREADS / 64];
    long long cnt = 0;
    const long long stride = (long long)gridDim.x * LOSS_THREADS;
    const long long tid = (long long)blockIdx.x * LOSS_THREADS + threadIdx.x;
    const long long n16 = vec ? n / 16 : 0;
    const unsigned span = (unsigned)(hi - lo);
    auto count16 = [&](const u32x4_s w) {
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
        int c = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c += ((((ww[k] >> (8 * j)) & 0xFF) - (unsigned)lo) <= span);
        }
        return c;
    };
    long long i = tid;
    for (; i + 3 * stride < n16; i += 4 * stride) {            // 4 x 16 B in flight per lane
        u32x4_s w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = __builtin_nontemporal_load((const u32x4_s*)v + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) cnt += count16(w[u]);
    }
    for (; i < n16; i += stride) cnt += count16(__builtin_nontemporal_load((const u32x4_s*)v + i));
    for (long long k = n16 * 16 + tid; k < n; k += stride) cnt += (((unsigned)v[k] - (unsigned)lo) <= span);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long c = 0;
        for (int k = 0; k < LOSS_THREADS / 64; ++k) c += s_cnt[k];
        partials[blockIdx.x] = c;
    }
}

// *count = sum of the partials; *mean_scale (optional) = weight / count as fp32: the
// correctly rounded division ATen's `weight / count` performs
__global__ __launch_bounds__(COUNT_MAX_BLOCKS) void k_count_finalize(
    const long long* __restrict__ partials, int n, long long* __restrict__ count,
    float* __restrict__ mean_scale, float weight)
{
    __shared__ long long s_cnt[COUNT_MAX_BLOCKS / 64];
    long long c = (threadIdx.x < n) ? partials[threadIdx.x] : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long t = 0;
        for (int k = 0; k < COUNT_MAX_BLOCKS / 64; ++k) t += s_cnt[k];
        *count = t;
        if (mean_scale) *mean_scale = weight / (float)t;
    }
}

template <int DTYPE, int KIND /* 0 mse, 1 l1, 2 center focal */>
__global__ __launch_bounds__(LOSS_THREADS) void k_elem_fwd(
    const void* __restrict__ pred, const float* __restrict__ target, const uint8_t* __restrict__ mask,
    int C, int P, int vec, LossPartial* __restrict__ partials)
{
    const int b = blockIdx.y;
    double acc = 0.0; long long cnt = 0;
    const float invC = 1.0f / C;
    for (int p0 = (blockIdx.x * LOSS_THREADS + threadIdx.x) * 4; p0 < P; p0 += gridDim.x * LOSS_THREADS * 4) {
        const int nvalid = min(4, P - p0);
        bool mk[4];
        ld_mask4(mask, (size_t)b * P + p0, nvalid, vec, mk);
        float part = 0.f;
        for (int c = 0; c < C; ++c) {
            const size_t off = ((size_t)b * C + c) * P + p0;
            const float4 x = ld4<DTYPE>(pred, off, nvalid, vec);
            const float4 y = ld4<NMSA_F32>(target, off, nvalid, vec);
            const float xv[4] = {x.x, x.y, x.z, x.w}, yv[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= nvalid) continue;
                if (KIND == 2) {
                    if (mk[j]) { part += focal_value(xv[j], yv[j]); cnt += (yv[j] == 1.0f); }
                    continue;
                }
                const float d = (mk[j] ? xv[j] : 0.f) - yv[j];       // pred*mask - target
                part += (KIND == 0) ? d * d : fabsf(d);
            }
        }
        acc += part * invC;
        if (KIND != 2) for (int j = 0; j < 4; ++j) cnt += (j < nvalid) && mk[j];
    }
    block_partial(acc, 0.0, cnt, partials);
}

template <int DTYPE, int KIND>
__global__ __launch_bounds__(LOSS_THREADS) void k_elem_bwd(
    const void* __restrict__ pred, const float* __restrict__ target, const uint8_t* __restrict__ mask,
    int C, int P, int vec, const float* __restrict__ gscale, void* __restrict__ grad,
    const float* __restrict__ computed_for, int* __restrict__ counters)
{
    if (grad_already_computed(gscale, computed_for, counters)) return;
    const int b = blockIdx.y;
    const float g = *gscale / C;
    for (int p0 = (blockIdx.x * LOSS_THREADS + threadIdx.x) * 4; p0 < P; p0 += gridDim.x * LOSS_THREADS * 4) {
        const int nvalid = min(4, P - p0);
        bool mk[4];
        ld_mask4(mask, (size_t)b * P + p0, nvalid, vec, mk);
        for (int c = 0; c < C; ++c) {
            const size_t off = ((size_t)b * C + c) * P + p0;
            const float4 x = ld4<DTYPE>(pred, off, nvalid, vec);
            const float4 y = ld4<NMSA_F32>(target, off, nvalid, vec);
            const float xv[4] = {x.x, x.y, x.z, x.w}, yv[4] = {y.x, y.y, y.z, y.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = (mk[j] ? xv[j] : 0.f) - yv[j];
                const float dd = (KIND == 2) ? focal_grad(xv[j], yv[j])
                               : (KIND == 0) ? 2.0f * d : (float)((d > 0.f) - (d < 0.f));
                o[j] = mk[j] ? g * dd : 0.f;                          // through the mask multiply
            }
            st4<DTYPE>(grad, off, nvalid, vec, o);
        }
    }
}

template <int DTYPE, int KIND>
__global__ __launch_bounds__(LOSS_THREADS) void k_elem_fused(
    const void* __restrict__ pred, const float* __restrict__ target, const uint8_t* __restrict__ mask,
    int C, int P, int vec, const float* __restrict__ expected_gscale, void* __restrict__ grad,
    LossPartial* __restrict__ partials)
{
    elem_fused_body<DTYPE, KIND, 0>(pred, target, mask, C, P, vec, *expected_gscale, grad,
                                       partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x,
                                       blockIdx.x, gridDim.x, blockIdx.y);
}

// =================================================================================
// a8: von Mises (biternion):  sum over masked px of 1 - exp(kappa (x0 y0 + x1 y1 - 1))
//   planar layout [B,2,P] + mask (the task helper's gather folded in)
// =================================================================================
template <int DTYPE>
__global__ __launch_bounds__(LOSS_THREADS) void k_vm_fwd(
    const void* __restrict__ pred, const float* __restrict__ target, const uint8_t* __restrict__ mask,
    int P, float kappa, int vec, LossPartial* __restrict__ partials)
{
    const int b = blockIdx.y;
    double acc = 0.0; long long cnt = 0;
    for (int p0 = (blockIdx.x * LOSS_THREADS + threadIdx.x) * 4; p0 < P; p0 += gridDim.x * LOSS_THREADS * 4) {
        const int nvalid = min(4, P - p0);
        bool mk[4];
        ld_mask4(mask, (size_t)b * P + p0, nvalid, vec, mk);
        const size_t o0 = ((size_t)b * 2) * P + p0, o1 = o0 + P;
        const float4 x0 = ld4<DTYPE>(pred, o0, nvalid, vec), x1 = ld4<DTYPE>(pred, o1, nvalid, vec);
        const float4 y0 = ld4<NMSA_F32>(target, o0, nvalid, vec), y1 = ld4<NMSA_F32>(target, o1, nvalid, vec);
        const float a0[4] = {x0.x, x0.y, x0.z, x0.w}, a1[4] = {x1.x, x1.y, x1.z, x1.w};
        const float b0[4] = {y0.x, y0.y, y0.z, y0.w}, b1[4] = {y1.x, y1.y, y1.z, y1.w};
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!mk[j]) continue;
            const float dot = fmaf(a1[j], b1[j], a0[j] * b0[j]);
            part += 1.0f - __expf(kappa * (dot - 1.0f));
            ++cnt;
        }
        acc += part;
    }
    block_partial(acc, 0.0, cnt, partials);
}

template <int DTYPE>
__global__ __launch_bounds__(LOSS_THREADS) void k_vm_bwd(
    const void* __restrict__ pred, const float* __restrict__ target, const uint8_t* __restrict__ mask,
    int P, float kappa, int vec, const float* __restrict__ gscale, void* __restrict__ grad,
    const float* __restrict__ computed_for, int* __restrict__ counters)
{
    if (grad_already_computed(gscale, computed_for, counters)) return;
    const int b = blockIdx.y;
    const float g = *gscale;
    for (int p0 = (blockIdx.x * LOSS_THREADS + threadIdx.x) * 4; p0 < P; p0 += gridDim.x * LOSS_THREADS * 4) {
        const int nvalid = min(4, P - p0);
        bool mk[4];
        ld_mask4(mask, (size_t)b * P + p0, nvalid, vec, mk);
        const size_t o0 = ((size_t)b * 2) * P + p0, o1 = o0 + P;
        const float4 x0 = ld4<DTYPE>(pred, o0, nvalid, vec), x1 = ld4<DTYPE>(pred, o1, nvalid, vec);
        const float4 y0 = ld4<NMSA_F32>(target, o0, nvalid, vec), y1 = ld4<NMSA_F32>(target, o1, nvalid, vec);
        const float a0[4] = {x0.x, x0.y, x0.z, x0.w}, a1[4] = {x1.x, x1.y, x1.z, x1.w};
        const float b0[4] = {y0.x, y0.y, y0.z, y0.w}, b1[4] = {y1.x, y1.y, y1.z, y1.w};
        float g0[4], g1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dot = fmaf(a1[j], b1[j], a0[j] * b0[j]);
            const float e = mk[j] ? -g * kappa * __expf(kappa * (dot - 1.0f)) : 0.f;
            g0[j] = e * b0[j]; g1[j] = e * b1[j];
        }
        st4<DTYPE>(grad, o0, nvalid, vec, g0);
        st4<DTYPE>(grad, o1, nvalid, vec, g1);
    }
}

template <int DTYPE>
__global__ __launch_bounds__(LOSS_THREADS) void k_vm_fused(
    const void* __restrict__ pred, const float* __restrict__ target, const uint8_t* __restrict__ mask,
    int P, float kappa, int vec, const float* __restrict__ expected_gscale, void* __restrict__ grad,
    LossPartial* __restrict__ partials)
{
    vm_fused_body<DTYPE, 0>(pred, target, mask, P, kappa, vec, *expected_gscale, grad,
                               partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x,
                               blockIdx.x, gridDim.x, blockIdx.y);
}

// =================================================================================
// a9: cosine embedding, planar prediction [B,D,P] + per-image LUT [L,D] + indices
//   per valid px (index != 0):  1 - x.y / sqrt((|x|^2 + eps)(|y|^2 + eps)), eps = 1e-12
//   (ATen cosine_embedding_loss); y = lut[b][index-1]
// one px per lane: plane reads are coalesced across lanes, LUT rows come from L2
// =================================================================================
template <int DTYPE>
__device__ __forceinline__ float ld1(const void* base, size_t off)
{
    if (DTYPE == NMSA_F32) return ((const float*)base)[off];
    const uint16_t h = ((const uint16_t*)base)[off];
    return (DTYPE == NMSA_BF16) ? bf16_to_f32(h) : f16_to_f32(h);
}
template <int DTYPE>
__device__ __forceinline__ void st1(void* base, size_t off, float v)
{
    if (DTYPE == NMSA_F32) ((float*)base)[off] = v;
    else ((uint16_t*)base)[off] = (DTYPE == NMSA_BF16) ? f32_to_bf16(v) : f32_to_f16(v);
}

template <int DTYPE, bool BWD>
__global__ __launch_bounds__(LOSS_THREADS) void k_cos_emb(
    const void* __restrict__ pred, const int32_t* __restrict__ indices, const float* __restrict__ lut,
    int D, int P, int L, const float* __restrict__ gscale, void* __restrict__ grad,
    LossPartial* __restrict__ partials, int* __restrict__ status, const int* __restrict__ gate,
    int skip_nan_scale)
{
    if (gate && *gate == 0) return;                    // the fallback of k_cos_parts: only when it gave up
    const int b = blockIdx.y;
    const float EPS = 1e-12f;
    double acc = 0.0; long long cnt = 0;
    bool bad = false;
    const float g = BWD ? *gscale : 0.f;
    if (BWD && skip_nan_scale && g != g) return;       // "no expectation": the forward call writes no gradient
    for (int p = blockIdx.x * LOSS_THREADS + threadIdx.x; p < P; p += gridDim.x * LOSS_THREADS) {
        const int ix = indices[(size_t)b * P + p];
        const bool on = ix > 0 && ix <= L;
        if (ix < 0 || ix > L) bad = true;
        const float* y = lut + ((size_t)b * L + (on ? ix - 1 : 0)) * D;
        float xy = 0.f, xx = 0.f, yy = 0.f;
        if (on) {
            for (int d = 0; d < D; ++d) {
                const float xv = ld1<DTYPE>(pred, ((size_t)b * D + d) * P + p);
                const float yv = y[d];
                xy = fmaf(xv, yv, xy); xx = fmaf(xv, xv, xx); yy = fmaf(yv, yv, yy);
            }
        }
        const float den = sqrtf((xx + EPS) * (yy + EPS));
        if (!BWD) {
            if (on) { acc += 1.0f - xy / den; ++cnt; }
        } else {
            // d/dx (1 - xy/den) = -y/den + xy x / ((xx + eps) den)
            const float k1 = on ? -g / den : 0.f;
            const float k2 = on ? g * xy / ((xx + EPS) * den) : 0.f;
            for (int d = 0; d < D; ++d) {
                const size_t off = ((size_t)b * D + d) * P + p;
                const float xv = on ? ld1<DTYPE>(pred, off) : 0.f;
                st1<DTYPE>(grad, off, on ? fmaf(k2, xv, k1 * y[d]) : 0.f);
            }
        }
    }
    if (!BWD) {
        if (bad) atomicOr(status, 8);
        block_partial(acc, 0.0, cnt, partials);
    }
}

// ---- LDS-resident LUT variant (the fast path) ---------------------------------------
// The per-image LUT [L, D] is staged in LDS with a row stride of DC+1 words: lanes that need
// different rows at the same d then hit different banks, lanes that need the same row are a
// broadcast.  Each lane owns PXT consecutive pixels (16-B plane loads, U planes in flight);
// the only HBM traffic is the prediction.  A LUT that does not fit (D=768, L=64: 197 KB as
// fp32, the CU has 160 KB) is staged in `nchunks` column chunks of DC embedding dimensions
// per pixel tile: the dot products accumulate over the chunks in registers, the restaged
// bytes come from L2 (L*D*4 per tile of COS_THREADS*PXT px — 1.5 % of the tile's prediction
// bytes at D=768 bf16).  One chunk (everything up to D=512, L=64) is staged once per
// workgroup.  The LUT stays fp32: rounding it to 16 bits would cost ~1e-4 relative on the
// loss, beyond the 1e-5 parity bar.
constexpr int COS_THREADS = 1024;

template <int DTYPE, int PXT, bool BWD>
__global__ __launch_bounds__(COS_THREADS) void k_cos_emb_lds(
    const void* __restrict__ pred, const int32_t* __restrict__ indices, const float* __restrict__ lut,
    int D, int P, int L, int DC, int px_per_block, int vec,
    const float* __restrict__ gscale, void* __restrict__ grad,
    LossPartial* __restrict__ partials, int* __restrict__ status,
    float* __restrict__ dots /* [B,2,P]: x.y and |x|^2 per pixel; fwd writes, bwd reads; or null */,
    const int* __restrict__ gate /* non-null: return at once unless *gate != 0 (fallback of k_cos_parts) */,
    int skip_nan_scale /* backward: write nothing when *gscale is a NaN (a forward call without expectation) */)
{
    extern __shared__ float s_lut[];                   // [L][DC + 1], then yy[L]
    if (gate && *gate == 0) return;
    if (BWD && skip_nan_scale && *gscale != *gscale) return;
    const int b = blockIdx.y;
    const int ld = DC + 1;
    const int nchunks = (D + DC - 1) / DC;
    float* s_yy = s_lut + (size_t)L * ld;
    const float* lut_b = lut + (size_t)b * L * D;
    // |y|^2 per LUT row straight from global memory: one wave per row, lanes stride over D
    for (int r = threadIdx.x >> 6; r < L; r += COS_THREADS / 64) {
        float yy = 0.f;
        for (int d = lane_id(); d < D; d += 64) { const float v = lut_b[(size_t)r * D + d]; yy = fmaf(v, v, yy); }
        yy = wave_reduce_sum(yy);
        if (lane_id() == 0) s_yy[r] = yy;
    }
    auto stage = [&](int c) {
        const int d0 = c * DC, n = min(DC, D - d0);
        for (int i = threadIdx.x; i < L * n; i += COS_THREADS) {
            const int r = i / n, d = i - r * n;
            s_lut[r * ld + d] = lut_b[(size_t)r * D + d0 + d];
        }
    };
    if (nchunks == 1) stage(0);
    __syncthreads();

    const float EPS = 1e-12f;
    const float g = BWD ? *gscale : 0.f;
    const size_t img = (size_t)b * D * P;
    double acc = 0.0; long long cnt = 0;
    bool bad = false;
    constexpr int U = (BWD || PXT == 4) ? 8 : 4;       // 16-B plane loads in flight per lane (bwd: 8 measured 8 % faster)
    const int start = blockIdx.x * px_per_block;
    const int end = min(start + px_per_block, P);
    // the trip count is uniform over the workgroup (barriers inside when the LUT is chunked)
    for (int t0 = start; t0 < end; t0 += COS_THREADS * PXT) {
        const int p0 = t0 + threadIdx.x * PXT;
        const int nvalid = max(0, min(PXT, end - p0));
        int row[PXT], ridx[PXT];
        bool on[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const int ix = (j < nvalid) ? indices[(size_t)b * P + p0 + j] : 0;
            if (ix < 0 || ix > L) bad = true;
            on[j] = ix > 0 && ix <= L;
            ridx[j] = on[j] ? ix - 1 : 0;
            row[j] = ridx[j] * ld;
        }
        float xy[PXT], xx[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j) { xy[j] = 0.f; xx[j] = 0.f; }
        // backward with the forward's per-pixel dot products at hand: no first pass over the
        // prediction (it is read once, for the gradient)
        const bool have_dots = BWD && dots != nullptr;
        if (have_dots && nvalid > 0) {
            float* q = dots + (size_t)b * 2 * P + p0;
            ldpx<NMSA_F32, 4, true>(q, 0, min(nvalid, 4), vec, xy);
            ldpx<NMSA_F32, 4, true>(q + P, 0, min(nvalid, 4), vec, xx);
            if constexpr (PXT == 8) {
                ldpx<NMSA_F32, 4, true>(q + 4, 0, max(0, nvalid - 4), vec, xy + 4);
                ldpx<NMSA_F32, 4, true>(q + P + 4, 0, max(0, nvalid - 4), vec, xx + 4);
            }
        }
        for (int c = 0; c < (have_dots ? 0 : nchunks); ++c) {
            if (nchunks > 1) { __syncthreads(); stage(c); __syncthreads(); }
            if (nvalid == 0) continue;
            const int d0 = c * DC, n = min(DC, D - d0);
            const size_t base = img + (size_t)d0 * P + p0;
            int d = 0;
            for (; d + U <= n; d += U) {
                float v[U][PXT];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    ldpx<DTYPE, PXT, !BWD>(pred, base + (size_t)(d + u) * P, nvalid, vec, v[u]);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int j = 0; j < PXT; ++j) {
                        xy[j] = fmaf(v[u][j], s_lut[row[j] + d + u], xy[j]);
                        xx[j] = fmaf(v[u][j], v[u][j], xx[j]);
                    }
            }
            for (; d < n; ++d) {
                float v[PXT];
                ldpx<DTYPE, PXT, !BWD>(pred, base + (size_t)d * P, nvalid, vec, v);
#pragma unroll
                for (int j = 0; j < PXT; ++j) {
                    xy[j] = fmaf(v[j], s_lut[row[j] + d], xy[j]);
                    xx[j] = fmaf(v[j], v[j], xx[j]);
                }
            }
        }
        if (!BWD) {
            if (dots != nullptr && nvalid > 0) {
                float* q = dots + (size_t)b * 2 * P + p0;
                stpx<NMSA_F32, 4>(q, 0, min(nvalid, 4), vec, xy);
                stpx<NMSA_F32, 4>(q + P, 0, min(nvalid, 4), vec, xx);
                if constexpr (PXT == 8) {
                    stpx<NMSA_F32, 4>(q + 4, 0, max(0, nvalid - 4), vec, xy + 4);
                    stpx<NMSA_F32, 4>(q + P + 4, 0, max(0, nvalid - 4), vec, xx + 4);
                }
            }
            float part = 0.f;
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                if (!on[j]) continue;
                const float den = sqrtf((xx[j] + EPS) * (s_yy[ridx[j]] + EPS));
                part += 1.0f - xy[j] / den;
                ++cnt;
            }
            acc += part;
        } else {
            // d/dx (1 - xy/den) = -y/den + xy x / ((xx + eps) den); second pass re-reads the tile (L2)
            float k1[PXT], k2[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const float den = sqrtf((xx[j] + EPS) * (s_yy[ridx[j]] + EPS));
                k1[j] = on[j] ? -g / den : 0.f;
                k2[j] = on[j] ? g * xy[j] / ((xx[j] + EPS) * den) : 0.f;
            }
            for (int c = 0; c < nchunks; ++c) {
                if (nchunks > 1) { __syncthreads(); stage(c); __syncthreads(); }
                if (nvalid == 0) continue;
                const int d0 = c * DC, n = min(DC, D - d0);
                const size_t base = img + (size_t)d0 * P + p0;
                // U plane loads in flight, then U plane stores (a load-store-load-store chain
                // would leave one request per lane in flight)
                int d = 0;
                for (; d + U <= n; d += U) {
                    float v[U][PXT];
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        ldpx<DTYPE, PXT, true>(pred, base + (size_t)(d + u) * P, nvalid, vec, v[u]);
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        float o[PXT];
#pragma unroll
                        for (int j = 0; j < PXT; ++j)
                            o[j] = fmaf(k2[j], v[u][j], k1[j] * s_lut[row[j] + d + u]);
                        stpx<DTYPE, PXT, GRAD_NT>(grad, base + (size_t)(d + u) * P, nvalid, vec, o);
                    }
                }
                for (; d < n; ++d) {
                    float v[PXT], o[PXT];
                    ldpx<DTYPE, PXT, true>(pred, base + (size_t)d * P, nvalid, vec, v);
#pragma unroll
                    for (int j = 0; j < PXT; ++j) o[j] = fmaf(k2[j], v[j], k1[j] * s_lut[row[j] + d]);
                    stpx<DTYPE, PXT, GRAD_NT>(grad, base + (size_t)d * P, nvalid, vec, o);
                }
            }
        }
    }
    if (!BWD) {
        if (bad) atomicOr(status, 8);
        // block_partial() is written for LOSS_THREADS; reduce the 16 waves here
        __shared__ double r_sum[COS_THREADS / 64];
        __shared__ long long r_cnt[COS_THREADS / 64];
        acc = wave_reduce_sum(acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
        if (lane_id() == 0) { r_sum[threadIdx.x >> 6] = acc; r_cnt[threadIdx.x >> 6] = cnt; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double a = 0; long long c = 0;
            for (int k = 0; k < COS_THREADS / 64; ++k) { a += r_sum[k]; c += r_cnt[k]; }
            LossPartial pr; pr.sum = a; pr.aux = 0; pr.count = c; pr.pad = 0;
            partials[blockIdx.y * gridDim.x + blockIdx.x] = pr;
        }
    }
}


}  // namespace nmsa

using namespace nmsa;

namespace nmsa {

int loss_env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

bool loss_bad_shape(int B, int H, int W)
{
    return B <= 0 || H <= 0 || W <= 0 || (int64_t)H * W > ((int64_t)1 << 30) || B > 65535;
}

int loss_finalize(const LossPartial* partials, int n, double* sum, double* aux, int64_t* count,
                  hipStream_t stream)
{
    hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(FIN_THREADS), 0, stream, partials, n, sum, aux,
                       (long long*)count);
    return check_launch();
}

}  // namespace nmsa

namespace {

int grid_x(int P, int px_per_thread) { return loss_grid_x(P, px_per_thread); }

bool bad_shape(int B, int H, int W) { return loss_bad_shape(B, H, W); }

int finalize(const LossPartial* partials, int n, double* sum, double* aux, int64_t* count,
             hipStream_t stream)
{
    return loss_finalize(partials, n, sum, aux, count, stream);
}

}  // namespace

extern "C" size_t nmsa_loss_workspace_bytes(int B, int H, int W)
{
    if (bad_shape(B, H, W)) return 0;
    // one partial per block of the widest launch (128 px per block: k_ce_split on f32 logits)
    return (size_t)B * 2 * grid_x(H * W, 1) * sizeof(LossPartial);
}


static int ce_fwd_impl(const void* logits, int dtype, const uint8_t* target, int wide,
                       const float* weights, int B, int C, int H, int W,
                       float label_smoothing,
                       double* loss_sum, int64_t* n_elements, double* weight_sum,
                       float* lse2_out,
                       int32_t* status, void* workspace, size_t workspace_bytes,
                       nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!logits || !target || !loss_sum || !n_elements || !status || !workspace) return NMSA_ERR_ARG;
    if (bad_shape(B, H, W) || C <= 0 || C > 4096) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_loss_workspace_bytes(B, H, W)) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    const int pxt = (dtype == NMSA_F32) ? 4 : 8;
    const int vec = (P % pxt == 0) && ((((uintptr_t)logits | (uintptr_t)lse2_out) & 15) == 0);
    const int gx = grid_x(P, pxt);
    const bool smooth = label_smoothing != 0.0f;
    LossPartial* partials = (LossPartial*)workspace;
    static const int fwd_u = loss_env_int("NMSA_CE_FWD_U", 0);      // tuning knob: 4 | 8 plane loads in flight
    const int uu = fwd_u ? fwd_u : ((dtype == NMSA_F32) ? 8 : 4);
#define CE_FWD_U(DT, PX, SM, UU) hipLaunchKernelGGL((k_ce_fwd<DT, PX, SM, UU>), dim3(gx, B), dim3(LOSS_THREADS), \
        C * sizeof(float), stream, logits, target, weights, C, P, label_smoothing, vec, partials, status, \
        lse2_out, wide)
#define CE_FWD(DT, PX, SM) do { if (uu == 8) CE_FWD_U(DT, PX, SM, 8); else CE_FWD_U(DT, PX, SM, 4); } while (0)
    switch (dtype) {
        case NMSA_F32: if (smooth) CE_FWD(NMSA_F32, 4, true); else CE_FWD(NMSA_F32, 4, false); break;
        case NMSA_BF16: if (smooth) CE_FWD(NMSA_BF16, 8, true); else CE_FWD(NMSA_BF16, 8, false); break;
        case NMSA_F16: if (smooth) CE_FWD(NMSA_F16, 8, true); else CE_FWD(NMSA_F16, 8, false); break;
        default: return NMSA_ERR_ARG;
    }
#undef CE_FWD
#undef CE_FWD_U
    int rc = check_launch();
    if (rc) return rc;
    return finalize(partials, gx * B, loss_sum, weight_sum, n_elements, stream);
}

extern "C" int nmsa_loss_ce_fwd(const void* logits, int dtype, const uint8_t* target,
                                const float* weights, int B, int C, int H, int W,
                                float label_smoothing,
                                double* loss_sum, int64_t* n_elements, double* weight_sum,
                                float* lse2_out,
                                int32_t* status, void* workspace, size_t workspace_bytes,
                                nmsa_stream_t stream_)
{
    return ce_fwd_impl(logits, dtype, target, 0, weights, B, C, H, W, label_smoothing, loss_sum, n_elements,
                       weight_sum, lse2_out, status, workspace, workspace_bytes, stream_);
}

// more than 255 classes (the reference's CrossEntropyLoss takes any number, ce.py:40-68): the
// labels travel as int16 (0 = void, 1..C, C <= 4096) through the same two kernels
extern "C" int nmsa_loss_ce_fwd_i16(const void* logits, int dtype, const int16_t* target,
                                    const float* weights, int B, int C, int H, int W,
                                    float label_smoothing,
                                    double* loss_sum, int64_t* n_elements, double* weight_sum,
                                    float* lse2_out,
                                    int32_t* status, void* workspace, size_t workspace_bytes,
                                    nmsa_stream_t stream_)
{
    if (((uintptr_t)target) & 1) return NMSA_ERR_ARG;
    return ce_fwd_impl(logits, dtype, (const uint8_t*)target, 1, weights, B, C, H, W, label_smoothing, loss_sum,
                       n_elements, weight_sum, lse2_out, status, workspace, workspace_bytes, stream_);
}

static int ce_bwd_impl(const void* logits, int dtype, const uint8_t* target,
                       const float* weights, int B, int C, int H, int W,
                       float label_smoothing, const float* grad_scale, const float* lse2,
                       void* grad_logits, const float* computed_for, int32_t* counters,
                       nmsa_stream_t stream_, int wide = 0)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!logits || !target || !grad_scale || !grad_logits) return NMSA_ERR_ARG;
    if (bad_shape(B, H, W) || C <= 0 || C > 4096) return NMSA_ERR_ARG;
    const int P = H * W;
    const int pxt = (dtype == NMSA_F32) ? 4 : 8;
    const int vec = (P % pxt == 0) &&
                    ((((uintptr_t)logits | (uintptr_t)grad_logits | (uintptr_t)lse2) & 15) == 0);
    const int gx = grid_x(P, pxt);
    const bool smooth = label_smoothing != 0.0f;
    static const int bwd_u = loss_env_int("NMSA_CE_BWD_U", 0);      // tuning knob: 1 | 4 | 8
    const int ub = bwd_u ? bwd_u : 4;
#define CE_BWD_U(DT, PX, SM, UU) hipLaunchKernelGGL((k_ce_bwd<DT, PX, SM, UU>), dim3(gx, B), dim3(LOSS_THREADS), \
        C * sizeof(float), stream, logits, target, weights, C, P, label_smoothing, vec, grad_scale, \
        grad_logits, lse2, computed_for, counters, (LossPartial*)nullptr, (int*)nullptr, wide)
#define CE_BWD(DT, PX, SM) do { if (ub == 8) CE_BWD_U(DT, PX, SM, 8); else if (ub == 1) CE_BWD_U(DT, PX, SM, 1); \
                                else CE_BWD_U(DT, PX, SM, 4); } while (0)
    switch (dtype) {
        case NMSA_F32: if (smooth) CE_BWD(NMSA_F32, 4, true); else CE_BWD(NMSA_F32, 4, false); break;
        case NMSA_BF16: if (smooth) CE_BWD(NMSA_BF16, 8, true); else CE_BWD(NMSA_BF16, 8, false); break;
        case NMSA_F16: if (smooth) CE_BWD(NMSA_F16, 8, true); else CE_BWD(NMSA_F16, 8, false); break;
        default: return NMSA_ERR_ARG;
    }
#undef CE_BWD
#undef CE_BWD_U
    return check_launch();
}

extern "C" int nmsa_loss_ce_bwd(const void* logits, int dtype, const uint8_t* target,
                                const float* weights, int B, int C, int H, int W,
                                float label_smoothing, const float* grad_scale, const float* lse2,
                                void* grad_logits, nmsa_stream_t stream)
{
    return ce_bwd_impl(logits, dtype, target, weights, B, C, H, W, label_smoothing, grad_scale, lse2,
                       grad_logits, nullptr, nullptr, stream);
}

extern "C" int nmsa_loss_ce_bwd_i16(const void* logits, int dtype, const int16_t* target,
                                    const float* weights, int B, int C, int H, int W,
                                    float label_smoothing, const float* grad_scale, const float* lse2,
                                    void* grad_logits, nmsa_stream_t stream)
{
    if (((uintptr_t)target) & 1) return NMSA_ERR_ARG;
    return ce_bwd_impl(logits, dtype, (const uint8_t*)target, weights, B, C, H, W, label_smoothing, grad_scale,
                       lse2, grad_logits, nullptr, nullptr, stream, 1);
}


// 49 .. 256 classes: k_ce_split (column over the four lane rows); loss = false: the confirming /
// recomputing backward launch
extern "C" int nmsa_loss_ce_bwd_unless(const void* logits, int dtype, const uint8_t* target,
                                       const float* weights, int B, int C, int H, int W,
                                       float label_smoothing, const float* grad_scale,
                                       void* grad_logits, const float* computed_for,
                                       int32_t* counters, nmsa_stream_t stream_)
{
    if (!computed_for) return NMSA_ERR_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (!logits || !target || !grad_scale || !grad_logits) return NMSA_ERR_ARG;
    if (bad_shape(B, H, W) || C <= 0 || C > 4096) return NMSA_ERR_ARG;
    if (dtype != NMSA_F32 && dtype != NMSA_BF16 && dtype != NMSA_F16) return NMSA_ERR_ARG;
    const int P = H * W;
    const bool smooth = label_smoothing != 0.0f;
    // a miss recomputes with the SAME single-pass kernels that wrote the expected gradient
    // (logits read once, gradient written once): never dearer than the two-kernel backward
    if (C <= CE_FUSED_MAX_C) {
        const int pxt = (dtype == NMSA_F32) ? 2 : 4;
        const int vec = (P % pxt == 0) && ((((uintptr_t)logits | (uintptr_t)grad_logits) & 7) == 0);
        const int gx = grid_x(P, pxt);
        const int ng = ce_fused_ng(C);
#define CE_REDO_L(DT, NG, SM) hipLaunchKernelGGL((k_ce_fused<DT, NG, SM, false>), dim3(gx, B), dim3(LOSS_THREADS), \
        C * sizeof(float), stream, logits, target, weights, C, P, label_smoothing, vec, \
        grad_scale, grad_logits, (LossPartial*)nullptr, (int*)nullptr, computed_for, counters)
#define CE_REDO_NG(DT, SM) do { if (ng == 3) CE_REDO_L(DT, 3, SM); else if (ng == 5) CE_REDO_L(DT, 5, SM); \
                                else CE_REDO_L(DT, 6, SM); } while (0)
#define CE_REDO(DT) do { if (smooth) CE_REDO_NG(DT, true); else CE_REDO_NG(DT, false); } while (0)
        NMSA_DISPATCH_DTYPE(dtype, CE_REDO)
#undef CE_REDO
#undef CE_REDO_NG
#undef CE_REDO_L
        return check_launch();
    }
    static const int use_split = loss_env_int("NMSA_CE_SPLIT", 1);
    if (C <= CE_SPLIT_MAX_C && use_split)
        return launch_ce_split(false, logits, dtype, target, weights, B, C, P, label_smoothing, grad_scale,
                               computed_for, counters, grad_logits, nullptr, nullptr, stream);
    return ce_bwd_impl(logits, dtype, target, weights, B, C, H, W, label_smoothing, grad_scale,
                       nullptr, grad_logits, computed_for, counters, stream);
}

extern "C" size_t nmsa_count_workspace_bytes(void)
{
    return (size_t)COUNT_MAX_BLOCKS * sizeof(long long);
}

extern "C" int nmsa_count_u8(const uint8_t* values, int64_t n, int lo, int hi, int64_t* count,
                             float* mean_scale, float weight, void* workspace,
                             size_t workspace_bytes, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if ((!values && n > 0) || !count || !workspace || n < 0 || lo < 0 || hi > 255 || lo > hi)
        return NMSA_ERR_ARG;                                        // an empty input counts 0
    if (workspace_bytes < nmsa_count_workspace_bytes()) return NMSA_ERR_WORKSPACE;
    const int vec = (((uintptr_t)values) & 15) == 0;
    int64_t blocks = (n / 16 + LOSS_THREADS * 4 - 1) / (LOSS_THREADS * 4);       // ~4 loads per lane
    if (blocks < 1) blocks = 1;
    if (blocks > COUNT_MAX_BLOCKS) blocks = COUNT_MAX_BLOCKS;
    long long* partials = (long long*)workspace;
    hipLaunchKernelGGL(k_count_u8, dim3((unsigned)blocks), dim3(LOSS_THREADS), 0, stream, values,
                       (long long)n, lo, hi, vec, partials);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(k_count_finalize, dim3(1), dim3(COUNT_MAX_BLOCKS), 0, stream, partials,
                       (int)blocks, (long long*)count, mean_scale, weight);
    return check_launch();
}

// k_ce_fused forward + gradient into a caller-owned partial region, no finalize (used by the
// multi-loss call for a cross entropy whose variant differs from the one in its joint launch)
int nmsa::loss_ce_fwd_grad_partials(const void* logits, int dtype, const uint8_t* target,
                                          const float* weights, int B, int C, int P, float ls,
                                          const float* expected, void* grad, LossPartial* partials,
                                          int32_t* status, hipStream_t stream)
{
    const int pxt = (dtype == NMSA_F32) ? 2 : 4;
    const int vec = (P % pxt == 0) && ((((uintptr_t)logits | (uintptr_t)grad) & 7) == 0);
    const int gx = grid_x(P, pxt);
    const bool smooth = ls != 0.0f;
    const int ng = ce_fused_ng(C);
#define CE_P_L(DT, NG, SM) hipLaunchKernelGGL((k_ce_fused<DT, NG, SM>), dim3(gx, B), dim3(LOSS_THREADS), \
        C * sizeof(float), stream, logits, target, weights, C, P, ls, vec, expected, grad, partials, status)
#define CE_P_NG(DT, SM) do { if (ng == 3) CE_P_L(DT, 3, SM); else if (ng == 5) CE_P_L(DT, 5, SM); else CE_P_L(DT, 6, SM); } while (0)
#define CE_P(DT) do { if (smooth) CE_P_NG(DT, true); else CE_P_NG(DT, false); } while (0)
    NMSA_DISPATCH_DTYPE(dtype, CE_P)
#undef CE_P
#undef CE_P_NG
#undef CE_P_L
    return check_launch();
}

extern "C" int nmsa_loss_ce_fwd_grad_supported(int dtype, int C)
{
    // C <= 48: register-resident column (k_ce_fused); above: two walks in one launch
    return (dtype == NMSA_F32 || dtype == NMSA_BF16 || dtype == NMSA_F16) && C >= 1 && C <= 4096;
}

extern "C" int nmsa_loss_ce_fwd_grad(const void* logits, int dtype, const uint8_t* target,
                                     const float* weights, int B, int C, int H, int W,
                                     float label_smoothing, const float* expected_grad_scale,
                                     double* loss_sum, int64_t* n_elements, double* weight_sum,
                                     void* grad_logits, int32_t* status, void* workspace,
                                     size_t workspace_bytes, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!logits || !target || !loss_sum || !n_elements || !status || !workspace ||
        !expected_grad_scale || !grad_logits) return NMSA_ERR_ARG;
    if (bad_shape(B, H, W) || C <= 0 || C > 4096) return NMSA_ERR_ARG;
    if (dtype != NMSA_F32 && dtype != NMSA_BF16 && dtype != NMSA_F16) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_loss_workspace_bytes(B, H, W)) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    static const int use_split = loss_env_int("NMSA_CE_SPLIT", 1);
    static const int force_split = loss_env_int("NMSA_CE_FORCE_SPLIT", 0);     // experiments
    if (((C > CE_FUSED_MAX_C && use_split) || (force_split && C >= 4)) && C <= CE_SPLIT_MAX_C) {
        // 49 .. 256 classes: the column spread over the four lane rows of a wave (k_ce_split):
        // logits read once, gradient written once, no LDS tile, no barrier
        LossPartial* partials = (LossPartial*)workspace;
        int rc2 = launch_ce_split(true, logits, dtype, target, weights, B, C, P, label_smoothing,
                                  expected_grad_scale, nullptr, nullptr, grad_logits, partials, status, stream);
        if (rc2) return rc2;
        return finalize(partials, ce_split_blocks(P, dtype) * B, loss_sum, weight_sum, n_elements, stream);
    }
    if (C > CE_FUSED_MAX_C) {
        // more than 256 classes (or NMSA_CE_SPLIT=0): the two-walk backward kernel also sums the loss
        const int pxt = (dtype == NMSA_F32) ? 4 : 8;
        const int vec = (P % pxt == 0) && ((((uintptr_t)logits | (uintptr_t)grad_logits) & 15) == 0);
        const int gx = grid_x(P, pxt);
        LossPartial* partials = (LossPartial*)workspace;
        // Fewer workgroups per CU keep more of the tiles between their two walks inside the
        // Infinity Cache: 50 KB of (unused) dynamic LDS admit 3 workgroups = 12 waves per CU —
        // B=16 C=150 1024x768: 2.02 ms unpadded, 1.89 (32 KB), 1.75 (50 KB), 1.76 (64 KB);
        // C=64: 0.592 / 0.582 / 0.576 / 0.610.  NMSA_CE_TWO_LDS overrides (bytes).
        static const int lds_pad = loss_env_int("NMSA_CE_TWO_LDS", 50000);
        const size_t lds_two = (size_t)C * sizeof(float) > (size_t)lds_pad ? (size_t)C * sizeof(float)
                                                                           : (size_t)lds_pad;
#define CE_TWO(DT, PX, SM) hipLaunchKernelGGL((k_ce_bwd<DT, PX, SM, 4, true>), dim3(gx, B), dim3(LOSS_THREADS), \
        lds_two, stream, logits, target, weights, C, P, label_smoothing, vec, \
        expected_grad_scale, grad_logits, (const float*)nullptr, (const float*)nullptr, (int*)nullptr, \
        partials, status)
        const bool smooth2 = label_smoothing != 0.0f;
        switch (dtype) {
            case NMSA_F32: if (smooth2) CE_TWO(NMSA_F32, 4, true); else CE_TWO(NMSA_F32, 4, false); break;
            case NMSA_BF16: if (smooth2) CE_TWO(NMSA_BF16, 8, true); else CE_TWO(NMSA_BF16, 8, false); break;
            default: if (smooth2) CE_TWO(NMSA_F16, 8, true); else CE_TWO(NMSA_F16, 8, false); break;
        }
#undef CE_TWO
        int rc2 = check_launch();
        if (rc2) return rc2;
        return finalize(partials, gx * B, loss_sum, weight_sum, n_elements, stream);
    }
    const int pxt = (dtype == NMSA_F32) ? 2 : 4;
    const int vec = (P % pxt == 0) && ((((uintptr_t)logits | (uintptr_t)grad_logits) & 7) == 0);
    const int gx = grid_x(P, pxt);
    const bool smooth = label_smoothing != 0.0f;
    LossPartial* partials = (LossPartial*)workspace;
    const int ng = ce_fused_ng(C);
#define CE_FUSED_L(DT, NG, SM) hipLaunchKernelGGL((k_ce_fused<DT, NG, SM>), dim3(gx, B), dim3(LOSS_THREADS), \
        C * sizeof(float), stream, logits, target, weights, C, P, label_smoothing, vec, \
        expected_grad_scale, grad_logits, partials, status)
#define CE_FUSED_NG(DT, SM) do { if (ng == 3) CE_FUSED_L(DT, 3, SM); else if (ng == 5) CE_FUSED_L(DT, 5, SM); \
                                 else CE_FUSED_L(DT, 6, SM); } while (0)
#define CE_FUSED(DT) do { if (smooth) CE_FUSED_NG(DT, true); else CE_FUSED_NG(DT, false); } while (0)
    NMSA_DISPATCH_DTYPE(dtype, CE_FUSED)
#undef CE_FUSED
#undef CE_FUSED_NG
#undef CE_FUSED_L
    int rc = check_launch();
    if (rc) return rc;
    return finalize(partials, gx * B, loss_sum, weight_sum, n_elements, stream);
}

extern "C" int nmsa_loss_masked_fwd(const void* pred, int dtype, const float* target,
                                    const uint8_t* mask, int B, int C, int H, int W, int kind,
                                    double* loss_sum, int64_t* n_mask,
                                    void* workspace, size_t workspace_bytes, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !target || !loss_sum || !n_mask || !workspace) return NMSA_ERR_ARG;
    if (bad_shape(B, H, W) || C <= 0 || kind < 0 || kind > 2) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_loss_workspace_bytes(B, H, W)) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    const int vec = (P % 4 == 0) && ((((uintptr_t)pred | (uintptr_t)target | (uintptr_t)mask) & 15) == 0);
    const int gx = grid_x(P, 8);
    LossPartial* partials = (LossPartial*)workspace;
#define CALL(DT)                                                                                     \
    if (kind == 0) hipLaunchKernelGGL((k_elem_fwd<DT, 0>), dim3(gx, B), dim3(LOSS_THREADS), 0, stream, \
                                      pred, target, mask, C, P, vec, partials);                      \
    else if (kind == 1) hipLaunchKernelGGL((k_elem_fwd<DT, 1>), dim3(gx, B), dim3(LOSS_THREADS), 0,   \
                                           stream, pred, target, mask, C, P, vec, partials);         \
    else hipLaunchKernelGGL((k_elem_fwd<DT, 2>), dim3(gx, B), dim3(LOSS_THREADS), 0, stream, pred,    \
                            target, mask, C, P, vec, partials)
    NMSA_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
    int rc = check_launch();
    if (rc) return rc;
    return finalize(partials, gx * B, loss_sum, nullptr, n_mask, stream);
}

static int masked_bwd_impl(const void* pred, int dtype, const float* target,
                           const uint8_t* mask, int B, int C, int H, int W, int kind,
                           const float* grad_scale, void* grad_pred, const float* computed_for,
                           int32_t* counters, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !target || !grad_scale || !grad_pred) return NMSA_ERR_ARG;
    if (bad_shape(B, H, W) || C <= 0 || kind < 0 || kind > 2) return NMSA_ERR_ARG;
    const int P = H * W;
    const int vec = (P % 4 == 0) &&
                    ((((uintptr_t)pred | (uintptr_t)target | (uintptr_t)mask | (uintptr_t)grad_pred) & 15) == 0);
    const int gx = grid_x(P, 8);
#define CALL(DT)                                                                                     \
    if (kind == 0) hipLaunchKernelGGL((k_elem_bwd<DT, 0>), dim3(gx, B), dim3(LOSS_THREADS), 0, stream, \
                                      pred, target, mask, C, P, vec, grad_scale, grad_pred,          \
                                      computed_for, counters);                                       \
    else if (kind == 1) hipLaunchKernelGGL((k_elem_bwd<DT, 1>), dim3(gx, B), dim3(LOSS_THREADS), 0,   \
                                           stream, pred, target, mask, C, P, vec, grad_scale,        \
                                           grad_pred, computed_for, counters);                       \
    else hipLaunchKernelGGL((k_elem_bwd<DT, 2>), dim3(gx, B), dim3(LOSS_THREADS), 0, stream, pred,    \
                            target, mask, C, P, vec, grad_scale, grad_pred, computed_for, counters)
    NMSA_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
    return check_launch();
}

extern "C" int nmsa_loss_masked_bwd(const void* pred, int dtype, const float* target,
                                    const uint8_t* mask, int B, int C, int H, int W, int kind,
                                    const float* grad_scale, void* grad_pred, nmsa_stream_t stream)
{
    return masked_bwd_impl(pred, dtype, target, mask, B, C, H, W, kind, grad_scale, grad_pred,
                           nullptr, nullptr, stream);
}

extern "C" int nmsa_loss_masked_bwd_unless(const void* pred, int dtype, const float* target,
                                           const uint8_t* mask, int B, int C, int H, int W, int kind,
                                           const float* grad_scale, void* grad_pred,
                                           const float* computed_for, int32_t* counters,
                                           nmsa_stream_t stream)
{
    if (!computed_for) return NMSA_ERR_ARG;
    return masked_bwd_impl(pred, dtype, target, mask, B, C, H, W, kind, grad_scale, grad_pred,
                           computed_for, counters, stream);
}

extern "C" int nmsa_loss_masked_fwd_grad(const void* pred, int dtype, const float* target,
                                         const uint8_t* mask, int B, int C, int H, int W, int kind,
                                         const float* expected_grad_scale,
                                         double* loss_sum, int64_t* n_mask, void* grad_pred,
                                         void* workspace, size_t workspace_bytes,
                                         nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !target || !loss_sum || !n_mask || !workspace || !expected_grad_scale || !grad_pred)
        return NMSA_ERR_ARG;
    if (bad_shape(B, H, W) || C <= 0 || kind < 0 || kind > 2) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_loss_workspace_bytes(B, H, W)) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    const int vec = (P % 4 == 0) &&
                    ((((uintptr_t)pred | (uintptr_t)target | (uintptr_t)mask | (uintptr_t)grad_pred) & 15) == 0);
    const int gx = grid_x(P, 8);
    LossPartial* partials = (LossPartial*)workspace;
#define CALL(DT)                                                                                       \
    if (kind == 0) hipLaunchKernelGGL((k_elem_fused<DT, 0>), dim3(gx, B), dim3(LOSS_THREADS), 0, stream, \
                                      pred, target, mask, C, P, vec, expected_grad_scale, grad_pred,   \
                                      partials);                                                       \
    else if (kind == 1) hipLaunchKernelGGL((k_elem_fused<DT, 1>), dim3(gx, B), dim3(LOSS_THREADS), 0,   \
                                           stream, pred, target, mask, C, P, vec, expected_grad_scale, \
                                           grad_pred, partials);                                       \
    else hipLaunchKernelGGL((k_elem_fused<DT, 2>), dim3(gx, B), dim3(LOSS_THREADS), 0, stream, pred,    \
                            target, mask, C, P, vec, expected_grad_scale, grad_pred, partials)
    NMSA_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
    int rc = check_launch();
    if (rc) return rc;
    return finalize(partials, gx * B, loss_sum, nullptr, n_mask, stream);
}

extern "C" int nmsa_loss_vonmises_fwd(const void* pred, int dtype, const float* target,
                                      const uint8_t* mask, int B, int H, int W, float kappa,
                                      double* loss_sum, int64_t* n_rows,
                                      void* workspace, size_t workspace_bytes, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !target || !loss_sum || !n_rows || !workspace) return NMSA_ERR_ARG;
    if (bad_shape(B, H, W)) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_loss_workspace_bytes(B, H, W)) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    const int vec = (P % 4 == 0) && ((((uintptr_t)pred | (uintptr_t)target | (uintptr_t)mask) & 15) == 0);
    const int gx = grid_x(P, 8);
    LossPartial* partials = (LossPartial*)workspace;
#define CALL(DT) hipLaunchKernelGGL((k_vm_fwd<DT>), dim3(gx, B), dim3(LOSS_THREADS), 0, stream, pred, \
                                    target, mask, P, kappa, vec, partials)
    NMSA_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
    int rc = check_launch();
    if (rc) return rc;
    return finalize(partials, gx * B, loss_sum, nullptr, n_rows, stream);
}

static int vonmises_bwd_impl(const void* pred, int dtype, const float* target,
                             const uint8_t* mask, int B, int H, int W, float kappa,
                             const float* grad_scale, void* grad_pred, const float* computed_for,
                             int32_t* counters, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !target || !grad_scale || !grad_pred) return NMSA_ERR_ARG;
    if (bad_shape(B, H, W)) return NMSA_ERR_ARG;
    const int P = H * W;
    const int vec = (P % 4 == 0) &&
                    ((((uintptr_t)pred | (uintptr_t)target | (uintptr_t)mask | (uintptr_t)grad_pred) & 15) == 0);
    const int gx = grid_x(P, 8);
#define CALL(DT) hipLaunchKernelGGL((k_vm_bwd<DT>), dim3(gx, B), dim3(LOSS_THREADS), 0, stream, pred, \
                                    target, mask, P, kappa, vec, grad_scale, grad_pred, computed_for, \
                                    counters)
    NMSA_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
    return check_launch();
}

extern "C" int nmsa_loss_vonmises_bwd(const void* pred, int dtype, const float* target,
                                      const uint8_t* mask, int B, int H, int W, float kappa,
                                      const float* grad_scale, void* grad_pred, nmsa_stream_t stream)
{
    return vonmises_bwd_impl(pred, dtype, target, mask, B, H, W, kappa, grad_scale, grad_pred,
                             nullptr, nullptr, stream);
}

extern "C" int nmsa_loss_vonmises_bwd_unless(const void* pred, int dtype, const float* target,
                                             const uint8_t* mask, int B, int H, int W, float kappa,
                                             const float* grad_scale, void* grad_pred,
                                             const float* computed_for, int32_t* counters,
                                             nmsa_stream_t stream)
{
    if (!computed_for) return NMSA_ERR_ARG;
    return vonmises_bwd_impl(pred, dtype, target, mask, B, H, W, kappa, grad_scale, grad_pred,
                             computed_for, counters, stream);
}

extern "C" int nmsa_loss_vonmises_fwd_grad(const void* pred, int dtype, const float* target,
                                           const uint8_t* mask, int B, int H, int W, float kappa,
                                           const float* expected_grad_scale,
                                           double* loss_sum, int64_t* n_rows, void* grad_pred,
                                           void* workspace, size_t workspace_bytes,
                                           nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !target || !loss_sum || !n_rows || !workspace || !expected_grad_scale || !grad_pred)
        return NMSA_ERR_ARG;
    if (bad_shape(B, H, W)) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_loss_workspace_bytes(B, H, W)) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    const int vec = (P % 4 == 0) &&
                    ((((uintptr_t)pred | (uintptr_t)target | (uintptr_t)mask | (uintptr_t)grad_pred) & 15) == 0);
    const int gx = grid_x(P, 8);
    LossPartial* partials = (LossPartial*)workspace;
#define CALL(DT) hipLaunchKernelGGL((k_vm_fused<DT>), dim3(gx, B), dim3(LOSS_THREADS), 0, stream, pred, \
                                    target, mask, P, kappa, vec, expected_grad_scale, grad_pred, partials)
    NMSA_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
    int rc = check_launch();
    if (rc) return rc;
    return finalize(partials, gx * B, loss_sum, nullptr, n_rows, stream);
}

namespace {

// workgroups per image for the LDS-LUT kernels: ~2 per CU over the whole batch
int cos_blocks_per_image(int B, int P, int pxt)
{
    int per_img = (2 * device_geometry().cus + B - 1) / B;
    const int max_useful = (P + COS_THREADS * pxt - 1) / (COS_THREADS * pxt);
    if (per_img > max_useful) per_img = max_useful;
    if (per_img < 1) per_img = 1;
    return per_img;
}

// column chunk of the LUT that is LDS-resident at a time: the whole row when it fits in
// COS_LDS_BUDGET, otherwise D split into the fewest equal chunks that do (multiple of 8);
// 0 = the LUT has too many rows for a useful chunk -> generic kernel (LUT rows from L2)
constexpr size_t COS_LDS_BUDGET = 150 * 1024;
int cos_chunk(int L, int D)
{
    const long long dc_max = (long long)(COS_LDS_BUDGET / sizeof(float) - L) / L - 1;
    if (dc_max < 32) return 0;
    if (D <= dc_max) return D;
    int n = (int)((D + dc_max - 1) / dc_max);
    for (;; ++n) {
        const int dc = (((D + n - 1) / n) + 7) / 8 * 8;
        if (dc <= dc_max) return dc;
    }
}
size_t cos_lds_bytes(int L, int DC) { return ((size_t)L * (DC + 1) + L) * sizeof(float); }

struct CosWalkGeometry { int DC, pxt, ppb, gx; size_t lds; };

CosWalkGeometry cos_walk_geometry(int dtype, int B, int D, int P, int L)
{
    CosWalkGeometry g;
    g.DC = cos_chunk(L, D);
    if (g.DC > 0) {
        g.lds = cos_lds_bytes(L, g.DC);
        g.pxt = (dtype == NMSA_F32) ? 4 : 8;
        const int per_img = cos_blocks_per_image(B, P, g.pxt);
        int ppb = (P + per_img - 1) / per_img;
        g.ppb = ((ppb + g.pxt - 1) / g.pxt) * g.pxt;                // keep 16-B alignment of block starts
        g.gx = (P + g.ppb - 1) / g.ppb;
    } else {
        g.lds = 0; g.pxt = 1; g.ppb = 0;
        g.gx = grid_x(P, 1);
    }
    return g;
}

}  // namespace

namespace nmsa {

// The two-walk cosine kernels (forward: sums; backward: a second read for the gradient).  They
// are the path of shapes no one-pass kernel takes AND the device-gated fallback of k_cos_parts
// (`gate` non-null: every workgroup returns at once unless *gate != 0).
int cos_two_walk_blocks(int dtype, int B, int D, int P, int L)       // partial slots (whole batch)
{
    return cos_walk_geometry(dtype, B, D, P, L).gx * B;
}

int launch_cos_two_walk_fwd(const void* pred, int dtype, const int32_t* indices, const float* lut,
                            int B, int D, int P, int L, LossPartial* partials, int32_t* status,
                            float* dots_out, const int* gate, hipStream_t stream)
{
    const CosWalkGeometry g = cos_walk_geometry(dtype, B, D, P, L);
    if (g.DC > 0) {
        const int vec = (P % g.pxt == 0) && ((((uintptr_t)pred) & 15) == 0);
        // the per-pixel dot products are only kept on the aligned path (16-B float stores)
        const bool dots_vec = dots_out && vec && ((((uintptr_t)dots_out) & 15) == 0);
        if (dots_out && !dots_vec) return NMSA_ERR_ARG;
#define COS_FWD(DT, PX) do { if (allow_dynamic_lds(k_cos_emb_lds<DT, PX, false>, COS_LDS_BUDGET + 1024)) return NMSA_ERR_LAUNCH; \
        hipLaunchKernelGGL((k_cos_emb_lds<DT, PX, false>), dim3(g.gx, B), dim3(COS_THREADS), g.lds, \
        stream, pred, indices, lut, D, P, L, g.DC, g.ppb, vec, (const float*)nullptr, (void*)nullptr, partials, status, \
        dots_vec ? dots_out : (float*)nullptr, gate, 0); } while (0)
        switch (dtype) {
            case NMSA_F32: COS_FWD(NMSA_F32, 4); break;
            case NMSA_BF16: COS_FWD(NMSA_BF16, 8); break;
            case NMSA_F16: COS_FWD(NMSA_F16, 8); break;
            default: return NMSA_ERR_ARG;
        }
#undef COS_FWD
        return check_launch();
    }
    if (dots_out) return NMSA_ERR_ARG;
#define CALL(DT) hipLaunchKernelGGL((k_cos_emb<DT, false>), dim3(g.gx, B), dim3(LOSS_THREADS), 0, stream, \
                                    pred, indices, lut, D, P, L, (const float*)nullptr, (void*)nullptr, partials, status, gate, 0)
    NMSA_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
    return check_launch();
}

int launch_cos_two_walk_bwd(const void* pred, int dtype, const int32_t* indices, const float* lut,
                            int B, int D, int P, int L, const float* grad_scale, const float* dots,
                            void* grad_pred, const int* gate, int skip_nan_scale, hipStream_t stream)
{
    const CosWalkGeometry g = cos_walk_geometry(dtype, B, D, P, L);
    if (dots && (g.DC <= 0 || ((uintptr_t)dots & 15) != 0)) return NMSA_ERR_ARG;
    if (g.DC > 0) {
        const int vec = (P % g.pxt == 0) && ((((uintptr_t)pred | (uintptr_t)grad_pred) & 15) == 0);
        if (dots && !vec) return NMSA_ERR_ARG;
#define COS_BWD(DT, PX) do { if (allow_dynamic_lds(k_cos_emb_lds<DT, PX, true>, COS_LDS_BUDGET + 1024)) return NMSA_ERR_LAUNCH; \
        hipLaunchKernelGGL((k_cos_emb_lds<DT, PX, true>), dim3(g.gx, B), dim3(COS_THREADS), g.lds, \
        stream, pred, indices, lut, D, P, L, g.DC, g.ppb, vec, grad_scale, grad_pred, (LossPartial*)nullptr, (int*)nullptr, \
        (float*)dots, gate, skip_nan_scale); } while (0)
        switch (dtype) {
            case NMSA_F32: COS_BWD(NMSA_F32, 4); break;
            case NMSA_BF16: COS_BWD(NMSA_BF16, 8); break;
            case NMSA_F16: COS_BWD(NMSA_F16, 8); break;
            default: return NMSA_ERR_ARG;
        }
#undef COS_BWD
        return check_launch();
    }
#define CALL(DT) hipLaunchKernelGGL((k_cos_emb<DT, true>), dim3(g.gx, B), dim3(LOSS_THREADS), 0, stream, \
                                    pred, indices, lut, D, P, L, grad_scale, grad_pred,                  \
                                    (LossPartial*)nullptr, (int*)nullptr, gate, skip_nan_scale)
    NMSA_DISPATCH_DTYPE(dtype, CALL)
#undef CALL
    return check_launch();
}

}  // namespace nmsa

extern "C" int nmsa_loss_cos_emb_fwd(const void* pred, int dtype, const int32_t* indices,
                                     const float* lut, int B, int D, int H, int W, int L,
                                     double* loss_sum, int64_t* n_rows, float* dots_out,
                                     int32_t* status,
                                     void* workspace, size_t workspace_bytes, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !indices || !lut || !loss_sum || !n_rows || !status || !workspace) return NMSA_ERR_ARG;
    if (bad_shape(B, H, W) || D <= 0 || L <= 0) return NMSA_ERR_ARG;
    if (dtype != NMSA_F32 && dtype != NMSA_BF16 && dtype != NMSA_F16) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_loss_workspace_bytes(B, H, W)) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    LossPartial* partials = (LossPartial*)workspace;
    const int n = cos_two_walk_blocks(dtype, B, D, P, L);
    if ((size_t)n * sizeof(LossPartial) > workspace_bytes) return NMSA_ERR_WORKSPACE;
    int rc = launch_cos_two_walk_fwd(pred, dtype, indices, lut, B, D, P, L, partials, status, dots_out, nullptr,
                                     stream);
    if (rc) return rc;
    return finalize(partials, n, loss_sum, nullptr, n_rows, stream);
}

extern "C" int nmsa_loss_cos_emb_can_keep_dots(int D, int H, int W, int L)
{
    if (D <= 0 || L <= 0 || H <= 0 || W <= 0) return 0;
    return cos_chunk(L, D) > 0 && ((int64_t)H * W) % 8 == 0;
}

extern "C" int nmsa_loss_cos_emb_bwd(const void* pred, int dtype, const int32_t* indices,
                                     const float* lut, int B, int D, int H, int W, int L,
                                     const float* grad_scale, const float* dots, void* grad_pred,
                                     nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !indices || !lut || !grad_scale || !grad_pred) return NMSA_ERR_ARG;
    if (bad_shape(B, H, W) || D <= 0 || L <= 0) return NMSA_ERR_ARG;
    return launch_cos_two_walk_bwd(pred, dtype, indices, lut, B, D, H * W, L, grad_scale, dots, grad_pred,
                                   nullptr, 0, stream);
}

