// loss_bodies.hpp — device-side pieces of the loss kernels that more than one translation unit
// instantiates (losses.hip: one kernel per loss; losses_split.hip: wide-column cross entropy;
// losses_multi.hip: all losses of a task helper in one launch): typed loads / stores, the
// block-partial reduction, and the per-workgroup BODIES of the forward(+gradient) kernels.
#pragma once
#include "loss_common.hpp"

namespace nmsa {

// ---- typed 4-px helpers --------------------------------------------------------------
template <int DTYPE>
__device__ __forceinline__ float4 ld4(const void* base, size_t off, int nvalid, bool vec)
{
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DTYPE == NMSA_F32) {
        const float* p = (const float*)base + off;
        if (vec) return *(const float4*)p;
        if (nvalid > 0) r.x = p[0];
        if (nvalid > 1) r.y = p[1];
        if (nvalid > 2) r.z = p[2];
        if (nvalid > 3) r.w = p[3];
    } else {
        const uint16_t* p = (const uint16_t*)base + off;
        uint16_t h[4] = {0, 0, 0, 0};
        if (vec) { const ushort4 u = *(const ushort4*)p; h[0] = u.x; h[1] = u.y; h[2] = u.z; h[3] = u.w; }
        else for (int j = 0; j < 4; ++j) if (j < nvalid) h[j] = p[j];
        if (DTYPE == NMSA_BF16) {
            r.x = bf16_to_f32(h[0]); r.y = bf16_to_f32(h[1]); r.z = bf16_to_f32(h[2]); r.w = bf16_to_f32(h[3]);
        } else {
            r.x = f16_to_f32(h[0]); r.y = f16_to_f32(h[1]); r.z = f16_to_f32(h[2]); r.w = f16_to_f32(h[3]);
        }
    }
    return r;
}

template <int DTYPE>
__device__ __forceinline__ void st4(void* base, size_t off, int nvalid, bool vec, const float v[4])
{
    if (DTYPE == NMSA_F32) {
        float* p = (float*)base + off;
        if (vec) *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
        else for (int j = 0; j < nvalid; ++j) p[j] = v[j];
    } else {
        uint16_t* p = (uint16_t*)base + off;
        uint16_t h[4];
        for (int j = 0; j < 4; ++j) h[j] = (DTYPE == NMSA_BF16) ? f32_to_bf16(v[j]) : f32_to_f16(v[j]);
        if (vec) *(ushort4*)p = make_ushort4(h[0], h[1], h[2], h[3]);
        else for (int j = 0; j < nvalid; ++j) p[j] = h[j];
    }
}

__device__ __forceinline__ void ld_mask4(const uint8_t* m, size_t off, int nvalid, bool vec, bool out[4])
{
    if (!m) { for (int j = 0; j < 4; ++j) out[j] = j < nvalid; return; }
    if (vec) {
        const uchar4 u = *(const uchar4*)(m + off);
        out[0] = u.x != 0; out[1] = u.y != 0; out[2] = u.z != 0; out[3] = u.w != 0;
    } else {
        for (int j = 0; j < 4; ++j) out[j] = (j < nvalid) && (m[off + j] != 0);
    }
}

// block reduction -> one LossPartial per block (fixed order: lane tree, then wave order)
__device__ __forceinline__ void block_partial_at(double sum, double aux, long long count,
                                                 LossPartial* __restrict__ slot)
{
    __shared__ double s_sum[LOSS_THREADS / 64], s_aux[LOSS_THREADS / 64];
    __shared__ long long s_cnt[LOSS_THREADS / 64];
    sum = wave_reduce_sum(sum);
    aux = wave_reduce_sum(aux);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) count += __shfl_down(count, o);
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) { s_sum[w] = sum; s_aux[w] = aux; s_cnt[w] = count; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0; long long c = 0;
        for (int k = 0; k < LOSS_THREADS / 64; ++k) { a += s_sum[k]; b += s_aux[k]; c += s_cnt[k]; }
        LossPartial p; p.sum = a; p.aux = b; p.count = c; p.pad = 0;
        *slot = p;
    }
}

__device__ __forceinline__ void block_partial(double sum, double aux, long long count,
                                              LossPartial* __restrict__ partials)
{
    block_partial_at(sum, aux, count, partials + (size_t)blockIdx.y * gridDim.x + blockIdx.x);
}

// =================================================================================
// a6: cross entropy (weights, ignore void, label smoothing)
//   per px (t = label-1 >= 0):  (1-ls)*w_t*(lse - x_t) + (ls/C)*(lse*W - sum_c w_c x_c)
//   outputs: sum, n = #non-void px, aux = sum_px w_t  (divisor of the ESANet
//   "weighted_reduction", ce.py:57-68)
//
// PXT pixels per lane: 4 for f32, 8 for bf16 / f16 — always 16-B loads.  The class loop
// works in groups of U planes: group maximum with v_max3, ONE rescale of the running sum
// per group, then 3 VALU per element (fma into the base-2 domain, v_exp_f32, add).  The
// target logit x_t is fetched with one gather per pixel after the loop (the tile was just
// streamed, the gather hits L2) instead of a compare/select per class.
// =================================================================================
#ifndef NMSA_GRAD_NT
#define NMSA_GRAD_NT 1
#endif
constexpr bool GRAD_NT = NMSA_GRAD_NT != 0;     // gradient planes are written once: streaming stores

template <int DTYPE, int PXT, bool NT = true>
__device__ __forceinline__ void ldpx(const void* base, size_t off, int nvalid, bool vec, float out[PXT])
{
    if (DTYPE == NMSA_F32) {
        const float* p = (const float*)base + off;
        if (vec) {
            const f32x4_s v = NT ? __builtin_nontemporal_load((const f32x4_s*)p) : *(const f32x4_s*)p;
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        } else {
            for (int j = 0; j < PXT; ++j) out[j] = (j < nvalid) ? p[j] : 0.f;
        }
    } else {
        const uint16_t* p = (const uint16_t*)base + off;
        uint16_t h[PXT];
        if (vec) {
            const u32x4_s v = NT ? __builtin_nontemporal_load((const u32x4_s*)p) : *(const u32x4_s*)p;
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { h[2 * j] = (uint16_t)(w[j] & 0xFFFF); h[2 * j + 1] = (uint16_t)(w[j] >> 16); }
        } else {
            for (int j = 0; j < PXT; ++j) h[j] = (j < nvalid) ? p[j] : 0;
        }
#pragma unroll
        for (int j = 0; j < PXT; ++j) out[j] = (DTYPE == NMSA_BF16) ? bf16_to_f32(h[j]) : f16_to_f32(h[j]);
    }
}

template <int DTYPE, int PXT, bool NT = false>
__device__ __forceinline__ void stpx(void* base, size_t off, int nvalid, bool vec, const float v[PXT])
{
    if (DTYPE == NMSA_F32) {
        float* p = (float*)base + off;
        if (vec) {
            const f32x4_s w = {v[0], v[1], v[2], v[3]};
            if (NT) __builtin_nontemporal_store(w, (f32x4_s*)p); else *(f32x4_s*)p = w;
        }
        else for (int j = 0; j < nvalid; ++j) p[j] = v[j];
    } else {
        uint16_t* p = (uint16_t*)base + off;
        uint16_t h[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j) h[j] = (DTYPE == NMSA_BF16) ? f32_to_bf16(v[j]) : f32_to_f16(v[j]);
        if (vec) {
            u32x4_s w;
            w.x = h[0] | ((uint32_t)h[1] << 16); w.y = h[2] | ((uint32_t)h[3] << 16);
            w.z = h[4] | ((uint32_t)h[5] << 16); w.w = h[6] | ((uint32_t)h[7] << 16);
            if (NT) __builtin_nontemporal_store(w, (u32x4_s*)p); else *(u32x4_s*)p = w;
        } else {
            for (int j = 0; j < nvalid; ++j) p[j] = h[j];
        }
    }
}

template <int DTYPE>
__device__ __forceinline__ float ld_scalar(const void* base, size_t off)
{
    if (DTYPE == NMSA_F32) return ((const float*)base)[off];
    const uint16_t h = ((const uint16_t*)base)[off];
    return (DTYPE == NMSA_BF16) ? bf16_to_f32(h) : f16_to_f32(h);
}

// streaming max / sum-of-exp2 over the classes for PXT pixels; SMOOTH adds sum_c w_c x_c
template <int DTYPE, int PXT, int U, bool SMOOTH, bool TRACK_T, bool NT>
__device__ __forceinline__ void ce_scan(const void* logits, size_t img, int P, int p0, int nvalid,
                                        bool vec, int C, const float* s_w, const int t[PXT],
                                        float m[PXT], float s[PXT], float swx[PXT], float xt[PXT])
{
#pragma unroll
    for (int j = 0; j < PXT; ++j) { m[j] = -INFINITY; s[j] = 0.f; swx[j] = 0.f; xt[j] = 0.f; }
    int c = 0;
    for (; c + U <= C; c += U) {
        float v[U][PXT];
#pragma unroll
        for (int u = 0; u < U; ++u) ldpx<DTYPE, PXT, NT>(logits, img + (size_t)(c + u) * P + p0, nvalid, vec, v[u]);
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const int tj = TRACK_T ? t[j] - c : 0;
            float g = v[0][j];
#pragma unroll
            for (int u = 1; u < U; ++u) g = fmaxf(g, v[u][j]);       // (the compiler folds these into v_max3_f32)
            const float mn = fmaxf(m[j], g);
            const float k = -mn * LOG2E;
            float acc = s[j] * __builtin_amdgcn_exp2f(fmaf(m[j], LOG2E, k));        // rescale once per group
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc += __builtin_amdgcn_exp2f(fmaf(v[u][j], LOG2E, k));
                if (SMOOTH) swx[j] = fmaf(s_w[c + u], v[u][j], swx[j]);
                if (TRACK_T) xt[j] = (tj == u) ? v[u][j] : xt[j];
            }
            s[j] = acc; m[j] = mn;
        }
    }
    for (; c < C; ++c) {
        float v[PXT];
        ldpx<DTYPE, PXT, NT>(logits, img + (size_t)c * P + p0, nvalid, vec, v);
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const float mn = fmaxf(m[j], v[j]);
            const float k = -mn * LOG2E;
            s[j] = s[j] * __builtin_amdgcn_exp2f(fmaf(m[j], LOG2E, k)) + __builtin_amdgcn_exp2f(fmaf(v[j], LOG2E, k));
            m[j] = mn;
            if (SMOOTH) swx[j] = fmaf(s_w[c], v[j], swx[j]);
            if (TRACK_T) xt[j] = (t[j] == c) ? v[j] : xt[j];
        }
    }
}

// body of workgroup bx (of nbx per image) of image b: shared by k_ce_fwd and the forward-only
// cross-entropy items of the multi-loss launch (k_multi_loss, MODE 1)
// label 0..C (0 = void) of pixel i: uint8, or int16 for more than 255 classes (`wide`)
__device__ __forceinline__ int ce_label(const uint8_t* __restrict__ target, int wide, size_t i)
{
    return wide ? (int)((const int16_t*)target)[i] : (int)target[i];
}

template <int DTYPE, int PXT, bool SMOOTH, int U>
__device__ __forceinline__ void ce_fwd_body(
    const void* __restrict__ logits, const uint8_t* __restrict__ target,
    const float* __restrict__ weights, int C, int P, float ls, int vec,
    LossPartial* __restrict__ slot, int* __restrict__ status, float* __restrict__ lse2_out,
    float* s_w, int bx, int nbx, int b, int wide = 0)
{
    for (int c = threadIdx.x; c < C; c += LOSS_THREADS) s_w[c] = weights ? weights[c] : 1.0f;
    __syncthreads();
    float wsum = 0.f;
    if (SMOOTH) for (int c = 0; c < C; ++c) wsum += s_w[c];
    const size_t img = (size_t)b * C * P;
    double acc = 0.0, accw = 0.0;
    long long cnt = 0;
    bool bad = false;
    for (int p0 = (bx * LOSS_THREADS + threadIdx.x) * PXT; p0 < P; p0 += nbx * LOSS_THREADS * PXT) {
        const int nvalid = min(PXT, P - p0);
        float m[PXT], s[PXT], swx[PXT], xts[PXT];
        int tt[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j)
            tt[j] = (j < nvalid) ? ce_label(target, wide, (size_t)b * P + p0 + j) - 1 : -1;       // ce.py:46
        ce_scan<DTYPE, PXT, U, SMOOTH, true, true>(logits, img, P, p0, nvalid, vec, C, s_w, tt,
                                                   m, s, swx, xts);
        if (lse2_out) {
            // log2-domain log-sum-exp per pixel, kept for the backward pass (one read of the
            // logits there instead of two)
            float k0[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j) k0[j] = -(fmaf(m[j], LOG2E, __log2f(s[j])));
            float* q = lse2_out + (size_t)b * P + p0;
            if (vec && nvalid == PXT) {
#pragma unroll
                for (int j = 0; j < PXT; j += 4)
                    *(float4*)(q + j) = make_float4(k0[j], k0[j + 1], k0[j + 2], k0[j + 3]);
            } else {
                for (int j = 0; j < nvalid; ++j) q[j] = k0[j];
            }
        }
        float part = 0.f, partw = 0.f;
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const int t = tt[j];
            if (t < 0) continue;                                            // void: ignore_index
            if (t >= C) { bad = true; continue; }
            const float xt = xts[j];
            const float lse = fmaf(__log2f(s[j]), LN2, m[j]);
            const float wt = s_w[t];
            float l = (1.0f - ls) * wt * (lse - xt);
            if (SMOOTH) l += (ls / C) * (lse * wsum - swx[j]);
            part += l;
            partw += wt;
            ++cnt;
        }
        acc += part; accw += partw;
    }
    if (bad) atomicOr(status, 8);
    block_partial_at(acc, accw, cnt, slot);
}

// ---- forward + gradient in ONE pass over the logits ------------------------------------------
// The gradient of the summed loss needs the upstream scale g, which autograd hands over only
// in backward.  The callers know what it is going to be (1 / n for a mean, w / sum_scales n in
// the task helpers; n comes from a 1 B/px count over the labels, k_count_u8), so the forward
// kernel writes g_expected * d loss / d logits right away and the backward launch only confirms
// it (grad_already_computed).  Per px: logits 2C|4C read once + gradient written once instead
// of forward read + log-sum-exp write + backward read + log-sum-exp read + gradient write.
//
// A lane keeps its pixels' WHOLE class column in registers (8 B per plane and lane: 4 px of a
// 16-bit dtype, 2 px of f32; 8*NG planes -> 16*NG VGPRs), so the maximum, the sum of
// exponentials and the softmax each walk registers, not memory.  C <= 48 (NG <= 6); larger C
// falls back to the two-kernel path.

template <int DTYPE>
__device__ __forceinline__ u32x2_s ld_plane8(const void* base, size_t off, int nvalid, bool vec)
{
    if (DTYPE == NMSA_F32) {
        const float* p = (const float*)base + off;
        if (vec) return __builtin_nontemporal_load((const u32x2_s*)p);
        u32x2_s r = {0u, 0u};
        if (nvalid > 0) r.x = __float_as_uint(p[0]);
        if (nvalid > 1) r.y = __float_as_uint(p[1]);
        return r;
    }
    const uint16_t* p = (const uint16_t*)base + off;
    if (vec) return __builtin_nontemporal_load((const u32x2_s*)p);
    uint16_t h[4] = {0, 0, 0, 0};
    for (int j = 0; j < 4; ++j) if (j < nvalid) h[j] = p[j];
    u32x2_s r = {h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16)};
    return r;
}

template <int DTYPE>
__device__ __forceinline__ float plane_px(const u32x2_s r, int j)
{
    if (DTYPE == NMSA_F32) return __uint_as_float(j == 0 ? r.x : r.y);
    const uint32_t w = (j < 2) ? r.x : r.y;
    if (DTYPE == NMSA_BF16) return __uint_as_float((j & 1) ? (w & 0xFFFF0000u) : (w << 16));
    return f16_to_f32((uint16_t)((j & 1) ? (w >> 16) : (w & 0xFFFFu)));
}

// The register tile after the sum-of-exp2 walk: the exponentials e = 2^((x - max) log2e) REPLACE the
// logits, so the gradient walk multiplies instead of calling v_exp_f32 a second time per element —
// f32 tiles: as they are (exact); bf16 tiles: as fp16 pairs of e * 2^14 (CE_EXP_SHIFT: the scale is
// folded into the exponent's offset, the sum is scaled back exactly and the per-pixel factor
// abg / s carries the 2^-14), so every e >= 2^-28 is an fp16 NORMAL with relative error 2^-11 —
// three bits below the half ulp of the bf16 gradient it ends up in; unscaled, classes more than
// 9.7 below the maximum would land in fp16 subnormals (3 % off at e = 1e-6, zero below 3e-8).
// Smaller e (gaps beyond 19.4) fade out through the subnormals: absolute error <= 2^-39 of the
// maximum's probability.  f16 tiles keep their logits and
// exp_px computes e again: an fp16 copy would be no finer than the f16 gradient itself.  The
// target class, where p - 1 would cancel, is computed from its logit in full precision either way.
constexpr float CE_EXP_SHIFT = 14.0f;                  // bf16 tiles: e is kept as e * 2^14 in fp16
typedef _Float16 f16x2_s __attribute__((ext_vector_type(2)));
typedef float f32x2_s __attribute__((ext_vector_type(2)));
template <int DTYPE>
__device__ __forceinline__ void pack_exps(u32x2_s& r, const float* e)
{
    if (DTYPE == NMSA_F32) { r.x = __float_as_uint(e[0]); r.y = __float_as_uint(e[1]); }
    if (DTYPE == NMSA_BF16) {
        const f32x2_s a = {e[0], e[1]}, b = {e[2], e[3]};
        r.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, f16x2_s));
        r.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, f16x2_s));
    }
}
// e of pixel j; k0 = -max log2e (only the f16 tile, which still holds the logit, needs it)
template <int DTYPE>
__device__ __forceinline__ float exp_px(const u32x2_s r, int j, float k0)
{
    if (DTYPE == NMSA_F32) return __uint_as_float(j == 0 ? r.x : r.y);
    const f16x2_s h = __builtin_bit_cast(f16x2_s, (j < 2) ? r.x : r.y);
    const float v = (float)((j & 1) ? h.y : h.x);
    if (DTYPE == NMSA_BF16) return v;
    return __builtin_amdgcn_exp2f(fmaf(v, LOG2E, k0));
}

template <int DTYPE>
__device__ __forceinline__ void st_plane8(void* base, size_t off, int nvalid, bool vec, const float* v)
{
    if (DTYPE == NMSA_F32) {
        float* p = (float*)base + off;
        if (vec) {
            const u32x2_s w = {__float_as_uint(v[0]), __float_as_uint(v[1])};
            if (GRAD_NT) __builtin_nontemporal_store(w, (u32x2_s*)p); else *(u32x2_s*)p = w;
        } else {
            for (int j = 0; j < nvalid; ++j) p[j] = v[j];
        }
        return;
    }
    uint16_t* p = (uint16_t*)base + off;
    uint16_t h[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) h[j] = (DTYPE == NMSA_BF16) ? f32_to_bf16(v[j]) : f32_to_f16(v[j]);
    if (vec) {
        const u32x2_s w = {h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16)};
        if (GRAD_NT) __builtin_nontemporal_store(w, (u32x2_s*)p); else *(u32x2_s*)p = w;
    } else {
        for (int j = 0; j < nvalid; ++j) p[j] = h[j];
    }
}

constexpr int CE_FUSED_MAX_C = 48;
#ifndef NMSA_CE_FUSED_KEEP_PACKED
#define NMSA_CE_FUSED_KEEP_PACKED 1
#endif

// Between the three walks over the register tile the compiler would rather keep the UNPACKED
// fp32 values of a 16-bit tile (4 px x 40 planes = 160 more VGPRs, 1-2 waves per SIMD) than
// unpack again (one shift / and per element); this makes the tile opaque so it stays packed.
template <int NP>
__device__ __forceinline__ void keep_packed(u32x2_s (&r)[NP])
{
#if NMSA_CE_FUSED_KEEP_PACKED
#pragma unroll
    for (int c = 0; c < NP; ++c) { asm volatile("" : "+v"(r[c].x), "+v"(r[c].y)); }
#endif
}

// LOSS = false: the confirming backward launch — returns at once when the gradient written by
// the forward launch was computed for the real upstream scale, otherwise recomputes it with the
// same single pass (a miss costs one read of the logits + one gradient write, no more than the
// backward of the two-kernel path)
// body of workgroup (bx, b): shared by k_ce_fused and the multi-loss launch (k_multi_loss).
// g = upstream scale the gradient is written for; a NaN g (no expectation) writes no gradient.
// MODE 0: loss + gradient (a NaN g: loss only, at the price of the gradient arithmetic), 1: loss
// only (forward-only calls: no third walk over the registers), 2: gradient only
template <int DTYPE, int NG, bool SMOOTH, int MODE>
__device__ __forceinline__ void ce_fused_body(
    const void* __restrict__ logits, const uint8_t* __restrict__ target,
    const float* __restrict__ weights, int C, int P, float ls, int vec, float g,
    void* __restrict__ grad, LossPartial* __restrict__ slot, int* __restrict__ status,
    float* s_w, int bx, int b)
{
    constexpr int PXT = (DTYPE == NMSA_F32) ? 2 : 4;
    constexpr int NP = 8 * NG;
    constexpr bool LOSS = MODE != 2;
    for (int c = threadIdx.x; c < C; c += LOSS_THREADS) s_w[c] = weights ? weights[c] : 1.0f;
    __syncthreads();
    float wsum = 0.f;
    if (SMOOTH) for (int c = 0; c < C; ++c) wsum += s_w[c];
    const bool write_grad = MODE != 1 && (MODE == 2 || g == g) && grad != nullptr;
    const size_t img = (size_t)b * C * P;
    double acc = 0.0, accw = 0.0;
    long long cnt = 0;
    bool bad = false;
    const int p0 = (bx * LOSS_THREADS + threadIdx.x) * PXT;
    if (p0 < P) {
        const int nvalid = min(PXT, P - p0);
        u32x2_s r[NP];
        // (`c < C` is compared again in every walk — a scalar compare and branch per plane; one
        // compare kept across the four walks is NP lane masks in scalar registers: 108 spilled)
        int C0 = C;
        asm volatile("" : "+s"(C0));
#pragma unroll
        for (int c = 0; c < NP; ++c)
            if (c < C0) r[c] = ld_plane8<DTYPE>(logits, img + (size_t)c * P + p0, nvalid, vec);
        int t[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j)
            t[j] = (j < nvalid) ? (int)target[(size_t)b * P + p0 + j] - 1 : -1;       // ce.py:46
        float m[PXT], s[PXT], swx[PXT], xt[PXT], k0[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j) { m[j] = -INFINITY; s[j] = 0.f; swx[j] = 0.f; xt[j] = 0.f; }
        int C1 = C;
        asm volatile("" : "+s"(C1));
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            if (c < C1) {
#pragma unroll
                for (int j = 0; j < PXT; ++j) m[j] = vmax(m[j], plane_px<DTYPE>(r[c], j));
            }
        }
#pragma unroll
        for (int j = 0; j < PXT; ++j) k0[j] = -m[j] * LOG2E;
        if (DTYPE != NMSA_F32) keep_packed(r);
        int C2 = C, C3 = C;
        asm volatile("" : "+s"(C2));
        if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                if (c < C2) {
#pragma unroll
                    for (int j = 0; j < PXT; ++j) {
                        const float x = plane_px<DTYPE>(r[c], j);
                        s[j] += __builtin_amdgcn_exp2f(fmaf(x, LOG2E, k0[j]));
                        if (SMOOTH) swx[j] = fmaf(s_w[c], x, swx[j]);
                        xt[j] = (t[j] == c) ? x : xt[j];                   // forward only: no third walk
                    }
                }
            }
        } else {
            // sum of exp2; the exponentials take the logits' place in the tile (pack_exps; bf16
            // tiles: scaled by 2^14 through the exponent offset, see pack_exps)
            constexpr bool SHIFTED = DTYPE == NMSA_BF16;
            float k0s[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j) k0s[j] = SHIFTED ? k0[j] + CE_EXP_SHIFT : k0[j];
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                if (c < C2) {
                    float e[PXT];
#pragma unroll
                    for (int j = 0; j < PXT; ++j) {
                        const float x = plane_px<DTYPE>(r[c], j);
                        e[j] = __builtin_amdgcn_exp2f(fmaf(x, LOG2E, k0s[j]));
                        s[j] += e[j];
                        if (SMOOTH) swx[j] = fmaf(s_w[c], x, swx[j]);
                        xt[j] = (t[j] == c) ? x : xt[j];
                    }
                    pack_exps<DTYPE>(r[c], e);
                }
            }
            // per pixel: abg / s for the walk, and the target class' gradient from its logit
            float abgs[PXT], qt[PXT];
            bool smooth_on[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                if (SHIFTED) s[j] *= 0x1p-14f;                             // exact: s >= 1
                const float k1 = -(fmaf(m[j], LOG2E, __log2f(s[j])));      // p = 2^(x log2e + k1)
                const bool on = t[j] >= 0 && t[j] < C;
                const float a = on ? (1.0f - ls) * s_w[t[j]] : 0.f;
                const float ag = g * a;
                const float abg = on ? g * (a + (SMOOTH ? (ls / C) * wsum : 0.f)) : 0.f;
                smooth_on[j] = SMOOTH && abg != 0.f;
                abgs[j] = SHIFTED ? (abg / s[j]) * 0x1p-14f : abg / s[j];  // the tile holds e * 2^14
                const float pt = __builtin_amdgcn_exp2f(fmaf(xt[j], LOG2E, k1));
                qt[j] = fmaf(abg, pt, smooth_on[j] ? -(g * (ls / C) * s_w[on ? t[j] : 0]) : 0.f) - ag;
            }
            if (DTYPE != NMSA_F32) keep_packed(r);
            asm volatile("" : "+s"(C3));
#pragma unroll
            for (int c = 0; c < NP; ++c) {
                if (c < C3) {
                    float o[PXT];
                    const float bjg = SMOOTH ? g * (ls / C) * s_w[c] : 0.f;
#pragma unroll
                    for (int j = 0; j < PXT; ++j) {
                        const float q = fmaf(abgs[j], exp_px<DTYPE>(r[c], j, k0[j]), smooth_on[j] ? -bjg : 0.f);
                        o[j] = (t[j] == c) ? qt[j] : q;
                    }
                    if (write_grad) st_plane8<DTYPE>(grad, img + (size_t)c * P + p0, nvalid, vec, o);
                }
            }
        }
        float part = 0.f, partw = 0.f;
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            if (t[j] < 0) continue;                                         // void: ignore_index
            if (t[j] >= C) { bad = true; continue; }
            const float lse = fmaf(__log2f(s[j]), LN2, m[j]);
            const float wt = s_w[t[j]];
            float l = (1.0f - ls) * wt * (lse - xt[j]);
            if (SMOOTH) l += (ls / C) * (lse * wsum - swx[j]);
            part += l;
            partw += wt;
            ++cnt;
        }
        acc = part; accw = partw;
    }
    if (LOSS) {
        if (bad) atomicOr(status, 8);
        block_partial_at(acc, accw, cnt, slot);
    }
}

// =================================================================================
// a7: masked MSE / L1 with channel mean (C = 1: center, C = 2: offset)
//   loss = sum_px mean_c f(pred*mask - target);  n = sum(mask)
// =================================================================================
// KIND 2 — center focal loss (EXTENSION: the reference has only MSE / L1 for the center heat-map;
// the penalty-reduced focal loss of CenterNet, alpha = 2, beta = 4, on p = clamp(pred, 1e-4,
// 1 - 1e-4)):  -(1-p)^2 log p where target == 1,  -(1-target)^4 p^2 log(1-p) elsewhere; masked-out
// pixels contribute nothing and the count is the number of positive (target == 1) masked pixels.
__device__ __forceinline__ float focal_value(float x, float y)
{
    const float p = fminf(fmaxf(x, 1e-4f), 1.0f - 1e-4f);
    if (y == 1.0f) return -(1.0f - p) * (1.0f - p) * __logf(p);
    const float w = (1.0f - y) * (1.0f - y);
    return -w * w * p * p * __logf(1.0f - p);
}
__device__ __forceinline__ float focal_grad(float x, float y)
{
    if (!(x > 1e-4f && x < 1.0f - 1e-4f)) return 0.f;     // clamped: no gradient
    const float q = 1.0f - x;
    if (y == 1.0f) return 2.0f * q * __logf(x) - q * q / x;
    const float w = (1.0f - y) * (1.0f - y);
    return -w * w * (2.0f * x * __logf(q) - x * x / q);
}

// forward + gradient for the expected upstream scale (see k_ce_fused)
// body of workgroup bx (of nbx) of image b; gs = upstream scale (NaN: no gradient is written);
// LOSS = false: gradient only (the recomputing backward launch of k_multi_loss)
template <int DTYPE, int KIND, int MODE>                   // MODE as in ce_fused_body
__device__ __attribute__((noinline)) void elem_fused_body(
    const void* __restrict__ pred, const float* __restrict__ target, const uint8_t* __restrict__ mask,
    int C, int P, int vec, float gs, void* __restrict__ grad, LossPartial* __restrict__ slot,
    int bx, int nbx, int b)
{
    constexpr bool LOSS = MODE != 2;
    double acc = 0.0; long long cnt = 0;
    const float invC = 1.0f / C;
    const float g = gs / C;
    const bool write_grad = MODE != 1 && (MODE == 2 || gs == gs) && grad != nullptr;
    for (int p0 = (bx * LOSS_THREADS + threadIdx.x) * 4; p0 < P; p0 += nbx * LOSS_THREADS * 4) {
        const int nvalid = min(4, P - p0);
        bool mk[4];
        ld_mask4(mask, (size_t)b * P + p0, nvalid, vec, mk);
        float part = 0.f;
        for (int c = 0; c < C; ++c) {
            const size_t off = ((size_t)b * C + c) * P + p0;
            const float4 x = ld4<DTYPE>(pred, off, nvalid, vec);
            const float4 y = ld4<NMSA_F32>(target, off, nvalid, vec);
            const float xv[4] = {x.x, x.y, x.z, x.w}, yv[4] = {y.x, y.y, y.z, y.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = (mk[j] ? xv[j] : 0.f) - yv[j];       // pred*mask - target
                const float dd = (KIND == 2) ? focal_grad(xv[j], yv[j])
                               : (KIND == 0) ? 2.0f * d : (float)((d > 0.f) - (d < 0.f));
                o[j] = mk[j] ? g * dd : 0.f;
                if (j >= nvalid) continue;
                if (KIND == 2) {
                    if (mk[j]) { part += focal_value(xv[j], yv[j]); cnt += (yv[j] == 1.0f); }
                    continue;
                }
                part += (KIND == 0) ? d * d : fabsf(d);
            }
            if (write_grad) st4<DTYPE>(grad, off, nvalid, vec, o);
        }
        acc += part * invC;
        if (KIND != 2) for (int j = 0; j < 4; ++j) cnt += (j < nvalid) && mk[j];
    }
    if (LOSS) block_partial_at(acc, 0.0, cnt, slot);
}

// forward + gradient for the expected upstream scale (see k_ce_fused)
template <int DTYPE, int MODE>                             // MODE as in ce_fused_body
__device__ __attribute__((noinline)) void vm_fused_body(
    const void* __restrict__ pred, const float* __restrict__ target, const uint8_t* __restrict__ mask,
    int P, float kappa, int vec, float g, void* __restrict__ grad, LossPartial* __restrict__ slot,
    int bx, int nbx, int b)
{
    constexpr bool LOSS = MODE != 2;
    const bool write_grad = MODE != 1 && (MODE == 2 || g == g) && grad != nullptr;
    double acc = 0.0; long long cnt = 0;
    for (int p0 = (bx * LOSS_THREADS + threadIdx.x) * 4; p0 < P; p0 += nbx * LOSS_THREADS * 4) {
        const int nvalid = min(4, P - p0);
        bool mk[4];
        ld_mask4(mask, (size_t)b * P + p0, nvalid, vec, mk);
        const size_t o0 = ((size_t)b * 2) * P + p0, o1 = o0 + P;
        const float4 x0 = ld4<DTYPE>(pred, o0, nvalid, vec), x1 = ld4<DTYPE>(pred, o1, nvalid, vec);
        const float4 y0 = ld4<NMSA_F32>(target, o0, nvalid, vec), y1 = ld4<NMSA_F32>(target, o1, nvalid, vec);
        const float a0[4] = {x0.x, x0.y, x0.z, x0.w}, a1[4] = {x1.x, x1.y, x1.z, x1.w};
        const float b0[4] = {y0.x, y0.y, y0.z, y0.w}, b1[4] = {y1.x, y1.y, y1.z, y1.w};
        float g0[4], g1[4];
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dot = fmaf(a1[j], b1[j], a0[j] * b0[j]);
            const float ex = __expf(kappa * (dot - 1.0f));
            const float e = mk[j] ? -g * kappa * ex : 0.f;
            g0[j] = e * b0[j]; g1[j] = e * b1[j];
            if (mk[j]) { part += 1.0f - ex; ++cnt; }
        }
        acc += part;
        if (write_grad) {
            st4<DTYPE>(grad, o0, nvalid, vec, g0);
            st4<DTYPE>(grad, o1, nvalid, vec, g1);
        }
    }
    if (LOSS) block_partial_at(acc, 0.0, cnt, slot);
}

// register-resident cross entropy: groups of 8 class planes per lane for C classes
inline int ce_fused_ng(int C) { return (C <= 24) ? 3 : (C <= 40) ? 5 : 6; }

inline int loss_grid_x(int P, int px_per_thread)
{
    const int64_t per_block = (int64_t)LOSS_THREADS * px_per_thread;
    int64_t g = (P + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace nmsa
