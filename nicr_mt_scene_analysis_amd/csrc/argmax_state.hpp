// argmax_state.hpp — per-pixel running class argmax + online softmax denominator,
// shared by panoptic.hip (network resolution) and resize.hip (full resolution).
#pragma once
#include "nmsa_common.hpp"

namespace nmsa {

// ---- per-pixel class argmax state ---------------------------------------------
struct ArgmaxState {
    float m[4];
    int am[4];
    float se[4];     // running sum of exp(x - m) (only when WITH_SCORE)
    float nf[4];     // stays 0 while every logit is finite, NaN as soon as one is NaN / +-inf
};

__device__ __forceinline__ void argmax_init(ArgmaxState& s)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) { s.m[j] = -INFINITY; s.am[j] = 0; s.se[j] = 0.f; s.nf[j] = 0.f; }
}

template <bool WITH_SCORE>
__device__ __forceinline__ void argmax_step(ArgmaxState& s, int j, float v, int c)
{
    s.nf[j] = fmaf(v, 0.0f, s.nf[j]);          // finite: += +-0 ; NaN / inf: NaN
    if (WITH_SCORE) {
        // online softmax denominator: one exp per class
        // (-inf logits contribute exp(-inf) = 0; avoid the NaN of -inf - -inf)
        const float e = (v == -INFINITY) ? 0.f : __expf(-fabsf(v - s.m[j]));
        s.se[j] = (v > s.m[j]) ? fmaf(s.se[j], e, 1.0f) : (s.se[j] + e);
    }
    const bool above = v > s.m[j];             // selects, not a branch: 1 compare + 2 cndmask
    s.am[j] = above ? c : s.am[j];
    s.m[j] = above ? v : s.m[j];
}

// Four classes c0..c0+3 of pixel j at once (values already in registers): one running-max
// update and ONE rescale of the softmax denominator per group instead of per class.
// No `nf` bookkeeping: a NaN / +inf logit, or a prefix of nothing but -inf, turns `se` into NaN,
// and the caller re-evaluates such a column exactly (a superset of the degenerate columns).
__device__ __forceinline__ void argmax_group4_score(ArgmaxState& s, int j, const float v[4], int c0)
{
    constexpr float kLog2e = 1.4426950408889634f;
    const float gm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    const float nm = fmaxf(s.m[j], gm);
    const float rescale = __builtin_amdgcn_exp2f((s.m[j] - nm) * kLog2e);
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) e += __builtin_amdgcn_exp2f((v[i] - nm) * kLog2e);
    s.se[j] = fmaf(s.se[j], rescale, e);
    const int first = (v[0] == gm) ? 0 : (v[1] == gm) ? 1 : (v[2] == gm) ? 2 : 3;
    s.am[j] = (gm > s.m[j]) ? c0 + first : s.am[j];
    s.m[j] = nm;
}

// ---- max(softmax(x)) vs argmax(x): where two DISTINCT logits get the same fp32 probability ----
// The reference takes max(softmax(x), dim=1) (semantic.py:52-53).  softmax is monotone, so that is
// argmax(x) with first-index ties — except where a class just below the maximum ends up with the
// maximum's probability after the fp32 roundings of ATen's softmax; then the LOWER index wins.
// ATen's CPU kernel for a non-last dim (vec_softmax, vectorised over the pixels) computes per
// pixel, sequentially over the classes:  e_c = Vectorized<float>::exp(x_c - max) — Sleef's
// expf_u10 as bundled with torch (FMA build):  q = rint(d * log2e),  s = fma(q, -L2U, d),
// s = fma(q, -L2L, s),  u = Horner(s) with FMAs,  u = fma(s*s, u, s) + 1,  u * 2^q —,
// S = ((e_0 + e_1) + ...) in fp32,  p_c = e_c / S (IEEE division).  aten_vec_expf and
// same_probability_as_max restate exactly that; tests/golden/argmax_ties.npz (run by the
// reference) pins it: all 1680 near-tie columns — the 229 with gaps in (2^-25, 2^-23] included —
// and both whole maps.  (Not reproduced: the last inner % 16 pixels of each host THREAD's chunk
// take ATen's scalar tail with libm's expf — a property of the host's thread count, not of the
// reference)
// A lower class can only share the maximum's probability when its e is 1 or 1 - 2^-24, i.e. its
// logit is less than 1.5 * 2^-24 below the maximum (e <= 1 - 2^-23 otherwise, and quotients a
// relative 2^-23 apart are at least one ulp apart: never the same float); two distinct values
// are only that close where the format's spacing is <= 2^-24: the per-pixel trigger below.
__device__ __forceinline__ float aten_vec_expf(float d)            // d <= 0 (or NaN handled by the caller)
{
    if (!(d >= -104.0f)) return 0.0f;                              // Sleef: d < -104 -> 0 (also -inf)
    const float q = rintf(__fmul_rn(d, 1.442695040888963407359924681001892137426645954152985934135449406931f));
    float s = __fmaf_rn(q, -0.693145751953125f, d);
    s = __fmaf_rn(q, -1.428606765330187045e-06f, s);
    float u = 0.000198527617612853646278381f;
    u = __fmaf_rn(u, s, 0.00139304355252534151077271f);
    u = __fmaf_rn(u, s, 0.00833336077630519866943359f);
    u = __fmaf_rn(u, s, 0.0416664853692054748535156f);
    u = __fmaf_rn(u, s, 0.166666671633720397949219f);
    u = __fmaf_rn(u, s, 0.5f);
    u = __fadd_rn(__fmaf_rn(__fmul_rn(s, s), u, s), 1.0f);
    const int qi = (int)q, qh = qi >> 1;                           // ldexp2kf: two exact power-of-two factors
    u = __fmul_rn(__fmul_rn(u, __int_as_float((qh + 127) << 23)), __int_as_float((qi - qh + 127) << 23));
    return u;
}

constexpr float TIE_CANDIDATE_GAP = -0x1p-23f;     // superset of the gaps that can tie (< 1.5 * 2^-24)

// column with maximum m at class am (first index of the maximum logit), `ld(c)` = logit of class c:
// the class the reference returns, and the maximum's probability (its `semantic_segmentation_score`)
template <typename LD>
__device__ __forceinline__ int class_by_probability(LD ld, int C, float m, int am, int first_candidate,
                                                    float* p_max)
{
    float S = 0.f;
    for (int c = 0; c < C; ++c) S = __fadd_rn(S, aten_vec_expf(__fsub_rn(ld(c), m)));
    const float pm = __fdiv_rn(1.0f, S);                           // e of the maximum is exactly 1
    if (p_max) *p_max = pm;
    for (int c = first_candidate; c < am; ++c) {
        const float d = __fsub_rn(ld(c), m);
        if (d >= TIE_CANDIDATE_GAP && __fdiv_rn(aten_vec_expf(d), S) == pm) return c;
    }
    return am;
}

template <int DTYPE>
__device__ __forceinline__ float tie_band_magnitude()
{
    // largest |max| whose lower neighbour is <= 2^-24 away: f32 [0.5, 1]; bf16 (8-bit
    // significand) [2^-17, 2^-16]; f16 (11-bit; 2^-24 is its subnormal spacing) up to 2^-13
    return (DTYPE == NMSA_F32) ? 1.0f : (DTYPE == NMSA_BF16) ? 0x1p-16f : 0x1p-13f;
}

template <int DTYPE>
__device__ __forceinline__ bool may_tie_in_probability(float m)
{
    // (<=: a maximum of exactly 1.0 has its lower neighbour 2^-24 away)
    return fabsf(m) <= tie_band_magnitude<DTYPE>();
}

}  // namespace nmsa
