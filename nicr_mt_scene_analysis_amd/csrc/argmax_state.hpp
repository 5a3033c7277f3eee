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

// ---- max(softmax(x)) vs argmax(x): the probability-tie trigger (see panoptic.hip) ----------
template <int DTYPE>
__device__ __forceinline__ float tie_band_magnitude()
{
    // f32: spacing 2^-25 in [0.25, 0.5); bf16 (8-bit significand): below 2^-17; f16: its finest
    // spacing is 2^-24, never
    return (DTYPE == NMSA_F32) ? 0.5f : (DTYPE == NMSA_BF16) ? 0x1p-17f : 0.0f;
}

template <int DTYPE>
__device__ __forceinline__ bool may_tie_in_probability(float m)
{
    // (<=: a maximum of exactly 0.5 has its lower neighbour 2^-25 away)
    return fabsf(m) <= tie_band_magnitude<DTYPE>();
}

}  // namespace nmsa
