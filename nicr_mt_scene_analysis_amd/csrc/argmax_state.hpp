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
    if (v > s.m[j]) { s.m[j] = v; s.am[j] = c; }
}

}  // namespace nmsa
