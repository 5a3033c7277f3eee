// losses_split.hip — cross entropy for class columns too wide for one lane's registers
// (49..256 classes): forward + gradient in one pass, the column split over the four waves of a
// workgroup.  Reference: CrossEntropyLossSemantic._compute_loss, loss/ce.py:40-68.
#include <stdlib.h>
#include "loss_bodies.hpp"

namespace nmsa {

// ---- the same for wider class columns: the column SPLIT OVER THE FOUR WAVES OF THE WORKGROUP -----
// k_ce_fused keeps a pixel's whole column in one lane's registers (C <= 48).  Here the four waves
// of a workgroup look at the SAME 64 x PXT pixels and wave w holds the classes [w CQ, (w + 1) CQ),
// CQ = ceil(C / 4) <= 64, so every wave-instruction still moves one contiguous 512-byte piece of a
// class plane (the access pattern of k_ce_fused; 128-byte row segments — a column spread over the
// lanes of ONE wave — ran at 3.6 instead of 5.4 TB/s).  The waves exchange their per-pixel maximum
// and sum of exp2 (and sum_c w_c x_c) through 8-12 KB of LDS, two barriers per workgroup; each
// wave then writes the gradient of its own classes from the same registers.  Logits read once,
// gradient written once for C <= 256.

template <int DTYPE, int NG, bool SMOOTH, int MODE>           // MODE as in ce_fused_body
__global__ __launch_bounds__(LOSS_THREADS) void k_ce_split(
    const void* __restrict__ logits, const uint8_t* __restrict__ target,
    const float* __restrict__ weights, int C, int P, float ls, int vec,
    const float* __restrict__ expected_gscale, void* __restrict__ grad,
    LossPartial* __restrict__ partials, int* __restrict__ status,
    const float* __restrict__ computed_for, int* __restrict__ counters, int tiles_per_wg)
{
    constexpr int PXT = (DTYPE == NMSA_F32) ? 2 : 4;
    constexpr int NP = 8 * NG;                         // class planes per wave
    constexpr int NWV = LOSS_THREADS / 64;             // 4
    constexpr bool LOSS = MODE != 2;
    constexpr int TPX = 64 * PXT;                      // pixels per workgroup
    extern __shared__ float s_w[];                     // [C] weights, then the exchange buffers
    if (!LOSS && grad_already_computed(expected_gscale, computed_for, counters)) return;
    float* s_m = s_w + ((C + 3) & ~3);                 // [NWV][TPX] maxima
    float* s_s = s_m + NWV * TPX;                      // [NWV][TPX] sums of exp2
    float* s_x = s_s + NWV * TPX;                      // [NWV][TPX] sum_c w_c x_c (label smoothing)
    for (int c = threadIdx.x; c < C; c += LOSS_THREADS) s_w[c] = weights ? weights[c] : 1.0f;
    __syncthreads();
    float wsum = 0.f;
    if (SMOOTH) for (int c = 0; c < C; ++c) wsum += s_w[c];
    // (no gradient buffer: forward only; a NaN expectation writes no gradient either)
    const float g = grad ? *expected_gscale : __int_as_float(0x7fc00000);
    const bool write_grad = (MODE == 2 || g == g) && grad != nullptr;
    const int b = blockIdx.y;
    const size_t img = (size_t)b * C * P;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id();
    const int CQ = (C + NWV - 1) / NWV;
    const int c0 = w * CQ;                             // my classes: c0 .. min(c0 + CQ, C) - 1
    const int nc = max(0, min(CQ, C - c0));            // wave-uniform
    // a workgroup walks a RUN of consecutive pixel tiles: its C class planes are C different pages,
    // and one 512-byte piece per page and workgroup left the address translation as the limit
    // (4.6 TB/s with every exp removed; runs of tiles: the pages are reused tile after tile)
    double acc = 0.0, accw = 0.0;
    long long cnt = 0;
    bool bad = false;
    const int n_tiles = (P + TPX - 1) / TPX;
    const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(n_tiles, t_begin + tiles_per_wg);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int p0 = (tile * 64 + l) * PXT;
    const bool alive = p0 < P;
    const int nvalid = alive ? min(PXT, P - p0) : 0;
    u32x2_s r[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        r[i] = u32x2_s{0u, 0u};
        if (i < nc && alive) r[i] = ld_plane8<DTYPE>(logits, img + (size_t)(c0 + i) * P + p0, nvalid, vec);
    }
    int t[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) t[j] = (j < nvalid) ? (int)target[(size_t)b * P + p0 + j] - 1 : -1;   // ce.py:46
    float m[PXT], s[PXT], swx[PXT], xt[PXT], k0[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) { m[j] = -INFINITY; s[j] = 0.f; swx[j] = 0.f; xt[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < nc) {
#pragma unroll
            for (int j = 0; j < PXT; ++j) m[j] = vmax(m[j], plane_px<DTYPE>(r[i], j));
        }
    }
    // ---- the column maximum over the four waves ----------------------------------------------
#pragma unroll
    for (int j = 0; j < PXT; ++j) s_m[w * TPX + l * PXT + j] = m[j];
    __syncthreads();                                   // (also: the last tile's sums have been read)
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        float mm = s_m[l * PXT + j];
#pragma unroll
        for (int ww = 1; ww < NWV; ++ww) mm = vmax(mm, s_m[ww * TPX + l * PXT + j]);
        m[j] = mm;
        k0[j] = -mm * LOG2E;
    }
    if (DTYPE != NMSA_F32) keep_packed(r);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if (i < nc) {
            const float wc = SMOOTH ? s_w[c0 + i] : 0.f;
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const float x = plane_px<DTYPE>(r[i], j);
                s[j] += __builtin_amdgcn_exp2f(fmaf(x, LOG2E, k0[j]));
                if (SMOOTH) swx[j] = fmaf(wc, x, swx[j]);
                if (MODE == 1) xt[j] = (t[j] == c0 + i) ? x : xt[j];            // forward only: no third walk
            }
        }
    }
    // ---- the sum of exp2 (and sum_c w_c x_c) over the four waves, in wave order -------------------
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        s_s[w * TPX + l * PXT + j] = s[j];
        if (SMOOTH) s_x[w * TPX + l * PXT + j] = swx[j];
    }
    __syncthreads();
    float ag[PXT], abg[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        float ss = s_s[l * PXT + j], sx = SMOOTH ? s_x[l * PXT + j] : 0.f;
#pragma unroll
        for (int ww = 1; ww < NWV; ++ww) {
            ss += s_s[ww * TPX + l * PXT + j];
            if (SMOOTH) sx += s_x[ww * TPX + l * PXT + j];
        }
        s[j] = ss; swx[j] = sx;
        k0[j] = -(fmaf(m[j], LOG2E, __log2f(ss)));                         // p = 2^(x log2e + k0)
        const bool on = t[j] >= 0 && t[j] < C;
        const float a = on ? (1.0f - ls) * s_w[t[j]] : 0.f;
        ag[j] = g * a;
        abg[j] = on ? g * (a + (SMOOTH ? (ls / C) * wsum : 0.f)) : 0.f;
    }
    if (DTYPE != NMSA_F32) keep_packed(r);
    if (MODE != 1) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (i < nc) {
                const int c = c0 + i;
                float o[PXT];
                const float bjg = SMOOTH ? g * (ls / C) * s_w[c] : 0.f;
#pragma unroll
                for (int j = 0; j < PXT; ++j) {
                    const float x = plane_px<DTYPE>(r[i], j);
                    const float pj = __builtin_amdgcn_exp2f(fmaf(x, LOG2E, k0[j]));
                    float qv = fmaf(abg[j], pj, (SMOOTH && abg[j] != 0.f) ? -bjg : 0.f);
                    const bool hit = t[j] == c;
                    qv -= hit ? ag[j] : 0.f;
                    xt[j] = hit ? x : xt[j];
                    o[j] = qv;
                }
                if (alive && write_grad) st_plane8<DTYPE>(grad, img + (size_t)c * P + p0, nvalid, vec, o);
            }
        }
    }
    if (LOSS) {
        float part = 0.f, partw = 0.f;
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            if (t[j] < 0) continue;                                         // void: ignore_index
            if (t[j] >= C) { bad = true; continue; }
            const float lse = fmaf(__log2f(s[j]), LN2, m[j]);
            const float wt = s_w[t[j]];
            // the wave holding the target class adds the pixel's main term, wave 0 counts the pixel
            if (t[j] >= c0 && t[j] < c0 + nc) part += (1.0f - ls) * wt * (lse - xt[j]);
            if (w == 0) {
                if (SMOOTH) part += (ls / C) * (lse * wsum - swx[j]);
                partw += wt;
                ++cnt;
            }
        }
        acc += part; accw += partw;
    }
  }
    if (LOSS) {
        if (bad) atomicOr(status, 8);
        block_partial(acc, accw, cnt, partials);
    }
}

}  // namespace nmsa

using namespace nmsa;

namespace nmsa {

int ce_split_blocks(int P, int dtype)
{
    const int pxt = (dtype == NMSA_F32) ? 2 : 4;
    const int n_tiles = (P + 64 * pxt - 1) / (64 * pxt);
    static const int run = loss_env_int("NMSA_CE_SPLIT_RUN", 4);
    const int tpw = run < 1 ? 1 : run;
    return (n_tiles + tpw - 1) / tpw;
}

int launch_ce_split(bool loss, const void* logits, int dtype, const uint8_t* target,
                           const float* weights, int B, int C, int P, float ls, const float* gscale,
                           const float* computed_for, int32_t* counters, void* grad,
                           LossPartial* partials, int32_t* status, hipStream_t stream)
{
    const int pxt = (dtype == NMSA_F32) ? 2 : 4;
    const int vec = (P % pxt == 0) && ((((uintptr_t)logits | (uintptr_t)grad) & 7) == 0);
    const int n_tiles = (P + 64 * pxt - 1) / (64 * pxt);     // the four waves of a block share 64 x pxt pixels
    static const int run = loss_env_int("NMSA_CE_SPLIT_RUN", 4);
    const int tpw = run < 1 ? 1 : run;
    const int gx = (n_tiles + tpw - 1) / tpw;
    const bool smooth = ls != 0.0f;
    const size_t lds = ((size_t)((C + 3) & ~3) + (size_t)(smooth ? 3 : 2) * 4 * 64 * pxt) * sizeof(float);
    const int per_lane = (C + 3) / 4;
    const int ng = per_lane <= 24 ? 3 : per_lane <= 32 ? 4 : per_lane <= 40 ? 5 : per_lane <= 48 ? 6 : 8;
#define CE_SPLIT_L(DT, NG, SM, LS) hipLaunchKernelGGL((k_ce_split<DT, NG, SM, LS>), dim3(gx, B), dim3(LOSS_THREADS), \
        lds, stream, logits, target, weights, C, P, ls, vec, gscale, grad, partials, status, \
        computed_for, counters, tpw)
#define CE_SPLIT_NG(DT, SM, LS) do { if (ng == 3) CE_SPLIT_L(DT, 3, SM, LS); else if (ng == 4) CE_SPLIT_L(DT, 4, SM, LS); \
        else if (ng == 5) CE_SPLIT_L(DT, 5, SM, LS); else if (ng == 6) CE_SPLIT_L(DT, 6, SM, LS); \
        else CE_SPLIT_L(DT, 8, SM, LS); } while (0)
    // MODE 0: loss + gradient, 1: loss only (no gradient buffer), 2: gradient only
#define CE_SPLIT_M(DT, LS) do { if (smooth) CE_SPLIT_NG(DT, true, LS); else CE_SPLIT_NG(DT, false, LS); } while (0)
#define CE_SPLIT(DT) do { if (!loss) CE_SPLIT_M(DT, 2); else if (grad) CE_SPLIT_M(DT, 0); else CE_SPLIT_M(DT, 1); } while (0)
    NMSA_DISPATCH_DTYPE(dtype, CE_SPLIT)
#undef CE_SPLIT
#undef CE_SPLIT_M
#undef CE_SPLIT_NG
#undef CE_SPLIT_L
    return check_launch();
}

}  // namespace nmsa
