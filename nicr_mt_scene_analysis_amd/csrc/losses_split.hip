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

// Issue slots (the kernel was VALU-bound: ~21 slots per element, two quarter-rate exponentials each):
// * the exponentials of the sum walk REPLACE the logits in the register tile (pack_exps, as in
//   ce_fused_body: f32 as they are, bf16 as fp16 pairs of e * 2^14), so the gradient walk multiplies
//   instead of a second v_exp_f32 per element; the target class is computed from its logit;
// * the target-class selects (compare + conditional move per element, in the sum walk and in the
//   gradient walk) run only on the planes a wave's 64 x PXT pixels actually have as target: a
//   64-bit wave-uniform `present` mask per tile, one scalar bit test per plane — label maps are
//   piecewise constant, most planes have no hit.  (Fetching the target logit with its own 2-byte
//   load per pixel instead was tried and is far slower: 4.0 instead of 1.6 ms — 64 different lines
//   per wave instruction with random labels, and a dependent load behind the whole tile);
// * VEC is a template parameter (no per-plane branch, no ragged code in the hot instantiation),
//   dead lanes of the last tile read the image's first pixels instead of carrying a lane mask into
//   every load, and the plane addresses are ONE running scalar base per walk + one lane offset per
//   tile (the 2 x NP loop-invariant plane offsets and NP "plane in range" masks the compiler hoisted
//   out of the tile loop before cost 330-700 SGPR spills to lane registers, read back per plane);
// * the planes are walked in groups of eight: a group that is entirely in range (all but the last
//   one or two) is straight-line code — eight loads issued back to back, 32 independent
//   exponentials per basic block — only the ragged last group tests every plane.
// MODE 0: gradient, and the loss when `partials` is given (without: the confirming / recomputing
// backward launch); MODE 1: loss only (forward-only calls: no gradient walk)
// (occupancy: the register tile is 16 NG VGPRs; the long straight-line groups would otherwise be
// scheduled into 2 waves per SIMD at NG = 5)
template <int DTYPE, int NG, bool SMOOTH, int MODE, bool VEC>
__global__ __launch_bounds__(LOSS_THREADS) __attribute__((amdgpu_waves_per_eu(NG <= 5 ? 3 : 2))) void k_ce_split(
    const void* __restrict__ logits, const uint8_t* __restrict__ target,
    const float* __restrict__ weights, int C, int P, float ls,
    const float* __restrict__ expected_gscale, void* __restrict__ grad,
    LossPartial* __restrict__ partials, int* __restrict__ status,
    const float* __restrict__ computed_for, int* __restrict__ counters, int tiles_per_wg)
{
    constexpr int PXT = (DTYPE == NMSA_F32) ? 2 : 4;
    constexpr int ESZ = (DTYPE == NMSA_F32) ? 4 : 2;
    constexpr int NP = 8 * NG;                         // class planes per wave
#ifndef NMSA_CE_ABL
#define NMSA_CE_ABL 0          // diagnostics: 1 no exponentials, 2 no exchange barriers, 4 no stores
#endif
#ifndef NMSA_CE_SPLIT_SUM1
#define NMSA_CE_SPLIT_SUM1 1
#endif
#ifndef NMSA_CE_SPLIT_GS
#define NMSA_CE_SPLIT_GS 1          // (with runs of two tiles per workgroup 1, 4 and 8 time the same: 1 is the least code)
#endif
    constexpr int GS = NMSA_CE_SPLIT_GS;               // planes per straight-line group
    constexpr int NWV = LOSS_THREADS / 64;             // 4
    const bool LOSS = partials != nullptr;
    constexpr int TPX = 64 * PXT;                      // pixels per workgroup
    constexpr bool SHIFTED = DTYPE == NMSA_BF16 && MODE != 1;      // the tile holds e * 2^14 (pack_exps)
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
    const bool write_grad = (!LOSS || g == g) && grad != nullptr;
    const int b = blockIdx.y;
    const size_t img = (size_t)b * C * P;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id();
    const int CQ = (C + NWV - 1) / NWV;
    const int c0 = w * CQ;                             // my classes: c0 .. min(c0 + CQ, C) - 1
    const int nc = max(0, min(CQ, C - c0));            // wave-uniform
    const char* lbase = (const char*)logits + (img + (size_t)c0 * P) * ESZ;
    char* gbase = (char*)grad + (img + (size_t)c0 * P) * ESZ;
    const size_t pstride = (size_t)P * ESZ;
    // a workgroup walks a RUN of consecutive pixel tiles: its C class planes are C different pages,
    // and one 512-byte piece per page and workgroup left the address translation as the limit
    // (4.6 TB/s with every exp removed; runs of tiles: the pages are reused tile after tile).
    // Runs of 2 since round 4 (NMSA_CE_SPLIT_RUN): 1.47 / 1.545 ms against 1.56 / 1.59 with runs
    // of 4 and 1.56 with single tiles, six and four processes each on two boxes — the spread
    // between processes (+-3-5 %, where the tensors land) had hidden it in single runs
    double acc = 0.0, accw = 0.0;
    long long cnt = 0;
    bool bad = false;
    // planes 0 .. n - 1 of this wave, GS at a time (n wave-uniform)
    auto for_planes = [&](int n, auto&& body) {
#pragma unroll
        for (int gq = 0; gq < NP / GS; ++gq) {
            if (VEC && gq * GS + GS <= n) {
#pragma unroll
                for (int k = 0; k < GS; ++k) body(gq * GS + k);
            } else {
#pragma unroll
                for (int k = 0; k < GS; ++k) if (gq * GS + k < n) body(gq * GS + k);
            }
        }
    };
    // one plane per (scalar) branch for the walks that DEFINE the register tile (the loads, the sum
    // walk that packs the exponentials into it): in the two-path form above every r[i] has two
    // definitions and the register allocator keeps copies — 168 registers + 100 B of scratch per lane
    // (8 % more HBM writes) against 139 and none
    auto for_planes_sum = [&](int n, auto&& body) {        // the sum walk rewrites the tile (pack_exps)
        if (MODE != 1 && NMSA_CE_SPLIT_SUM1) {
#pragma unroll
            for (int i = 0; i < NP; ++i) if (i < n) body(i);
        } else {
            for_planes(n, body);
        }
    };
    const int n_tiles = (P + TPX - 1) / TPX;
    const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(n_tiles, t_begin + tiles_per_wg);
    // lane offset of a tile inside a plane (P * ESZ < 2^32); VEC: dead lanes of the last tile read
    // (and drop) the image's first pixels
    auto lane_off = [&](int tile_) -> uint32_t {
        const int q = (tile_ * 64 + l) * PXT;
        return (uint32_t)(q < P ? q : 0) * ESZ;
    };
    u32x2_s r[NP];
    // the planes of a tile -> the register tile (ragged rows: pixel by pixel, zero-filled)
    auto request_plane = [&](int i, size_t po, int tile_) {
        if (VEC) r[i] = __builtin_nontemporal_load((const u32x2_s*)(lbase + po + lane_off(tile_)));
        else {
            const int q = (tile_ * 64 + l) * PXT;
            r[i] = u32x2_s{0u, 0u};
            if (q < P) r[i] = ld_plane8<DTYPE>(lbase + po + lane_off(tile_), 0, min(PXT, P - q), false);
        }
    };
  for (int tile = t_begin; tile < t_end; ++tile) {
    {
        size_t po = 0;                                 // wave-uniform running plane offset
        int ncl = nc;
        asm volatile("" : "+s"(ncl));                  // (compares per tile, not NP hoisted lane masks)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (i < ncl) {
                request_plane(i, po, tile);
                po += pstride;
                asm volatile("" : "+s"(po));
            }
        }
    }
    const int p0 = (tile * 64 + l) * PXT;
    const bool alive = p0 < P;
    const int nvalid = alive ? (VEC ? PXT : min(PXT, P - p0)) : 0;
    const uint32_t voff = lane_off(tile);
    int t[PXT];
    if (VEC) {
        uint32_t tw = 0u;
        if (alive) tw = (PXT == 4) ? *(const uint32_t*)(target + (size_t)b * P + p0)
                                   : (uint32_t)*(const uint16_t*)(target + (size_t)b * P + p0);
#pragma unroll
        for (int j = 0; j < PXT; ++j) t[j] = alive ? (int)((tw >> (8 * j)) & 0xFFu) - 1 : -1;       // ce.py:46
    } else {
#pragma unroll
        for (int j = 0; j < PXT; ++j) t[j] = (j < nvalid) ? (int)target[(size_t)b * P + p0 + j] - 1 : -1;
    }
    // planes of this wave that are the target of at least one of its pixels
    uint32_t plo = 0u, phi = 0u;
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        const int d = t[j] - c0;
        if (d >= 0 && d < nc) { if (d < 32) plo |= 1u << d; else phi |= 1u << (d - 32); }
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        plo |= (uint32_t)__shfl_xor((int)plo, o);
        if (NP > 32) phi |= (uint32_t)__shfl_xor((int)phi, o);
    }
    const uint64_t present = ((uint64_t)(NP > 32 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)phi) : 0u) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)plo);
    float m[PXT], s[PXT], swx[PXT], xt[PXT], k0[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) { m[j] = -INFINITY; s[j] = 0.f; swx[j] = 0.f; xt[j] = 0.f; }
    {
        int ncl = nc;
        asm volatile("" : "+s"(ncl));
        for_planes(ncl, [&](int i) {
#pragma unroll
            for (int j = 0; j < PXT; ++j) m[j] = vmax(m[j], plane_px<DTYPE>(r[i], j));
        });
    }
    // ---- the column maximum over the four waves ----------------------------------------------
#pragma unroll
    for (int j = 0; j < PXT; ++j) s_m[w * TPX + l * PXT + j] = m[j];
    if (!(NMSA_CE_ABL & 2)) __syncthreads();           // (also: the last tile's sums have been read)
    float k0s[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        float mm = s_m[l * PXT + j];
#pragma unroll
        for (int ww = 1; ww < NWV; ++ww) mm = vmax(mm, s_m[ww * TPX + l * PXT + j]);
        m[j] = mm;
        k0[j] = -mm * LOG2E;
        k0s[j] = SHIFTED ? k0[j] + CE_EXP_SHIFT : k0[j];
    }
    if (DTYPE != NMSA_F32) keep_packed(r);
    // ---- sum of exp2; MODE 0 / 2: the exponentials take the logits' place in the tile ---------------
    {
        int ncl = nc;
        asm volatile("" : "+s"(ncl));
        for_planes_sum(ncl, [&](int i) {
            const float wc = SMOOTH ? s_w[c0 + i] : 0.f;
            const bool hits = (present >> i) & 1ull;   // wave-uniform
            float e[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const float x = plane_px<DTYPE>(r[i], j);
                e[j] = (NMSA_CE_ABL & 1) ? fmaf(x, LOG2E, k0s[j]) : __builtin_amdgcn_exp2f(fmaf(x, LOG2E, k0s[j]));
                s[j] += e[j];
                if (SMOOTH) swx[j] = fmaf(wc, x, swx[j]);
            }
            if (hits) {
#pragma unroll
                for (int j = 0; j < PXT; ++j) xt[j] = (t[j] == c0 + i) ? plane_px<DTYPE>(r[i], j) : xt[j];
            }
            if (MODE != 1) pack_exps<DTYPE>(r[i], e);
        });
    }
    // ---- the sum of exp2 (and sum_c w_c x_c) over the four waves, in wave order -------------------
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        s_s[w * TPX + l * PXT + j] = s[j];
        if (SMOOTH) s_x[w * TPX + l * PXT + j] = swx[j];
    }
    if (!(NMSA_CE_ABL & 2)) __syncthreads();
    float abgs[PXT], qt[PXT];
    bool smooth_on[PXT];
#pragma unroll
    for (int j = 0; j < PXT; ++j) {
        float ss = s_s[l * PXT + j], sx = SMOOTH ? s_x[l * PXT + j] : 0.f;
#pragma unroll
        for (int ww = 1; ww < NWV; ++ww) {
            ss += s_s[ww * TPX + l * PXT + j];
            if (SMOOTH) sx += s_x[ww * TPX + l * PXT + j];
        }
        if (SHIFTED) ss *= 0x1p-14f;                                       // exact: the sum is >= 2^14
        s[j] = ss; swx[j] = sx;
        const float k1 = -(fmaf(m[j], LOG2E, __log2f(ss)));                // p = 2^(x log2e + k1)
        const bool on = t[j] >= 0 && t[j] < C;
        const float a = on ? (1.0f - ls) * s_w[t[j]] : 0.f;
        const float ag = g * a;
        const float abg = on ? g * (a + (SMOOTH ? (ls / C) * wsum : 0.f)) : 0.f;
        smooth_on[j] = SMOOTH && abg != 0.f;
        abgs[j] = SHIFTED ? (abg / ss) * 0x1p-14f : abg / ss;              // the tile holds e (* 2^14)
        // the target class, where p - 1 would cancel, from its logit (only the wave that holds it
        // has the logit; the others never select this value)
        const float pt = __builtin_amdgcn_exp2f(fmaf(xt[j], LOG2E, k1));
        qt[j] = fmaf(abg, pt, smooth_on[j] ? -(g * (ls / C) * s_w[on ? t[j] : 0]) : 0.f) - ag;
    }
    if (MODE != 1) {
        if (DTYPE != NMSA_F32) keep_packed(r);
        const bool store = alive && write_grad && !(NMSA_CE_ABL & 4);
        {
            size_t po = 0;
            int ncl = nc;
            asm volatile("" : "+s"(ncl));
            for_planes(ncl, [&](int i) {
                const int c = c0 + i;
                const bool hits = (present >> i) & 1ull;
                float o[PXT];
                const float bjg = SMOOTH ? g * (ls / C) * s_w[c] : 0.f;
#pragma unroll
                for (int j = 0; j < PXT; ++j)
                    o[j] = fmaf(abgs[j], exp_px<DTYPE>(r[i], j, k0[j]), smooth_on[j] ? -bjg : 0.f);
                if (hits) {
#pragma unroll
                    for (int j = 0; j < PXT; ++j) o[j] = (t[j] == c) ? qt[j] : o[j];
                }
                if (store) st_plane8<DTYPE>(gbase + po + voff, 0, nvalid, VEC, o);
                po += pstride;
                asm volatile("" : "+s"(po));
            });
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
    static const int run = loss_env_int("NMSA_CE_SPLIT_RUN", 2);
    const int tpw = run < 1 ? 1 : run;
    return (n_tiles + tpw - 1) / tpw;
}

int launch_ce_split(bool loss, const void* logits, int dtype, const uint8_t* target,
                           const float* weights, int B, int C, int P, float ls, const float* gscale,
                           const float* computed_for, int32_t* counters, void* grad,
                           LossPartial* partials, int32_t* status, hipStream_t stream)
{
    const int pxt = (dtype == NMSA_F32) ? 2 : 4;
    const bool vec = (P % pxt == 0) && ((((uintptr_t)logits | (uintptr_t)grad) & 7) == 0) &&
                     ((((uintptr_t)target) & (uintptr_t)(pxt - 1)) == 0);
    const int n_tiles = (P + 64 * pxt - 1) / (64 * pxt);     // the four waves of a block share 64 x pxt pixels
    static const int run = loss_env_int("NMSA_CE_SPLIT_RUN", 2);
    const int tpw = run < 1 ? 1 : run;
    const int gx = (n_tiles + tpw - 1) / tpw;
    const bool smooth = ls != 0.0f;
    const size_t lds = ((size_t)((C + 3) & ~3) + (size_t)(smooth ? 3 : 2) * 4 * 64 * pxt) * sizeof(float);
    // planes per wave in groups of eight: 4 / 5 / 6 / 8 groups (<= 128 / 160 / 192 / 256 classes; the
    // register tile decides the occupancy: 3 waves per SIMD up to 5 groups, 2 above); rows that cannot be
    // read as 8-byte pieces (VEC = false) take the widest instantiation whatever C is
    const int per_lane = (C + 3) / 4;
    const int ng = !vec ? 8 : per_lane <= 32 ? 4 : per_lane <= 40 ? 5 : per_lane <= 48 ? 6 : 8;
    LossPartial* parts = loss ? partials : nullptr;    // none: the confirming / recomputing backward launch
    if (loss && !partials) return NMSA_ERR_ARG;
#define CE_SPLIT_V(DT, NG, SM, MD, V) hipLaunchKernelGGL((k_ce_split<DT, NG, SM, MD, V>), dim3(gx, B), dim3(LOSS_THREADS), \
        lds, stream, logits, target, weights, C, P, ls, gscale, grad, parts, status, \
        computed_for, counters, tpw)
#define CE_SPLIT_NG(DT, SM, MD) do { if (!vec) CE_SPLIT_V(DT, 8, SM, MD, false); else if (ng == 4) CE_SPLIT_V(DT, 4, SM, MD, true); \
        else if (ng == 5) CE_SPLIT_V(DT, 5, SM, MD, true); else if (ng == 6) CE_SPLIT_V(DT, 6, SM, MD, true); else CE_SPLIT_V(DT, 8, SM, MD, true); } while (0)
    // MODE 0: gradient (+ loss with partials), 1: loss only (no gradient buffer)
#define CE_SPLIT_M(DT, MD) do { if (smooth) CE_SPLIT_NG(DT, true, MD); else CE_SPLIT_NG(DT, false, MD); } while (0)
#define CE_SPLIT(DT) do { if (loss && !grad) CE_SPLIT_M(DT, 1); else CE_SPLIT_M(DT, 0); } while (0)
    NMSA_DISPATCH_DTYPE(dtype, CE_SPLIT)
#undef CE_SPLIT
#undef CE_SPLIT_M
#undef CE_SPLIT_NG
#undef CE_SPLIT_V
    return check_launch();
}

}  // namespace nmsa
