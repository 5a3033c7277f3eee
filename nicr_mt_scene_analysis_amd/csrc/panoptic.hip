// panoptic.hip — semantic argmax + offset grouping + panoptic merge for gfx950.
//
// Replaces, behind the C ABI of include/nmsa.h, the ATen chains of
//   SemanticPostprocessing._postprocess_inference    (semantic.py:52-53)
//   InstancePostprocessing._get_instance_segmentation (instance.py:187-253)
//   PanopticPostprocessing._postprocess_inference     (panoptic.py:105-160)
//   deeplab_merge_semantic_and_instance               (panoptic_merge.py:172-225)
//
// Data flow (B images, P = H*W pixels each, C classes):
//   k_panoptic_fused  reads  logits 4C B/px (f32) + offset 8 B/px
//                     writes sem u8 1 B/px + inst u8 1 B/px (+ votes, tiny)
//   k_assign          1024 threads / image over the [256 x (C+1)] vote table (staged in LDS)
//   k_paint           reads sem+inst 2 B/px, writes panoptic i64 8 B/px
// All three are HBM-bound streaming kernels: 16-B loads per lane, >= 8 loads
// in flight per lane, no LDS staging of the big tensors (each byte is used
// once), LDS only for the center list and the privatised vote histogram.
//
// Exactness (ids must be bit-identical to the reference CPU path):
//  * loc = float(y) + off*scale  : explicit __fmul_rn / __fadd_rn (no FMA);
//  * distance^2 s = fma(dx,dx, dy*dy) — the exact rounding sequence of ATen's
//    norm (verified in oracle/gen_golden.py::check_norm_formula);
//  * the reference compares d = sqrt_rn(s) and takes the lowest index among
//    equal d.  sqrt_rn is monotone, so argmin is found WITHOUT a per-center
//    sqrt: pass 1 takes s_min = min_i s_i; d_min = sqrt_rn(s_min); U = largest
//    float with sqrt_rn(U) == d_min (from the exact fp64 square of the
//    rounding midpoint); pass 2 (descending i) takes the lowest i with
//    s_i <= U.  Both passes are branch-free compare/select.
#include <stdlib.h>
#include <type_traits>

#include "nmsa_common.hpp"
#include "argmax_state.hpp"

namespace nmsa {

constexpr int FUSED_THREADS = 256;
constexpr int FUSED_VOTE_SLOTS = 256;
constexpr int PX_PER_THREAD = 4;
constexpr int PX_PER_ITER = FUSED_THREADS * PX_PER_THREAD;   // 1024

// ---- typed 4-pixel loads -------------------------------------------------------
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));

template <int DTYPE, bool VEC, bool NT = false>
__device__ __forceinline__ float4 load_px4(const void* base, size_t elem_off, int nvalid)
{
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (DTYPE == NMSA_F32) {
        const float* p = (const float*)base + elem_off;
        if (VEC && NT) {
            // streamed once: non-temporal hint (no reuse -> do not keep the lines)
            const f32x4_t v = __builtin_nontemporal_load((const f32x4_t*)p);
            return make_float4(v.x, v.y, v.z, v.w);
        }
        if (VEC) return *(const float4*)p;
        if (nvalid > 0) r.x = p[0];
        if (nvalid > 1) r.y = p[1];
        if (nvalid > 2) r.z = p[2];
        if (nvalid > 3) r.w = p[3];
        return r;
    } else {
        const uint16_t* p = (const uint16_t*)base + elem_off;
        uint16_t h[4] = {0, 0, 0, 0};
        if (VEC && NT) {
            const u16x4_t u = __builtin_nontemporal_load((const u16x4_t*)p);
            h[0] = u.x; h[1] = u.y; h[2] = u.z; h[3] = u.w;
        } else if (VEC) {
            const ushort4 u = *(const ushort4*)p;
            h[0] = u.x; h[1] = u.y; h[2] = u.z; h[3] = u.w;
        } else {
            for (int j = 0; j < 4; ++j) if (j < nvalid) h[j] = p[j];
        }
        if (DTYPE == NMSA_BF16) {
            r.x = bf16_to_f32(h[0]); r.y = bf16_to_f32(h[1]);
            r.z = bf16_to_f32(h[2]); r.w = bf16_to_f32(h[3]);
        } else {
            r.x = f16_to_f32(h[0]); r.y = f16_to_f32(h[1]);
            r.z = f16_to_f32(h[2]); r.w = f16_to_f32(h[3]);
        }
        return r;
    }
}

// The reference takes max(softmax(x)) (semantic.py:52-53), the kernels argmax(x).  They differ
// on two kinds of columns, both re-evaluated exactly by the functions below (rare, re-read):
//  * a NaN, a +inf, or nothing but -inf: softmax is all-NaN and torch.max returns index 0;
//    a column with some -inf entries is an ordinary one;
//  * a class BELOW the maximum's index whose logit is so close to the maximum (< 1.5 * 2^-24)
//    that ATen's fp32 softmax gives both the same probability: the lower index wins.  Decided
//    with the reference's own arithmetic (argmax_state.hpp: class_by_probability); such gaps only
//    exist where the format's spacing is <= 2^-24, i.e. for |max| <= tie_band_magnitude — the
//    trigger, one compare per pixel.
// class of one column by the reference's rule, given its maximum m (the running maximum of the
// fast path; with a NaN / +inf in the column the answer is 0 whatever m is): one walk unless an
// earlier class is a candidate for a probability tie
template <int DTYPE>
__device__ __noinline__ int column_class(const void* logits, size_t col0, int P, int C, float m)
{
    auto ld = [&](int c) -> float {
        if (DTYPE == NMSA_F32) return ((const float*)logits)[col0 + (size_t)c * P];
        const uint16_t h = ((const uint16_t*)logits)[col0 + (size_t)c * P];
        return (DTYPE == NMSA_BF16) ? bf16_to_f32(h) : f16_to_f32(h);
    };
    bool nan_or_pinf = false, any_finite = false;
    int first = C, am = C;                          // lowest candidate / first index of the maximum
    int c = 0;
    for (; c + 4 <= C; c += 4) {                    // 4 loads in flight
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld(c + u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            nan_or_pinf |= (v[u] != v[u]) || (v[u] == INFINITY);
            any_finite |= fabsf(v[u]) < INFINITY;
            if (first == C && __fsub_rn(v[u], m) >= TIE_CANDIDATE_GAP) first = c + u;
            if (am == C && v[u] == m) am = c + u;
        }
    }
    for (; c < C; ++c) {
        const float v = ld(c);
        nan_or_pinf |= (v != v) || (v == INFINITY);
        any_finite |= fabsf(v) < INFINITY;
        if (first == C && __fsub_rn(v, m) >= TIE_CANDIDATE_GAP) first = c;
        if (am == C && v == m) am = c;
    }
    if (nan_or_pinf || !any_finite || am == C) return 0;
    if (first >= am) return am;                     // nobody below the maximum's index comes close
    return class_by_probability(ld, C, m, am, first, nullptr);
}

// Exact argmax + score of one column (the group-wise fast path of the WITH_SCORE kernels came
// out with a NaN denominator, or the maximum is small enough for a probability tie).
// returns (score, class index as int bits): by value, so that no scratch slot is needed
template <int DTYPE>
__device__ __noinline__ float2 column_exact(const void* logits, size_t col0, int P, int C)
{
    auto ld = [&](int c) -> float {
        if (DTYPE == NMSA_F32) return ((const float*)logits)[col0 + (size_t)c * P];
        const uint16_t h = ((const uint16_t*)logits)[col0 + (size_t)c * P];
        return (DTYPE == NMSA_BF16) ? bf16_to_f32(h) : f16_to_f32(h);
    };
    bool nan_or_pinf = false, any_finite = false;
    float m = -INFINITY;
    int am = 0;
    for (int c = 0; c < C; ++c) {
        const float v = ld(c);
        if (v != v || v == INFINITY) nan_or_pinf = true;
        if (fabsf(v) < INFINITY) any_finite = true;
        if (v > m) { m = v; am = c; }
    }
    if (nan_or_pinf || !any_finite) return make_float2(__int_as_float(0x7fc00000), __int_as_float(0));
    float se = 0.f;
    int first = am;
    for (int c = 0; c < C; ++c) {
        const float v = ld(c);
        se += (v == -INFINITY) ? 0.f : __expf(v - m);
        if (c < first && __fsub_rn(v, m) >= TIE_CANDIDATE_GAP) first = c;
    }
    if (first >= am) return make_float2(1.0f / se, __int_as_float(am));
    float pm;
    const int cls = class_by_probability(ld, C, m, am, first, &pm);
    return make_float2(pm, __int_as_float(cls));
}

// classes c0 .. c0+3 of the 4 pixels of a lane (WITH_SCORE: one rescale per group, see
// argmax_state.hpp; otherwise the per-class step)
// `nclasses` < 4: the last, padded group (planes beyond it hold -inf).  The score path takes the
// padding as it is (exp(-inf) = 0); the argmax-only path must SKIP it: its finite-tracker
// `nf += v * 0` would turn NaN on the -inf padding and send every pixel through the exact
// column re-check (measured 5x on the whole kernel at C = 150).
template <bool WITH_SCORE>
__device__ __forceinline__ void argmax_quad(ArgmaxState& st, const float4& a, const float4& b,
                                            const float4& c, const float4& d, int c0,
                                            int nclasses = 4)
{
    if (WITH_SCORE) {
        const float p0[4] = {a.x, b.x, c.x, d.x}, p1[4] = {a.y, b.y, c.y, d.y};
        const float p2[4] = {a.z, b.z, c.z, d.z}, p3[4] = {a.w, b.w, c.w, d.w};
        argmax_group4_score(st, 0, p0, c0);
        argmax_group4_score(st, 1, p1, c0);
        argmax_group4_score(st, 2, p2, c0);
        argmax_group4_score(st, 3, p3, c0);
    } else {
        const float4 q[4] = {a, b, c, d};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u >= nclasses) break;                     // wave-uniform
            argmax_step<false>(st, 0, q[u].x, c0 + u);
            argmax_step<false>(st, 1, q[u].y, c0 + u);
            argmax_step<false>(st, 2, q[u].z, c0 + u);
            argmax_step<false>(st, 3, q[u].w, c0 + u);
        }
    }
}

// ---- exact nearest-center search for 4 pixels -----------------------------------
__device__ __forceinline__ float sqdist(float cy, float cx, float ly, float lx)
{
    const float dy = __fsub_rn(cy, ly);
    const float dx = __fsub_rn(cx, lx);
    return __fmaf_rn(dx, dx, __fmul_rn(dy, dy));
}

// largest float U with sqrt_rn(U) == d   (d = sqrt_rn(s) >= 0, or inf / NaN)
__device__ __forceinline__ float sqrt_tie_upper(float d)
{
    if (!(d < INFINITY)) return d;                        // inf -> inf, NaN -> NaN
    const float dn = __uint_as_float(__float_as_uint(d) + 1u);   // next float up
    const double m = 0.5 * ((double)d + (double)dn);      // rounding midpoint (exact)
    const double m2 = m * m;                              // exact (<= 50 bits)
    float u = (float)m2;
    if ((double)u >= m2) u = __uint_as_float(__float_as_uint(u) - 1u);   // largest float < m2
    return u;
}

// all-lanes minimum (fminf: a NaN operand is ignored)
__device__ __forceinline__ float wave_all_min(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// Candidate centers of one wave (n <= 64): bit i set = center i can be the nearest center of
// some active pixel of this wave.  The wave's locations lie in a bounding box; lane i bounds
// the squared distance from center i to any point of the box from below (lo) and above (hi).
// With U = min_i hi_i every pixel has a center within sqrt(U), so a center with lo > U is
// farther from EVERY pixel than that pixel's nearest center — by the relative margin 1e-5,
// far outside the sqrt-rounding tie band (2^-22) that the exact search resolves.  A non-finite
// location switches the culling off for the wave (NaN / inf distances select index 0).
// center tables: LdsCenters = float2 array in LDS (any n); LaneCenters = lane i of the wave holds
// center i in registers (n <= 64): a wave-uniform index is a v_readlane, no memory access
struct LdsCenters {
    const float2* cen;
    __device__ __forceinline__ float2 at(int i) const { return cen[i]; }
    __device__ __forceinline__ float2 of_lane(int n) const { return cen[min(lane_id(), n - 1)]; }
};
struct LaneCenters {
    float cy, cx;
    __device__ __forceinline__ float2 at(int i) const
    {
        return make_float2(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(cy), i)),
                           __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cx), i)));
    }
    __device__ __forceinline__ float2 of_lane(int) const { return make_float2(cy, cx); }
};

template <typename Centers>
__device__ __forceinline__ uint64_t wave_candidate_centers(const Centers& cen, int n,
                                                           const float ly[4], const float lx[4],
                                                           const bool act[4])
{
    float lo_y = INFINITY, lo_x = INFINITY, nhi_y = INFINITY, nhi_x = INFINITY;   // nhi = -max
    bool finite = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (act[j]) {
            finite = finite && (fabsf(ly[j]) < INFINITY) && (fabsf(lx[j]) < INFINITY);
            lo_y = fminf(lo_y, ly[j]);  nhi_y = fminf(nhi_y, -ly[j]);
            lo_x = fminf(lo_x, lx[j]);  nhi_x = fminf(nhi_x, -lx[j]);
        }
    }
    const uint64_t all = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    if (__any(!finite)) return all;
    lo_y = wave_all_min(lo_y);  nhi_y = wave_all_min(nhi_y);
    lo_x = wave_all_min(lo_x);  nhi_x = wave_all_min(nhi_x);
    if (!(lo_y < INFINITY)) return 0ull;                 // no active pixel in this wave
    const float hi_y = -nhi_y, hi_x = -nhi_x;
    const int lane = lane_id();
    const float2 c = cen.of_lane(n);
    const float a0 = c.x - lo_y, a1 = hi_y - c.x;         // >= 0 inside the box
    const float b0 = c.y - lo_x, b1 = hi_x - c.y;
    const float near_y = fmaxf(0.f, fmaxf(-a0, -a1)), near_x = fmaxf(0.f, fmaxf(-b0, -b1));
    const float far_y = fmaxf(fabsf(a0), fabsf(a1)), far_x = fmaxf(fabsf(b0), fabsf(b1));
    const float lo = near_y * near_y + near_x * near_x;
    const float hi = (lane < n) ? far_y * far_y + far_x * far_x : INFINITY;
    const float U = wave_all_min(hi);
    return __ballot(lane < n && lo <= U * 1.00001f) & all;
}

template <bool CULL = true, typename Centers = LdsCenters>
__device__ __forceinline__ void group4(const Centers& cen, int n,
                                       const float ly[4], const float lx[4],
                                       const bool act[4], int use_thr, float thr,
                                       uint32_t id[4])
{
    // pass 1: first index of the smallest squared distance, and `prev` = the smallest
    // squared distance among the centers BEFORE that index (= the running minimum at the
    // moment of the last update)
    float smin[4], prev[4];
    int imin[4];
    // ascending index order over the wave's candidates keeps the first-index rule
    const bool cull = CULL && n <= 64;
    uint64_t m = cull ? wave_candidate_centers(cen, n, ly, lx, act) : 0ull;
    int first = 0;
    if (cull) {
        if (m == 0ull) {                                  // wave without an active pixel
#pragma unroll
            for (int j = 0; j < 4; ++j) id[j] = 0u;
            return;
        }
        first = __ffsll((unsigned long long)m) - 1;
        m &= m - 1;
    }
    {
        const float2 c0 = cen.at(first);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            smin[j] = sqdist(c0.x, c0.y, ly[j], lx[j]);
            prev[j] = INFINITY;
            imin[j] = first;
        }
    }
    auto visit = [&](const int i) {
        const float2 c = cen.at(i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = sqdist(c.x, c.y, ly[j], lx[j]);
            const bool upd = s < smin[j];
            prev[j] = upd ? smin[j] : prev[j];
            imin[j] = upd ? i : imin[j];
            smin[j] = upd ? s : smin[j];
        }
    };
    if (cull) {
        while (m) {
            const int i = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            visit(i);
        }
    } else {
        for (int i = 1; i < n; ++i) visit(i);
    }
    // The reference compares d = sqrt_rn(s) and takes the LOWEST index among equal d.
    // imin is already that index unless an earlier center has s in (smin, U], U = largest
    // float with the same rounded square root.  U <= smin (1 + 2^-22); test a superset
    // with one fma and only then pay for the exact tie interval and the second pass.
    bool maybe_tie = false;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        maybe_tie = maybe_tie || (act[j] && !(prev[j] > fmaf(smin[j], 4.8e-7f, smin[j])));
    int best[4] = {imin[0], imin[1], imin[2], imin[3]};
    if (__any(maybe_tie)) {
        float U[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) U[j] = sqrt_tie_upper(sqrtf(smin[j]));
        for (int i = n - 1; i >= 0; --i) {
            const float2 c = cen.at(i);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (sqdist(c.x, c.y, ly[j], lx[j]) <= U[j]) best[j] = i;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) best[j] = (smin[j] != smin[j]) ? 0 : best[j];    // NaN loc -> index 0
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t v = (uint32_t)(best[j] + 1) & 0xFFu;               // uint8 wrap (instance.py:236)
        if (use_thr && sqrtf(smin[j]) > thr) v = 0;                 // instance.py:246-247
        id[j] = act[j] ? v : 0u;
    }
}

// =================================================================================
// fused: argmax + fg + grouping + class votes
// dynamic LDS: float2 centers[max_centers] (used with > 64 centers only) | i32 vote_key[FUSED_VOTE_SLOTS]
//              | u32 vote_cnt[FUSED_VOTE_SLOTS]
// `ablate`: diagnostics only (NMSA_FUSED_ABLATE; bits 1 no votes, 2 no search, 4 no offset loads)
// occupancy: capped at 5 waves per SIMD.  The kernel alone runs as fast with 5 as with 7 or 8
// (0.2766-0.277 vs 0.2776-0.2784 ms isolated, same box), and the free wave slots let the small
// kernels of the same step (NMS rows, metric chain) start beside it: whole step +0.6-0.9 % at
// K = 50 in three interleaved same-box passes; 4 waves: -7 %.
// =================================================================================
template <int DTYPE, bool VEC, bool WITH_SCORE, int UNROLL = 8, bool NT = true,
          bool EARLY_OFFSETS = false, bool TILED = false, int TILE_LOG2W = 6>
__global__ __launch_bounds__(FUSED_THREADS) __attribute__((amdgpu_waves_per_eu(5, WITH_SCORE ? 8 : 5))) void k_panoptic_fused(
    const void* __restrict__ logits, const float* __restrict__ offset,
    const int32_t* __restrict__ centers_yx, const int32_t* __restrict__ n_centers,
    const uint8_t* __restrict__ is_thing,
    int C, int H, int W, int max_centers, int iters,
    float scale_y, float scale_x, int use_thr, float thr,
    uint8_t* __restrict__ sem_u8, uint8_t* __restrict__ inst, uint8_t* __restrict__ fg_out,
    float* __restrict__ score, uint32_t* __restrict__ votes, int ablate)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float2* cen = (float2*)smem;
    // (instance, class) vote counters of this workgroup: a 1024-pixel chunk meets a handful of
    // (id, class) pairs, so a small LDS hash table (one slot per thread: cleared and flushed with
    // one LDS access each) replaces a dense [ids, classes] histogram
    int* vote_key = (int*)(cen + max_centers);
    uint32_t* vote_cnt = (uint32_t*)(vote_key + FUSED_VOTE_SLOTS);
    const int NC = C + 1;

    const int b = blockIdx.y;
    const int P = H * W;
    const int n = min(n_centers[b], max_centers);

    // No workgroup barrier between the class stream and the pixel decisions: with n <= 64
    // centers (top-k 64: the usual case) every wave keeps the whole center table in registers —
    // lane i holds center i, a wave-uniform index is a v_readlane — and the thing LUT as 64-bit
    // ballot masks.  Both are requested from HBM/L2 before the first class plane and consumed
    // behind the class loop.  Only the vote table is shared: cleared here, one barrier at the
    // START of the kernel (all waves arrive at once), one before the flush.  More than 64
    // centers (ties at the k-th value): LDS center table + a barrier behind the first chunk.
    static_assert(FUSED_THREADS == 256, "one LUT entry per thread");
    static_assert(FUSED_VOTE_SLOTS == FUSED_THREADS, "one vote slot per thread");
    vote_key[threadIdx.x] = -1;
    vote_cnt[threadIdx.x] = 0;
    const bool lane_centers = n <= 64 && H <= 65535 && W <= 65535;   // workgroup-uniform; (y, x) packed in 32 bits
    // two registers live across the class loop: the lane's center as (y << 16 | x) (image
    // sides < 2^16, else the LDS table is used) and its four thing-LUT bytes
    uint32_t lane_cyx, thing_b = 0u;
    {
        const int2 c2 = *(const int2*)(centers_yx +
            ((size_t)b * max_centers + min(lane_id(), max_centers - 1)) * 2);
        lane_cyx = ((uint32_t)c2.x << 16) | ((uint32_t)c2.y & 0xFFFFu);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k * 64 < C)
                thing_b |= (is_thing[min(k * 64 + lane_id(), C - 1)] != 0 ? 1u : 0u) << k;
    }
    __syncthreads();

    const size_t img_logits = (size_t)b * C * P;
    const float* offy = offset + (size_t)b * 2 * P;
    const float* offx = offy + P;
    uint32_t* votes_b = votes + (size_t)b * 256 * NC;

    const int chunk_start = blockIdx.x * iters * PX_PER_ITER;
    // one chunk of 1024 pixels; the first one (FIRST) also fills the LDS tables
    auto chunk = [&](const int it, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        int p0;
        bool active;                            // whole thread out of range in the tail chunk
        if (TILED) {
            // a workgroup covers a TW x (1024 / TW) pixel tile; TW = 64: 64 x 4 per wave (four
            // 128-B row pieces per wave load of 16-bit logits), TW = 128: 128 x 2 per wave (two
            // 256-B pieces: whole cache-line pairs)
            constexpr int TW = 1 << TILE_LOG2W, TH = 1024 >> TILE_LOG2W, LPR = TW / 4;
            const int tiles_x = (W + TW - 1) >> TILE_LOG2W;
            const int tile = blockIdx.x * iters + it;
            const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
            const int row = ty * TH + (int)threadIdx.x / LPR;
            const int col = tx * TW + ((int)threadIdx.x % LPR) * 4;
            active = row < H && col < W;        // W % 4 == 0: all 4 pixels or none
            p0 = active ? row * W + col : 0;
        } else {
            p0 = chunk_start + it * PX_PER_ITER + threadIdx.x * PX_PER_THREAD;
            active = p0 < P;
        }
        const int nvalid = active ? min(4, P - p0) : 0;

        // offsets of this thread's pixels.  Default: requested after the argmax and only by
        // threads that hold a foreground pixel — background regions are large, so whole sectors
        // of the offset planes are never fetched (f32 logits: 1.5 % faster than requesting them
        // up front, EARLY_OFFSETS; 16-bit logits: no difference).
        float4 oy = make_float4(0.f, 0.f, 0.f, 0.f), ox = oy;
        if (EARLY_OFFSETS && active) {
            oy = load_px4<NMSA_F32, VEC, NT>(offy, (size_t)p0, nvalid);
            ox = load_px4<NMSA_F32, VEC, NT>(offx, (size_t)p0, nvalid);
        }


        // ---- a1: argmax over classes (first index of the maximum) --------------------
        ArgmaxState st;
        argmax_init(st);
        if (active) {
            int c = 0;
            static_assert(UNROLL % 4 == 0, "class groups of 4");
            for (; c + UNROLL <= C; c += UNROLL) {
                float4 v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u)
                    v[u] = load_px4<DTYPE, VEC, NT>(logits, img_logits + (size_t)(c + u) * P + p0, nvalid);
#pragma unroll
                for (int u = 0; u < UNROLL; u += 4)
                    argmax_quad<WITH_SCORE>(st, v[u], v[u + 1], v[u + 2], v[u + 3], c + u);
            }
            for (; c < C; c += 4) {                       // tail: pad the last group with -inf
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    v[u] = (c + u < C)
                        ? load_px4<DTYPE, VEC, NT>(logits, img_logits + (size_t)(c + u) * P + p0, nvalid)
                        : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
                argmax_quad<WITH_SCORE>(st, v[0], v[1], v[2], v[3], c, min(4, C - c));
            }
        }
        if (FIRST && !lane_centers) {           // > 64 centers: LDS table (workgroup-uniform branch)
            for (int i = threadIdx.x; i < n; i += FUSED_THREADS) {
                const int2 c2 = *(const int2*)(centers_yx + ((size_t)b * max_centers + i) * 2);
                cen[i] = make_float2((float)c2.x, (float)c2.y);
            }
            __syncthreads();
        }
        // thing LUT as ballot masks: bit c of thing_m[c >> 6] (wave-uniform, SGPRs)
        uint64_t thing_m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            thing_m[k] = __ballot(((thing_b >> k) & 1u) != 0u && k * 64 + lane_id() < C);
        // threads beyond the image (tail chunk) stay in the wave: nvalid == 0 keeps them out of
        // every pixel decision, and the wave-level steps below need all lanes
        int cls[4];
        bool fg[4];
        bool any_fg = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // softmax of a column with a NaN / +inf / all -inf is all-NaN and torch.max then
            // returns index 0 (semantic.py:52-53).  A non-finite logit shows as NaN in `nf`
            // (argmax only) or in the softmax denominator (with score): exact re-check, rare.
            cls[j] = st.am[j];
            if (WITH_SCORE) {
                float sc = 1.0f / st.se[j];
                if ((st.se[j] != st.se[j] || may_tie_in_probability<DTYPE>(st.m[j])) && j < nvalid) {
                    const float2 ex = column_exact<DTYPE>(logits, img_logits + p0 + j, P, C);
                    sc = ex.x;
                    cls[j] = __float_as_int(ex.y);
                }
                st.se[j] = sc;
            } else if ((st.nf[j] != st.nf[j] || may_tie_in_probability<DTYPE>(st.m[j])) && j < nvalid) {
                cls[j] = column_class<DTYPE>(logits, img_logits + p0 + j, P, C, st.m[j]);
            }
            uint64_t tm = thing_m[0];
            if (C > 64) {                                  // workgroup-uniform
                const int q = cls[j] >> 6;
                tm = (q == 0) ? thing_m[0] : (q == 1) ? thing_m[1] : (q == 2) ? thing_m[2] : thing_m[3];
            }
            fg[j] = (j < nvalid) && ((tm >> (cls[j] & 63)) & 1ull);     // panoptic.py:123-127
            any_fg = any_fg || fg[j];
        }

        // ---- a3: offset grouping ------------------------------------------------------
        uint32_t id[4] = {0u, 0u, 0u, 0u};
        if (__any(any_fg) && n > 0) {           // wave-uniform: group4 culls centers per wave
            if (!EARLY_OFFSETS && any_fg && !(ablate & 4)) {
                oy = load_px4<NMSA_F32, VEC, NT>(offy, (size_t)p0, nvalid);
                ox = load_px4<NMSA_F32, VEC, NT>(offx, (size_t)p0, nvalid);
            }
            const float oyv[4] = {oy.x, oy.y, oy.z, oy.w};
            const float oxv[4] = {ox.x, ox.y, ox.z, ox.w};
            float ly[4], lx[4];
            int y = p0 / W, x = p0 - y * W;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // de-normalise (panoptic.py:108-109) then add the int grid (instance.py:194):
                // two separate roundings, never an FMA
                ly[j] = __fadd_rn((float)y, __fmul_rn(oyv[j], scale_y));
                lx[j] = __fadd_rn((float)x, __fmul_rn(oxv[j], scale_x));
                if (++x == W) { x = 0; ++y; }
            }
            if (ablate & 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) id[j] = fg[j] ? 1u : 0u;
            } else if (lane_centers) {
                group4(LaneCenters{(float)(lane_cyx >> 16), (float)(lane_cyx & 0xFFFFu)}, n, ly, lx, fg,
                       use_thr, thr, id);
            } else {
                group4(LdsCenters{cen}, n, ly, lx, fg, use_thr, thr, id);
            }
        }

        // ---- stores --------------------------------------------------------------------
        if (VEC && active) {
            typedef unsigned char u8x4_t __attribute__((ext_vector_type(4)));
            const u8x4_t s4 = {(uint8_t)cls[0], (uint8_t)cls[1], (uint8_t)cls[2], (uint8_t)cls[3]};
            const u8x4_t i4 = {(uint8_t)id[0], (uint8_t)id[1], (uint8_t)id[2], (uint8_t)id[3]};
            const u8x4_t f4 = {(uint8_t)fg[0], (uint8_t)fg[1], (uint8_t)fg[2], (uint8_t)fg[3]};
            if (ablate & 8) {                   // experiment: streaming stores for the u8 maps
                __builtin_nontemporal_store(s4, (u8x4_t*)(sem_u8 + (size_t)b * P + p0));
                __builtin_nontemporal_store(i4, (u8x4_t*)(inst + (size_t)b * P + p0));
                if (fg_out) __builtin_nontemporal_store(f4, (u8x4_t*)(fg_out + (size_t)b * P + p0));
            } else {
                *(u8x4_t*)(sem_u8 + (size_t)b * P + p0) = s4;
                *(u8x4_t*)(inst + (size_t)b * P + p0) = i4;
                if (fg_out) *(u8x4_t*)(fg_out + (size_t)b * P + p0) = f4;
            }
            if (WITH_SCORE)
                *(float4*)(score + (size_t)b * P + p0) =
                    make_float4(st.se[0], st.se[1], st.se[2], st.se[3]);
        } else {
            for (int j = 0; j < nvalid; ++j) {
                sem_u8[(size_t)b * P + p0 + j] = (uint8_t)cls[j];
                inst[(size_t)b * P + p0 + j] = (uint8_t)id[j];
                if (fg_out) fg_out[(size_t)b * P + p0 + j] = fg[j];
                if (WITH_SCORE) score[(size_t)b * P + p0 + j] = st.se[j];
            }
        }

        // ---- a5 (votes): votes[id][class+1] += 1 --------------------------------------------
        // Lanes whose 4 pixels agree (all but the few on a segment boundary) form RUNS of equal
        // keys across the wave: every run head adds 4 x its run length with one LDS atomic, all
        // heads in parallel (two ballots, no loop over the distinct keys).  Boundary lanes add
        // their pixels one by one.
        const bool wave_has_inst = __any((id[0] | id[1] | id[2] | id[3]) != 0u);
        if (wave_has_inst && !(ablate & 1)) {
            int key[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                key[j] = (id[j] != 0u) ? (int)(id[j] * NC + cls[j] + 1) : -1;
            auto add = [&](int k, uint32_t cnt) {
                const int slot = lds_hash_slot(vote_key, FUSED_VOTE_SLOTS, k);
                if (slot >= 0) atomicAdd(&vote_cnt[slot], cnt);
                else atomicAdd(&votes_b[k], cnt);            // table full around that slot
            };
            const bool uniform = key[0] == key[1] && key[1] == key[2] && key[2] == key[3];
            int run_len, run_last;
            if (wave_run_head(uniform ? key[0] : -1, run_len, run_last))
                add(key[0], 4u * (uint32_t)run_len);
            if (!uniform) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (key[j] >= 0) add(key[j], 1u);
            }
        }
    };
    chunk(0, std::true_type{});
    for (int it = 1; it < iters; ++it) chunk(it, std::false_type{});

    __syncthreads();
    {
        const int k = vote_key[threadIdx.x];
        if (k >= 0) atomicAdd(&votes_b[k], vote_cnt[threadIdx.x]);
    }
}

// =================================================================================
// standalone a1: argmax / score without grouping (SemanticPostprocessing only)
// =================================================================================
template <int DTYPE, bool VEC, bool WITH_SCORE>
__global__ __launch_bounds__(FUSED_THREADS) void k_semantic_argmax(
    const void* __restrict__ logits, int C, int P,
    uint8_t* __restrict__ idx_u8, int64_t* __restrict__ idx_i64, float* __restrict__ score)
{
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * FUSED_THREADS + threadIdx.x) * PX_PER_THREAD;
    if (p0 >= P) return;
    const int nvalid = min(4, P - p0);
    const size_t img = (size_t)b * C * P;
    ArgmaxState st;
    argmax_init(st);
    int c = 0;
    for (; c + 8 <= C; c += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = load_px4<DTYPE, VEC, true>(logits, img + (size_t)(c + u) * P + p0, nvalid);
        argmax_quad<WITH_SCORE>(st, v[0], v[1], v[2], v[3], c);
        argmax_quad<WITH_SCORE>(st, v[4], v[5], v[6], v[7], c + 4);
    }
    for (; c < C; c += 4) {                           // tail: pad the last group with -inf
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = (c + u < C) ? load_px4<DTYPE, VEC, true>(logits, img + (size_t)(c + u) * P + p0, nvalid)
                               : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        argmax_quad<WITH_SCORE>(st, v[0], v[1], v[2], v[3], c, min(4, C - c));
    }
    int cls[4];
    float sc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cls[j] = st.am[j];
        sc[j] = 0.f;
        if (WITH_SCORE) {
            sc[j] = 1.0f / st.se[j];
            if ((st.se[j] != st.se[j] || may_tie_in_probability<DTYPE>(st.m[j])) && j < nvalid) {
                const float2 ex = column_exact<DTYPE>(logits, img + p0 + j, P, C);
                sc[j] = ex.x;
                cls[j] = __float_as_int(ex.y);
            }
        } else if ((st.nf[j] != st.nf[j] || may_tie_in_probability<DTYPE>(st.m[j])) && j < nvalid) {
            cls[j] = column_class<DTYPE>(logits, img + p0 + j, P, C, st.m[j]);
        }
    }
    const size_t o = (size_t)b * P + p0;
    if (VEC) {                                  // one store instruction per output and thread
        if (idx_u8)
            *(uchar4*)(idx_u8 + o) = make_uchar4((uint8_t)cls[0], (uint8_t)cls[1], (uint8_t)cls[2],
                                                 (uint8_t)cls[3]);
        if (idx_i64) {
            *(longlong2*)(idx_i64 + o) = make_longlong2(cls[0], cls[1]);
            *(longlong2*)(idx_i64 + o + 2) = make_longlong2(cls[2], cls[3]);
        }
        if (WITH_SCORE) *(float4*)(score + o) = make_float4(sc[0], sc[1], sc[2], sc[3]);
    } else {
        for (int j = 0; j < nvalid; ++j) {
            if (idx_u8) idx_u8[o + j] = (uint8_t)cls[j];
            if (idx_i64) idx_i64[o + j] = cls[j];
            if (WITH_SCORE) score[o + j] = sc[j];
        }
    }
}

// softmax over the class axis, f32 out (semantic.py:52 'semantic_softmax_scores')
// ATen's max reduction propagates NaN; fmaxf drops it
__device__ __forceinline__ float max_nan(float m, float v)
{
    return (v != v) ? v : ((m != m) ? m : fmaxf(m, v));
}

// exact three-pass column (max, sum of exp, normalise): the arithmetic of F.softmax incl. its
// NaN results for columns with a NaN, a +inf or nothing but -inf
template <int DTYPE>
__device__ __noinline__ void softmax_column_3pass(const void* logits, size_t col0, int P, int C,
                                                  float* probs)
{
    auto ld = [&](int c) -> float {
        if (DTYPE == NMSA_F32) return ((const float*)logits)[col0 + (size_t)c * P];
        const uint16_t h = ((const uint16_t*)logits)[col0 + (size_t)c * P];
        return (DTYPE == NMSA_BF16) ? bf16_to_f32(h) : f16_to_f32(h);
    };
    float m = -INFINITY, sum = 0.f;
    for (int c = 0; c < C; ++c) m = max_nan(m, ld(c));
    for (int c = 0; c < C; ++c) sum += __expf(ld(c) - m);
    for (int c = 0; c < C; ++c) probs[col0 + (size_t)c * P] = __expf(ld(c) - m) / sum;
}

// generic: TWO passes over the logits (online max / sum of exp, then normalise) = 3 x 4C bytes
// per pixel instead of the 4 x 4C of max / sum / normalise.  Columns holding a non-finite logit
// (rare) are redone by the exact three-pass routine.
template <int DTYPE, bool VEC>
__global__ __launch_bounds__(FUSED_THREADS) void k_semantic_softmax(
    const void* __restrict__ logits, int C, int P, float* __restrict__ probs)
{
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * FUSED_THREADS + threadIdx.x) * PX_PER_THREAD;
    if (p0 >= P) return;
    const int nvalid = min(4, P - p0);
    const size_t img = (size_t)b * C * P;
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float sum[4] = {0.f, 0.f, 0.f, 0.f}, nf[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
        const float4 v4 = load_px4<DTYPE, VEC>(logits, img + (size_t)c * P + p0, nvalid);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            nf[j] = fmaf(v[j], 0.0f, nf[j]);                      // NaN once a logit is not finite
            const float e = __expf(-fabsf(v[j] - m[j]));
            sum[j] = (v[j] > m[j]) ? fmaf(sum[j], e, 1.0f) : (sum[j] + e);
            m[j] = fmaxf(m[j], v[j]);
        }
    }
    float inv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) inv[j] = 1.0f / sum[j];
    for (int c = 0; c < C; ++c) {
        const float4 v4 = load_px4<DTYPE, VEC>(logits, img + (size_t)c * P + p0, nvalid);
        const float r[4] = {__expf(v4.x - m[0]) * inv[0], __expf(v4.y - m[1]) * inv[1],
                            __expf(v4.z - m[2]) * inv[2], __expf(v4.w - m[3]) * inv[3]};
        float* o = probs + img + (size_t)c * P + p0;
        if (VEC) {
            f32x4_t q; q.x = r[0]; q.y = r[1]; q.z = r[2]; q.w = r[3];
            __builtin_nontemporal_store(q, (f32x4_t*)o);
        } else {
            for (int j = 0; j < nvalid; ++j) o[j] = r[j];
        }
    }
    for (int j = 0; j < nvalid; ++j)
        if (nf[j] != nf[j]) softmax_column_3pass<DTYPE>(logits, img + p0 + j, P, C, probs);
}

// C <= CMAX: the whole column of 4 pixels stays in registers — ONE read and one write of the
// tensor (2 x 4C bytes per pixel).  The arithmetic is the plain max / sum / normalise, so the
// non-finite cases need no special handling.
template <int DTYPE, int CMAX>
__global__ __launch_bounds__(FUSED_THREADS) void k_semantic_softmax_reg(
    const void* __restrict__ logits, int C, int P, float* __restrict__ probs)
{
    const int b = blockIdx.y;
    const int p0 = (blockIdx.x * FUSED_THREADS + threadIdx.x) * PX_PER_THREAD;
    if (p0 >= P) return;
    const size_t img = (size_t)b * C * P;
    float4 v[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) v[c] = load_px4<DTYPE, true, true>(logits, img + (size_t)c * P + p0, 4);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) {
            m[0] = max_nan(m[0], v[c].x); m[1] = max_nan(m[1], v[c].y);
            m[2] = max_nan(m[2], v[c].z); m[3] = max_nan(m[3], v[c].w);
        }
    float sum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) {
            v[c].x = __expf(v[c].x - m[0]); v[c].y = __expf(v[c].y - m[1]);
            v[c].z = __expf(v[c].z - m[2]); v[c].w = __expf(v[c].w - m[3]);
            sum[0] += v[c].x; sum[1] += v[c].y; sum[2] += v[c].z; sum[3] += v[c].w;
        }
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
        if (c < C) {
            f32x4_t q;
            q.x = v[c].x / sum[0]; q.y = v[c].y / sum[1]; q.z = v[c].z / sum[2]; q.w = v[c].w / sum[3];
            __builtin_nontemporal_store(q, (f32x4_t*)(probs + img + (size_t)c * P + p0));
        }
}

// =================================================================================
// standalone a3: grouping with a given foreground mask (+ per-id area)
// dynamic LDS: float2 centers[max_centers] | u32 area_hist[256]
// =================================================================================
// TILED (needs VEC): a workgroup covers a 64 x 16 pixel tile instead of 1024 consecutive pixels,
// a wave 64 x 4 — compact in the image, so the wave's locations cluster around 1-3 centers and
// the candidate culling of group4 leaves just those (a 256-pixel row segment spans ~10 of 24).
template <bool VEC, bool TILED = false>
__global__ __launch_bounds__(FUSED_THREADS) void k_group_offsets(
    const float* __restrict__ offset, const uint8_t* __restrict__ fgmask,
    const int32_t* __restrict__ centers_yx, const int32_t* __restrict__ n_centers,
    int H, int W, int max_centers, int iters,
    float scale_y, float scale_x, int use_thr, float thr,
    uint8_t* __restrict__ inst, int32_t* __restrict__ area)
{
    extern __shared__ __align__(16) unsigned char smem[];
    float2* cen = (float2*)smem;
    uint32_t* ahist = (uint32_t*)(cen + max_centers);
    const int b = blockIdx.y;
    const int P = H * W;
    const int n = min(n_centers[b], max_centers);
    for (int i = threadIdx.x; i < n; i += FUSED_THREADS) {
        const int32_t cy = centers_yx[((size_t)b * max_centers + i) * 2 + 0];
        const int32_t cx = centers_yx[((size_t)b * max_centers + i) * 2 + 1];
        cen[i] = make_float2((float)cy, (float)cx);
    }
    for (int i = threadIdx.x; i < 256; i += FUSED_THREADS) ahist[i] = 0;
    __syncthreads();

    const float* offy = offset + (size_t)b * 2 * P;
    const float* offx = offy + P;
    const int chunk_start = blockIdx.x * iters * PX_PER_ITER;
    const int tiles_x = (W + 63) >> 6;
    for (int it = 0; it < iters; ++it) {
        int p0;
        bool active;                            // tail threads stay for the wave-level steps
        if (TILED) {
            const int tile = blockIdx.x * iters + it;
            const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
            const int row = ty * 16 + (int)(threadIdx.x >> 4);        // 16 lanes per tile row
            const int col = tx * 64 + (int)(threadIdx.x & 15) * 4;
            active = row < H && col < W;                               // W % 4 == 0: 4 pixels or none
            p0 = active ? row * W + col : 0;
        } else {
            p0 = chunk_start + it * PX_PER_ITER + threadIdx.x * PX_PER_THREAD;
            active = p0 < P;
        }
        const int nvalid = active ? min(4, P - p0) : 0;
        bool fg[4] = {false, false, false, false};
        if (VEC && active) {
            const uchar4 f = *(const uchar4*)(fgmask + (size_t)b * P + p0);
            fg[0] = f.x != 0; fg[1] = f.y != 0; fg[2] = f.z != 0; fg[3] = f.w != 0;
        } else {
            for (int j = 0; j < nvalid; ++j) fg[j] = fgmask[(size_t)b * P + p0 + j] != 0;
        }
        const bool any_fg = fg[0] || fg[1] || fg[2] || fg[3];
        uint32_t id[4] = {0u, 0u, 0u, 0u};
        if (__any(any_fg) && n > 0) {           // wave-uniform: group4 culls centers per wave
            float4 oy = make_float4(0.f, 0.f, 0.f, 0.f), ox = oy;
            if (any_fg) {
                oy = load_px4<NMSA_F32, VEC, true>(offy, (size_t)p0, nvalid);
                ox = load_px4<NMSA_F32, VEC, true>(offx, (size_t)p0, nvalid);
            }
            const float oyv[4] = {oy.x, oy.y, oy.z, oy.w};
            const float oxv[4] = {ox.x, ox.y, ox.z, ox.w};
            float ly[4], lx[4];
            int y = p0 / W, x = p0 - y * W;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ly[j] = __fadd_rn((float)y, __fmul_rn(oyv[j], scale_y));
                lx[j] = __fadd_rn((float)x, __fmul_rn(oxv[j], scale_x));
                if (++x == W) { x = 0; ++y; }
            }
            group4(LdsCenters{cen}, n, ly, lx, fg, use_thr, thr, id);
        }
        if (VEC && active) {
            *(uchar4*)(inst + (size_t)b * P + p0) =
                make_uchar4((uint8_t)id[0], (uint8_t)id[1], (uint8_t)id[2], (uint8_t)id[3]);
        } else {
            for (int j = 0; j < nvalid; ++j) inst[(size_t)b * P + p0 + j] = (uint8_t)id[j];
        }
        if (area && n > 0 && __any(any_fg)) {
            // bincount over the foreground pixels, id 0 included (instance.py:253): runs of lanes
            // whose 4 pixels agree add 4 x run length at the run head, boundary lanes per pixel
            int key[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) key[j] = fg[j] ? (int)id[j] : -1;
            const bool uniform = key[0] == key[1] && key[1] == key[2] && key[2] == key[3];
            int run_len, run_last;
            if (wave_run_head(uniform ? key[0] : -1, run_len, run_last))
                atomicAdd(&ahist[key[0]], 4u * (uint32_t)run_len);
            if (!uniform) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (key[j] >= 0) atomicAdd(&ahist[key[j]], 1u);
            }
        }
    }
    if (area) {
        __syncthreads();
        for (int i = threadIdx.x; i < 256; i += FUSED_THREADS) {
            const uint32_t v = ahist[i];
            if (v) atomicAdd(&area[(size_t)b * 256 + i], (int32_t)v);
        }
    }
}

// =================================================================================
// a5: per-instance class (mode, smallest on ties) + running per-class counter
// =================================================================================
// The image's vote table [256 x NC] is staged through LDS in blocks of whole rows with
// coalesced loads (thread-per-row global reads are 64 different cache lines per wave load:
// 46 us at NC = 151, 10 us at NC = 41); non-zero words are zeroed on the way (the table is left
// clean for the next call); thread r then scans row r in LDS (row stride NC words).
constexpr int ASSIGN_LDS_WORDS = 36 * 1024;       // 144 KB of dynamic LDS
// 256 threads: a 1024-thread workgroup does not fit next to the other batch's fused kernel (7 of
// 8 wave slots per SIMD taken) and waits for it to drain — the step lost 9 % with it
constexpr int ASSIGN_THREADS = 256;
__global__ __launch_bounds__(ASSIGN_THREADS) void k_assign(
    uint32_t* __restrict__ votes, int NC, int clear_votes, int64_t max_inst, int64_t void_label,
    int64_t* __restrict__ pan_of_inst, int32_t* __restrict__ area,
    int64_t* __restrict__ ids_pan, int64_t* __restrict__ ids_ins, int32_t* __restrict__ n_ids,
    int lds_words)
{
    extern __shared__ uint32_t s_rows[];  // rows_per_pass x NC
    __shared__ int s_vcls[256];          // classes of the valid instances, ascending id
    __shared__ int s_wcnt[4];
    __shared__ uint32_t s_total[256];
    __shared__ int s_cls[256];
    const int b = blockIdx.x, t = threadIdx.x;
    uint32_t* tab = votes + (size_t)b * 256 * NC;
    const int rows_per_pass = max(1, min(256, lds_words / NC));
    for (int r0 = 0; r0 < 256; r0 += rows_per_pass) {
        const int nr = min(rows_per_pass, 256 - r0);
        const int nwords = nr * NC;
        uint32_t* src = tab + (size_t)r0 * NC;
        // 8 independent 16-byte loads in flight per thread (a load -> LDS store chain per word
        // would pay one memory latency per iteration; with 4-byte loads the 42 KB of a 41-class
        // table were six round trips for the 256 threads, now two): rows start 16-byte aligned
        // (256 NC words per image; checked here, an unaligned pass reads word by word)
        const int nquads = ((((uintptr_t)src) & 15) == 0) ? (nwords >> 2) : 0;
        for (int i0 = t; i0 < nquads; i0 += 8 * ASSIGN_THREADS) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * ASSIGN_THREADS;
                v[u] = (i < nquads) ? ((const uint4*)src)[i] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * ASSIGN_THREADS;
                if (i < nquads) {
                    s_rows[4 * i] = v[u].x; s_rows[4 * i + 1] = v[u].y;
                    s_rows[4 * i + 2] = v[u].z; s_rows[4 * i + 3] = v[u].w;
                    if (clear_votes && (v[u].x | v[u].y | v[u].z | v[u].w)) ((uint4*)src)[i] = make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
        for (int i = 4 * nquads + t; i < nwords; i += ASSIGN_THREADS) {        // (ragged end / unaligned pass)
            const uint32_t v = src[i];
            s_rows[i] = v;
            if (clear_votes && v) src[i] = 0;
        }
        __syncthreads();
        // four lanes per row: each scans a contiguous quarter of the classes, then the quarters
        // are combined (larger count wins, equal counts -> the smaller class = torch.mode)
        for (int rb = 0; rb < nr; rb += ASSIGN_THREADS / 4) {          // uniform trip count
            const int r = rb + (t >> 2), part = t & 3;
            const int q = (NC + 3) >> 2;
            const int c0 = part * q, c1 = min(c0 + q, NC);
            uint32_t tot = 0;
            int64_t bestc = -1;
            int c_best = 0;
            if (r < nr) {
                const uint32_t* row = s_rows + r * NC;
                for (int c = c0; c < c1; ++c) {
                    const uint32_t v = row[c];
                    tot += v;
                    if ((int64_t)v > bestc) { bestc = v; c_best = c; }
                }
            }
#pragma unroll
            for (int o = 1; o <= 2; o <<= 1) {
                const uint32_t tot_o = __shfl_xor(tot, o);
                const int64_t best_o = __shfl_xor(bestc, o);
                const int c_o = __shfl_xor(c_best, o);
                tot += tot_o;
                if (best_o > bestc || (best_o == bestc && c_o < c_best)) { bestc = best_o; c_best = c_o; }
            }
            if (part == 0 && r < nr) { s_total[r0 + r] = tot; s_cls[r0 + r] = c_best; }
        }
        __syncthreads();
    }
    // the ordered compaction below is one thread per row (threads 256.. only keep the barriers)
    const bool rowt = t < 256;
    const uint32_t total = rowt ? s_total[t] : 0u;
    const int cls = rowt ? s_cls[t] : 0;
    // skip id 0, empty masks (panoptic_merge.py:195-200) and void majority (:203-204)
    const bool valid = rowt && (t > 0) && (total > 0) && (cls != 0);
    // order-preserving compaction of the valid instances (ballot + popcount)
    const unsigned long long m = __ballot(valid);
    const int w = t >> 6, l = t & 63;
    if (l == 0 && w < 4) s_wcnt[w] = __popcll(m);
    __syncthreads();
    int pos = __popcll(m & ((1ull << l) - 1ull));
    for (int k = 0; k < min(w, 4); ++k) pos += s_wcnt[k];
    if (valid) s_vcls[pos] = cls;
    __syncthreads();
    // running per-class counter in ascending instance-id order (:206-207)
    int rank = 1;
    if (valid)
        for (int j = 0; j < pos; ++j) rank += (s_vcls[j] == cls);
    const int64_t pid = (int64_t)cls * max_inst + rank;        // :208
    if (rowt) pan_of_inst[(size_t)b * 256 + t] = valid ? pid : void_label;
    if (area && rowt) area[(size_t)b * 256 + t] = (t > 0) ? (int32_t)total : 0;
    if (valid) {
        ids_pan[(size_t)b * 256 + pos] = pid;                   // dict insertion order (:209)
        ids_ins[(size_t)b * 256 + pos] = t;
    }
    if (t == 0) n_ids[b] = s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
}

// =================================================================================
// a5: paint.  8 px / thread: 8-B loads of sem & inst, 4 x 16-B stores of i64.
// =================================================================================
template <bool VEC>
__global__ __launch_bounds__(256) void k_paint(
    const uint8_t* __restrict__ sem_u8, const uint8_t* __restrict__ inst,
    const int64_t* __restrict__ pan_of_inst, const uint8_t* __restrict__ is_thing,
    int C, int P, int iters, int64_t max_inst, int64_t void_label,
    int64_t* __restrict__ pan, int64_t* __restrict__ pan_sem)
{
    __shared__ int64_t s_inst[256];
    __shared__ int64_t s_stuff[256];
    const int b = blockIdx.y, t = threadIdx.x;
    s_inst[t] = pan_of_inst[(size_t)b * 256 + t];
    // stuff paste (panoptic_merge.py:213-223): class value = idx + 1, thing classes stay void
    s_stuff[t] = (t < C && !is_thing[t]) ? (int64_t)(t + 1) * max_inst : void_label;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        const int p0 = ((blockIdx.x * iters + it) * 256 + t) * 8;
        if (p0 >= P) break;
        const size_t o = (size_t)b * P + p0;
        uint8_t s[8], in[8];
        const int nvalid = min(8, P - p0);
        if (VEC) {
            const uint2 sv = *(const uint2*)(sem_u8 + o);
            const uint2 iv = *(const uint2*)(inst + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] = (sv.x >> (8 * j)) & 0xFF; s[4 + j] = (sv.y >> (8 * j)) & 0xFF;
                in[j] = (iv.x >> (8 * j)) & 0xFF; in[4 + j] = (iv.y >> (8 * j)) & 0xFF;
            }
        } else {
            for (int j = 0; j < 8; ++j) {
                s[j] = (j < nvalid) ? sem_u8[o + j] : 0;
                in[j] = (j < nvalid) ? inst[o + j] : 0;
            }
        }
        int64_t r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = in[j] ? s_inst[in[j]] : s_stuff[s[j]];
        if (VEC) {
#pragma unroll
            for (int j = 0; j < 8; j += 2)
                *(longlong2*)(pan + o + j) = make_longlong2(r[j], r[j + 1]);
            if (pan_sem) {
#pragma unroll
                for (int j = 0; j < 8; j += 2)
                    *(longlong2*)(pan_sem + o + j) = make_longlong2(r[j] / max_inst, r[j + 1] / max_inst);
            }
        } else {
            for (int j = 0; j < nvalid; ++j) {
                pan[o + j] = r[j];
                if (pan_sem) pan_sem[o + j] = r[j] / max_inst;
            }
        }
    }
}

// variant: 2 px per lane and step -> every store instruction of a wave writes one
// contiguous 1-KiB run (whole 128-B lines), loads are 2-B per lane
__global__ __launch_bounds__(256) void k_paint2(
    const uint8_t* __restrict__ sem_u8, const uint8_t* __restrict__ inst,
    const int64_t* __restrict__ pan_of_inst, const uint8_t* __restrict__ is_thing,
    int C, int P, int steps, int64_t max_inst, int64_t void_label,
    int64_t* __restrict__ pan, int64_t* __restrict__ pan_sem)
{
    __shared__ int64_t s_inst[256];
    __shared__ int64_t s_stuff[256];
    const int b = blockIdx.y, t = threadIdx.x;
    s_inst[t] = pan_of_inst[(size_t)b * 256 + t];
    s_stuff[t] = (t < C && !is_thing[t]) ? (int64_t)(t + 1) * max_inst : void_label;
    __syncthreads();
    const int base = blockIdx.x * steps * 512;
    constexpr int U = 4;
    for (int k0 = 0; k0 < steps; k0 += U) {
        uint16_t sv[U], iv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p0 = base + ((k0 + u) * 256 + t) * 2;
            const bool ok = (k0 + u) < steps && p0 < P;
            sv[u] = ok ? *(const uint16_t*)(sem_u8 + (size_t)b * P + p0) : 0;
            iv[u] = ok ? *(const uint16_t*)(inst + (size_t)b * P + p0) : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p0 = base + ((k0 + u) * 256 + t) * 2;
            if ((k0 + u) >= steps || p0 >= P) continue;
            const int i0 = iv[u] & 0xFF, i1 = iv[u] >> 8;
            const int64_t r0 = i0 ? s_inst[i0] : s_stuff[sv[u] & 0xFF];
            const int64_t r1 = i1 ? s_inst[i1] : s_stuff[sv[u] >> 8];
            *(longlong2*)(pan + (size_t)b * P + p0) = make_longlong2(r0, r1);
            if (pan_sem)
                *(longlong2*)(pan_sem + (size_t)b * P + p0) = make_longlong2(r0 / max_inst, r1 / max_inst);
        }
    }
}

// =================================================================================
// generic a5 (deeplab_merge_batch with arbitrary integer dtypes; GT path)
// =================================================================================
__device__ __forceinline__ int64_t load_int(const void* p, int dtype, size_t i)
{
    switch (dtype) {
        case NMSA_U8: return ((const uint8_t*)p)[i];
        case NMSA_I16: return ((const int16_t*)p)[i];
        case NMSA_I32: return ((const int32_t*)p)[i];
        default: return ((const int64_t*)p)[i];
    }
}

// 4 consecutive integer labels; VEC = 4-element aligned rows (one wide load per tensor)
template <bool VEC>
__device__ __forceinline__ void load_int4(const void* p, int dtype, size_t o, int nvalid, int64_t out[4])
{
    if (VEC) {
        switch (dtype) {
            case NMSA_U8: {
                const uchar4 v = *(const uchar4*)((const uint8_t*)p + o);
                out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
                return;
            }
            case NMSA_I16: {
                const short4 v = *(const short4*)((const int16_t*)p + o);
                out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
                return;
            }
            case NMSA_I32: {
                const int4 v = *(const int4*)((const int32_t*)p + o);
                out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
                return;
            }
            default: {
                const longlong2 a = *(const longlong2*)((const int64_t*)p + o);
                const longlong2 c = *(const longlong2*)((const int64_t*)p + o + 2);
                out[0] = a.x; out[1] = a.y; out[2] = c.x; out[3] = c.y;
                return;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = (j < nvalid) ? load_int(p, dtype, o + j) : 0;
}

constexpr int MG_PX_PER_BLOCK = 256 * 4;

template <bool VEC>
__global__ __launch_bounds__(256) void k_merge_votes(
    const void* __restrict__ sem, int sem_dtype, const void* __restrict__ ins, int ins_dtype,
    const uint8_t* __restrict__ thing_seg, int NC, int P, uint32_t* __restrict__ votes)
{
    // per-workgroup (instance, class) counters in an LDS hash table, one global atomic per used
    // slot at the end (see lds_hash_slot)
    __shared__ int s_key[256];
    __shared__ uint32_t s_cnt[256];
    const int b = blockIdx.y;
    uint32_t* votes_b = votes + (size_t)b * 256 * NC;
    s_key[threadIdx.x] = -1;
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    auto add = [&](int kk, uint32_t cnt) {
        const int slot = lds_hash_slot(s_key, 256, kk);
        if (slot >= 0) atomicAdd(&s_cnt[slot], cnt);
        else atomicAdd(&votes_b[kk], cnt);
    };
    // block-uniform trip count so that every lane reaches the wave-level aggregation
    for (int p0 = (blockIdx.x * 256 + threadIdx.x) * 4; p0 - (int)threadIdx.x * 4 < P;
         p0 += gridDim.x * MG_PX_PER_BLOCK) {
        int key[4] = {-1, -1, -1, -1};
        if (p0 < P) {
            const int nvalid = min(4, P - p0);
            const size_t o = (size_t)b * P + p0;
            int64_t i4[4], s4[4], t4[4];
            load_int4<VEC>(ins, ins_dtype, o, nvalid, i4);
            load_int4<VEC>(sem, sem_dtype, o, nvalid, s4);
            load_int4<VEC>(thing_seg, NMSA_U8, o, nvalid, t4);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                // is_thing = (ins > 0) & thing_seg  (panoptic_merge.py:182,198)
                if (j < nvalid && i4[j] > 0 && i4[j] < 256 && t4[j] && s4[j] >= 0 && s4[j] < NC)
                    key[j] = (int)(i4[j] * NC + s4[j]);
        }
        // lanes whose 4 pixels agree form runs across the wave: the run head adds 4 x run length
        // (two ballots, no loop over the distinct keys); boundary lanes add their own pixels
        const bool same4 = key[0] == key[1] && key[1] == key[2] && key[2] == key[3];
        int run_len, run_last;
        if (wave_run_head(same4 ? key[0] : -1, run_len, run_last)) add(key[0], 4u * (uint32_t)run_len);
        if (!same4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (key[j] >= 0) add(key[j], 1u);
        }
    }
    __syncthreads();
    if (s_key[threadIdx.x] >= 0 && s_cnt[threadIdx.x])
        atomicAdd(&votes_b[s_key[threadIdx.x]], s_cnt[threadIdx.x]);
}

template <bool VEC>
__global__ __launch_bounds__(256) void k_merge_paint(
    const void* __restrict__ sem, int sem_dtype, const void* __restrict__ ins, int ins_dtype,
    const uint8_t* __restrict__ thing_seg, const uint8_t* __restrict__ is_thing_class,
    const int64_t* __restrict__ pan_of_inst, int NC, int P,
    int64_t max_inst, int64_t void_label, int64_t* __restrict__ pan)
{
    __shared__ int64_t s_inst[256];
    const int b = blockIdx.y;
    s_inst[threadIdx.x] = pan_of_inst[(size_t)b * 256 + threadIdx.x];
    __syncthreads();
    for (int p0 = (blockIdx.x * 256 + threadIdx.x) * 4; p0 < P; p0 += gridDim.x * MG_PX_PER_BLOCK) {
        const int nvalid = min(4, P - p0);
        const size_t o = (size_t)b * P + p0;
        int64_t i4[4], s4[4], t4[4], r[4];
        load_int4<VEC>(ins, ins_dtype, o, nvalid, i4);
        load_int4<VEC>(sem, sem_dtype, o, nvalid, s4);
        load_int4<VEC>(thing_seg, NMSA_U8, o, nvalid, t4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            r[j] = void_label;
            if (i4[j] != 0) {
                if (i4[j] > 0 && i4[j] < 256 && t4[j]) r[j] = s_inst[i4[j]];
            } else if (s4[j] > 0 && s4[j] < NC && !is_thing_class[s4[j]]) {
                r[j] = s4[j] * max_inst;                       // panoptic_merge.py:222-223
            }
        }
        if (VEC) {
            *(longlong2*)(pan + o) = make_longlong2(r[0], r[1]);
            *(longlong2*)(pan + o + 2) = make_longlong2(r[2], r[3]);
        } else {
            for (int j = 0; j < nvalid; ++j) pan[o + j] = r[j];
        }
    }
}

// =================================================================================
// next-1: per-instance biternion sums (instance.py:300-313)
// =================================================================================
template <bool VEC>
__global__ __launch_bounds__(256) void k_orientation_sums(
    const float* __restrict__ orientation, const uint8_t* __restrict__ inst,
    const uint8_t* __restrict__ mask, int P, double* __restrict__ sums, int32_t* __restrict__ count)
{
    __shared__ double s_sum[256 * 2];
    __shared__ int s_cnt[256];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 512; i += 256) s_sum[i] = 0.0;
    s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const float* o0 = orientation + (size_t)b * 2 * P;
    const float* o1 = o0 + P;
    // 4 consecutive pixels per lane.  A lane whose pixels belong to one instance adds its fp64
    // partial sums with one LDS atomic per component; boundary lanes add their pixels one by one.
    for (int p0 = (blockIdx.x * 256 + threadIdx.x) * 4; p0 - (int)threadIdx.x * 4 < P;
         p0 += gridDim.x * MG_PX_PER_BLOCK) {
        int id[4] = {0, 0, 0, 0};
        float a[4] = {0.f, 0.f, 0.f, 0.f}, c[4] = {0.f, 0.f, 0.f, 0.f};
        if (p0 < P) {
            const int nvalid = min(4, P - p0);
            const size_t o = (size_t)b * P + p0;
            int64_t i4[4], m4[4] = {1, 1, 1, 1};
            load_int4<VEC>(inst, NMSA_U8, o, nvalid, i4);
            if (mask) load_int4<VEC>(mask, NMSA_U8, o, nvalid, m4);
            if (VEC) {
                const float4 x = *(const float4*)(o0 + p0), y = *(const float4*)(o1 + p0);
                a[0] = x.x; a[1] = x.y; a[2] = x.z; a[3] = x.w;
                c[0] = y.x; c[1] = y.y; c[2] = y.z; c[3] = y.w;
            } else {
                for (int j = 0; j < nvalid; ++j) { a[j] = o0[p0 + j]; c[j] = o1[p0 + j]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) id[j] = (j < nvalid && m4[j]) ? (int)i4[j] : 0;
        }
        const bool same4 = id[0] == id[1] && id[1] == id[2] && id[2] == id[3];
        const int lid = same4 ? id[0] : 0;
        const double l0 = (double)a[0] + (double)a[1] + (double)a[2] + (double)a[3];
        const double l1 = (double)c[0] + (double)c[1] + (double)c[2] + (double)c[3];
        if (lid) {      // LDS fp64 atomics: same-address lanes serialise in the LDS unit (~1 / clk),
                        // cheaper than fp64 wave reductions through ds_bpermute per distinct id
            atomicAdd(&s_sum[lid * 2 + 0], l0);
            atomicAdd(&s_sum[lid * 2 + 1], l1);
        }
        // counts: runs of equal ids add 4 x run length at the run head
        int run_len, run_last;
        if (wave_run_head(lid ? lid : -1, run_len, run_last)) atomicAdd(&s_cnt[lid], 4 * run_len);
        if (!same4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!id[j]) continue;
                atomicAdd(&s_sum[id[j] * 2 + 0], (double)a[j]);
                atomicAdd(&s_sum[id[j] * 2 + 1], (double)c[j]);
                atomicAdd(&s_cnt[id[j]], 1);
            }
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (s_cnt[t]) {
        atomicAdd(&sums[((size_t)b * 256 + t) * 2 + 0], s_sum[t * 2 + 0]);
        atomicAdd(&sums[((size_t)b * 256 + t) * 2 + 1], s_sum[t * 2 + 1]);
        atomicAdd(&count[(size_t)b * 256 + t], s_cnt[t]);
    }
}

}  // namespace nmsa

using namespace nmsa;

// ---------------------------------------------------------------------------------
namespace {

int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// dynamic LDS of k_assign: whole rows of the vote table, at most ASSIGN_LDS_WORDS words; the
// grant above 64 KB is asked for per device — when it is denied the kernel stages fewer rows per
// pass inside the default 64 KB
constexpr int ASSIGN_LDS_WORDS_SMALL = 12 * 1024;
int assign_lds_words(int NC)
{
    int words = ASSIGN_LDS_WORDS;
    if ((size_t)256 * NC * sizeof(uint32_t) > 48 * 1024 &&
        allow_dynamic_lds(k_assign, ASSIGN_LDS_WORDS * sizeof(uint32_t)) != NMSA_OK)
        words = ASSIGN_LDS_WORDS_SMALL;
    return words;
}
size_t assign_lds_bytes(int NC, int words)
{
    int rows = words / NC;
    rows = rows > 256 ? 256 : (rows < 1 ? 1 : rows);
    return (size_t)rows * NC * sizeof(uint32_t);
}

int fused_iters(int P)
{
    // 1024 px per workgroup: 300 workgroups per 640x480 image -> 9600 at B=32, >> 256 CUs x 8
    // resident (measured on MI355X, profiles/r01_tune_fused.txt: 1 < 2 < 4 iterations)
    (void)P;
    static const int it = env_int("NMSA_FUSED_ITERS", 1);     // tuning knob
    return it > 0 ? it : 1;
}

template <int DTYPE>
int launch_fused(const void* logits, const float* offset, const int32_t* centers_yx,
                 const int32_t* n_centers, const uint8_t* is_thing,
                 int B, int C, int H, int W, int max_centers,
                 float sy, float sx, int use_thr, float thr,
                 uint8_t* sem_u8, uint8_t* inst, uint8_t* fg_out, float* score,
                 uint32_t* votes, int vote_rows_hint, hipStream_t stream)
{
    const int P = H * W;
    const int iters = fused_iters(P);
    const int chunks = (P + iters * PX_PER_ITER - 1) / (iters * PX_PER_ITER);
    (void)vote_rows_hint;       // the LDS vote table is a fixed-size hash now: no sizing hint needed
    static const int ablate = env_int("NMSA_FUSED_ABLATE", 0);   // diagnostics: 1 no votes, 2 no search, 4 no offsets
    const size_t lds = (size_t)max_centers * sizeof(float2) + (size_t)FUSED_VOTE_SLOTS * 8 + 256;
    if (lds > 64 * 1024) return NMSA_ERR_ARG;
    const bool vec = (P % 4 == 0) &&
                     (((uintptr_t)logits | (uintptr_t)offset | (uintptr_t)sem_u8 |
                       (uintptr_t)inst | (uintptr_t)fg_out | (uintptr_t)score) % 16 == 0);
    dim3 grid(chunks, B), block(FUSED_THREADS);
#define NMSA_LAUNCH_FUSED(V, S)                                                              \
    hipLaunchKernelGGL((k_panoptic_fused<DTYPE, V, S>), grid, block, lds, stream, logits,    \
                       offset, centers_yx, n_centers, is_thing, C, H, W, max_centers, iters, \
                       sy, sx, use_thr, thr, sem_u8, inst, fg_out, score, votes, ablate)
    // 128 x 8 pixel tiles per workgroup (a wave covers 128 x 2: compact in the image, so the
    // center culling leaves 1-3 candidates; every wave load is two whole row pieces of 256 B /
    // 512 B).  Measured on one box, B=32 640x480 C=40, 24 | 64 centers: 16-bit logits 146 us vs
    // 153 us with 64 x 16 tiles vs 172 | 203 us with 1024 consecutive pixels; f32 286 | 286 us vs
    // 299 | 299 (64 x 16) vs 293 | 300 (consecutive).
    static const int tiled = env_int("NMSA_FUSED_TILED", 2);           // 0: never, 1: 16-bit, 2: all
    static const int tile_w = env_int("NMSA_FUSED_TILE_W", 128);
    if (vec && !score && W % 4 == 0 && (tiled == 2 || (tiled == 1 && DTYPE != NMSA_F32))) {
        if (tile_w == 64) {
            const int tiles = ((W + 63) / 64) * ((H + 15) / 16);
            hipLaunchKernelGGL((k_panoptic_fused<DTYPE, true, false, 8, true, false, true, 6>),
                               dim3((tiles + iters - 1) / iters, B), block, lds, stream, logits, offset,
                               centers_yx, n_centers, is_thing, C, H, W, max_centers, iters, sy, sx,
                               use_thr, thr, sem_u8, inst, fg_out, score, votes, ablate);
        } else if (tile_w == 256) {
            const int tiles = ((W + 255) / 256) * ((H + 3) / 4);
            hipLaunchKernelGGL((k_panoptic_fused<DTYPE, true, false, 8, true, false, true, 8>),
                               dim3((tiles + iters - 1) / iters, B), block, lds, stream, logits, offset,
                               centers_yx, n_centers, is_thing, C, H, W, max_centers, iters, sy, sx,
                               use_thr, thr, sem_u8, inst, fg_out, score, votes, ablate);
        } else {
            const int tiles = ((W + 127) / 128) * ((H + 7) / 8);
            // 8 class planes in flight per lane: 4 / 12 / 20 measured 169 / 166 / 197 us vs 161 (bf16)
            hipLaunchKernelGGL((k_panoptic_fused<DTYPE, true, false, 8, true, false, true, 7>),
                               dim3((tiles + iters - 1) / iters, B), block, lds, stream, logits, offset,
                               centers_yx, n_centers, is_thing, C, H, W, max_centers, iters, sy, sx,
                               use_thr, thr, sem_u8, inst, fg_out, score, votes, ablate);
        }
    } else if (vec && score && W % 4 == 0 && tiled == 2 && env_int("NMSA_FUSED_SCORE_TILED", 1)) {
        // the with-score variant on the same 128 x 8 tiles
        const int tiles = ((W + 127) / 128) * ((H + 7) / 8);
        hipLaunchKernelGGL((k_panoptic_fused<DTYPE, true, true, 8, true, false, true, 7>),
                           dim3((tiles + iters - 1) / iters, B), block, lds, stream, logits, offset,
                           centers_yx, n_centers, is_thing, C, H, W, max_centers, iters, sy, sx,
                           use_thr, thr, sem_u8, inst, fg_out, score, votes, ablate);
    } else if (vec) { if (score) NMSA_LAUNCH_FUSED(true, true); else NMSA_LAUNCH_FUSED(true, false); }
    else { if (score) NMSA_LAUNCH_FUSED(false, true); else NMSA_LAUNCH_FUSED(false, false); }
#undef NMSA_LAUNCH_FUSED
    return check_launch();
}

template <int DTYPE>
int launch_argmax(const void* logits, int B, int C, int P, uint8_t* idx_u8, int64_t* idx_i64,
                  float* score, hipStream_t stream)
{
    const bool vec = (P % 4 == 0) &&
                     (((uintptr_t)logits | (uintptr_t)idx_u8 | (uintptr_t)idx_i64 |
                       (uintptr_t)score) % 16 == 0);
    dim3 grid((P + PX_PER_ITER - 1) / PX_PER_ITER, B), block(FUSED_THREADS);
#define NMSA_LAUNCH_ARGMAX(V, S)                                                          \
    hipLaunchKernelGGL((k_semantic_argmax<DTYPE, V, S>), grid, block, 0, stream, logits,  \
                       C, P, idx_u8, idx_i64, score)
    if (vec) { if (score) NMSA_LAUNCH_ARGMAX(true, true); else NMSA_LAUNCH_ARGMAX(true, false); }
    else { if (score) NMSA_LAUNCH_ARGMAX(false, true); else NMSA_LAUNCH_ARGMAX(false, false); }
#undef NMSA_LAUNCH_ARGMAX
    return check_launch();
}

template <int DTYPE>
int launch_softmax(const void* logits, int B, int C, int P, float* probs, hipStream_t stream)
{
    const bool vec = (P % 4 == 0) && (((uintptr_t)logits | (uintptr_t)probs) % 16 == 0);
    dim3 grid((P + PX_PER_ITER - 1) / PX_PER_ITER, B), block(FUSED_THREADS);
    if (vec && C <= 48 && getenv("NMSA_SOFTMAX_2PASS") == nullptr)
        hipLaunchKernelGGL((k_semantic_softmax_reg<DTYPE, 48>), grid, block, 0, stream, logits, C, P, probs);
    else if (vec) hipLaunchKernelGGL((k_semantic_softmax<DTYPE, true>), grid, block, 0, stream, logits, C, P, probs);
    else hipLaunchKernelGGL((k_semantic_softmax<DTYPE, false>), grid, block, 0, stream, logits, C, P, probs);
    return check_launch();
}

bool bad_dims(int B, int H, int W)
{
    return B <= 0 || H <= 0 || W <= 0 || (int64_t)H * W > ((int64_t)1 << 30) || B > 65535;
}

}  // namespace

extern "C" int nmsa_group_offsets(const float* offset, const uint8_t* fg,
                                  const int32_t* centers_yx, const int32_t* n_centers,
                                  int B, int H, int W, int max_centers,
                                  float scale_y, float scale_x,
                                  int use_dist_thr, float dist_thr,
                                  uint8_t* inst, int32_t* area, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!offset || !fg || !centers_yx || !n_centers || !inst) return NMSA_ERR_ARG;
    if (bad_dims(B, H, W) || max_centers <= 0) return NMSA_ERR_ARG;
    const int P = H * W;
    const size_t lds = (size_t)max_centers * sizeof(float2) + 256 * 4;
    if (lds > 64 * 1024) return NMSA_ERR_ARG;
    int rc;
    if (area) {
        rc = check_hip(hipMemsetAsync(area, 0, (size_t)B * 256 * sizeof(int32_t), stream));
        if (rc) return rc;
    }
    const int iters = 2;
    const int chunks = (P + iters * PX_PER_ITER - 1) / (iters * PX_PER_ITER);
    const bool vec = (P % 4 == 0) &&
                     (((uintptr_t)offset | (uintptr_t)fg | (uintptr_t)inst) % 16 == 0);
    dim3 grid(chunks, B), block(FUSED_THREADS);
    static const int tiled_env = getenv("NMSA_GROUP_TILED") ? atoi(getenv("NMSA_GROUP_TILED")) : 1;
    if (vec && W % 4 == 0 && tiled_env) {
        const int tiles = ((W + 63) / 64) * ((H + 15) / 16);
        hipLaunchKernelGGL((k_group_offsets<true, true>), dim3((tiles + iters - 1) / iters, B), block, lds,
                           stream, offset, fg, centers_yx, n_centers, H, W, max_centers, iters,
                           scale_y, scale_x, use_dist_thr, dist_thr, inst, area);
    } else if (vec)
        hipLaunchKernelGGL(k_group_offsets<true>, grid, block, lds, stream, offset, fg, centers_yx,
                           n_centers, H, W, max_centers, iters, scale_y, scale_x, use_dist_thr,
                           dist_thr, inst, area);
    else
        hipLaunchKernelGGL(k_group_offsets<false>, grid, block, lds, stream, offset, fg, centers_yx,
                           n_centers, H, W, max_centers, iters, scale_y, scale_x, use_dist_thr,
                           dist_thr, inst, area);
    return check_launch();
}

extern "C" int nmsa_semantic_argmax(const void* logits, int logits_dtype,
                                    int B, int C, int H, int W,
                                    uint8_t* idx_u8, int64_t* idx_i64, float* score,
                                    nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!logits || bad_dims(B, H, W) || C <= 0) return NMSA_ERR_ARG;
    if (idx_u8 && C > 256) return NMSA_ERR_ARG;
    const int P = H * W;
    switch (logits_dtype) {
        case NMSA_F32: return launch_argmax<NMSA_F32>(logits, B, C, P, idx_u8, idx_i64, score, stream);
        case NMSA_BF16: return launch_argmax<NMSA_BF16>(logits, B, C, P, idx_u8, idx_i64, score, stream);
        case NMSA_F16: return launch_argmax<NMSA_F16>(logits, B, C, P, idx_u8, idx_i64, score, stream);
        default: return NMSA_ERR_ARG;
    }
}

extern "C" int nmsa_semantic_softmax(const void* logits, int logits_dtype,
                                     int B, int C, int H, int W, float* probs,
                                     nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!logits || !probs || bad_dims(B, H, W) || C <= 0) return NMSA_ERR_ARG;
    const int P = H * W;
    switch (logits_dtype) {
        case NMSA_F32: return launch_softmax<NMSA_F32>(logits, B, C, P, probs, stream);
        case NMSA_BF16: return launch_softmax<NMSA_BF16>(logits, B, C, P, probs, stream);
        case NMSA_F16: return launch_softmax<NMSA_F16>(logits, B, C, P, probs, stream);
        default: return NMSA_ERR_ARG;
    }
}

extern "C" int nmsa_panoptic_fused(const void* logits, int logits_dtype, const float* offset,
                                   const int32_t* centers_yx, const int32_t* n_centers,
                                   const uint8_t* is_thing,
                                   int B, int C, int H, int W, int max_centers,
                                   float scale_y, float scale_x,
                                   int use_dist_thr, float dist_thr,
                                   uint8_t* sem_u8, uint8_t* inst, uint8_t* fg_out, float* score,
                                   uint32_t* votes, int votes_are_zero, int vote_rows_hint,
                                   nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!logits || !offset || !centers_yx || !n_centers || !is_thing || !sem_u8 || !inst || !votes)
        return NMSA_ERR_ARG;
    if (bad_dims(B, H, W) || C <= 0 || C > 256 || max_centers <= 0) return NMSA_ERR_ARG;
    if (!votes_are_zero) {
        int rc = check_hip(hipMemsetAsync(votes, 0, (size_t)B * 256 * (C + 1) * sizeof(uint32_t), stream));
        if (rc) return rc;
    }
    switch (logits_dtype) {
        case NMSA_F32:
            return launch_fused<NMSA_F32>(logits, offset, centers_yx, n_centers, is_thing, B, C, H, W,
                                          max_centers, scale_y, scale_x, use_dist_thr, dist_thr,
                                          sem_u8, inst, fg_out, score, votes, vote_rows_hint, stream);
        case NMSA_BF16:
            return launch_fused<NMSA_BF16>(logits, offset, centers_yx, n_centers, is_thing, B, C, H, W,
                                           max_centers, scale_y, scale_x, use_dist_thr, dist_thr,
                                           sem_u8, inst, fg_out, score, votes, vote_rows_hint, stream);
        case NMSA_F16:
            return launch_fused<NMSA_F16>(logits, offset, centers_yx, n_centers, is_thing, B, C, H, W,
                                          max_centers, scale_y, scale_x, use_dist_thr, dist_thr,
                                          sem_u8, inst, fg_out, score, votes, vote_rows_hint, stream);
        default: return NMSA_ERR_ARG;
    }
}

extern "C" int nmsa_panoptic_assign(uint32_t* votes, int B, int n_vote_classes, int clear_votes,
                                    int64_t max_instances_per_category, int64_t void_label,
                                    int64_t* pan_of_inst, int32_t* area,
                                    int64_t* ids_pan, int64_t* ids_ins, int32_t* n_ids,
                                    nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!votes || !pan_of_inst || !ids_pan || !ids_ins || !n_ids) return NMSA_ERR_ARG;
    if (B <= 0 || n_vote_classes <= 0 || n_vote_classes > 4096) return NMSA_ERR_ARG;
    const int lds_words = assign_lds_words(n_vote_classes);
    hipLaunchKernelGGL(k_assign, dim3(B), dim3(ASSIGN_THREADS), assign_lds_bytes(n_vote_classes, lds_words),
                       stream, votes, n_vote_classes, clear_votes,
                       max_instances_per_category, void_label, pan_of_inst, area,
                       ids_pan, ids_ins, n_ids, lds_words);
    return check_launch();
}

extern "C" int nmsa_panoptic_paint(const uint8_t* sem_u8, const uint8_t* inst,
                                   const int64_t* pan_of_inst, const uint8_t* is_thing,
                                   int B, int C, int H, int W,
                                   int64_t max_instances_per_category, int64_t void_label,
                                   int64_t* pan, int64_t* pan_sem, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!sem_u8 || !inst || !pan_of_inst || !is_thing || !pan) return NMSA_ERR_ARG;
    if (bad_dims(B, H, W) || C <= 0 || C > 256 || max_instances_per_category <= 0) return NMSA_ERR_ARG;
    const int P = H * W;
    const bool vec = (P % 8 == 0) &&
                     (((uintptr_t)sem_u8 | (uintptr_t)inst | (uintptr_t)pan | (uintptr_t)pan_sem) % 16 == 0);
    const int iters = 1;
    dim3 grid((P + 2048 * iters - 1) / (2048 * iters), B), block(256);
    // whole-line stores (k_paint2) measured 16.8 us vs 27.4 us for 4 px per lane at B=32 640x480
    if (vec) {
        const int steps = 4 * iters;                       // 512 px per step and block
        hipLaunchKernelGGL(k_paint2, grid, block, 0, stream, sem_u8, inst, pan_of_inst, is_thing,
                           C, P, steps, max_instances_per_category, void_label, pan, pan_sem);
    }
    else
        hipLaunchKernelGGL(k_paint<false>, grid, block, 0, stream, sem_u8, inst, pan_of_inst, is_thing,
                           C, P, iters, max_instances_per_category, void_label, pan, pan_sem);
    return check_launch();
}

extern "C" int nmsa_panoptic_merge(const void* sem, int sem_dtype, const void* ins, int ins_dtype,
                                   const uint8_t* thing_seg, const uint8_t* is_thing_class,
                                   int B, int n_classes, int H, int W,
                                   int64_t max_instances_per_category, int64_t void_label,
                                   uint32_t* votes, int64_t* pan_of_inst,
                                   int64_t* pan, int64_t* ids_pan, int64_t* ids_ins, int32_t* n_ids,
                                   nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!sem || !ins || !thing_seg || !is_thing_class || !votes || !pan_of_inst || !pan ||
        !ids_pan || !ids_ins || !n_ids)
        return NMSA_ERR_ARG;
    if (bad_dims(B, H, W) || n_classes <= 0 || n_classes > 4096) return NMSA_ERR_ARG;
    if (sem_dtype < NMSA_U8 || sem_dtype > NMSA_I64 || ins_dtype < NMSA_U8 || ins_dtype > NMSA_I64)
        return NMSA_ERR_ARG;
    const int P = H * W;
    int rc = check_hip(hipMemsetAsync(votes, 0, (size_t)B * 256 * n_classes * sizeof(uint32_t), stream));
    if (rc) return rc;
    int gx = (P + MG_PX_PER_BLOCK - 1) / MG_PX_PER_BLOCK;
    if (gx > 1024) gx = 1024;
    const size_t esz[4] = {1, 2, 4, 8};
    const bool vec = P % 4 == 0 && (uintptr_t)sem % (4 * esz[sem_dtype]) == 0 &&
                     (uintptr_t)ins % (4 * esz[ins_dtype]) == 0 && (uintptr_t)thing_seg % 4 == 0 &&
                     (uintptr_t)pan % 16 == 0;
    if (vec) hipLaunchKernelGGL(k_merge_votes<true>, dim3(gx, B), dim3(256), 0, stream, sem, sem_dtype, ins,
                                ins_dtype, thing_seg, n_classes, P, votes);
    else hipLaunchKernelGGL(k_merge_votes<false>, dim3(gx, B), dim3(256), 0, stream, sem, sem_dtype, ins,
                            ins_dtype, thing_seg, n_classes, P, votes);
    rc = check_launch();
    if (rc) return rc;
    const int lds_words = assign_lds_words(n_classes);
    hipLaunchKernelGGL(k_assign, dim3(B), dim3(ASSIGN_THREADS), assign_lds_bytes(n_classes, lds_words),
                       stream, votes, n_classes, 0,
                       max_instances_per_category, void_label, pan_of_inst, (int32_t*)nullptr,
                       ids_pan, ids_ins, n_ids, lds_words);
    rc = check_launch();
    if (rc) return rc;
    if (vec) hipLaunchKernelGGL(k_merge_paint<true>, dim3(gx, B), dim3(256), 0, stream, sem, sem_dtype, ins,
                                ins_dtype, thing_seg, is_thing_class, pan_of_inst, n_classes, P,
                                max_instances_per_category, void_label, pan);
    else hipLaunchKernelGGL(k_merge_paint<false>, dim3(gx, B), dim3(256), 0, stream, sem, sem_dtype, ins,
                            ins_dtype, thing_seg, is_thing_class, pan_of_inst, n_classes, P,
                            max_instances_per_category, void_label, pan);
    return check_launch();
}

extern "C" int nmsa_instance_orientation(const float* orientation, const uint8_t* inst,
                                         const uint8_t* mask, int B, int H, int W,
                                         double* sums, int32_t* count, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!orientation || !inst || !sums || !count || bad_dims(B, H, W)) return NMSA_ERR_ARG;
    const int P = H * W;
    int rc = check_hip(hipMemsetAsync(sums, 0, (size_t)B * 256 * 2 * sizeof(double), stream));
    if (rc) return rc;
    rc = check_hip(hipMemsetAsync(count, 0, (size_t)B * 256 * sizeof(int32_t), stream));
    if (rc) return rc;
    const int gx = (P + 4 * MG_PX_PER_BLOCK - 1) / (4 * MG_PX_PER_BLOCK);
    const bool vec = P % 4 == 0 && (uintptr_t)orientation % 16 == 0 && (uintptr_t)inst % 4 == 0 &&
                     (uintptr_t)mask % 4 == 0;
    if (vec) hipLaunchKernelGGL(k_orientation_sums<true>, dim3(gx, B), dim3(256), 0, stream, orientation,
                                inst, mask, P, sums, count);
    else hipLaunchKernelGGL(k_orientation_sums<false>, dim3(gx, B), dim3(256), 0, stream, orientation,
                            inst, mask, P, sums, count);
    return check_launch();
}

// ---------------------------------------------------------------------------------
// host hand-over of the small per-image tables: ONE launch packs the first `columns` entries
// of every table into one f64 row per image (every value is exact in f64: ids < 2^53, f32
// scores), so that the host needs a single device->host copy per batch.
// row = [n_centers, n_ids, centers_yx[columns][2], scores[columns], area[columns + 1],
//        ids_pan[columns], ids_ins[columns]]
// ---------------------------------------------------------------------------------
namespace nmsa {
__global__ __launch_bounds__(256) void k_pack_tables(
    const int32_t* __restrict__ n_centers, const int32_t* __restrict__ n_ids,
    const int32_t* __restrict__ centers_yx, const float* __restrict__ scores,
    const int32_t* __restrict__ area, const int64_t* __restrict__ ids_pan,
    const int64_t* __restrict__ ids_ins, int max_centers, int kc, int ka, int ki,
    double* __restrict__ out)
{
    const int b = blockIdx.x;
    const int row = 2 + 3 * kc + ka + 2 * ki;
    double* o = out + (size_t)b * row;
    for (int i = threadIdx.x; i < row; i += blockDim.x) {
        int k = i;
        double v;
        if (k == 0) v = n_centers[b];
        else if (k == 1) v = n_ids ? n_ids[b] : 0;
        else if ((k -= 2) < 2 * kc) v = centers_yx[(size_t)b * max_centers * 2 + k];
        else if ((k -= 2 * kc) < kc) v = scores[(size_t)b * max_centers + k];
        else if ((k -= kc) < ka) v = area[(size_t)b * 256 + k];
        else if ((k -= ka) < ki) v = (double)ids_pan[(size_t)b * 256 + k];
        else v = (double)ids_ins[(size_t)b * 256 + (k - ki)];
        o[i] = v;
    }
}
}  // namespace nmsa

extern "C" int nmsa_pack_tables(const int32_t* n_centers, const int32_t* n_ids,
                                const int32_t* centers_yx, const float* scores,
                                const int32_t* area, const int64_t* ids_pan, const int64_t* ids_ins,
                                int B, int max_centers, int columns, double* out,
                                nmsa_stream_t stream_)
{
    if (!n_centers || !centers_yx || !scores || !area || !out) return NMSA_ERR_ARG;
    const bool with_ids = n_ids && ids_pan && ids_ins;      // NULL id tables: no id columns
    if (B <= 0 || max_centers <= 0 || columns <= 0) return NMSA_ERR_ARG;
    const int kc = columns < max_centers ? columns : max_centers;
    const int ka = kc + 1 < 256 ? kc + 1 : 256;
    const int ki = with_ids ? (kc < 256 ? kc : 256) : 0;
    hipLaunchKernelGGL(nmsa::k_pack_tables, dim3(B), dim3(256), 0, (hipStream_t)stream_, n_centers, n_ids,
                       centers_yx, scores, area, ids_pan, ids_ins, max_centers, kc, ka, ki, out);
    return nmsa::check_launch();
}
