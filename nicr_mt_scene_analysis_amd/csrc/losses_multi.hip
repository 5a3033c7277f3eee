// losses_multi.hip — all losses of a task helper (every loss, every supervision scale) in one call.
#include <stdlib.h>
#include "loss_bodies.hpp"

namespace nmsa {

// =================================================================================
// a10: the losses of a task helper in ONE forward launch (task_helper/instance.py:92-269,
// task_helper/semantic.py:57-90, task_helper/base.py:161-182): every (loss, scale) pair is an
// ITEM, items whose sums the caller adds before dividing by the summed counts form a TOTAL.
//   k_multi_count    counts the labels / mask bytes of every item (1 B/px); its LAST workgroup
//                    (a ticket) forms the counts per item, the divisor per total and the EXPECTED
//                    upstream gradient of the total's loss sums: w / n with w from the total's
//                    spec record (multi_expect_body: a launch of its own until round 4)
//   k_multi_loss     all items in one launch (block ranges): forward sums + gradients
//   k_multi_finalize block partials -> sums / counts per item, fixed order
// = 3 launches with gradients, 2 without (no count: the divisors come out of the finalized counts).
// backward: ONE launch — every workgroup of k_multi_loss<..., MODE 2> turns the upstream gradients
// of the three outputs into the upstream scale per item (a handful of loads), workgroup 0 also
// compares them with the expectation, keeps the record's `w` up to date and stores the scales
// (multi_spec_update: the kernel k_multi_spec until round 4, still launched when no item is part
// of the joint launch); the walk recomputes only the items whose upstream gradient differs
// bit-wise from the expectation.
// The two tickets (count, finalize) live behind the caller's spec records — spec[8 T + 0 / 1] —
// zero between calls: the last workgroup of each launch resets its ticket, so no launch is spent
// on zeroing them.
//
// Spec record (int32[8], device): [0] confirmed [1] recomputed [2] w (fp32 bits): the factor the
// caller multiplies the total with before backward (loss weights, AMP scale: learned, see
// k_multi_spec) [3] last upstream gradient [4] last divisor [5] flags (bit 0: expectation
// switched off) [6] misses in a row [7] agreeing estimates in a row while switched off.
// =================================================================================
constexpr int MULTI_MAX_ITEMS = NMSA_MULTI_MAX_ITEMS;
constexpr int MULTI_MAX_TOTALS = NMSA_MULTI_MAX_TOTALS;
constexpr int MULTI_COUNT_MAX_BLOCKS = 256;            // per item

struct MultiItem {
    const void* pred; const void* target; const uint8_t* mask; const float* weights; void* grad;
    int kind, dtype, B, C, P, vec, total, clamp;
    float param;
    int block0, nbx;                                   // first block of the item, blocks per image
    int cblock0, cnblocks;                             // count pass: first block, blocks
    int count_mode;                                    // 0: B * P, 1: bytes of `mask` in [lo, hi], 2: none, 3: int32 words of `mask` in [lo, hi]
    int lo, hi;
    int in_launch;                                     // 1: part of k_multi_loss, 0: own kernel (wide CE)
    int first_of_total;
    int L;                                             // cosine embedding: LUT rows per image
};
struct MultiArgs { MultiItem it[MULTI_MAX_ITEMS]; int n_items, n_totals, n_blocks; };

// divisor of total t from per-item counts (accumulate_losses: max(sum of counts, 1) as float32;
// an item with clamp enters as max(count, 1), task_helper/instance.py:206-211)
__device__ inline float multi_divisor(const MultiArgs& a, const long long* counts, int t)
{
    long long n = 0;
    for (int i = 0; i < a.n_items; ++i)
        if (a.it[i].total == t) n += a.it[i].clamp == 1 ? max(counts[i], 1LL) : counts[i];
    return (float)max(n, 1LL);
}

// counts per item from the count partials, divisor + expected upstream gradient per total; one
// workgroup of LOSS_THREADS (the last one of k_multi_count: `partials` were written by other
// workgroups, read them past the L1)
__device__ inline void multi_expect_body(const MultiArgs& a, long long* partials,
                                         const int32_t* __restrict__ spec, float* __restrict__ expect)
{
    __shared__ long long s_count[MULTI_MAX_ITEMS];
    const int w = threadIdx.x >> 6, l = lane_id();
    for (int i = w; i < a.n_items; i += LOSS_THREADS / 64) {
        const MultiItem& it = a.it[i];
        const bool counted = it.count_mode == 1 || it.count_mode == 3;
        long long c = 0;
        if (counted) {
            for (int k = l; k < it.cnblocks; k += 64)
                c += __hip_atomic_load(&partials[it.cblock0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
        if (l == 0) s_count[i] = counted ? c : (long long)it.B * it.P;
    }
    __syncthreads();
    if ((int)threadIdx.x < a.n_totals) {
        const int t = threadIdx.x;
        bool known = true;
        for (int i = 0; i < a.n_items; ++i)
            if (a.it[i].total == t && a.it[i].count_mode == 2) known = false;   // the divisor is no count of mask bytes
        const float nf = multi_divisor(a, s_count, t);
        const float wv = __int_as_float(spec[8 * t + 2]);
        const bool off = (spec[8 * t + 5] & 1) || !known;
        expect[2 * t] = off ? __int_as_float(0x7fc00000) : wv / nf;     // NaN: the forward writes no gradient
        expect[2 * t + 1] = nf;                                         // (!known: k_multi_finalize corrects it)
    }
}

// number of bytes / int32 words of the item's mask inside [lo, hi] that fall to this workgroup
__device__ inline long long multi_count_block(const MultiItem& it, int bi)
{
    const long long n = (long long)it.B * it.P;
    if (it.count_mode == 3) {
        // int32 indices (cosine embedding: 0 = no target): 4 words per 16-byte load
        const int32_t* v32 = (const int32_t*)it.mask;
        const long long per4 = ((n + it.cnblocks - 1) / it.cnblocks + 3) / 4 * 4;
        const long long b4 = min(n, per4 * bi), e4 = min(n, b4 + per4);
        const unsigned lo3 = (unsigned)it.lo, span3 = (unsigned)(it.hi - it.lo);
        long long c3 = 0;
        long long q = b4 + (long long)threadIdx.x * 4;
        if ((((uintptr_t)v32) & 15) == 0) {
            for (; q + 4 <= e4; q += LOSS_THREADS * 4) {
                const u32x4_s w = __builtin_nontemporal_load((const u32x4_s*)(v32 + q));
                c3 += ((w.x - lo3) <= span3) + ((w.y - lo3) <= span3) + ((w.z - lo3) <= span3) + ((w.w - lo3) <= span3);
            }
        }
        for (; q < e4; q += LOSS_THREADS * 4)
            for (long long j = q; j < min(e4, q + 4); ++j) c3 += (((unsigned)v32[j] - lo3) <= span3);
        return c3;
    }
    const long long per = ((n + it.cnblocks - 1) / it.cnblocks + 15) / 16 * 16;
    const long long begin = min(n, per * bi), end = min(n, begin + per);
    const uint8_t* v = it.mask;
    const unsigned lo = (unsigned)it.lo, span = (unsigned)(it.hi - it.lo);
    const bool vec = (((uintptr_t)v) & 15) == 0;
    long long cnt = 0;
    long long k = begin + (long long)threadIdx.x * 16;
    auto count16 = [&](const u32x4_s w) {
        const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
        int c = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) c += ((((ww[q] >> (8 * j)) & 0xFF) - lo) <= span);
        return c;
    };
    if (vec) {
        constexpr long long STEP = LOSS_THREADS * 16;
        for (; k + 3 * STEP + 16 <= end; k += 4 * STEP) {          // 4 x 16 B in flight per lane
            u32x4_s w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) w[u] = __builtin_nontemporal_load((const u32x4_s*)(v + k + u * STEP));
#pragma unroll
            for (int u = 0; u < 4; ++u) cnt += count16(w[u]);
        }
        for (; k + 16 <= end; k += STEP) cnt += count16(__builtin_nontemporal_load((const u32x4_s*)(v + k)));
    }
    for (; k < end; k += LOSS_THREADS * 16)
        for (long long j = k; j < min(end, k + 16); ++j) cnt += (((unsigned)v[j] - lo) <= span);
    return cnt;
}

// the first launch of a call that writes gradients.  `tickets` = spec + 8 n_totals: [0] this
// launch's, [1] k_multi_finalize's; both are zero between calls (the last workgroup resets its own)
__global__ __launch_bounds__(LOSS_THREADS) void k_multi_count(MultiArgs a, long long* __restrict__ partials,
                                                              const int32_t* __restrict__ spec,
                                                              float* __restrict__ expect,
                                                              unsigned int* __restrict__ tickets)
{
    __shared__ long long s_cnt[LOSS_THREADS / 64];
    __shared__ bool s_last;
    int i = 0;
    while (i + 1 < a.n_items && (int)blockIdx.x >= a.it[i].cblock0 + a.it[i].cnblocks) ++i;
    const MultiItem& it = a.it[i];
    long long cnt = 0;
    if ((it.count_mode == 1 || it.count_mode == 3) && (int)blockIdx.x >= it.cblock0)
        cnt = multi_count_block(it, blockIdx.x - it.cblock0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane_id() == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        long long c = 0;
        for (int q = 0; q < LOSS_THREADS / 64; ++q) c += s_cnt[q];
        // my partial before my ticket: ONE write-through (sc1) 8-byte store, drained — not a release
        // fence: a thousand workgroups writing back their XCD's L2 at once cost 26 us here
        __hip_atomic_store(&partials[blockIdx.x], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(&tickets[0], 1u) == gridDim.x - 1;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other workgroups' partials after it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!s_last) return;
    multi_expect_body(a, partials, spec, expect);
    if (threadIdx.x == 0) tickets[0] = 0u;                 // zero again for the next call
}

// upstream weight behind an observed gradient g = fl(w / n): candidates around fl(g n) that
// reproduce g, preferring one that also explains the previous observation, then the "roundest"
__device__ inline float spec_estimate_w(float g, float n, float g_prev, float n_prev)
{
    const float c0 = g * n;
    float best = c0;
    int best_score = -1;
    for (int k = -4; k <= 4; ++k) {
        const float c = __int_as_float(__float_as_int(c0) + k);
        if (!(c / n == g)) continue;
        int score = 1 + __builtin_ctz((unsigned)__float_as_int(c) | 0x800000u);     // trailing zero bits of the significand
        if (n_prev > 0.f && c / n_prev == g_prev) score += 64;
        if (score > best_score) { best_score = score; best = c; }
    }
    return best;
}

struct MultiBwd {                                      // what the backward launch turns into upstream scales
    const float* grad_sums; const float* grad_items; const float* grad_totals;
    const long long* counts;
    int32_t* spec; int32_t* counters;
    float* gs;                                         // out: upstream scale per item
};

// upstream scale of item i's raw loss sum from the gradients of the three outputs (what
// autograd's division backward gives: grad / divisor, float32)
__device__ inline float multi_upstream(const MultiArgs& a, const MultiBwd& bw, const float* __restrict__ expect, int i)
{
    const MultiItem& it = a.it[i];
    float g = bw.grad_sums ? bw.grad_sums[i] : 0.f;
    const float gi = bw.grad_items ? bw.grad_items[i] : 0.f, gt = bw.grad_totals ? bw.grad_totals[it.total] : 0.f;
    if (gi != 0.f) g += gi / (float)(it.clamp ? max(bw.counts[i], 1LL) : bw.counts[i]);
    if (gt != 0.f) g += gt / expect[2 * it.total + 1];
    return g;
}

// thread t < n_totals of ONE workgroup: judge the expectation of total t against the upstream
// scale of its first item with a gradient and keep the record up to date
__device__ inline void multi_spec_update(const MultiArgs& a, const MultiBwd& bw, const float* __restrict__ expect, int t)
{
    int32_t* spec = bw.spec;
    int32_t* counters = bw.counters;
    int first = -1;
    for (int i = 0; i < a.n_items; ++i) if (a.it[i].total == t && a.it[i].grad && first < 0) first = i;
    if (first < 0 || !spec) return;          // no record: a recompute nobody predicted (retained graph)
    int32_t* r = spec + 8 * t;
    const float g = multi_upstream(a, bw, expect, first), e = expect[2 * t], nf = expect[2 * t + 1];
    const float g_prev = __int_as_float(r[3]), n_prev = __int_as_float(r[4]);
    const bool same = e == e && __float_as_int(g) == __float_as_int(e);     // NaN: no expectation
    if (counters) atomicAdd(&counters[same ? 0 : 1], 1);       // per-device tally (statistics only)
    if (same) { r[0] += 1; r[6] = 0; }
    else {
        r[1] += 1;
        const float w_old = __int_as_float(r[2]);
        const float w_new = (g == g && g != 0.f) ? spec_estimate_w(g, nf, g_prev, n_prev) : w_old;
        if (r[5] & 1) {
            // switched off: back on once the estimate has been stable for a few steps
            r[7] = (__float_as_int(w_new) == __float_as_int(w_old)) ? r[7] + 1 : 0;
            if (r[7] >= 3) { r[5] &= ~1; r[6] = 0; r[7] = 0; }
        } else {
            r[6] += 1;
            if (r[6] >= 8) { r[5] |= 1; r[7] = 0; }     // the caller's factor keeps changing: stop guessing
        }
        r[2] = __float_as_int(w_new);
    }
    r[3] = __float_as_int(g);
    r[4] = __float_as_int(nf);
}

// calls whose items all run as launches of their own (no joint launch to ride in)
__global__ void k_multi_spec(MultiArgs a, MultiBwd bw, const float* __restrict__ expect)
{
    if ((int)threadIdx.x < a.n_items) bw.gs[threadIdx.x] = multi_upstream(a, bw, expect, threadIdx.x);
    if ((int)threadIdx.x < a.n_totals) multi_spec_update(a, bw, expect, threadIdx.x);
}

// all items of one call: block ranges [block0, block0 + nbx * B) per item.  CE_* select the ONE
// cross-entropy variant compiled into this instantiation (CE_NG = 0: no CE item in the launch);
// the element-wise and von Mises bodies are selected at run time (they are small).
// (up to 40 class planes in registers: 4 waves per SIMD as in k_ce_fused — the calls of the small
// bodies cost the allocator 16 registers otherwise and one wave per SIMD with them)
#ifndef NMSA_MULTI_FWD_U
#define NMSA_MULTI_FWD_U 4          // 16-bit logits: class planes per group of the forward-only walk
#endif
#ifndef NMSA_MULTI_FWD_WAVES
#define NMSA_MULTI_FWD_WAVES 5      // forward-only launches: the streaming CE walk needs few registers
                                    // (96 without label smoothing; with it 4 waves: no scratch)
#endif
template <int CE_DT, int CE_NG, bool CE_SM, int MODE>        // MODE as in ce_fused_body
__global__ __launch_bounds__(LOSS_THREADS)
__attribute__((amdgpu_waves_per_eu((MODE == 1) ? (CE_SM ? 4 : NMSA_MULTI_FWD_WAVES) : (CE_NG <= 5) ? 4 : 3, (MODE != 1 && CE_NG > 5) ? 3 : 8))) void k_multi_loss(
    MultiArgs a, const float* __restrict__ expect, MultiBwd bw,
    LossPartial* __restrict__ partials, int* __restrict__ status)
{
    extern __shared__ float s_w[];
    constexpr bool LOSS = MODE != 2;
    if (!LOSS) {
        // every workgroup forms the upstream scales itself (n_items <= 16: a handful of scalar
        // loads per item); workgroup 0 also judges the expectations, updates the records and stores
        // the scales for the launches that follow (wide cross entropies, cosine items)
        if (blockIdx.x == 0) {
            if ((int)threadIdx.x < a.n_items) bw.gs[threadIdx.x] = multi_upstream(a, bw, expect, threadIdx.x);
            if ((int)threadIdx.x < a.n_totals) multi_spec_update(a, bw, expect, threadIdx.x);
        }
        // the recomputing launch is a small grid walking the block list: when every item's
        // gradient stands (the usual case) its workgroups are gone after this check
    }
    // (lane i of every wave holds item i's upstream scale and whether its gradient stands: the
    // loads of all items in flight at once, then lane reads with wave-uniform indices)
    float g_lane = 0.f;
    bool redo_lane = false;
    if (!LOSS) {
        const int i = lane_id();
        if (i < a.n_items) {
            const MultiItem& it = a.it[i];
            const float e = expect[2 * it.total];
            g_lane = multi_upstream(a, bw, expect, i);
            redo_lane = it.in_launch && it.grad && (!(e == e) || __float_as_int(g_lane) != __float_as_int(e));
        }
        if (!__any(redo_lane)) return;
    }
    for (int blk = blockIdx.x; blk < a.n_blocks; blk += gridDim.x) {     // LOSS: exactly one pass
        int i = 0;
        while (i + 1 < a.n_items && (!a.it[i].in_launch || blk >= a.it[i].block0 + a.it[i].nbx * a.it[i].B)) ++i;
        const MultiItem& it = a.it[i];
        const int local = blk - it.block0;
        if (!it.in_launch || local < 0) continue;
        const int bx = local % it.nbx, b = local / it.nbx;
        float g = __int_as_float(0x7fc00000);              // NaN: no gradient wanted / no expectation
        if (MODE != 1 && it.grad && expect) g = expect[2 * it.total];
        if (!LOSS) {
            if (!it.grad) continue;
            const float gr = __shfl(g_lane, i);                                // (i is wave-uniform)
            if (g == g && __float_as_int(gr) == __float_as_int(g)) continue;   // the forward's gradient stands
            g = gr;
        }
        LossPartial* slot = partials ? partials + blk : nullptr;
#define MULTI_DT(CALL) switch (it.dtype) { case NMSA_F32: CALL(NMSA_F32); break; case NMSA_BF16: CALL(NMSA_BF16); break; \
                                            default: CALL(NMSA_F16); break; }
        switch (it.kind) {
            case NMSA_LOSS_CE:
                if constexpr (CE_NG != 0 && MODE == 1) {
                    // forward only: the streaming walk of k_ce_fwd (16-byte loads, 4 planes in
                    // flight, few registers) is faster than the register-resident column
                    constexpr int FPX = (CE_DT == NMSA_F32) ? 4 : 8;
                    const int vec16 = (it.P % FPX == 0) && ((((uintptr_t)it.pred) & 15) == 0);
                    // (two inlined copies with `vec` folded to a constant — no per-plane branch in the
                    // hot one — were measured SLOWER: 0.51 vs 0.45 ms for the forward-only call)
                    ce_fwd_body<CE_DT, FPX, CE_SM, (CE_DT == NMSA_F32) ? 8 : NMSA_MULTI_FWD_U>(
                        it.pred, (const uint8_t*)it.mask, it.weights, it.C, it.P, it.param, vec16, slot, status,
                        nullptr, s_w, bx, it.nbx, b);
                } else if constexpr (CE_NG != 0) {
                    ce_fused_body<CE_DT, CE_NG, CE_SM, MODE>(it.pred, (const uint8_t*)it.mask, it.weights, it.C, it.P,
                                                             it.param, it.vec, g, it.grad, slot, status, s_w, bx, b);
                }
                break;
            case NMSA_LOSS_MSE:
#define CALL(DT) elem_fused_body<DT, 0, MODE>(it.pred, (const float*)it.target, it.mask, it.C, it.P, it.vec, g, it.grad, slot, bx, it.nbx, b)
                MULTI_DT(CALL)
#undef CALL
                break;
            case NMSA_LOSS_L1:
#define CALL(DT) elem_fused_body<DT, 1, MODE>(it.pred, (const float*)it.target, it.mask, it.C, it.P, it.vec, g, it.grad, slot, bx, it.nbx, b)
                MULTI_DT(CALL)
#undef CALL
                break;
            case NMSA_LOSS_FOCAL:
#define CALL(DT) elem_fused_body<DT, 2, MODE>(it.pred, (const float*)it.target, it.mask, it.C, it.P, it.vec, g, it.grad, slot, bx, it.nbx, b)
                MULTI_DT(CALL)
#undef CALL
                break;
            default:
#define CALL(DT) vm_fused_body<DT, MODE>(it.pred, (const float*)it.target, it.mask, it.P, it.param, it.vec, g, it.grad, slot, bx, it.nbx, b)
                MULTI_DT(CALL)
#undef CALL
                break;
        }
#undef MULTI_DT
        if (!LOSS) __syncthreads();                        // s_w is rewritten by the next block of the walk
    }
}

// MULTI_FIN_SPLIT workgroups per item reduce slices of its block partials (fixed order); the LAST
// workgroup of the launch to finish (ticket) adds the slices per item, again in a fixed order, and
// forms the outputs: sums / counts / aux per item; divisors of totals k_multi_expect could not
// know (forward-only calls, focal items: counts that only the loss kernels produce); and
// out[0 .. n): the sums as float32, [n .. 2n): sum / count per item, [2n .. 2n + T): per total the
// float32 sums of its items added in item order, divided by the total's divisor
// (accumulate_losses, task_helper/base.py:161-182)
constexpr int MULTI_FIN_SPLIT = 16;
constexpr int MULTI_FIN_THREADS = 256;
static_assert(MULTI_MAX_ITEMS * MULTI_FIN_SPLIT <= MULTI_FIN_THREADS, "one thread per slice in the last workgroup");

__global__ __launch_bounds__(MULTI_FIN_THREADS) void k_multi_finalize(
    MultiArgs a, const LossPartial* __restrict__ partials, LossPartial* __restrict__ slices,
    unsigned int* __restrict__ ticket, int late_divisors, double* __restrict__ sums,
    long long* __restrict__ counts, double* __restrict__ aux, float* __restrict__ expect,
    float* __restrict__ out)
{
    __shared__ double s_sum[MULTI_FIN_THREADS], s_aux[MULTI_FIN_THREADS];
    __shared__ long long s_cnt[MULTI_FIN_THREADS];
    __shared__ bool s_last;
    const int item = blockIdx.x / MULTI_FIN_SPLIT, sl = blockIdx.x % MULTI_FIN_SPLIT;
    const MultiItem& it = a.it[item];
    const LossPartial* p = partials + it.block0;
    const int n = it.nbx * it.B;
    const int per = (n + MULTI_FIN_SPLIT - 1) / MULTI_FIN_SPLIT;
    const int begin = min(n, sl * per), end = min(n, begin + per);
    double x = 0, y = 0; long long c = 0;
    int k = begin + threadIdx.x;
    for (; k + 3 * MULTI_FIN_THREADS < end; k += 4 * MULTI_FIN_THREADS) {       // 4 independent loads per round
        LossPartial q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = p[k + u * MULTI_FIN_THREADS];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x += q[u].sum; y += q[u].aux; c += q[u].count; }
    }
    for (; k < end; k += MULTI_FIN_THREADS) { x += p[k].sum; y += p[k].aux; c += p[k].count; }
    s_sum[threadIdx.x] = x; s_aux[threadIdx.x] = y; s_cnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = MULTI_FIN_THREADS / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
            s_aux[threadIdx.x] += s_aux[threadIdx.x + o];
            s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // the slice before the ticket: write-through 8-byte stores, drained (see k_multi_count)
        unsigned long long* q = (unsigned long long*)(slices + blockIdx.x);
        __hip_atomic_store(q + 0, (unsigned long long)__double_as_longlong(s_sum[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, (unsigned long long)__double_as_longlong(s_aux[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 2, (unsigned long long)s_cnt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other workgroups' slices after it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!s_last) return;
    __shared__ long long s_count[MULTI_MAX_ITEMS];
    __shared__ float s_f[MULTI_MAX_ITEMS];
    const int ni = a.n_items, t = threadIdx.x;
    if (t < ni * MULTI_FIN_SPLIT) {                        // every slice by its own thread, then item by item
        unsigned long long* q = (unsigned long long*)(slices + t);
        s_sum[t] = __longlong_as_double((long long)__hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        s_aux[t] = __longlong_as_double((long long)__hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        s_cnt[t] = (long long)__hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (t < ni) {
        double sx = 0, sy = 0; long long sc = 0;
        for (int k = 0; k < MULTI_FIN_SPLIT; ++k) {
            sx += s_sum[t * MULTI_FIN_SPLIT + k]; sy += s_aux[t * MULTI_FIN_SPLIT + k]; sc += s_cnt[t * MULTI_FIN_SPLIT + k];
        }
        sums[t] = sx; counts[t] = sc;
        if (aux) aux[t] = sy;
        s_count[t] = sc;
        s_f[t] = (float)sx;
        if (out) {
            const long long cc = a.it[t].clamp ? max(sc, 1LL) : sc;
            out[t] = (float)sx;
            out[ni + t] = (float)sx / (float)cc;
        }
    }
    __syncthreads();
    if (t < a.n_totals) {
        bool known = !late_divisors;
        for (int i = 0; i < ni; ++i) if (a.it[i].total == t && a.it[i].count_mode == 2) known = false;
        float nf = expect[2 * t + 1];
        if (!known) { nf = multi_divisor(a, s_count, t); expect[2 * t + 1] = nf; }
        if (late_divisors) expect[2 * t] = __int_as_float(0x7fc00000);      // a call without gradients: no expectation
        if (out) {
            float acc = 0.f;
            for (int i = 0; i < ni; ++i) if (a.it[i].total == t) acc += s_f[i];
            out[2 * ni + t] = acc / nf;
        }
    }
    if (t == 0) *ticket = 0u;                              // zero again for the next call
}

}  // namespace nmsa

using namespace nmsa;

// ---------------------------------------------------------------------------------------------
// a10: all losses of a task helper in one call (see k_multi_loss)
namespace {

struct MultiPlan {
    MultiArgs args;
    int n_count_blocks;
    int ce_dt, ce_ng, ce_sm;                           // the cross-entropy variant inside the launch (ng 0: none)
    int max_c;
    size_t lds;
};

int multi_plan(const nmsa_loss_item* items, int n_items, int n_totals, MultiPlan& pl)
{
    if (!items || n_items <= 0 || n_items > MULTI_MAX_ITEMS || n_totals <= 0 || n_totals > MULTI_MAX_TOTALS)
        return NMSA_ERR_ARG;
    MultiArgs& a = pl.args;
    a.n_items = n_items; a.n_totals = n_totals; a.n_blocks = 0;
    pl.ce_ng = 0; pl.ce_dt = NMSA_F32; pl.ce_sm = 0; pl.max_c = 0;
    int block = 0, cblock = 0;
    bool seen_total[MULTI_MAX_TOTALS] = {false};
    for (int pass = 0; pass < 2; ++pass) {             // pass 0: the items of the joint launch, pass 1: the others
        for (int i = 0; i < n_items; ++i) {
            const nmsa_loss_item& s = items[i];
            MultiItem& it = a.it[i];
            if (pass == 0) {
                if (!s.pred || s.total < 0 || s.total >= n_totals || loss_bad_shape(s.B, s.H, s.W) || s.C <= 0)
                    return NMSA_ERR_ARG;
                if (s.dtype != NMSA_F32 && s.dtype != NMSA_BF16 && s.dtype != NMSA_F16) return NMSA_ERR_ARG;
                if (s.kind < NMSA_LOSS_CE || s.kind > NMSA_LOSS_COS_EMB) return NMSA_ERR_ARG;
                if (s.kind != NMSA_LOSS_CE && !s.target) return NMSA_ERR_ARG;
                if ((s.kind == NMSA_LOSS_CE || s.kind == NMSA_LOSS_COS_EMB) && !s.mask) return NMSA_ERR_ARG;
                if (s.kind == NMSA_LOSS_VONMISES && s.C != 2) return NMSA_ERR_ARG;
                if (s.clamp_count < 0 || s.clamp_count > 2) return NMSA_ERR_ARG;
                it.pred = s.pred; it.target = s.target; it.mask = (const uint8_t*)s.mask; it.weights = s.weights;
                it.grad = s.grad;
                it.kind = s.kind; it.dtype = s.dtype; it.B = s.B; it.C = s.C; it.P = s.H * s.W;
                it.total = s.total; it.clamp = s.clamp_count; it.param = s.param;
                it.L = s.reserved;
                it.first_of_total = !seen_total[s.total];
                seen_total[s.total] = true;
                const uintptr_t al = (uintptr_t)s.pred | (uintptr_t)s.grad | (uintptr_t)s.target;
                it.in_launch = 1;
                if (s.kind == NMSA_LOSS_CE) {
                    if (s.C > CE_SPLIT_MAX_C) return NMSA_ERR_UNSUPPORTED;
                    const int pxt = (s.dtype == NMSA_F32) ? 2 : 4;
                    it.vec = (it.P % pxt == 0) && ((((uintptr_t)s.pred | (uintptr_t)s.grad) & 7) == 0);
                    it.count_mode = 1; it.lo = 1; it.hi = s.C < 255 ? s.C : 255;
                    if (s.C > CE_FUSED_MAX_C) {
                        it.in_launch = 0;
                        it.nbx = ce_split_blocks(it.P, s.dtype);
                    } else {
                        const int ng = ce_fused_ng(s.C), sm = s.param != 0.0f;
                        if (pl.ce_ng == 0) { pl.ce_ng = ng; pl.ce_dt = s.dtype; pl.ce_sm = sm; }
                        if (ng != pl.ce_ng || s.dtype != pl.ce_dt || sm != pl.ce_sm) it.in_launch = 0;
                        it.nbx = loss_grid_x(it.P, pxt);
                        if (it.in_launch && s.C > pl.max_c) pl.max_c = s.C;
                    }
                } else if (s.kind == NMSA_LOSS_COS_EMB) {
                    // pred [B, D = C, H, W], target = LUT f32 [B, L = reserved, D], mask = indices i32
                    if (s.reserved <= 0 || s.clamp_count < 0 || s.clamp_count > 2) return NMSA_ERR_ARG;
                    if (!nmsa_loss_cos_emb_fwd_grad_supported(s.dtype, s.C, s.H, s.W, s.reserved))
                        return NMSA_ERR_UNSUPPORTED;
                    const int pxt = (s.dtype == NMSA_F32) ? 2 : 4;
                    if (it.P % pxt != 0 || ((((uintptr_t)s.pred | (uintptr_t)s.grad) & 7) != 0))
                        return NMSA_ERR_UNSUPPORTED;
                    it.vec = 1;
                    it.in_launch = 0;
                    it.nbx = cos_split_blocks(s.B, s.C, it.P, s.reserved, s.dtype);
                    it.count_mode = 3; it.lo = 1; it.hi = s.reserved;
                } else {
                    it.vec = (it.P % 4 == 0) && (((al | (uintptr_t)s.mask) & 15) == 0);
                    it.nbx = loss_grid_x(it.P, 8);
                    it.count_mode = s.kind == NMSA_LOSS_FOCAL ? 2 : (s.mask ? 1 : 0);
                    it.lo = 1; it.hi = 255;
                }
                if (it.count_mode == 1 || it.count_mode == 3) {
                    const long long n = (long long)it.B * it.P * (it.count_mode == 3 ? 4 : 1);    // bytes
                    long long cb = (n / 16 + LOSS_THREADS * 4 - 1) / (LOSS_THREADS * 4);
                    it.cnblocks = (int)(cb < 1 ? 1 : cb > MULTI_COUNT_MAX_BLOCKS ? MULTI_COUNT_MAX_BLOCKS : cb);
                } else {
                    it.cnblocks = 1;
                }
                it.cblock0 = cblock;
                cblock += it.cnblocks;
            }
            if ((pass == 0) == (it.in_launch != 0)) {
                it.block0 = block;
                block += it.nbx * it.B;
                if (pass == 0) a.n_blocks = block;      // blocks of the joint launch
            }
            if (block < 0 || block > (1 << 28)) return NMSA_ERR_ARG;
        }
    }
    pl.n_count_blocks = cblock;
    pl.lds = (size_t)(pl.max_c > 0 ? pl.max_c : 1) * sizeof(float);
    return NMSA_OK;
}

size_t multi_partial_blocks(const MultiPlan& pl)
{
    size_t n = 0;
    for (int i = 0; i < pl.args.n_items; ++i) n += (size_t)pl.args.it[i].nbx * pl.args.it[i].B;
    return n;
}

template <int MODE>
int multi_launch_joint(const MultiPlan& pl, const float* expect, const MultiBwd& bw, LossPartial* partials,
                       int* status, hipStream_t stream)
{
    const MultiArgs& a = pl.args;
    if (a.n_blocks <= 0) return NMSA_OK;
    // (the recomputing launch: a small grid that walks the block list, see k_multi_loss)
    const int grid = MODE != 2 ? a.n_blocks : (a.n_blocks < 4096 ? a.n_blocks : 4096);
#define ML(DT, NG, SM) hipLaunchKernelGGL((k_multi_loss<DT, NG, SM, MODE>), dim3(grid), dim3(LOSS_THREADS), \
        pl.lds, stream, a, expect, bw, partials, status)
#define ML_NG(DT, SM) do { if (pl.ce_ng == 3) ML(DT, 3, SM); else if (pl.ce_ng == 5) ML(DT, 5, SM); else ML(DT, 6, SM); } while (0)
#define ML_DT(DT) do { if (pl.ce_sm) ML_NG(DT, true); else ML_NG(DT, false); } while (0)
    if (pl.ce_ng == 0) ML(NMSA_F32, 0, false);
    else switch (pl.ce_dt) {
        case NMSA_F32: ML_DT(NMSA_F32); break;
        case NMSA_BF16: ML_DT(NMSA_BF16); break;
        default: ML_DT(NMSA_F16); break;
    }
#undef ML_DT
#undef ML_NG
#undef ML
    return check_launch();
}

}  // namespace

namespace {
// granule exchange buffer of the cosine items whose column spans several workgroups (k_cos_parts):
// their launches follow each other on the stream, so they share one buffer
size_t multi_xch_bytes(const MultiPlan& pl)
{
    size_t n = 0;
    for (int i = 0; i < pl.args.n_items; ++i) {
        const MultiItem& it = pl.args.it[i];
        if (it.kind != NMSA_LOSS_COS_EMB) continue;
        const size_t x = cos_split_xch_bytes(it.B, it.C, it.P, it.L, it.dtype);
        if (x > n) n = x;
    }
    return n;
}
size_t multi_main_bytes(const MultiPlan& pl)
{
    return ((multi_partial_blocks(pl) + (size_t)MULTI_MAX_ITEMS * MULTI_FIN_SPLIT) * sizeof(LossPartial) +
            (size_t)pl.n_count_blocks * sizeof(long long) + 64 + 63) & ~(size_t)63;
}
size_t multi_workspace_bytes(const MultiPlan& pl)
{
    return multi_main_bytes(pl) + multi_xch_bytes(pl);
}
}  // namespace

extern "C" size_t nmsa_multitask_loss_workspace_bytes(const nmsa_loss_item* items, int n_items)
{
    MultiPlan pl;
    int nt = 1;
    if (!items) return 0;
    for (int i = 0; i < n_items && i < MULTI_MAX_ITEMS; ++i) if (items[i].total + 1 > nt) nt = items[i].total + 1;
    if (nt > MULTI_MAX_TOTALS || multi_plan(items, n_items, nt, pl)) return 0;
    return multi_workspace_bytes(pl);
}

extern "C" int nmsa_multitask_loss_fwd_grad(const nmsa_loss_item* items, int n_items, int n_totals,
                                            int32_t* spec, float* expect, double* loss_sums,
                                            int64_t* counts, double* aux, float* out_f32, int32_t* status,
                                            void* workspace, size_t workspace_bytes, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!spec || !expect || !loss_sums || !counts || !status || !workspace) return NMSA_ERR_ARG;
    MultiPlan pl;
    int rc = multi_plan(items, n_items, n_totals, pl);
    if (rc) return rc;
    const size_t nb = multi_partial_blocks(pl);
    if (workspace_bytes < multi_workspace_bytes(pl)) return NMSA_ERR_WORKSPACE;
    LossPartial* partials = (LossPartial*)workspace;
    LossPartial* slices = partials + nb;
    long long* cpart = (long long*)(slices + (size_t)MULTI_MAX_ITEMS * MULTI_FIN_SPLIT);
    unsigned int* tickets = (unsigned int*)(spec + 8 * n_totals);     // [0] count, [1] finalize: zero between calls
    void* xch = (char*)workspace + multi_main_bytes(pl);
    const size_t xch_bytes = multi_xch_bytes(pl);
    const MultiArgs& a = pl.args;
    bool any_grad = false;
    for (int i = 0; i < n_items; ++i) any_grad = any_grad || a.it[i].grad != nullptr;
    // (forward only: nobody needs a count before the sums; k_multi_finalize then marks "no
    // expectation" and forms the divisors from the finalized counts)
    const MultiBwd none{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (any_grad) {
        hipLaunchKernelGGL(k_multi_count, dim3(pl.n_count_blocks), dim3(LOSS_THREADS), 0, stream, a, cpart,
                           (const int32_t*)spec, expect, tickets);
        rc = check_launch();
        if (rc) return rc;
    }
    rc = any_grad ? multi_launch_joint<0>(pl, expect, none, partials, status, stream)
                  : multi_launch_joint<1>(pl, expect, none, partials, status, stream);
    if (rc) return rc;
    for (int i = 0; i < n_items; ++i) {                 // cross entropies outside the joint launch
        const MultiItem& it = a.it[i];
        if (it.in_launch) continue;
        if (it.kind == NMSA_LOSS_COS_EMB) {
            rc = launch_cos_split(true, it.pred, it.dtype, (const int32_t*)it.mask, (const float*)it.target, it.B,
                                  it.C, it.P, it.L, expect + 2 * it.total, nullptr, nullptr, it.grad,
                                  partials + it.block0, status, xch, xch_bytes, stream);
        } else if (it.C > CE_FUSED_MAX_C) {
            rc = launch_ce_split(true, it.pred, it.dtype, it.mask, it.weights, it.B, it.C, it.P, it.param,
                                 expect + 2 * it.total, nullptr, nullptr, it.grad, partials + it.block0, status,
                                 stream);
        } else {
            // a second register-resident variant in one call: its own forward + gradient launch
            rc = loss_ce_fwd_grad_partials(it.pred, it.dtype, it.mask, it.weights, it.B, it.C, it.P,
                                                it.param, expect + 2 * it.total, it.grad, partials + it.block0,
                                                status, stream);
        }
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_multi_finalize, dim3(n_items * MULTI_FIN_SPLIT), dim3(MULTI_FIN_THREADS), 0, stream, a,
                       partials, slices, tickets + 1, any_grad ? 0 : 1, loss_sums, (long long*)counts, aux, expect,
                       out_f32);
    return check_launch();
}

extern "C" int nmsa_multitask_loss_bwd_unless(const nmsa_loss_item* items, int n_items, int n_totals,
                                              const float* grad_sums, const float* grad_item_losses,
                                              const float* grad_total_losses, const int64_t* counts,
                                              const float* expect, int32_t* spec, float* grad_scales,
                                              int32_t* counters, void* workspace, size_t workspace_bytes,
                                              nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!counts || !grad_scales || !expect) return NMSA_ERR_ARG;
    MultiPlan pl;
    int rc = multi_plan(items, n_items, n_totals, pl);
    if (rc) return rc;
    // the forward call's workspace (its contents are dead by now): only cosine items whose column
    // spans several workgroups use it, as their granule exchange buffer
    if (multi_xch_bytes(pl) > 0 && (!workspace || workspace_bytes < multi_xch_bytes(pl))) return NMSA_ERR_WORKSPACE;
    const MultiArgs& a = pl.args;
    const MultiBwd bw{grad_sums, grad_item_losses, grad_total_losses, (const long long*)counts, spec, counters,
                      grad_scales};
    if (a.n_blocks > 0) {
        rc = multi_launch_joint<2>(pl, expect, bw, nullptr, nullptr, stream);
    } else {
        hipLaunchKernelGGL(k_multi_spec, dim3(1), dim3(64), 0, stream, a, bw, expect);
        rc = check_launch();
    }
    if (rc) return rc;
    for (int i = 0; i < n_items; ++i) {
        const MultiItem& it = a.it[i];
        if (it.in_launch || !it.grad) continue;
        if (it.kind == NMSA_LOSS_COS_EMB) {
            rc = launch_cos_split(false, it.pred, it.dtype, (const int32_t*)it.mask, (const float*)it.target, it.B,
                                  it.C, it.P, it.L, grad_scales + i, expect + 2 * it.total, nullptr, it.grad,
                                  nullptr, nullptr, workspace, workspace_bytes, stream);
            if (rc) return rc;
            continue;
        }
        rc = nmsa_loss_ce_bwd_unless(it.pred, it.dtype, it.mask, it.weights, it.B, it.C, 1, it.P, it.param,
                                     grad_scales + i, it.grad, expect + 2 * it.total, nullptr, stream_);
        if (rc) return rc;
    }
    return NMSA_OK;
}
