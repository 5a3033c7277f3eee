// losses_tile.hip — forward + gradient of the two "whole column per pixel" losses in ONE pass
// over HBM, for gfx950:
//   CosineEmbeddingLoss._compute_loss   loss/cos_emb.py:21-56 (+ the LUT gather of
//                                       task_helper/dense_visual_embedding.py:110-171)
//   CrossEntropyLossSemantic._compute_loss  loss/ce.py:40-68 for class counts whose column does
//                                       not fit the registers of k_ce_fused (C > 48)
//
// Both gradients need a reduction over the pixel's whole column (x.y and |x|^2 over D; max and
// sum-of-exp over C) before the first gradient element can be written.  Walking the column twice
// through the caches moved 1.5x the algorithmic bytes (round 2: the second walk of a 614 KB
// workgroup tile leaves the L2).  Here the column stays ON CHIP between the reduction and the
// gradient — in the REGISTER FILE, the largest on-chip memory of a CU (512 KB against 160 KB LDS):
//
//   a tile = 64 pixels (128-byte row segments; 64-byte segments fetch every line twice) x all R
//          rows (R = D or C).  A wave holds KS "steps" of it, a step being one 16-byte load per
//          lane = 8 rows x 8 pieces: KS x 4 VGPRs per lane, packed as loaded (KS <= 32: 32 KB per
//          wave).  Columns beyond one wave's registers (D > 256) are shared by the 2-4 waves of a
//          SMALL workgroup;
//   pass 1 reduces the wave's rows in registers; the 8 lanes that hold rows of the same pixels
//          combine with `v_permlane32_swap` / `v_permlane16_swap` / DPP `row_ror:8` (a fixed
//          tree, no LDS traffic); the waves of the workgroup (if more than one) combine through
//          1-2 KB of LDS, wave 0 sums the partials per pixel in a FIXED order and publishes the
//          gradient coefficients;
//   pass 2 forms the gradient from the SAME registers and writes it once (16-byte non-temporal
//          stores; edge lanes / rows duplicate a valid lane's work and store identical bytes).
//
// No workgroup is larger than 4 waves, 8-12 waves from 3-12 independent workgroups share a CU:
// while one waits for its 32 KB per wave, the others compute or store — the overlap that wide
// workgroups marching through barriers in step did not give (an LDS-resident tile
// shared by 8-16 waves, single- or double-buffered, ran at the SUM of its memory and compute
// phases).  HBM traffic = prediction once + gradient once (+ labels / indices).
// The gradient is written for an EXPECTED upstream scale; the same kernel with LOSS = false is
// the confirming / recomputing backward launch (returns at once when the real upstream gradient
// is bit-equal to the expectation, see loss_common.hpp).
#include <stdlib.h>
#include "loss_common.hpp"

namespace nmsa {

template <int DTYPE, int PPL>
__device__ __forceinline__ void unpack_piece(const u32x4_s v, float x[PPL])
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    if constexpr (DTYPE == NMSA_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = __uint_as_float(w[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[2 * i] = unpack16<DTYPE>(w[i], 0); x[2 * i + 1] = unpack16<DTYPE>(w[i], 1); }
    }
}

template <int DTYPE, int PPL>
__device__ __forceinline__ u32x4_s pack_piece(const float o[PPL])
{
    u32x4_s r;
    if constexpr (DTYPE == NMSA_F32) {
        r.x = __float_as_uint(o[0]); r.y = __float_as_uint(o[1]);
        r.z = __float_as_uint(o[2]); r.w = __float_as_uint(o[3]);
    } else {
        r.x = pack16<DTYPE>(o[0], o[1]); r.y = pack16<DTYPE>(o[2], o[3]);
        r.z = pack16<DTYPE>(o[4], o[5]); r.w = pack16<DTYPE>(o[6], o[7]);
    }
    return r;
}

template <int N> struct int_c { static constexpr int value = N; };

// ---- combining the lanes of a pixel group (lanes l, l + LPR, l + 2 LPR, ...) ----------------------
// One swap + one add reduces TWO values over one lane bit: after `v_permlane32_swap a, b` the pair
// is ([a.lo, b.lo], [a.hi, b.hi]); their sum holds a (reduced over lane bit 5) in lanes 0-31 and
// b in lanes 32-63.  `v_permlane16_swap` does the same for lane bit 4 (rows of 16 lanes), DPP
// row_ror:8 + add for lane bit 3.  A lane ends up with N / 4 (LPR = 8, 16) or N / 2 (LPR = 32) of
// the N values; lane_value_id() says which.
struct OpAdd { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct OpMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };

template <int N, typename Op>
__device__ __forceinline__ void fold32(const float (&v)[N], float (&o)[N / 2], Op op)
{
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * i]), __float_as_uint(v[2 * i + 1]), false, false);
        o[i] = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
}
template <int N, typename Op>
__device__ __forceinline__ void fold16(const float (&v)[N], float (&o)[N / 2], Op op)
{
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[2 * i]), __float_as_uint(v[2 * i + 1]), false, false);
        o[i] = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
}
template <int N, typename Op>
__device__ __forceinline__ void fold8(float (&v)[N], Op op)
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float t = __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(
            0, (int)__float_as_uint(v[i]), 0x128 /* row_ror:8 */, 0xF, 0xF, false));
        v[i] = op(v[i], t);
    }
}
// the inverse of fold32 / fold16 for an ALL-reduce: every lane gets both values of the pair back
template <int N>
__device__ __forceinline__ void unfold32(const float (&o)[N / 2], float (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(o[i]), __float_as_uint(o[i]), false, false);
        v[2 * i] = __uint_as_float(r[0]); v[2 * i + 1] = __uint_as_float(r[1]);
    }
}
template <int N>
__device__ __forceinline__ void unfold16(const float (&o)[N / 2], float (&v)[N])
{
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(o[i]), __float_as_uint(o[i]), false, false);
        v[2 * i] = __uint_as_float(r[0]); v[2 * i + 1] = __uint_as_float(r[1]);
    }
}

// number of registers left per lane after group_reduce, and the index of the value register i holds
template <int LPR, int N> struct Reduced { static constexpr int n = (LPR == 64) ? N : (LPR == 32) ? N / 2 : N / 4; };
template <int LPR>
__device__ __forceinline__ int lane_value_id(int i, int lane)
{
    if (LPR == 64) return i;
    const int h = lane >> 5;
    if (LPR == 32) return 2 * i + h;
    return 4 * i + 2 * ((lane >> 4) & 1) + h;
}
template <int LPR, int N, typename Op>
__device__ __forceinline__ void group_reduce(const float (&v)[N], float (&o)[Reduced<LPR, N>::n], Op op)
{
    if constexpr (LPR == 64) {
#pragma unroll
        for (int i = 0; i < N; ++i) o[i] = v[i];
    } else if constexpr (LPR == 32) {
        fold32(v, o, op);
    } else {
        float a[N / 2];
        fold32(v, a, op);
        fold16(a, o, op);
        if constexpr (LPR == 8) fold8(o, op);
    }
}
// every lane gets the reduction of all N values over its pixel group (used for the maximum)
template <int LPR, int N, typename Op>
__device__ __forceinline__ void group_allreduce(float (&v)[N], Op op)
{
    if constexpr (LPR == 32) {
        float a[N / 2];
        fold32(v, a, op);
        unfold32(a, v);
    } else if constexpr (LPR < 32) {
        float a[N / 2], b[N / 4];
        fold32(v, a, op);
        fold16(a, b, op);
        if constexpr (LPR == 8) fold8(b, op);
        unfold16(b, a);
        unfold32(a, v);
    }
}

// block sum of (acc, aux, cnt) over the waves -> partials[block] (fixed order)
__device__ __forceinline__ void tile_block_partial(double acc, double aux, long long cnt, double* s_red,
                                                   int nw, LossPartial* __restrict__ partials)
{
    acc = wave_reduce_sum(acc);
    aux = wave_reduce_sum(aux);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();                                        // s_red aliases the per-tile scratch
    if (lane_id() == 0) { s_red[3 * w] = acc; s_red[3 * w + 1] = aux; s_red[3 * w + 2] = (double)cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, x = 0, c = 0;
        for (int k = 0; k < nw; ++k) { a += s_red[3 * k]; x += s_red[3 * k + 1]; c += s_red[3 * k + 2]; }
        LossPartial p; p.sum = a; p.aux = x; p.count = (long long)c; p.pad = 0;
        partials[blockIdx.y * gridDim.x + blockIdx.x] = p;
    }
}

// |y|^2 of every LUT row, once per call: one wave per row (rows are shared by whole segments,
// every pixel tile would otherwise redo them)
__global__ __launch_bounds__(256) void k_lut_norms(const float* __restrict__ lut, int rows, int D,
                                                   float* __restrict__ yy)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* y = lut + (size_t)r * D;
    float s = 0.f;
    for (int d = lane_id(); d < D; d += 64) { const float v = y[d]; s = fmaf(v, v, s); }
    s = wave_reduce_sum(s);
    if (lane_id() == 0) yy[r] = s;
}

// LUT rows of every pixel GROUP (the PPL pixels of one 16-byte piece), once per call: the first
// valid row `ra`, the first other row `rb`, which pixels take `ra` (bit j of mask_a; pixels
// without a target count as `ra`), which have a target (`on`), and how many LUT values the group
// needs per prediction row: 1 (one row), 2 (a segment boundary inside the group) or a gather per
// pixel (3).  The tile kernel then spends one 16-byte load and a few scalar tests per tile where
// it would decode PPL indices per lane in each of its 16 waves.
template <int PPL>
__global__ __launch_bounds__(256) void k_cos_rows(const int32_t* __restrict__ indices, long long n_groups,
                                                   int L, int4* __restrict__ rows)
{
    constexpr unsigned FULL = (1u << PPL) - 1u;
    const long long gidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gidx >= n_groups) return;
    const int4* ip = (const int4*)(indices + gidx * PPL);
    int ix[PPL];
#pragma unroll
    for (int k = 0; k < PPL / 4; ++k) {
        const int4 v = ip[k];
        ix[4 * k] = v.x; ix[4 * k + 1] = v.y; ix[4 * k + 2] = v.z; ix[4 * k + 3] = v.w;
    }
    int row[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) row[j] = (ix[j] > 0 && ix[j] <= L) ? ix[j] - 1 : -1;
    int ra = -1, rb = -1;
#pragma unroll
    for (int j = PPL - 1; j >= 0; --j) ra = (row[j] >= 0) ? row[j] : ra;              // first valid row
    unsigned on = 0, mask_a = 0;
#pragma unroll
    for (int j = 0; j < PPL; ++j) {
        on |= (row[j] >= 0) ? (1u << j) : 0u;
        mask_a |= (row[j] < 0 || row[j] == ra) ? (1u << j) : 0u;
    }
#pragma unroll
    for (int j = PPL - 1; j >= 0; --j) rb = !((mask_a >> j) & 1u) ? row[j] : rb;      // first other row
    bool two_ok = true;
#pragma unroll
    for (int j = 0; j < PPL; ++j) two_ok = two_ok && (((mask_a >> j) & 1u) || row[j] == rb);
    const int need = (mask_a == FULL) ? 1 : (two_ok ? 2 : 3);
    ra = max(ra, 0);
    rb = (rb < 0) ? ra : rb;
    rows[gidx] = make_int4(ra, rb, (int)(mask_a | (on << 8) | ((unsigned)need << 16)), 0);
}

// =================================================================================
// a9 fused: cosine embedding, planar prediction [B,D,P], per-image LUT [L,D] read through
// L1 / L2: a segment's row is shared by its pixels, so lanes whose 8 pixels agree on the row
// (MODE 1) need ONE LUT value per prediction row, lanes on a segment boundary two + a select
// (MODE 2); a lane owns KS consecutive rows, so these are wide loads of consecutive floats.
// Tiles with a lane holding three or more rows (MODE 3, rare) take a plain two-walk path.
//   per valid px (index != 0):  1 - x.y / sqrt((|x|^2 + eps)(|y|^2 + eps)), eps = 1e-12
//   d/dx = -y/den + (x.y) x / ((|x|^2 + eps) den)
// =================================================================================
constexpr int COL_LPR = 8;                             // lanes per row segment: 8 x 16 B = 128 B

template <int DTYPE, int KS, bool LOSS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((KS <= 16) ? 4 : 2, (KS <= 16) ? 4 : 2))) void k_cos_col(
    const void* __restrict__ pred, const int32_t* __restrict__ indices,
    const float* __restrict__ lut, const float* __restrict__ yy, const int4* __restrict__ rows,
    int D, int P, int L, int tiles_per_wg,
    const float* __restrict__ gscale, const float* __restrict__ computed_for,
    int* __restrict__ counters, void* __restrict__ grad, LossPartial* __restrict__ partials,
    int* __restrict__ status, int ablate)
{
    constexpr int ES = (DTYPE == NMSA_F32) ? 4 : 2;
    constexpr int PPL = 16 / ES;                       // pixels per lane (one 16-byte piece)
    constexpr int LPR = COL_LPR;
    constexpr int RPI = 64 / LPR;                      // rows per wave-instruction: 8
    constexpr int TP = LPR * PPL;                      // pixels per tile: 64 (32 for f32)
    constexpr unsigned FULL = (1u << PPL) - 1u;
    constexpr int NR = Reduced<LPR, 2 * PPL>::n;
    __shared__ float s_part[4 * TP * 2];               // [nw][TP][2]: x.y, |x|^2
    __shared__ float s_k[TP * 2];                      // [TP][2]: k1, k2
    __shared__ double s_red[4 * 3];
    if (!LOSS && grad_already_computed(gscale, computed_for, counters)) return;
    const int nw = blockDim.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = lane_id();
    const int q = l % LPR, rsub = l / LPR;
    const int b = blockIdx.y;
    const float g = *gscale;
    // uniform (scalar) bases + 32-bit lane offsets (the host checks that an image and a LUT stay
    // below 4 GB)
    const unsigned char* pred_b = (const unsigned char*)pred + (size_t)b * D * P * ES;
    unsigned char* grad_b = (unsigned char*)grad + (size_t)b * D * P * ES;
    const float* lut_b = lut + (size_t)b * L * D;
    const float* yy_b = yy + (size_t)b * L;
    const int32_t* idx_b = indices + (size_t)b * P;
    const int4* rows_b = rows + (size_t)b * (P / PPL);
    const int NT = (P + TP - 1) / TP;
    const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(NT, t_begin + tiles_per_wg);
    const int tr = threadIdx.x % TP;                   // the pixel this thread combines (threads < TP)
    const bool reducer = threadIdx.x < TP;
    const float EPS = 1e-12f;
    double acc = 0.0;
    long long cnt = 0;
    bool bad = false;
    // my rows: a wave owns KS * 8 consecutive prediction rows, lane group rsub the KS consecutive
    // ones from d0 on; step k covers row d0 + k of every lane group.  Rows >= D (the last wave) are
    // clamped: loaded again, excluded from the sums, their gradient stored a second time with
    // identical bytes.
    const int d0 = (w * RPI + rsub) * KS;
    const bool all_rows = (w + 1) * RPI * KS <= D;     // wave-uniform
    const bool y_vec = all_rows && (D % 4 == 0);       // LUT values as 16-byte loads
    auto drow = [&](int k) { return min(d0 + k, D - 1); };
    auto row_off = [&](int k) { return (uint32_t)drow(k) * (uint32_t)P * ES; };
    // lanes beyond the image in its last tile act as the tile's first pixel group: same loads,
    // same coefficients, same (duplicate) stores; their sums land on pixels >= P, which nobody reads
    auto q_eff = [&](int tile) { return (tile * TP + q * PPL < P) ? q : 0; };

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int p0 = tile * TP;
        const int qe = q_eff(tile);
        const uint32_t px = (uint32_t)(p0 + qe * PPL) * ES;
        // ---- the tile's requests: my KS pieces, the row-table entry, the reducer's index --------
        u32x4_s cur[KS];
        if (!(ablate & 4)) {
#pragma unroll
            for (int k = 0; k < KS; ++k)
                cur[k] = __builtin_nontemporal_load((const u32x4_s*)(pred_b + (size_t)(row_off(k) + px)));
        } else {
#pragma unroll
            for (int k = 0; k < KS; ++k) cur[k] = u32x4_s{0u, 0u, 0u, 0u};
        }
        const int4 r_cur = rows_b[(size_t)tile * LPR + qe];
        const int ixr = idx_b[min(p0 + tr, P - 1)];
        const bool on_r = ixr > 0 && ixr <= L;
        const float yyr = reducer ? yy_b[on_r ? ixr - 1 : 0] : 1.0f;
        const unsigned mask_a = (unsigned)r_cur.z & 0xFFu, on = ((unsigned)r_cur.z >> 8) & 0xFFu;
        const int need = r_cur.z >> 16;
        // wave-uniform: 1 = one LUT row per lane and every pixel has a target, 2 = two rows or
        // pixels without a target, 3 = a lane with three or more rows
        int mode = __any(need == 3) ? 3 : (__any(need == 2 || on != FULL) ? 2 : 1);
        if (nw > 1) {                                  // the waves of a workgroup must agree
            if (l == 0) s_k[w] = __int_as_float(mode);
            __syncthreads();
            int mm = 1;
            for (int ww = 0; ww < nw; ++ww) mm = max(mm, __float_as_int(s_k[ww]));
            mode = mm;
            __syncthreads();
        }
        const uint32_t ra_off = (uint32_t)r_cur.x * (uint32_t)D, rb_off = (uint32_t)r_cur.y * (uint32_t)D;
        // LUT values of step k .. k + 3 (consecutive rows of this lane)
        auto load_y4 = [&](uint32_t r_off, int k, float y4[4]) {
            if (y_vec) {
                const float4 v = *(const float4*)(lut_b + (size_t)(r_off + (uint32_t)(d0 + k)));
                y4[0] = v.x; y4[1] = v.y; y4[2] = v.z; y4[3] = v.w;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) y4[i] = lut_b[(size_t)(r_off + (uint32_t)drow(k + i))];
            }
        };

        float k1[PPL], k2[PPL];
        if (mode != 3) {
            // ---- pass 1: x.y and |x|^2 over my rows (registers only) ----------------------------
            float val[2 * PPL];                            // [2 j] = x.y, [2 j + 1] = |x|^2 of pixel j
#pragma unroll
            for (int j = 0; j < 2 * PPL; ++j) val[j] = 0.f;
            auto pass1 = [&](auto mode_c, auto tail_c) {
                constexpr int MODE = decltype(mode_c)::value;
                constexpr bool TAIL = decltype(tail_c)::value != 0;
#pragma unroll
                for (int k4 = 0; k4 < KS; k4 += 4) {
                    float ya[4], yb[4];
                    load_y4(ra_off, k4, ya);
                    if constexpr (MODE == 2) load_y4(rb_off, k4, yb);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = k4 + i;
                        float x[PPL];
                        unpack_piece<DTYPE, PPL>(cur[k], x);
#pragma unroll
                        for (int j = 0; j < PPL; ++j) {
                            const float yj = (MODE == 1 || ((mask_a >> j) & 1u)) ? ya[i] : yb[i];
                            const float xv = (!TAIL || d0 + k < D) ? x[j] : 0.f;
                            val[2 * j] = fmaf(xv, yj, val[2 * j]);
                            val[2 * j + 1] = fmaf(xv, xv, val[2 * j + 1]);
                        }
                    }
                }
            };
            if (!(ablate & 8)) {
                if (all_rows) { if (mode == 1) pass1(int_c<1>{}, int_c<0>{}); else pass1(int_c<2>{}, int_c<0>{}); }
                else { if (mode == 1) pass1(int_c<1>{}, int_c<1>{}); else pass1(int_c<2>{}, int_c<1>{}); }
            }
            // pass 2 unpacks the registers again (left alone the compiler keeps the unpacked fp32
            // values of pass 1 alive: PPL more registers per step)
#pragma unroll
            for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(cur[k].x), "+v"(cur[k].y), "+v"(cur[k].z), "+v"(cur[k].w));
            float red[NR];
            group_reduce<LPR>(val, red, OpAdd{});
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int v = lane_value_id<LPR>(i, l);
                s_part[((size_t)w * TP + q * PPL + (v >> 1)) * 2 + (v & 1)] = red[i];
            }
        } else {
            // ---- rare tiles: plain walk over my rows with a LUT gather per pixel ---------------------
            float xy[PPL], xx[PPL];
#pragma unroll
            for (int j = 0; j < PPL; ++j) { xy[j] = 0.f; xx[j] = 0.f; }
            const int4* ip = (const int4*)(idx_b + p0 + qe * PPL);
            int ixj[PPL];
#pragma unroll
            for (int kq = 0; kq < PPL / 4; ++kq) { const int4 iv = ip[kq]; ixj[4 * kq] = iv.x; ixj[4 * kq + 1] = iv.y; ixj[4 * kq + 2] = iv.z; ixj[4 * kq + 3] = iv.w; }
            for (int k = 0; k < KS; ++k) {
                if (d0 + k >= D) break;
                const u32x4_s v = *(const u32x4_s*)(pred_b + (size_t)(row_off(k) + px));
                float x[PPL];
                unpack_piece<DTYPE, PPL>(v, x);
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    const float yj = ((on >> j) & 1u) ? lut_b[(size_t)((uint32_t)(ixj[j] - 1) * (uint32_t)D + (uint32_t)(d0 + k))] : 0.f;
                    xy[j] = fmaf(x[j], yj, xy[j]);
                    xx[j] = fmaf(x[j], x[j], xx[j]);
                }
            }
            float val[2 * PPL], red[NR];
#pragma unroll
            for (int j = 0; j < PPL; ++j) { val[2 * j] = xy[j]; val[2 * j + 1] = xx[j]; }
            group_reduce<LPR>(val, red, OpAdd{});
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int v = lane_value_id<LPR>(i, l);
                s_part[((size_t)w * TP + q * PPL + (v >> 1)) * 2 + (v & 1)] = red[i];
            }
        }
        __syncthreads();
        // ---- one thread per pixel: combine the waves, loss term, gradient coefficients ------
        if (reducer) {
            float X = 0.f, XX = 0.f;
            for (int ww = 0; ww < nw; ++ww) {
                const float2 pr = *(const float2*)(s_part + ((size_t)ww * TP + tr) * 2);
                X += pr.x; XX += pr.y;
            }
            float c1 = 0.f, c2 = 0.f;
            if (p0 + tr < P) {
                if (ixr < 0 || ixr > L) bad = true;
                if (on_r) {
                    const float den = sqrtf((XX + EPS) * (yyr + EPS));
                    if (LOSS) { acc += 1.0f - X / den; ++cnt; }
                    c1 = -g / den;
                    c2 = g * X / ((XX + EPS) * den);
                }
            }
            *(float2*)(s_k + 2 * tr) = make_float2(c1, c2);
        }
        __syncthreads();
        {
            const float* src = s_k + qe * PPL * 2;
#pragma unroll
            for (int j = 0; j < PPL; ++j) { k1[j] = src[2 * j]; k2[j] = src[2 * j + 1]; }
        }
        // ---- pass 2: gradient from the same registers, written once -----------------------
        if (mode != 3) {
            auto pass2 = [&](auto mode_c) {
                constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
                for (int k4 = 0; k4 < KS; k4 += 4) {
                    float ya[4], yb[4];
                    load_y4(ra_off, k4, ya);
                    if constexpr (MODE == 2) load_y4(rb_off, k4, yb);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = k4 + i;
                        float x[PPL], o[PPL];
                        unpack_piece<DTYPE, PPL>(cur[k], x);
#pragma unroll
                        for (int j = 0; j < PPL; ++j) {
                            const float yj = (MODE == 1 || ((mask_a >> j) & 1u)) ? ya[i] : yb[i];
                            const float r = fmaf(k2[j], x[j], k1[j] * yj);
                            o[j] = (MODE == 1 || ((on >> j) & 1u)) ? r : 0.f;   // MODE 1: every pixel has a target
                        }
                        __builtin_nontemporal_store(pack_piece<DTYPE, PPL>(o), (u32x4_s*)(grad_b + (size_t)(row_off(k) + px)));
                    }
                }
            };
            if (!(ablate & 2)) { if (mode == 1) pass2(int_c<1>{}); else pass2(int_c<2>{}); }
        } else {
            const int4* ip = (const int4*)(idx_b + p0 + qe * PPL);
            int ixj[PPL];
#pragma unroll
            for (int kq = 0; kq < PPL / 4; ++kq) { const int4 iv = ip[kq]; ixj[4 * kq] = iv.x; ixj[4 * kq + 1] = iv.y; ixj[4 * kq + 2] = iv.z; ixj[4 * kq + 3] = iv.w; }
            for (int k = 0; k < KS; ++k) {
                if (d0 + k >= D) break;
                const u32x4_s v = *(const u32x4_s*)(pred_b + (size_t)(row_off(k) + px));
                float x[PPL], o[PPL];
                unpack_piece<DTYPE, PPL>(v, x);
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    const float yj = ((on >> j) & 1u) ? lut_b[(size_t)((uint32_t)(ixj[j] - 1) * (uint32_t)D + (uint32_t)(d0 + k))] : 0.f;
                    o[j] = ((on >> j) & 1u) ? fmaf(k2[j], x[j], k1[j] * yj) : 0.f;
                }
                __builtin_nontemporal_store(pack_piece<DTYPE, PPL>(o), (u32x4_s*)(grad_b + (size_t)(row_off(k) + px)));
            }
        }
    }
    if (LOSS) {
        if (bad) atomicOr(status, 8);
        tile_block_partial(acc, 0.0, cnt, s_red, nw, partials);
    }
}

// =================================================================================
// a6 fused for wide class columns: weighted, label-smoothed cross entropy
//   per px (t = label-1 >= 0):  (1-ls)*w_t*(lse - x_t) + (ls/C)*(lse*W - sum_c w_c x_c)
//   grad_c = g * [ (a + b W) p_c - a [c == t] - b_c ],  a = (1-ls) w_t, b_c = (ls/C) w_c
// ONE wave holds the whole column of its 64 pixels (C <= 256: at most 32 steps of 8 classes), so
// the kernel has no workgroup barrier at all: maximum and sum-of-exp2 are all-reduced over the 8
// lanes of a pixel group (swap tree), every lane then knows its pixels' log-sum-exp and writes
// its share of the gradient.  Step k covers classes 8 k + rsub.
// =================================================================================
template <int DTYPE, int KS, bool SMOOTH, bool LOSS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((KS <= 19) ? 4 : 2, (KS <= 19) ? 4 : 2))) void k_ce_col(
    const void* __restrict__ logits, const uint8_t* __restrict__ target,
    const float* __restrict__ weights, int C, int P, float ls, int tiles_per_wave,
    const float* __restrict__ gscale, const float* __restrict__ computed_for,
    int* __restrict__ counters, void* __restrict__ grad, LossPartial* __restrict__ partials,
    int* __restrict__ status, int lab_vec)
{
    constexpr int ES = (DTYPE == NMSA_F32) ? 4 : 2;
    constexpr int PPL = 16 / ES;
    constexpr int LPR = COL_LPR;
    constexpr int RPI = 64 / LPR;
    constexpr int TP = LPR * PPL;
    extern __shared__ float s_w[];                     // [C] class weights
    __shared__ double s_red[4 * 3];
    if (!LOSS && grad_already_computed(gscale, computed_for, counters)) return;
    const int nw = blockDim.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l = lane_id();
    const int q = l % LPR, rsub = l / LPR;
    const int b = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += blockDim.x) s_w[c] = weights ? weights[c] : 1.0f;
    __syncthreads();
    float wsum = 0.f;
    if (SMOOTH) for (int c = 0; c < C; ++c) wsum += s_w[c];
    const float g = *gscale;
    const unsigned char* log_b = (const unsigned char*)logits + (size_t)b * C * P * ES;
    unsigned char* grad_b = (unsigned char*)grad + (size_t)b * C * P * ES;
    const uint8_t* tgt_b = target + (size_t)b * P;
    const int NT = (P + TP - 1) / TP;
    // every WAVE walks its own run of tiles
    const int t_begin = (blockIdx.x * nw + w) * tiles_per_wave, t_end = min(NT, t_begin + tiles_per_wave);
    double acc = 0.0, accw = 0.0;
    long long cnt = 0;
    bool bad = false;
    const bool all_rows = KS * RPI <= C;               // no clamped class in any step
    auto crow = [&](int k) { return min(k * RPI + rsub, C - 1); };
    auto row_off = [&](int k) { return (uint32_t)crow(k) * (uint32_t)P * ES; };

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int p0 = tile * TP;
        const int qe = (p0 + q * PPL < P) ? q : 0;     // lanes beyond the image duplicate pixel group 0
        const uint32_t px = (uint32_t)(p0 + qe * PPL) * ES;
        u32x4_s cur[KS];
#pragma unroll
        for (int k = 0; k < KS; ++k)
            cur[k] = __builtin_nontemporal_load((const u32x4_s*)(log_b + (size_t)(row_off(k) + px)));
        int lab[PPL];                                  // labels of my pixels - 1 (ce.py:46)
        {
            const uint8_t* lp = tgt_b + p0 + qe * PPL;
            if (lab_vec) {                              // label map aligned to the lane's PPL bytes
                uint32_t v[2] = {0u, 0u};
                if constexpr (PPL == 8) { const uint2 u = *(const uint2*)lp; v[0] = u.x; v[1] = u.y; }
                else v[0] = *(const uint32_t*)lp;
#pragma unroll
                for (int j = 0; j < PPL; ++j) lab[j] = (int)((v[j >> 2] >> (8 * (j & 3))) & 0xFF) - 1;
            } else {
#pragma unroll
                for (int j = 0; j < PPL; ++j) lab[j] = (int)lp[j] - 1;
            }
        }
        // ---- pass 1a: column maximum -----------------------------------------------------------
        float m[PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) m[j] = -INFINITY;
        auto pass1a = [&](auto tail_c) {
            constexpr bool TAIL = decltype(tail_c)::value != 0;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                float x[PPL];
                unpack_piece<DTYPE, PPL>(cur[k], x);
                const bool rv = !TAIL || k * RPI + rsub < C;
#pragma unroll
                for (int j = 0; j < PPL; ++j) m[j] = fmaxf(m[j], rv ? x[j] : -INFINITY);
            }
        };
        if (all_rows) pass1a(int_c<0>{}); else pass1a(int_c<1>{});
        group_allreduce<LPR>(m, OpMax{});
        // ---- pass 1b: sum of exp2, target logit, (smoothing: sum_c w_c x_c) ---------------------
        constexpr int NVS = SMOOTH ? 3 : 2;
        float kk[PPL], val[NVS * PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) kk[j] = -m[j] * LOG2E;
#pragma unroll
        for (int j = 0; j < NVS * PPL; ++j) val[j] = 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(cur[k].x), "+v"(cur[k].y), "+v"(cur[k].z), "+v"(cur[k].w));
        auto pass1b = [&](auto tail_c) {
            constexpr bool TAIL = decltype(tail_c)::value != 0;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                float x[PPL];
                unpack_piece<DTYPE, PPL>(cur[k], x);
                const bool rv = !TAIL || k * RPI + rsub < C;
                const int c = rv ? k * RPI + rsub : -2;
                const float wc = SMOOTH ? s_w[crow(k)] : 0.f;
#pragma unroll
                for (int j = 0; j < PPL; ++j) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(x[j], LOG2E, kk[j]));
                    val[NVS * j] += rv ? e : 0.f;
                    val[NVS * j + 1] += (lab[j] == c) ? x[j] : 0.f;          // exactly one lane and step holds the target
                    if (SMOOTH) val[NVS * j + 2] = fmaf(wc, rv ? x[j] : 0.f, val[NVS * j + 2]);
                }
            }
        };
        if (all_rows) pass1b(int_c<0>{}); else pass1b(int_c<1>{});
#pragma unroll
        for (int k = 0; k < KS; ++k) asm volatile("" : "+v"(cur[k].x), "+v"(cur[k].y), "+v"(cur[k].z), "+v"(cur[k].w));
        group_allreduce<LPR>(val, OpAdd{});
        // ---- per pixel (every lane of the group computes the same): log-sum-exp, coefficients -----
        float kq[PPL], abg[PPL], ag[PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const float S = val[NVS * j], XT = val[NVS * j + 1];
            kq[j] = -(fmaf(m[j], LOG2E, __log2f(S)));                                // p = 2^(x log2e + kq)
            abg[j] = 0.f; ag[j] = 0.f;
            const bool px_valid = p0 + q * PPL + j < P;
            if (px_valid && lab[j] >= C) { bad = true; lab[j] = -1; }
            if (!px_valid) lab[j] = (q * PPL == qe * PPL) ? lab[j] : lab[j];        // duplicates keep pixel group 0's labels
            if (lab[j] >= 0 && lab[j] < C) {
                const float wt = s_w[lab[j]];
                const float a = (1.0f - ls) * wt;
                ag[j] = g * a;
                abg[j] = g * (a + (SMOOTH ? (ls / C) * wsum : 0.f));
                if (LOSS && rsub == 0 && px_valid) {                                // one lane per pixel adds it
                    const float lse = fmaf(__log2f(S), LN2, m[j]);
                    float lo = (1.0f - ls) * wt * (lse - XT);
                    if (SMOOTH) lo += (ls / C) * (lse * wsum - val[NVS * j + 2]);
                    acc += lo; accw += wt; ++cnt;
                }
            } else if (lab[j] >= C) {
                lab[j] = -1;
            }
        }
        // ---- pass 2: gradient -----------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            float x[PPL], o[PPL];
            unpack_piece<DTYPE, PPL>(cur[k], x);
            const int c = crow(k);
            const float bjg = SMOOTH ? g * (ls / C) * s_w[c] : 0.f;
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                const float pj = __builtin_amdgcn_exp2f(fmaf(x[j], LOG2E, kq[j]));
                float r = fmaf(abg[j], pj, (SMOOTH && abg[j] != 0.f) ? -bjg : 0.f);
                r -= (lab[j] == c) ? ag[j] : 0.f;
                o[j] = (lab[j] >= 0) ? r : 0.f;                    // void / invalid label: no gradient
            }
            __builtin_nontemporal_store(pack_piece<DTYPE, PPL>(o), (u32x4_s*)(grad_b + (size_t)(row_off(k) + px)));
        }
    }
    if (LOSS) {
        if (bad) atomicOr(status, 8);
        tile_block_partial(acc, accw, cnt, s_red, nw, partials);
    }
}

}  // namespace nmsa

using namespace nmsa;

namespace {

// ---- geometry chosen on the host -------------------------------------------------------------
// NS = ceil(rows / 8) steps hold a column; a wave keeps KS <= 32 of them (the instantiated KS that
// wastes least), NWV = ceil(NS / KS) <= 4 waves share a column (cosine) / one wave holds it (CE).
struct ColCfg { int ks, nwv; };

const int COS_KSS[4] = {8, 16, 24, 32};
const int CE_KSS[6] = {8, 12, 16, 19, 24, 32};

ColCfg col_config(int rows, const int* kss, int n_kss, int max_nwv)
{
    const int ns = (rows + 7) / 8;
    ColCfg best = {0, 0};
    int best_waste = 1 << 30;
    for (int nwv = 1; nwv <= max_nwv; ++nwv) {
        for (int j = 0; j < n_kss; ++j) {
            const int ks = kss[j];
            if (ks * nwv < ns) continue;
            // rows the last wave would hold beyond the column must leave it at least one real row
            if ((nwv - 1) * ks * 8 >= rows) continue;
            const int waste = ks * nwv - ns;
            if (waste < best_waste) { best_waste = waste; best = {ks, nwv}; }
            break;                                      // larger KS only waste more at this nwv
        }
        if (best.ks && best_waste * 8 <= ns) break;     // <= 12.5 % of phantom steps: good enough
    }
    return best;
}

int tiles_per_run(int B, int NT, int runs_wanted, int max_runs_per_image)
{
    int per_img = (runs_wanted + B - 1) / B;
    if (per_img > NT) per_img = NT;
    if (per_img > max_runs_per_image) per_img = max_runs_per_image;
    if (per_img < 1) per_img = 1;
    return (NT + per_img - 1) / per_img;
}

bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

size_t cos_ws_partials(int B, int P)
{
    const size_t nt = ((size_t)P + 31) / 32;
    return (size_t)B * (nt < 8192 ? nt : 8192) * sizeof(LossPartial);
}
size_t cos_ws_rows(int B, int P) { return ((size_t)B * ((size_t)P / 4 + 1) * sizeof(int4) + 255) / 256 * 256; }

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" int nmsa_loss_cos_emb_fwd_grad_supported(int dtype, int D, int H, int W, int L)
{
    if (D <= 0 || L <= 0 || H <= 0 || W <= 0) return 0;
    if (dtype != NMSA_F32 && dtype != NMSA_BF16 && dtype != NMSA_F16) return 0;
    const int es = dtype == NMSA_F32 ? 4 : 2;
    static const int on = loss_env_int("NMSA_COS_TILE", 0);         // experimental: off by default
    if (!on) return 0;
    if (((int64_t)H * W) % (16 / es) != 0) return 0;
    // 32-bit byte offsets inside one image / one LUT
    if ((int64_t)D * H * W * es >= ((int64_t)1 << 32) || (int64_t)L * D * 4 >= ((int64_t)1 << 32)) return 0;
    return col_config(D, COS_KSS, 4, 4).ks != 0;
}

extern "C" size_t nmsa_loss_cos_emb_fwd_grad_workspace_bytes(int B, int D, int H, int W, int L)
{
    (void)D;
    if (loss_bad_shape(B, H, W) || L <= 0) return 0;
    // block partials + the LUT rows of every pixel group (16 B per 4 px at most) + |y|^2 per LUT row
    return cos_ws_partials(B, H * W) + cos_ws_rows(B, H * W) + (size_t)B * L * sizeof(float);
}

namespace {

template <int DTYPE, int KS, bool LOSS>
int launch_cos_col(const ColCfg& cfg, const void* pred, const int32_t* indices, const float* lut,
                   const float* yy, const int4* rows, int B, int D, int P, int L, const float* gscale,
                   const float* computed_for, int* counters, void* grad, LossPartial* partials,
                   int* status, int* n_blocks, hipStream_t stream)
{
    constexpr int ES = (DTYPE == NMSA_F32) ? 4 : 2;
    constexpr int TP = COL_LPR * 16 / ES;
    const int NT = (P + TP - 1) / TP;
    // ~12 runs of tiles per CU over the batch: enough independent workgroups to fill the wave
    // slots several times over (they drift apart and overlap), long enough runs to amortise set-up
    static const int runs = loss_env_int("NMSA_TILE_WGS", 3072);
    const int tpw = tiles_per_run(B, NT, runs, 8192);
    const int gx = (NT + tpw - 1) / tpw;
    if (n_blocks) *n_blocks = gx * B;
    static const int ablate = loss_env_int("NMSA_TILE_ABLATE", 0);       // timing experiments only
    hipLaunchKernelGGL((k_cos_col<DTYPE, KS, LOSS>), dim3(gx, B), dim3(cfg.nwv * 64), 0, stream, pred, indices,
                       lut, yy, rows, D, P, L, tpw, gscale, computed_for, counters, grad, partials, status, ablate);
    return check_launch();
}

template <bool LOSS>
int dispatch_cos_col(int dtype, const ColCfg& cfg, const void* pred, const int32_t* indices,
                     const float* lut, const float* yy, const int4* rows, int B, int D, int P, int L,
                     const float* gscale, const float* computed_for, int* counters, void* grad,
                     LossPartial* partials, int* status, int* n_blocks, hipStream_t stream)
{
#define COS_COL(DT, KSV) launch_cos_col<DT, KSV, LOSS>(cfg, pred, indices, lut, yy, rows, B, D, P, L, \
        gscale, computed_for, counters, grad, partials, status, n_blocks, stream)
#define COS_COL_KS(DT) (cfg.ks == 32 ? COS_COL(DT, 32) : cfg.ks == 24 ? COS_COL(DT, 24) : cfg.ks == 16 ? COS_COL(DT, 16) : COS_COL(DT, 8))
    switch (dtype) {
        case NMSA_F32: return COS_COL_KS(NMSA_F32);
        case NMSA_BF16: return COS_COL_KS(NMSA_BF16);
        case NMSA_F16: return COS_COL_KS(NMSA_F16);
        default: return NMSA_ERR_ARG;
    }
#undef COS_COL_KS
#undef COS_COL
}

}  // namespace

extern "C" int nmsa_loss_cos_emb_fwd_grad(const void* pred, int dtype, const int32_t* indices,
                                          const float* lut, int B, int D, int H, int W, int L,
                                          const float* expected_grad_scale,
                                          double* loss_sum, int64_t* n_rows, void* grad_pred,
                                          int32_t* status, void* workspace, size_t workspace_bytes,
                                          nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !indices || !lut || !loss_sum || !n_rows || !status || !workspace ||
        !expected_grad_scale || !grad_pred) return NMSA_ERR_ARG;
    if (loss_bad_shape(B, H, W) || D <= 0 || L <= 0) return NMSA_ERR_ARG;
    if (!nmsa_loss_cos_emb_fwd_grad_supported(dtype, D, H, W, L)) return NMSA_ERR_UNSUPPORTED;
    if (!aligned16(pred) || !aligned16(grad_pred) || !aligned16(indices)) return NMSA_ERR_UNSUPPORTED;
    if (workspace_bytes < nmsa_loss_cos_emb_fwd_grad_workspace_bytes(B, D, H, W, L)) return NMSA_ERR_WORKSPACE;
    const int P = H * W;
    const int es = dtype == NMSA_F32 ? 4 : 2;
    const ColCfg cfg = col_config(D, COS_KSS, 4, 4);
    LossPartial* partials = (LossPartial*)workspace;
    int4* rows = (int4*)((char*)workspace + cos_ws_partials(B, P));
    float* yy = (float*)((char*)rows + cos_ws_rows(B, P));
    hipLaunchKernelGGL(k_lut_norms, dim3((B * L + 3) / 4), dim3(256), 0, stream, lut, B * L, D, yy);
    int rc = check_launch();
    if (rc) return rc;
    {
        // P % PPL == 0: the groups of all images are consecutive
        const long long n_groups = (long long)B * (P / (16 / es));
        const unsigned gb = (unsigned)((n_groups + 255) / 256);
        if (es == 4) hipLaunchKernelGGL(k_cos_rows<4>, dim3(gb), dim3(256), 0, stream, indices, n_groups, L, rows);
        else hipLaunchKernelGGL(k_cos_rows<8>, dim3(gb), dim3(256), 0, stream, indices, n_groups, L, rows);
        rc = check_launch();
        if (rc) return rc;
    }
    int n_blocks = 0;
    rc = dispatch_cos_col<true>(dtype, cfg, pred, indices, lut, yy, rows, B, D, P, L, expected_grad_scale,
                                nullptr, nullptr, grad_pred, partials, status, &n_blocks, stream);
    if (rc) return rc;
    return loss_finalize(partials, n_blocks, loss_sum, nullptr, n_rows, stream);
}

extern "C" int nmsa_loss_cos_emb_bwd_unless(const void* pred, int dtype, const int32_t* indices,
                                            const float* lut, int B, int D, int H, int W, int L,
                                            const float* grad_scale, void* grad_pred,
                                            const float* computed_for, int32_t* counters,
                                            void* workspace, size_t workspace_bytes,
                                            nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !indices || !lut || !grad_scale || !grad_pred || !computed_for || !workspace)
        return NMSA_ERR_ARG;
    if (loss_bad_shape(B, H, W) || D <= 0 || L <= 0) return NMSA_ERR_ARG;
    if (!nmsa_loss_cos_emb_fwd_grad_supported(dtype, D, H, W, L)) return NMSA_ERR_UNSUPPORTED;
    if (!aligned16(pred) || !aligned16(grad_pred) || !aligned16(indices)) return NMSA_ERR_UNSUPPORTED;
    if (workspace_bytes < nmsa_loss_cos_emb_fwd_grad_workspace_bytes(B, D, H, W, L)) return NMSA_ERR_WORKSPACE;
    const ColCfg cfg = col_config(D, COS_KSS, 4, 4);
    // the LUT-row table and the |y|^2 table of the forward call are still in the workspace
    const int4* rows = (const int4*)((const char*)workspace + cos_ws_partials(B, H * W));
    const float* yy = (const float*)((const char*)rows + cos_ws_rows(B, H * W));
    return dispatch_cos_col<false>(dtype, cfg, pred, indices, lut, yy, rows, B, D, H * W, L, grad_scale,
                                   computed_for, counters, grad_pred, nullptr, nullptr, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------
namespace {

template <int DTYPE, int KS, bool SMOOTH, bool LOSS>
int launch_ce_col(const void* logits, const uint8_t* target, const float* weights,
                  int B, int C, int P, float ls, const float* gscale, const float* computed_for,
                  int* counters, void* grad, LossPartial* partials, int* status, int* n_blocks,
                  int max_bpi, hipStream_t stream)
{
    constexpr int ES = (DTYPE == NMSA_F32) ? 4 : 2;
    constexpr int TP = COL_LPR * 16 / ES;
    constexpr int NWB = 4;                              // waves per workgroup (independent of each other)
    const int NT = (P + TP - 1) / TP;
    static const int runs = loss_env_int("NMSA_TILE_WGS", 3072);
    int tpw = tiles_per_run(B, NT, runs * NWB, 1 << 30);      // tiles per WAVE
    int gx = (NT + tpw * NWB - 1) / (tpw * NWB);
    if (gx > max_bpi) { gx = max_bpi; tpw = (NT + gx * NWB - 1) / (gx * NWB); }   // the caller's partial buffer
    if (n_blocks) *n_blocks = gx * B;
    const int lab_vec = (((uintptr_t)target) & 7) == 0;                  // P % PPL == 0 keeps the images aligned
    hipLaunchKernelGGL((k_ce_col<DTYPE, KS, SMOOTH, LOSS>), dim3(gx, B), dim3(NWB * 64), C * sizeof(float), stream,
                       logits, target, weights, C, P, ls, tpw, gscale, computed_for, counters, grad,
                       partials, status, lab_vec);
    return check_launch();
}

template <bool LOSS>
int dispatch_ce_col(int dtype, bool smooth, int ks, const void* logits, const uint8_t* target,
                    const float* weights, int B, int C, int P, float ls, const float* gscale,
                    const float* computed_for, int* counters, void* grad, LossPartial* partials,
                    int* status, int* n_blocks, int max_bpi, hipStream_t stream)
{
#define CE_COL(DT, KSV, SM) launch_ce_col<DT, KSV, SM, LOSS>(logits, target, weights, B, C, P, ls, \
        gscale, computed_for, counters, grad, partials, status, n_blocks, max_bpi, stream)
#define CE_COL_SM(DT, KSV) (smooth ? CE_COL(DT, KSV, true) : CE_COL(DT, KSV, false))
#define CE_COL_KS(DT) (ks == 32 ? CE_COL_SM(DT, 32) : ks == 24 ? CE_COL_SM(DT, 24) : ks == 19 ? CE_COL_SM(DT, 19) : \
                       ks == 16 ? CE_COL_SM(DT, 16) : ks == 12 ? CE_COL_SM(DT, 12) : CE_COL_SM(DT, 8))
    switch (dtype) {
        case NMSA_F32: return CE_COL_KS(NMSA_F32);
        case NMSA_BF16: return CE_COL_KS(NMSA_BF16);
        case NMSA_F16: return CE_COL_KS(NMSA_F16);
        default: return NMSA_ERR_ARG;
    }
#undef CE_COL_KS
#undef CE_COL_SM
#undef CE_COL
}

}  // namespace

namespace nmsa {

// used by nmsa_loss_ce_fwd_grad / nmsa_loss_ce_bwd_unless (losses.hip) for C > 48
bool ce_tile_supported(const void* logits, const void* grad, int dtype, int C, int P, float ls)
{
    (void)ls;
    const int es = dtype == NMSA_F32 ? 4 : 2;
    if (P % (16 / es) != 0 || !aligned16(logits) || !aligned16(grad)) return false;
    static const int on = loss_env_int("NMSA_CE_TILE", 0);          // experimental: off by default
    if (!on) return false;
    if ((int64_t)C * P * es >= ((int64_t)1 << 32)) return false;           // 32-bit offsets inside an image
    return col_config(C, CE_KSS, 6, 1).ks != 0;
}

int ce_tile_launch(bool loss, const void* logits, int dtype, const uint8_t* target, const float* weights,
                   int B, int C, int P, float ls, const float* gscale, const float* computed_for,
                   int* counters, void* grad, LossPartial* partials, int* status, int* n_blocks,
                   int max_blocks_per_image, hipStream_t stream)
{
    const bool smooth = ls != 0.0f;
    const ColCfg cfg = col_config(C, CE_KSS, 6, 1);
    if (!cfg.ks) return NMSA_ERR_UNSUPPORTED;
    if (loss) return dispatch_ce_col<true>(dtype, smooth, cfg.ks, logits, target, weights, B, C, P, ls, gscale,
                                           computed_for, counters, grad, partials, status, n_blocks,
                                           max_blocks_per_image, stream);
    return dispatch_ce_col<false>(dtype, smooth, cfg.ks, logits, target, weights, B, C, P, ls, gscale,
                                  computed_for, counters, grad, partials, status, n_blocks,
                                  max_blocks_per_image, stream);
}

}  // namespace nmsa
