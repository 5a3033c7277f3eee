// losses_cos.hip — cosine-embedding loss, forward + gradient in ONE pass over the prediction.
//   CosineEmbeddingLoss._compute_loss   loss/cos_emb.py:21-56 (+ the LUT gather of
//                                       task_helper/dense_visual_embedding.py:110-171)
//   per valid px (index != 0):  1 - x.y / sqrt((|x|^2 + eps)(|y|^2 + eps)), eps = 1e-12,
//   y = lut[b][index - 1];   d/dx = -y / den + (x.y) x / ((|x|^2 + eps) den)
//
// The gradient needs x.y and |x|^2 over the pixel's whole D-column (1 KB at D = 512) before its
// first element can be written; the two-kernel path reads the prediction twice (1.5x the
// algorithmic bytes).  Here the column stays in REGISTERS between the reduction and the gradient,
// in the layout that made the wide cross entropy byte-exact (k_ce_split): the NW = D / 64 waves
// of a workgroup look at the SAME 64 x PXT pixels, wave w keeps the planes [64 w, 64 w + 64), so
// every wave-instruction moves one contiguous 512-byte piece of a plane.  The waves exchange
// their partial x.y and |x|^2 through LDS (two barriers per tile, back to back), each wave then
// writes the gradient of its own planes from its registers — and, as a register is consumed,
// requests the same plane of the NEXT tile into it, so the memory pipe stays busy through the
// arithmetic although the CU holds a single workgroup (the fp32 LUT of the image, 128 KB at
// D = 512 / L = 64, fills its LDS).  Indices are piecewise constant: when the PXT pixels of every
// lane share one LUT row, one LDS read per plane serves them all.
#include <stdlib.h>
#include "loss_bodies.hpp"

namespace nmsa {

constexpr int COSS_NP = 64;                            // planes per wave
constexpr int COSS_MAX_WAVES = 8;                      // D <= 512: 8 waves = 2 per SIMD, 256 VGPRs each

// (loss, count) of the workgroup -> its partial slot; any number of waves up to COSS_MAX_WAVES
__device__ __forceinline__ void block_partial_wide(double acc, long long cnt, LossPartial* __restrict__ partials)
{
    __shared__ double r_sum[COSS_MAX_WAVES];
    __shared__ long long r_cnt[COSS_MAX_WAVES];
    acc = wave_reduce_sum(acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane_id() == 0) { r_sum[threadIdx.x >> 6] = acc; r_cnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0; long long c = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) { a += r_sum[k]; c += r_cnt[k]; }
        LossPartial pr; pr.sum = a; pr.aux = 0; pr.count = c; pr.pad = 0;
        partials[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = pr;
    }
}

template <int DTYPE, int MODE>                         // MODE 0: loss + gradient, 2: gradient only
__global__ __launch_bounds__(64 * COSS_MAX_WAVES) void k_cos_split(
    const void* __restrict__ pred, const int32_t* __restrict__ indices, const float* __restrict__ lut,
    int D, int P, int L, int vec, int tiles_per_wg,
    const float* __restrict__ expected_gscale, void* __restrict__ grad,
    LossPartial* __restrict__ partials, int* __restrict__ status,
    const float* __restrict__ computed_for, int* __restrict__ counters)
{
    constexpr int PXT = (DTYPE == NMSA_F32) ? 2 : 4;
    constexpr int NP = COSS_NP;
    constexpr int TPX = 64 * PXT;                      // pixels per tile
    constexpr bool LOSS = MODE != 2;
    extern __shared__ float s_mem[];                   // [L][D + 1] LUT | yy[L] | xy[NW][TPX] | xx[NW][TPX]
    if (!LOSS && grad_already_computed(expected_gscale, computed_for, counters)) return;
    const int NW = blockDim.x >> 6;
    const int ld = D + 1;
    float* s_lut = s_mem;
    float* s_yy = s_lut + (size_t)L * ld;
    float* s_xy = s_yy + ((L + 3) & ~3);
    float* s_xx = s_xy + NW * TPX;
    const int b = blockIdx.y;
    const float* lut_b = lut + (size_t)b * L * D;
    for (int i = threadIdx.x; i < L * D; i += blockDim.x) {
        const int r = i / D, d = i - r * D;
        s_lut[r * ld + d] = lut_b[i];
    }
    __syncthreads();
    for (int r = threadIdx.x >> 6; r < L; r += NW) {   // |y|^2 per row: one wave per row
        float yy = 0.f;
        for (int d = lane_id(); d < D; d += 64) { const float v = s_lut[r * ld + d]; yy = fmaf(v, v, yy); }
        yy = wave_reduce_sum(yy);
        if (lane_id() == 0) s_yy[r] = yy;
    }
    __syncthreads();

    const float EPS = 1e-12f;
    const float g = grad ? *expected_gscale : __int_as_float(0x7fc00000);
    const bool write_grad = grad && (MODE == 2 || g == g);
    const size_t img = (size_t)b * D * P;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id();
    const int c0 = w * NP;
    constexpr int nc = NP;                             // D % NP == 0 (the host checks): no per-plane branches
    double acc = 0.0;
    long long cnt = 0;
    bool bad = false;
    const int n_tiles = (P + TPX - 1) / TPX;
    const int t_begin = blockIdx.x * tiles_per_wg, t_end = min(n_tiles, t_begin + tiles_per_wg);
    if (t_begin >= t_end) { if (LOSS) block_partial_wide(0.0, 0, partials); return; }

    // `vec` layouts only (P % PXT == 0, 8-byte aligned planes: the host checks): every plane access
    // is ONE 8-byte load / store at a wave-uniform plane base + a 32-bit lane offset shared by all
    // planes — no per-plane 64-bit address registers, no edge branches inside the walks.  Lanes
    // past the end of the image (last tile) read the image's first pixels and store nothing.
    constexpr int ESIZE = (DTYPE == NMSA_F32) ? 4 : 2;
    const char* pred_b = (const char*)pred + img * ESIZE;
    char* grad_b = (char*)grad + img * ESIZE;
    u32x2_s r[NP];
    auto lane_offset = [&](int tile) -> uint32_t {
        const int p0 = (tile * 64 + l) * PXT;
        return (uint32_t)((p0 < P ? p0 : 0) * ESIZE);
    };
    // plane addresses: ONE running scalar offset per walk (`po`, advanced by a plane per step and
    // made opaque so that it stays a running sum) + the lane offset — NP loop-invariant plane
    // bases hoisted out of the tile loop were 128 scalar registers, 200-280 of them spilled to lane
    // registers and read back before every load and store
    const size_t pstride = (size_t)P * ESIZE;
    const size_t po0 = (size_t)c0 * pstride;
    auto request_plane = [&](int i, size_t po, uint32_t off) {
        r[i] = __builtin_nontemporal_load((const u32x2_s*)(pred_b + po + off));
    };
    // pixels without a target get exactly +0 (the reference gathers the valid rows only): the
    // masks clear their lanes of the packed words, so a non-finite prediction there cannot leak
    // a 0 * inf = NaN into the gradient
    auto store_plane = [&](int i, size_t po, uint32_t off, const float o[PXT], uint32_t mx, uint32_t my) {
        u32x2_s v;
        if (DTYPE == NMSA_F32) { v.x = __float_as_uint(o[0]); v.y = __float_as_uint(o[1]); }
        else { v.x = pack16<DTYPE>(o[0], o[1]); v.y = pack16<DTYPE>(o[2], o[3]); }
        v.x &= mx; v.y &= my;
        __builtin_nontemporal_store(v, (u32x2_s*)(grad_b + po + off));
    };
    auto request = [&](int tile) {                     // all planes of `tile` into r[]
        const uint32_t off = lane_offset(tile);
        size_t po = po0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            r[i] = u32x2_s{0u, 0u};
            if (i < nc) request_plane(i, po, off);
            po += pstride;
            asm volatile("" : "+s"(po));
        }
    };
#ifndef COSS_PREFETCH
#define COSS_PREFETCH 1
#endif
    if (COSS_PREFETCH) request(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        if (!COSS_PREFETCH) request(tile);
        const int p0 = (tile * 64 + l) * PXT;
        const bool alive = p0 < P;
        const int nvalid = alive ? min(PXT, P - p0) : 0;
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
        // ---- pass 1: partial x.y and |x|^2 over my planes --------------------------------------
        // (one LDS read per pixel and plane: lanes that share a row are a broadcast, different
        // rows sit in different banks — row stride D + 1; one code path for segment interiors and
        // boundaries keeps the walks straight-line)
        float xy[PXT], xx[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j) { xy[j] = 0.f; xx[j] = 0.f; row[j] += c0; }
        if (DTYPE != NMSA_F32) keep_packed(r);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                const float x = plane_px<DTYPE>(r[i], j);
                xy[j] = fmaf(x, s_lut[row[j] + i], xy[j]);
                xx[j] = fmaf(x, x, xx[j]);
            }
        }
        // ---- the sums over the waves, in wave order (the second barrier follows the first by a few
        // LDS reads: the LUT leaves no room for a second set of exchange buffers) --------------------
        float* bxy = s_xy;
        float* bxx = s_xx;
#pragma unroll
        for (int j = 0; j < PXT; ++j) { bxy[w * TPX + l * PXT + j] = xy[j]; bxx[w * TPX + l * PXT + j] = xx[j]; }
        __syncthreads();
        float k1[PXT], k2[PXT];
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            float sxy = bxy[l * PXT + j], sxx = bxx[l * PXT + j];
            for (int ww = 1; ww < NW; ++ww) { sxy += bxy[ww * TPX + l * PXT + j]; sxx += bxx[ww * TPX + l * PXT + j]; }
            const float den = sqrtf((sxx + EPS) * (s_yy[ridx[j]] + EPS));
            k1[j] = on[j] ? -g / den : 0.f;
            k2[j] = on[j] ? g * sxy / ((sxx + EPS) * den) : 0.f;
            if (LOSS && w == 0 && on[j]) { part += 1.0f - sxy / den; ++cnt; }
        }
        acc += part;
        __syncthreads();                                // every wave has read the sums: the buffers are free
        // ---- pass 2: the gradient of my planes; each register then takes the next tile's plane ----
        const bool more = COSS_PREFETCH && tile + 1 < t_end;        // wave-uniform
        const uint32_t off = lane_offset(tile), qoff = lane_offset(more ? tile + 1 : tile);
        const bool store = alive && write_grad;
        uint32_t mx, my;
        if (DTYPE == NMSA_F32) { mx = on[0] ? ~0u : 0u; my = on[1] ? ~0u : 0u; }
        else {
            mx = (on[0] ? 0xFFFFu : 0u) | (on[PXT > 2 ? 1 : 0] ? 0xFFFF0000u : 0u);
            my = (on[PXT > 2 ? 2 : 0] ? 0xFFFFu : 0u) | (on[PXT > 2 ? 3 : 0] ? 0xFFFF0000u : 0u);
        }
        if (DTYPE != NMSA_F32) keep_packed(r);
        size_t po = po0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            float o[PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j)
                o[j] = fmaf(k2[j], plane_px<DTYPE>(r[i], j), k1[j] * s_lut[row[j] + i]);
            if (store) store_plane(i, po, off, o, mx, my);
            if (COSS_PREFETCH) request_plane(i, po, qoff);     // (the last tile re-reads itself: no branch in the walk)
            po += pstride;
            asm volatile("" : "+s"(po));
        }
    }
    if (LOSS) {
        if (bad) atomicOr(status, 8);
        block_partial_wide(acc, cnt, partials);
    }
}


// ---------------------------------------------------------------------------------------------
// Columns beyond one workgroup (D = 768: a 197 KB LUT, 12 waves of 64 planes): k_cos_parts.
// The column is split over NS = ceil(D / 256) COOPERATING workgroups ("parts") of four waves;
// part q keeps the planes [256 q, 256 q + 256) of the same tiles in its registers and only those
// LUT columns in its LDS (66 KB at L = 64: two workgroups per CU, so one part's exchange wait is
// the other workgroup's streaming time).  What the parts owe each other per tile is 2 x PXT
// partial sums per lane.  They travel as 8-byte {value, tag} granules: one sc1 (write-through)
// store per granule by wave 0, polled with sc1 loads by the partners' wave 0 — a granule is
// written and read whole, so it validates itself and no fence is needed (MI355X_MICROARCH.md,
// hand-off price list: 1-3 us per hop against ~20 us of streaming per tile).  Every part adds the
// partial sums in part order: all parts get the same bits.  Two slots per part (tile parity): a
// part publishes tile t + 1 only after it has read its partners' tile t, which they published
// after reading tile t - 1 of everybody.
// Forward progress: the launch geometry (coss_geometry) never asks for more workgroups than the
// DEVICE holds at once — compute units as the HIP runtime reports them for this device (a
// partitioned MI355X or a CU mask shows fewer than 256: api.hip device_geometry) x the workgroups
// per CU that the LDS footprint, the launch bounds (2 waves per SIMD) and the occupancy query
// admit — so once foreign kernels have drained every workgroup of the grid is resident and every
// partner answers, whatever the dispatch order.  That is an argument about an otherwise idle
// device, not a guarantee: another stream's long kernel, or a device that shows the runtime more
// CUs than it schedules on, leaves partners un-dispatched.  So every wait is bounded (`timeout`
// ticks of the 100 MHz wall clock, 0.5 s by default, NMSA_COS_PARTS_TIMEOUT_MS): a wave 0 that
// gives up stops waiting for good, poisons its sums with NaN so that every wave still reaches the
// end, and raises the call's `gave_up` word (+ status bit 32, also from the gradient-only
// launch).  The launch function queues the two-walk kernels of losses.hip behind it, gated ON THE
// DEVICE by that word: they return at once after a healthy run (three empty launches, ~6 us behind
// a 4-8 ms kernel) and otherwise recompute the call's sums and gradient — a call never returns
// the poison.  (The test hooks NMSA_ASSUME_CUS — an overstated chip — and NMSA_COS_PARTS_ORDER=1
// — the parts of a group far apart in the workgroup order instead of neighbours — together strand
// every resident workgroup and drive exactly that path.)
constexpr int COSP_WAVES = 4;
constexpr int COSP_COLS = COSP_WAVES * COSS_NP;        // 256 planes per part
constexpr int COSP_MAX_PARTS = 4;                      // D <= 1024
constexpr int COSP_GRAN = 2 * 4 * 64;                  // granules per (slot, part): 2 sums x 4 px x 64 lanes
constexpr size_t COSP_HEAD = 64;                       // the call's gave_up word in front of the granules

// RAGGED: D is no multiple of 64 — the last wave that holds planes holds fewer than 64; its walks
// test every plane against the wave's plane count (wave-uniform branches).  Compiled apart so that
// the walks of the usual shapes stay free of branches.
template <int DTYPE, int MODE, bool RAGGED>            // MODE 0: loss + gradient, 2: gradient only
__global__ __launch_bounds__(64 * COSP_WAVES, 2) void k_cos_parts(
    const void* __restrict__ pred, const int32_t* __restrict__ indices, const float* __restrict__ lut,
    int B, int D, int P, int L, int NS, int tiles_per_wg,
    const float* __restrict__ expected_gscale, void* __restrict__ grad,
    LossPartial* __restrict__ partials, int* __restrict__ status,
    const float* __restrict__ computed_for, int* __restrict__ counters,
    unsigned long long* __restrict__ xch, int* __restrict__ gave_up, int part_major, long long timeout)
{
    constexpr int PXT = (DTYPE == NMSA_F32) ? 2 : 4;
    constexpr int NP = COSS_NP;
    constexpr int NW = COSP_WAVES;
    constexpr int TPX = 64 * PXT;
    constexpr bool LOSS = MODE != 2;
    extern __shared__ float s_mem[];       // [L][257] LUT columns of this part | yy[L] | xy[NW][TPX] | xx[NW][TPX] | tot[2 PXT][64]
    if (!LOSS && grad_already_computed(expected_gscale, computed_for, counters)) return;
    // the tiles of the whole batch form ONE sequence (image after image) cut into equal runs, one
    // per group: every workgroup slot of the chip gets the same amount of work whatever B is; a
    // run that crosses into the next image restages that image's LUT columns (from L2)
    // the NS parts of a group are neighbours in the workgroup order (dispatched together); the
    // test hook `part_major` puts them a whole grid / NS apart instead
    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    const int n_groups = (int)(gridDim.x * gridDim.y) / NS;
    const int part = part_major ? wg / n_groups : wg % NS, group = part_major ? wg % n_groups : wg / NS;
    const int d0 = part * COSP_COLS;
    const int DP = min(COSP_COLS, D - d0);             // my columns (a multiple of 64 unless RAGGED)
    constexpr int ld = COSP_COLS + 1;
    float* s_lut = s_mem;
    float* s_yy = s_lut + (size_t)L * ld;
    float* s_xy = s_yy + ((L + 3) & ~3);
    float* s_xx = s_xy + NW * TPX;
    float* s_tot = s_xx + NW * TPX;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = lane_id();
    auto stage_lut = [&](int b) {
        const float* lut_b = lut + (size_t)b * L * D;
        const int DPW = RAGGED ? ((DP + NP - 1) / NP) * NP : DP;       // zeros behind a ragged end: 0 * x = 0
        for (int i = threadIdx.x; i < L * DPW; i += blockDim.x) {
            const int r = i / DPW, d = i - r * DPW;
            s_lut[r * ld + d] = d < DP ? lut_b[(size_t)r * D + d0 + d] : 0.f;
        }
        for (int r = w; r < L; r += NW) {              // |y|^2 over the WHOLE row (all parts: same bits)
            float yy = 0.f;
            for (int d = l; d < D; d += 64) { const float v = lut_b[(size_t)r * D + d]; yy = fmaf(v, v, yy); }
            yy = wave_reduce_sum(yy);
            if (l == 0) s_yy[r] = yy;
        }
    };

    const float EPS = 1e-12f;
    const float g = grad ? *expected_gscale : __int_as_float(0x7fc00000);
    const bool write_grad = grad && (MODE == 2 || g == g);
    const int nwa = (DP + NP - 1) / NP;                // waves of this part that hold planes
    const bool active = w < nwa;                       // wave-uniform
    const int nc = RAGGED ? min(NP, DP - w * NP) : NP; // planes of this wave (<= 0: none)
    const int c0 = d0 + w * NP;                        // my first plane
    const int lc0 = w * NP;                            // ... as a column of s_lut
    double acc = 0.0;
    long long cnt = 0;
    bool bad = false;
    bool dead = false;                                 // a partner did not answer: stop waiting
    const int n_tiles = (P + TPX - 1) / TPX;           // per image
    const long long t_total = (long long)B * n_tiles;
    const long long t_begin = (long long)group * tiles_per_wg;
    const long long t_end = t_begin + tiles_per_wg < t_total ? t_begin + tiles_per_wg : t_total;
    if (t_begin >= t_end) { if (LOSS) block_partial_wide(0.0, 0, partials); return; }
    const size_t gid = (size_t)group;

    constexpr int ESIZE = (DTYPE == NMSA_F32) ? 4 : 2;
    const size_t img_bytes = (size_t)D * P * ESIZE;
    u32x2_s r[NP];
    auto lane_offset = [&](int tile) -> uint32_t {
        const int p0 = (tile * 64 + l) * PXT;
        return (uint32_t)((p0 < P ? p0 : 0) * ESIZE);
    };
    // (plane addresses as in k_cos_split: one running scalar offset per walk + the lane offset)
    const size_t pstride = (size_t)P * ESIZE;
    const size_t po0 = (size_t)c0 * pstride;
    auto request_plane = [&](int i, const char* pred_b, size_t po, uint32_t off) {
        r[i] = __builtin_nontemporal_load((const u32x2_s*)(pred_b + po + off));
    };
    auto store_plane = [&](int i, char* grad_b, size_t po, uint32_t off, const float o[PXT], uint32_t mx, uint32_t my) {
        char* gb = grad_b + po;
        u32x2_s v;
        if (DTYPE == NMSA_F32) { v.x = __float_as_uint(o[0]); v.y = __float_as_uint(o[1]); }
        else { v.x = pack16<DTYPE>(o[0], o[1]); v.y = pack16<DTYPE>(o[2], o[3]); }
        v.x &= mx; v.y &= my;
        __builtin_nontemporal_store(v, (u32x2_s*)(gb + off));
    };
#pragma unroll
    for (int i = 0; i < NP; ++i) r[i] = u32x2_s{0u, 0u};
    if (active) {
        const int b0 = (int)(t_begin / n_tiles);
        const uint32_t off = lane_offset((int)(t_begin - (long long)b0 * n_tiles));
        size_t po = po0;
        int ncw = nc;
        if (RAGGED) asm volatile("" : "+s"(ncw));
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (!RAGGED || i < ncw) request_plane(i, (const char*)pred + b0 * img_bytes, po, off);
            po += pstride;
            asm volatile("" : "+s"(po));
        }
    }
    int b_staged = -1;
    for (long long gt = t_begin; gt < t_end; ++gt) {
        const int b = (int)(gt / n_tiles), tile = (int)(gt - (long long)b * n_tiles);
        if (b != b_staged) {                            // first tile, or the run enters the next image
            if (b_staged >= 0) __syncthreads();         // every wave is done with the old LUT
            stage_lut(b);
            __syncthreads();
            b_staged = b;
        }
        const int p0 = (tile * 64 + l) * PXT;
        const bool alive = p0 < P;
        const int nvalid = alive ? min(PXT, P - p0) : 0;
        int row[PXT], ridx[PXT];
        bool on[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const int ix = (j < nvalid) ? indices[(size_t)b * P + p0 + j] : 0;
            if (ix < 0 || ix > L) bad = true;
            on[j] = ix > 0 && ix <= L;
            ridx[j] = on[j] ? ix - 1 : 0;
            row[j] = ridx[j] * ld + lc0;
        }
        // ---- pass 1: partial x.y and |x|^2 over my planes --------------------------------------
        float xy[PXT], xx[PXT];
#pragma unroll
        for (int j = 0; j < PXT; ++j) { xy[j] = 0.f; xx[j] = 0.f; }
        if (active) {
            if (DTYPE != NMSA_F32) keep_packed(r);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
#pragma unroll
                for (int j = 0; j < PXT; ++j) {
                    const float x = plane_px<DTYPE>(r[i], j);
                    xy[j] = fmaf(x, s_lut[row[j] + i], xy[j]);
                    xx[j] = fmaf(x, x, xx[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PXT; ++j) { s_xy[w * TPX + l * PXT + j] = xy[j]; s_xx[w * TPX + l * PXT + j] = xx[j]; }
        __syncthreads();
        // ---- wave 0: the part's sums -> granules -> the sums over all parts ----------------------
        if (w == 0) {
            float v[2 * PXT], tot[2 * PXT];
#pragma unroll
            for (int j = 0; j < PXT; ++j) {
                float a = s_xy[l * PXT + j], c = s_xx[l * PXT + j];
                for (int ww = 1; ww < nwa; ++ww) { a += s_xy[ww * TPX + l * PXT + j]; c += s_xx[ww * TPX + l * PXT + j]; }
                v[j] = a; v[PXT + j] = c;
            }
            const uint32_t seq = (uint32_t)(gt - t_begin) + 1u;
            unsigned long long* slot = xch + (gid * 2 + ((gt - t_begin) & 1)) * NS * COSP_GRAN;
            unsigned long long* mine = slot + (size_t)part * COSP_GRAN;
#pragma unroll
            for (int k = 0; k < 2 * PXT; ++k)
                __hip_atomic_store(mine + k * 64 + l, ((unsigned long long)seq << 32) | __float_as_uint(v[k]),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 2 * PXT; ++k) tot[k] = 0.f;
            for (int q = 0; q < NS; ++q) {             // in part order: every part adds the same way
                if (q == part) {
#pragma unroll
                    for (int k = 0; k < 2 * PXT; ++k) tot[k] += v[k];
                    continue;
                }
                const unsigned long long* theirs = slot + (size_t)q * COSP_GRAN;
                unsigned long long gr[2 * PXT];
                // poll the LAST granule of the partner's eight stores (one load per lane and
                // round), then read them all; every granule is checked — store order is no promise
                const long long t0 = dead ? 0 : (long long)wall_clock64();
                for (;;) {
                    const unsigned long long probe = __hip_atomic_load(theirs + (2 * PXT - 1) * 64 + l,
                                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all((uint32_t)(probe >> 32) == seq)) {
                        bool ok = true;
#pragma unroll
                        for (int k = 0; k < 2 * PXT; ++k)
                            gr[k] = __hip_atomic_load(theirs + k * 64 + l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int k = 0; k < 2 * PXT; ++k) ok = ok && (uint32_t)(gr[k] >> 32) == seq;
                        if (__all(ok)) break;
                    }
                    if (dead) break;                                                  // (wave-uniform)
                    __builtin_amdgcn_s_sleep(8);
                    if (__any((long long)wall_clock64() - t0 > timeout)) dead = true;         // 100 MHz ticks
                }
#pragma unroll
                for (int k = 0; k < 2 * PXT; ++k)
                    tot[k] += dead ? __int_as_float(0x7fc00000) : __uint_as_float((uint32_t)gr[k]);
            }
#pragma unroll
            for (int k = 0; k < 2 * PXT; ++k) s_tot[k * 64 + l] = tot[k];
        }
        __syncthreads();
        float k1[PXT], k2[PXT];
        float ploss = 0.f;
#pragma unroll
        for (int j = 0; j < PXT; ++j) {
            const float sxy = s_tot[j * 64 + l], sxx = s_tot[(PXT + j) * 64 + l];
            const float den = sqrtf((sxx + EPS) * (s_yy[ridx[j]] + EPS));
            k1[j] = on[j] ? -g / den : 0.f;
            k2[j] = on[j] ? g * sxy / ((sxx + EPS) * den) : 0.f;
            if (LOSS && w == 0 && part == 0 && on[j]) { ploss += 1.0f - sxy / den; ++cnt; }
        }
        acc += ploss;
        // ---- pass 2: the gradient of my planes; each register then takes the next tile's plane ----
        if (active) {
            const long long gn = gt + 1 < t_end ? gt + 1 : gt;          // (the last tile re-reads itself)
            const int bn = (int)(gn / n_tiles);
            const uint32_t off = lane_offset(tile), qoff = lane_offset((int)(gn - (long long)bn * n_tiles));
            const char* pred_n = (const char*)pred + bn * img_bytes;
            char* grad_b = (char*)grad + b * img_bytes;
            const bool store = alive && write_grad;
            uint32_t mx, my;
            if (DTYPE == NMSA_F32) { mx = on[0] ? ~0u : 0u; my = on[1] ? ~0u : 0u; }
            else {
                mx = (on[0] ? 0xFFFFu : 0u) | (on[PXT > 2 ? 1 : 0] ? 0xFFFF0000u : 0u);
                my = (on[PXT > 2 ? 2 : 0] ? 0xFFFFu : 0u) | (on[PXT > 2 ? 3 : 0] ? 0xFFFF0000u : 0u);
            }
            if (DTYPE != NMSA_F32) keep_packed(r);
            size_t po = po0;
            int ncw = nc;                                               // (compared per plane as a scalar,
            if (RAGGED) asm volatile("" : "+s"(ncw));                   //  not kept as NP lane masks)
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (!RAGGED || i < ncw) {                               // wave-uniform
                    float o[PXT];
#pragma unroll
                    for (int j = 0; j < PXT; ++j)
                        o[j] = fmaf(k2[j], plane_px<DTYPE>(r[i], j), k1[j] * s_lut[row[j] + i]);
                    if (store) store_plane(i, grad_b, po, off, o, mx, my);
                    request_plane(i, pred_n, po, qoff);
                }
                po += pstride;
                asm volatile("" : "+s"(po));
            }
        }
    }
    if (dead && l == 0) {                               // (wave 0 only; both launch modes)
        atomicOr(gave_up, 1);
        if (status) atomicOr(status, 32);
    }
    if (LOSS) {
        if (bad) atomicOr(status, 8);
        block_partial_wide(acc, cnt, partials);
    }
}

// After a k_cos_parts that gave up, the two-walk forward kernel has left the call's sums in
// `fb[0 .. n_fb)`: they replace the poisoned partials (slot 0 = their sum in a fixed order, the
// other slots 0), so whoever finalizes the partials — this file or the multi-loss call — sees the
// fallback's result.  One workgroup; returns at once when nobody gave up.
__global__ __launch_bounds__(256) void k_cos_adopt(const int* __restrict__ gave_up, const LossPartial* __restrict__ fb,
                                                   int n_fb, LossPartial* __restrict__ partials, int n)
{
    if (*gave_up == 0) return;
    __shared__ double s_sum[256];
    __shared__ long long s_cnt[256];
    double a = 0.0; long long c = 0;
    for (int i = threadIdx.x; i < n_fb; i += 256) { a += fb[i].sum; c += fb[i].count; }
    s_sum[threadIdx.x] = a; s_cnt[threadIdx.x] = c;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { s_sum[threadIdx.x] += s_sum[threadIdx.x + o]; s_cnt[threadIdx.x] += s_cnt[threadIdx.x + o]; }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += 256) {
        LossPartial pr; pr.sum = 0; pr.aux = 0; pr.count = 0; pr.pad = 0;
        if (i == 0) { pr.sum = s_sum[0]; pr.count = s_cnt[0]; }
        partials[i] = pr;
    }
}
}  // namespace nmsa

using namespace nmsa;

namespace {

size_t coss_lds_bytes(int D, int L, int nw, int pxt)
{
    return ((size_t)L * (D + 1) + ((L + 3) & ~3) + (size_t)2 * nw * 64 * pxt) * sizeof(float);
}

size_t cosp_lds_bytes(int L, int pxt)
{
    return ((size_t)L * (COSP_COLS + 1) + ((L + 3) & ~3) + (size_t)2 * COSP_WAVES * 64 * pxt +
            (size_t)2 * pxt * 64) * sizeof(float);
}

// which one-pass kernel takes this shape: 1 = k_cos_split (the whole column in one workgroup:
// D <= 512, the image's whole fp32 LUT in LDS), 2 = k_cos_parts (the column over up to four
// cooperating workgroups, D <= 1024), 0 = none.  NMSA_COS_SPLIT=0: none; NMSA_COS_PARTS=0: never
// the parts kernel, =1: the parts kernel wherever it can run (same-box A/B at D <= 512)
int cos_kernel(int dtype, int D, int L)
{
    if (dtype != NMSA_F32 && dtype != NMSA_BF16 && dtype != NMSA_F16) return 0;
    if (D <= 0 || L <= 0) return 0;
    const bool ragged = D % COSS_NP != 0;                          // (k_cos_split: whole groups of 64 planes only)
    static const int on = loss_env_int("NMSA_COS_SPLIT", 1);
    const char* pe = getenv("NMSA_COS_PARTS");         // read at every call: tests switch it
    const int parts = (pe && *pe) ? atoi(pe) : -1;
    if (!on) return 0;
    const int pxt = dtype == NMSA_F32 ? 2 : 4;
    const int nw = D / COSS_NP;
    const DeviceGeometry dg = device_geometry();
    const size_t wg_lds = dg.lds_per_block > 2048 ? dg.lds_per_block - 2048 : 0;    // (static LDS of the reductions)
    const bool split_ok = !ragged && nw <= COSS_MAX_WAVES && coss_lds_bytes(D, L, nw, pxt) <= wg_lds;
    const bool parts_ok = parts != 0 && D <= COSP_COLS * COSP_MAX_PARTS && cosp_lds_bytes(L, pxt) <= wg_lds;
    if (parts == 1 && parts_ok) return 2;
    // 512 planes fit one workgroup, but two cooperating half-columns with two workgroups per CU
    // stream 7 % faster (one's exchange wait is the other's streaming time): measured at B = 16,
    // same process, both kernels alternating: 4.61-4.72 vs 4.97-5.17 ms.  Narrower columns are as
    // fast or faster in one workgroup (D = 448: 4.28-4.36 vs 4.30-4.41, 384: 3.74 vs 3.94,
    // 256: 2.79 vs 2.82 ms)
    if (split_ok && parts_ok && parts == -1 && D >= 2 * COSP_COLS && 2 * cosp_lds_bytes(L, pxt) <= dg.lds_per_cu)
        return 2;
    if (split_ok) return 1;
    return parts_ok ? 2 : 0;
}

int cosp_parts(int D) { return (D + COSP_COLS - 1) / COSP_COLS; }

// workgroups of k_cos_parts that are resident per CU: 2 at most (its launch bounds keep the
// registers of 2 x 4 waves per SIMD pair free), fewer when the LDS of the device or the runtime's
// occupancy answer for this instantiation says so.  (The API over-reports only near 8 per CU —
// MI355X_MICROARCH.md, Residency — far from the 2 asked for here.)
int cosp_resident_per_cu(int dtype, bool ragged, int L)
{
    const int pxt = (dtype == NMSA_F32) ? 2 : 4;
    const size_t lds = cosp_lds_bytes(L, pxt);
    const DeviceGeometry dg = device_geometry();
    int per_cu = (int)(dg.lds_per_cu / lds) >= 2 ? 2 : 1;
    int api = 0;
    hipError_t e = hipErrorUnknown;
#define OCC_(DT, RG) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k_cos_parts<DT, 0, RG>, 64 * COSP_WAVES, lds)
#define OCC(DT) do { if (ragged) OCC_(DT, true); else OCC_(DT, false); } while (0)
    switch (dtype) {
        case NMSA_F32: OCC(NMSA_F32); break;
        case NMSA_BF16: OCC(NMSA_BF16); break;
        default: OCC(NMSA_F16); break;
    }
#undef OCC
#undef OCC_
    if (e == hipSuccess && api >= 1 && api < per_cu) per_cu = api;
    if (e != hipSuccess) (void)hipGetLastError();      // (no device: the LDS bound alone)
    return per_cu;
}

// workgroups per image and tiles per workgroup.  k_cos_split: one workgroup per CU is resident
// (the LUT fills the LDS), a few workgroups per CU over the whole batch, each walking a run of
// consecutive tiles of one image.  k_cos_parts: groups of NS workgroups; as many groups as fill
// the chip's workgroup slots `rounds` times (rounded DOWN: a few idle slots in the last round cost
// less than a round of their own).
void coss_geometry(int B, int D, int P, int L, int dtype, int* gx, int* tpw, int* ns)
{
    const int pxt = (dtype == NMSA_F32) ? 2 : 4;
    const int n_tiles = (P + 64 * pxt - 1) / (64 * pxt);
    int per_img;
    *ns = 1;
    if (cos_kernel(dtype, D, L) == 2) {
        // k_cos_parts cuts the tiles of the WHOLE batch into G equal runs, G = the groups of NS
        // workgroups that fill the chip's workgroup slots (`rounds` times); the grid is
        // (gx * NS, B) >= G * NS workgroups, the surplus ones find an empty run
        *ns = cosp_parts(D);
        const char* re = getenv("NMSA_COS_PARTS_ROUNDS");              // (per call: same-process A/B)
        const int rounds = (re && *re) ? atoi(re) : 1;
        const int per_cu = cosp_resident_per_cu(dtype, D % COSS_NP != 0, L);
        long long G = (long long)device_geometry().cus * per_cu * (rounds < 1 ? 1 : rounds) / *ns;
        const long long t_total = (long long)B * n_tiles;
        if (G > t_total) G = t_total;
        if (G < 1) G = 1;
        long long run_len = (t_total + G - 1) / G;
        const char* run = getenv("NMSA_COS_SPLIT_RUN");
        if (run && atoi(run) > 0) run_len = atoi(run);
        if ((t_total + run_len - 1) / run_len > (1 << 20)) run_len = (t_total + (1 << 20) - 1) >> 20;
        G = (t_total + run_len - 1) / run_len;
        *tpw = (int)run_len;
        *gx = (int)((G + B - 1) / B);                   // groups "per image": gx * B >= G
        return;
    } else {
        static const int per_cu = loss_env_int("NMSA_COS_SPLIT_WGS_PER_CU", 2);
        per_img = (device_geometry().cus * (per_cu < 1 ? 1 : per_cu) + B - 1) / B;
    }
    if (per_img > n_tiles) per_img = n_tiles;
    if (per_img > 4096) per_img = 4096;
    if (per_img < 1) per_img = 1;
    *tpw = (n_tiles + per_img - 1) / per_img;
    // NMSA_COS_SPLIT_RUN=k (read at every call: tests switch it inside one process): runs of k
    // tiles per workgroup whatever the image size, so that small shapes reach the tile hand-over
    // of the gradient walk, the ragged last tile inside a run and the clamped lane offsets
    const char* run = getenv("NMSA_COS_SPLIT_RUN");
    if (run && atoi(run) > 0) {
        *tpw = atoi(run);
        if ((n_tiles + *tpw - 1) / *tpw > 4096) *tpw = (n_tiles + 4095) / 4096;
    }
    *gx = (n_tiles + *tpw - 1) / *tpw;                  // groups per image
}

template <int MODE>
int coss_launch(const void* pred, int dtype, const int32_t* indices, const float* lut, int B, int D, int P,
                int L, const float* gscale, void* grad, LossPartial* partials, int32_t* status,
                const float* computed_for, int32_t* counters, void* xch, size_t xch_bytes, hipStream_t stream)
{
    const int pxt = (dtype == NMSA_F32) ? 2 : 4;
    const int which = cos_kernel(dtype, D, L);
    if (!which) return NMSA_ERR_UNSUPPORTED;
    // the kernels only have the 8-byte plane accesses: whole groups of PXT pixels, aligned planes
    if (P % pxt != 0 || ((((uintptr_t)pred | (uintptr_t)grad) & 7) != 0)) return NMSA_ERR_UNSUPPORTED;
    int gx, tpw, ns;
    coss_geometry(B, D, P, L, dtype, &gx, &tpw, &ns);
    if (which == 2) {
        // [ gave_up word (64 B) | granules | partials of the fallback's forward walk ]
        const size_t need = (size_t)B * gx * 2 * ns * COSP_GRAN * sizeof(unsigned long long);
        const int n_fb = cos_two_walk_blocks(dtype, B, D, P, L);
        if (!xch || xch_bytes < COSP_HEAD + need + (MODE == 0 ? (size_t)n_fb * sizeof(LossPartial) : 0) ||
            (((uintptr_t)xch) & 15)) return NMSA_ERR_WORKSPACE;
        int* gave_up = (int*)xch;
        unsigned long long* granules = (unsigned long long*)((char*)xch + COSP_HEAD);
        LossPartial* fb = (LossPartial*)((char*)xch + COSP_HEAD + need);
        // tags start at 1: a zeroed buffer holds no valid granule
        if (check_hip(hipMemsetAsync(xch, 0, COSP_HEAD + need, stream))) return NMSA_ERR_LAUNCH;
        const size_t lds = cosp_lds_bytes(L, pxt);
        const char* oe = getenv("NMSA_COS_PARTS_ORDER");               // (per call: a test hook)
        const int part_major = (oe && atoi(oe) == 1) ? 1 : 0;
        const char* te = getenv("NMSA_COS_PARTS_TIMEOUT_MS");
        const long long timeout = 100000ll * ((te && atoi(te) > 0) ? atoi(te) : 500);   // 100 MHz ticks
#define COSP_(DT, RG) do { int rc_ = allow_dynamic_lds(k_cos_parts<DT, MODE, RG>, lds); if (rc_) return rc_;       \
        hipLaunchKernelGGL((k_cos_parts<DT, MODE, RG>), dim3(gx * ns, B), dim3(64 * COSP_WAVES), lds, stream, pred, \
                           indices, lut, B, D, P, L, ns, tpw, gscale, grad, partials, status, computed_for,     \
                           counters, granules, gave_up, part_major, timeout); } while (0)
#define COSP(DT) do { if (D % COSS_NP != 0) COSP_(DT, true); else COSP_(DT, false); } while (0)
        NMSA_DISPATCH_DTYPE(dtype, COSP)
#undef COSP
#undef COSP_
        int rc = check_launch();
        if (rc) return rc;
        // the fallback, gated on the device by `gave_up` (see the kernel's header comment)
        if (MODE == 0) {
            rc = launch_cos_two_walk_fwd(pred, dtype, indices, lut, B, D, P, L, fb, status, nullptr, gave_up, stream);
            if (rc) return rc;
            hipLaunchKernelGGL(k_cos_adopt, dim3(1), dim3(256), 0, stream, gave_up, fb, n_fb, partials, B * gx * ns);
            rc = check_launch();
            if (rc) return rc;
        }
        if (grad) rc = launch_cos_two_walk_bwd(pred, dtype, indices, lut, B, D, P, L, gscale, nullptr, grad, gave_up,
                                               MODE == 0 ? 1 : 0, stream);
        return rc;
    }
    const int nw = D / COSS_NP;
    const size_t lds = coss_lds_bytes(D, L, nw, pxt);
#define COSS(DT) do { int rc_ = allow_dynamic_lds(k_cos_split<DT, MODE>, lds); if (rc_) return rc_;              \
        hipLaunchKernelGGL((k_cos_split<DT, MODE>), dim3(gx, B), dim3(64 * nw), lds, stream, pred, indices, lut, \
                           D, P, L, 1, tpw, gscale, grad, partials, status, computed_for, counters); } while (0)
    NMSA_DISPATCH_DTYPE(dtype, COSS)
#undef COSS
    return check_launch();
}

}  // namespace

// 1 when a one-pass kernel takes this shape: D % 64 == 0 and either the image's fp32 LUT and the
// exchange buffers fit the LDS of a CU with the column in the waves of one workgroup (D <= 512),
// or the column splits over up to four cooperating workgroups (D <= 1024)
extern "C" int nmsa_loss_cos_emb_fwd_grad_supported(int dtype, int D, int H, int W, int L)
{
    if (H <= 0 || W <= 0) return 0;
    return cos_kernel(dtype, D, L) != 0;
}

namespace nmsa {

int cos_split_blocks(int B, int D, int P, int L, int dtype)     // partial slots per image
{
    int gx, tpw, ns;
    coss_geometry(B, D, P, L, dtype, &gx, &tpw, &ns);
    return gx * ns;
}

// bytes of the exchange buffer of k_cos_parts (0: the shape runs in one workgroup)
size_t cos_split_xch_bytes(int B, int D, int P, int L, int dtype)
{
    if (cos_kernel(dtype, D, L) != 2) return 0;
    int gx, tpw, ns;
    coss_geometry(B, D, P, L, dtype, &gx, &tpw, &ns);
    // [ gave_up word | granules | partials of the fallback's forward walk ] (coss_launch)
    const size_t fb = ((size_t)cos_two_walk_blocks(dtype, B, D, P, L) * sizeof(LossPartial) + 63) & ~(size_t)63;
    return COSP_HEAD + (size_t)B * gx * 2 * ns * COSP_GRAN * sizeof(unsigned long long) + fb;
}

int launch_cos_split(bool loss, const void* pred, int dtype, const int32_t* indices, const float* lut,
                     int B, int D, int P, int L, const float* gscale, const float* computed_for,
                     int32_t* counters, void* grad, LossPartial* partials, int32_t* status,
                     void* xch, size_t xch_bytes, hipStream_t stream)
{
    return loss ? coss_launch<0>(pred, dtype, indices, lut, B, D, P, L, gscale, grad, partials, status,
                                 computed_for, counters, xch, xch_bytes, stream)
                : coss_launch<2>(pred, dtype, indices, lut, B, D, P, L, gscale, grad, partials, status,
                                 computed_for, counters, xch, xch_bytes, stream);
}

}  // namespace nmsa

namespace {
size_t cos_partial_bytes(int B, int D, int P, int L)
{
    // (dtype-independent bound: the f32 geometry has the most tiles)
    int nb = cos_split_blocks(B, D, P, L, NMSA_F32);
    const int nb16 = cos_split_blocks(B, D, P, L, NMSA_BF16);
    if (nb16 > nb) nb = nb16;
    return ((size_t)B * nb * sizeof(LossPartial) + 63) & ~(size_t)63;
}
size_t cos_xch_max(int B, int D, int P, int L)
{
    const size_t a = cos_split_xch_bytes(B, D, P, L, NMSA_F32), b = cos_split_xch_bytes(B, D, P, L, NMSA_BF16);
    return a > b ? a : b;
}
}  // namespace

// [ block partials | granule exchange of k_cos_parts ]; nmsa_loss_cos_emb_bwd_unless takes the
// same buffer (it only uses the exchange part)
extern "C" size_t nmsa_loss_cos_emb_fwd_grad_workspace_bytes(int B, int D, int H, int W, int L)
{
    if (B <= 0 || H <= 0 || W <= 0 || D <= 0 || L <= 0) return 0;
    return cos_partial_bytes(B, D, H * W, L) + cos_xch_max(B, D, H * W, L) + 64;
}

// forward sum + n_rows + the gradient for the EXPECTED upstream scale *expected_gscale (a NaN:
// no gradient is written), one pass over the prediction
extern "C" int nmsa_loss_cos_emb_fwd_grad(const void* pred, int dtype, const int32_t* indices, const float* lut,
                                          int B, int D, int H, int W, int L, const float* expected_gscale,
                                          double* loss_sum, int64_t* n_rows, void* grad_pred, int32_t* status,
                                          void* workspace, size_t workspace_bytes, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !indices || !lut || !expected_gscale || !loss_sum || !n_rows || !status || !workspace)
        return NMSA_ERR_ARG;
    if (loss_bad_shape(B, H, W) || D <= 0 || L <= 0) return NMSA_ERR_ARG;
    if (!nmsa_loss_cos_emb_fwd_grad_supported(dtype, D, H, W, L)) return NMSA_ERR_UNSUPPORTED;
    if (workspace_bytes < nmsa_loss_cos_emb_fwd_grad_workspace_bytes(B, D, H, W, L)) return NMSA_ERR_WORKSPACE;
    LossPartial* partials = (LossPartial*)workspace;
    const size_t pbytes = cos_partial_bytes(B, D, H * W, L);
    int rc = coss_launch<0>(pred, dtype, indices, lut, B, D, H * W, L, expected_gscale, grad_pred, partials,
                            status, nullptr, nullptr, (char*)workspace + pbytes, workspace_bytes - pbytes, stream);
    if (rc) return rc;
    return loss_finalize(partials, cos_split_blocks(B, D, H * W, L, dtype) * B, loss_sum, nullptr, n_rows, stream);
}

// confirms the gradient nmsa_loss_cos_emb_fwd_grad wrote (returns at once when *grad_scale is
// bit-equal to *computed_for) or recomputes it; `workspace`: the forward call's buffer (only
// columns beyond 512 use it: NULL is fine below)
extern "C" int nmsa_loss_cos_emb_bwd_unless(const void* pred, int dtype, const int32_t* indices, const float* lut,
                                            int B, int D, int H, int W, int L, const float* grad_scale,
                                            void* grad_pred, const float* computed_for, int32_t* counters,
                                            void* workspace, size_t workspace_bytes, nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!pred || !indices || !lut || !grad_scale || !grad_pred) return NMSA_ERR_ARG;
    if (loss_bad_shape(B, H, W) || D <= 0 || L <= 0) return NMSA_ERR_ARG;
    if (!nmsa_loss_cos_emb_fwd_grad_supported(dtype, D, H, W, L)) return NMSA_ERR_UNSUPPORTED;
    return coss_launch<2>(pred, dtype, indices, lut, B, D, H * W, L, grad_scale, grad_pred, nullptr, nullptr,
                          computed_for, counters, workspace, workspace_bytes, stream);
}
