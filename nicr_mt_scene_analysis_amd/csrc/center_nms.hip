// center_nms.hip — instance-center NMS + per-image top-k + ordered compaction.
//
// Replaces the ATen chain of InstancePostprocessing._get_instance_centers
// (reference model/postprocessing/instance.py:79-168):
//   F.threshold -> max_pool2d(return_indices) -> pad -> index test -> equality
//   test -> topk(k)[..., -1] -> clamp(min=0) -> [fg mask] -> >= kth -> nonzero()
//
// Two kernels, both HBM/L2-latency bound integer + compare work (no MFMA):
//   k_nms_candidates : LDS-staged (TH+2p)x(TW+2p) tile of the thresholded map,
//                      window scan per pixel, survivors with value >= 0 set a
//                      bit in a per-image bitmask (4 B/px read, ~0 written).
//   k_select_compact : one 1024-thread workgroup per image: 3-level radix
//                      select (11/11/10 bits) of the k-th largest candidate
//                      key, then an ORDER-PRESERVING compaction (block prefix
//                      sum over the bitmask) so that center index == raster
//                      order, exactly as `nonzero()` yields it.
//
// Why "value >= 0 candidates only": everything kept must pass `>= clamp(kth, 0)`,
// so negative survivors never matter, and for v >= 0 the IEEE bit pattern is a
// monotone unsigned key (with -0.0 mapped to key 0).
#include <stdlib.h>
#include "nmsa_common.hpp"

namespace nmsa {

constexpr int NMS_TW = 64;
constexpr int NMS_TH = 16;
constexpr int NMS_PAD_MAX = 4;  // LDS path for ksize <= 9, direct-global path beyond

__device__ __forceinline__ float threshold_m1(float x, float thr)
{
    // F.threshold(x, thr, -1): ATen evaluates `x <= thr ? -1 : x` (NaN is kept)
    return (x <= thr) ? -1.0f : x;
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_nms_candidates(
    const float* __restrict__ center, uint32_t* __restrict__ cand_bits,
    int H, int W, int words_per_image, float thr, int pad)
{
    __shared__ float tile[USE_LDS ? (NMS_TH + 2 * NMS_PAD_MAX) * (NMS_TW + 2 * NMS_PAD_MAX) : 1];
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * NMS_TW, y0 = blockIdx.y * NMS_TH;
    const float* img = center + (size_t)b * H * W;
    const int tw = NMS_TW + 2 * pad, th = NMS_TH + 2 * pad;

    if (USE_LDS) {
        for (int i = threadIdx.x; i < tw * th; i += blockDim.x) {
            const int ty = i / tw, tx = i - ty * tw;
            const int gy = y0 + ty - pad, gx = x0 + tx - pad;
            float v = -1.0f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = threshold_m1(img[(size_t)gy * W + gx], thr);
            tile[i] = v;
        }
        __syncthreads();
    }
    auto at = [&](int gy, int gx) -> float {
        if (USE_LDS) return tile[(gy - y0 + pad) * tw + (gx - x0 + pad)];
        return threshold_m1(img[(size_t)gy * W + gx], thr);
    };

    uint32_t* bits = cand_bits + (size_t)b * words_per_image;
    for (int i = threadIdx.x; i < NMS_TW * NMS_TH; i += blockDim.x) {
        const int ly = i / NMS_TW, lx = i - ly * NMS_TW;
        const int y = y0 + ly, x = x0 + lx;
        if (y >= H || x >= W) continue;
        const int self = y * W + x;
        const float h = at(y, x);
        bool survive;
        if (y < pad || y >= H - pad || x < pad || x >= W - pad) {
            // zero-padded pooled value / index (instance.py:104-109): only pixel 0
            // can pass the index test, and only with value exactly 0
            survive = (self == 0) && (h == 0.0f);
        } else {
            // ATen max_pool2d window scan: row-major, take on (v > max) || isnan(v)
            float pooled = -INFINITY;
            int pidx = (y - pad) * W + (x - pad);
            for (int dy = -pad; dy <= pad; ++dy)
                for (int dx = -pad; dx <= pad; ++dx) {
                    const float v = at(y + dy, x + dx);
                    if (v > pooled || v != v) { pooled = v; pidx = (y + dy) * W + (x + dx); }
                }
            survive = (pidx == self) && (h == pooled);
        }
        if (survive && h >= 0.0f) atomicOr(&bits[self >> 5], 1u << (self & 31));
    }
}

// Fast path (W % 32 == 0, ksize <= 9): a workgroup owns a strip of R full rows.  The
// strip (+ halo rows) is staged in LDS with 16-B loads; every wave then covers 64
// consecutive pixels of one row, so the survivors of a wave are ONE ballot and the
// candidate words are written whole — no atomics, no memset of the bitmask.
template <int PAD>
__global__ __launch_bounds__(1024) void k_nms_strip(
    const float* __restrict__ center, uint32_t* __restrict__ cand_bits,
    int H, int W, int R, int words_per_image, float thr)
{
    extern __shared__ float strip[];                 // [(R + 2 PAD)][W + 2 PAD]
    const int b = blockIdx.y;
    const int y0 = blockIdx.x * R;
    const int rows = min(R, H - y0);
    const float* img = center + (size_t)b * H * W;
    const int tw = W + 2 * PAD;
    const int trows = rows + 2 * PAD;

    // ---- stage: thresholded values, -1 outside the image ---------------------------
    if ((W & 3) == 0) {
        const int w4 = W >> 2;
        for (int i = threadIdx.x; i < trows * w4; i += blockDim.x) {
            const int r = i / w4, c4 = i - r * w4;
            const int gy = y0 - PAD + r;
            float4 v = make_float4(-1.f, -1.f, -1.f, -1.f);
            if (gy >= 0 && gy < H) {
                v = *(const float4*)(img + (size_t)gy * W + 4 * c4);
                v.x = threshold_m1(v.x, thr); v.y = threshold_m1(v.y, thr);
                v.z = threshold_m1(v.z, thr); v.w = threshold_m1(v.w, thr);
            }
            float* dst = strip + r * tw + PAD + 4 * c4;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
    } else {
        for (int i = threadIdx.x; i < trows * W; i += blockDim.x) {
            const int r = i / W, c = i - r * W;
            const int gy = y0 - PAD + r;
            strip[r * tw + PAD + c] = (gy >= 0 && gy < H) ? threshold_m1(img[(size_t)gy * W + c], thr) : -1.f;
        }
    }
    if (PAD > 0)
        for (int i = threadIdx.x; i < trows * 2 * PAD; i += blockDim.x) {
            const int r = i / (2 * PAD), k = i - r * 2 * PAD;
            strip[r * tw + (k < PAD ? k : W + k)] = -1.f;
        }
    __syncthreads();

    // A pixel survives iff it is the FIRST maximum of its window in row-major scan order
    // and the window holds no NaN (ATen takes `v > max || isnan(v)`):  every element before
    // it is strictly smaller, every element after it is <= (NaN fails both tests).
    // Each thread walks one column down the strip and keeps the (2 PAD + 1) window rows in
    // registers: (2 PAD + 1) LDS reads per pixel instead of (2 PAD + 1)^2.
    uint32_t* bits = cand_bits + (size_t)b * words_per_image;
    constexpr int K = 2 * PAD + 1;
    for (int xb = 0; xb < W; xb += blockDim.x) {
        const int x = xb + threadIdx.x;
        const bool in_x = x < W;
        const int xc = in_x ? x : 0;
        float win[K][K];
#pragma unroll
        for (int i = 0; i < K - 1; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j)
                win[i + 1][j] = strip[i * tw + xc + j];            // strip rows 0 .. K-2
        for (int r = 0; r < rows; ++r) {
            const int y = y0 + r;
#pragma unroll
            for (int i = 0; i < K - 1; ++i)
#pragma unroll
                for (int j = 0; j < K; ++j) win[i][j] = win[i + 1][j];
#pragma unroll
            for (int j = 0; j < K; ++j) win[K - 1][j] = strip[(r + K - 1) * tw + xc + j];
            bool cand = false;
            if (in_x) {
                const float h = win[PAD][PAD];
                bool survive;
                if (y < PAD || y >= H - PAD || x < PAD || x >= W - PAD) {
                    survive = (y * W + x == 0) && (h == 0.0f);     // zero-padded pool output
                } else {
                    survive = true;
#pragma unroll
                    for (int i = 0; i < K; ++i)
#pragma unroll
                        for (int j = 0; j < K; ++j) {
                            if (i == PAD && j == PAD) continue;
                            const bool before = (i < PAD) || (i == PAD && j < PAD);
                            survive = survive && (before ? (win[i][j] < h) : (win[i][j] <= h));
                        }
                    survive = survive && (h == h);
                }
                cand = survive && h >= 0.0f;
            }
            const unsigned long long m = __ballot(cand);
            if (lane_id() == 0) {
                const int x_wave = xb + (int)(threadIdx.x & ~63u);
                if (x_wave < W) {
                    const int word = (y * W + x_wave) >> 5;        // W % 32 == 0: word-aligned
                    bits[word] = (uint32_t)m;
                    if (x_wave + 32 < W) bits[word + 1] = (uint32_t)(m >> 32);
                }
            }
        }
    }
}

// 3x3 fast path (the reference's default kernel size, W % 32 == 0): no LDS, no barrier.  A wave
// owns a 256-pixel-wide column band and NR_ROWS rows; lane = 4 consecutive pixels (one 16-B
// load per row, 1 KiB per wave and row), all NR_ROWS + 2 row loads are issued before the first
// use.  Horizontal neighbours come from the adjacent lanes (two lane shifts per row; the band's
// outer columns from one extra 4-B load by lanes 0 / 63), vertical neighbours are the other
// rows' registers.  Eight lanes OR their 4 survivor bits into one 32-bit candidate word.

// HALF: the wave is two 32-lane halves, each a 128-pixel-wide band of its own row group — for
// widths that are a multiple of 128 but not of 256 (640 = 5 x 128: no idle lanes, where 256-px
// bands leave half of every third wave empty).
template <int NR_ROWS, bool HALF>
__global__ __launch_bounds__(256) void k_nms_rows3(
    const float* __restrict__ center, uint32_t* __restrict__ cand_bits,
    int H, int W, int words_per_image, float thr, int blocks_per_image, int n_blocks, int n_xcd)
{
    constexpr int LPB = HALF ? 32 : 64;                    // lanes per band
    constexpr int BW = LPB * 4;                            // band width in pixels
    // XCD-aware order: consecutive workgroup ids go round-robin over the device's XCDs (each with
    // its own L2; `n_xcd` as the runtime reports it: 8 on an unpartitioned MI355X); every XCD gets
    // one contiguous range of row groups, so that the halo rows a wave shares with the row groups
    // above and below are L2 hits instead of second fetches through the fabric (FETCH_SIZE 1.42x ->
    // see profiles/r04*).  Speed only: any n_xcd gives the same bits.
    const int per_xcd = n_blocks > 0 ? (n_blocks + n_xcd - 1) / n_xcd : 0;
    const int logical = n_blocks > 0 ? (int)(blockIdx.x % n_xcd) * per_xcd + (int)(blockIdx.x / n_xcd) : (int)blockIdx.x;
    if (n_blocks > 0 && logical >= n_blocks) return;
    const int b = n_blocks > 0 ? logical / blocks_per_image : (int)blockIdx.y;
    const int bx = n_blocks > 0 ? logical - b * blocks_per_image : (int)blockIdx.x;
    const int lane = lane_id();
    const int l = lane & (LPB - 1);                        // lane within its band
    const int sub = HALF ? (lane >> 5) : 0;
    const int bands = (W + BW - 1) / BW;
    const int wave = bx * 4 + (int)(threadIdx.x >> 6);
    const int seg = wave / bands, band = wave - seg * bands;
    const int y0 = (seg * (HALF ? 2 : 1) + sub) * NR_ROWS;
    if ((seg * (HALF ? 2 : 1)) * NR_ROWS >= H) return;     // whole wave
    const int xb = band * BW;
    const int x0 = xb + 4 * l;
    const bool in_x = x0 < W && y0 < H;                    // W % 4 == 0: all 4 pixels or none
    const float* img = center + (size_t)b * H * W;
    // band edge columns: the first lane fetches the pixel left of the band, the last one the
    // pixel right of it
    const int xe = (l == 0) ? xb - 1 : xb + BW;
    const bool edge_lane = (l == 0 || l == LPB - 1) && xe >= 0 && xe < W;

    float4 row[NR_ROWS + 2];
    float edge[NR_ROWS + 2];
#pragma unroll
    for (int r = 0; r < NR_ROWS + 2; ++r) {
        const int gy = min(max(y0 - 1 + r, 0), H - 1);     // rows outside the image: any value
        row[r] = in_x ? *(const float4*)(img + (size_t)gy * W + x0) : make_float4(-1.f, -1.f, -1.f, -1.f);
        edge[r] = edge_lane ? img[(size_t)gy * W + xe] : -1.f;
    }
    // thresholded row with its two horizontal neighbours: e[0] = left of pixel 0 ... e[5] = right of pixel 3
    auto extend = [&](int r, float e[6]) {
        const float4 v = row[r];
        e[1] = threshold_m1(v.x, thr); e[2] = threshold_m1(v.y, thr);
        e[3] = threshold_m1(v.z, thr); e[4] = threshold_m1(v.w, thr);
        const float ev = threshold_m1(edge[r], thr);
        const float lf = __shfl_up(e[4], 1), rr = __shfl_down(e[1], 1);
        e[0] = (l == 0) ? ev : lf;
        e[5] = (l == LPB - 1) ? ev : rr;
    };
    uint32_t* bits = cand_bits + (size_t)b * words_per_image;
    // hm[j] = max of the three horizontal neighbours around pixel j (NaN ignored): the pooled
    // maximum is max3 of three rows' hm, so almost every pixel is rejected by ONE compare
    // (h < pooled, or the thresholded background h == -1); the exact first-maximum rule runs
    // only in waves that hold a potential peak.
    auto hmax3 = [](const float e[6], float hm[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) hm[j] = fmaxf(fmaxf(e[j], e[j + 1]), e[j + 2]);
    };
    float top[6], mid[6], bot[6], htop[4], hmid[4], hbot[4];
    extend(0, mid);
    hmax3(mid, hmid);
    extend(1, bot);
    hmax3(bot, hbot);
#pragma unroll
    for (int r = 1; r <= NR_ROWS; ++r) {
        const int y = y0 + r - 1;
#pragma unroll
        for (int k = 0; k < 6; ++k) { top[k] = mid[k]; mid[k] = bot[k]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { htop[k] = hmid[k]; hmid[k] = hbot[k]; }
        extend(r + 1, bot);                                // wave-uniform: all lanes shift
        hmax3(bot, hbot);
        if (__all(y >= H)) break;                          // wave-uniform
        const bool row_ok = y < H;                         // the halves of a wave differ in y
        const bool border_y = y < 1 || y >= H - 1;
        bool maybe[4];
        bool any_maybe = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float h = mid[j + 1];
            const float pooled = fmaxf(fmaxf(htop[j], hmid[j]), hbot[j]);
            // NaN h: never a candidate (h >= 0 fails); NaN neighbours are caught by the exact rule
            maybe[j] = in_x && row_ok && h >= 0.0f &&
                       (border_y || h == pooled || x0 + j < 1 || x0 + j >= W - 1);
            any_maybe = any_maybe || maybe[j];
        }
        uint32_t nib = 0;
        if (__any(any_maybe)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = x0 + j;
                const float h = mid[j + 1];
                bool survive;
                if (border_y || x < 1 || x >= W - 1) {
                    survive = (y * W + x == 0) && (h == 0.0f);     // zero-padded pool output
                } else {
                    // first maximum in row-major window order, no NaN in the window
                    survive = (top[j] < h) && (top[j + 1] < h) && (top[j + 2] < h) && (mid[j] < h) &&
                              (mid[j + 2] <= h) && (bot[j] <= h) && (bot[j + 1] <= h) &&
                              (bot[j + 2] <= h) && (h == h);
                }
                if (maybe[j] && survive) nib |= 1u << j;
            }
        }
        uint32_t word = nib << (4 * (lane & 7));
        word |= __shfl_xor(word, 1);
        word |= __shfl_xor(word, 2);
        word |= __shfl_xor(word, 4);
        if ((lane & 7) == 0 && in_x && row_ok) bits[(y * W + x0) >> 5] = word;
    }
}

// ---- block-wide helpers (1024 threads = 16 waves) ---------------------------
__device__ __forceinline__ int wave_inclusive_scan(int v)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o);
        if (lane_id() >= o) v += t;
    }
    return v;
}

// inclusive prefix sum over the block; `scratch` holds >= 17 ints; returns the
// inclusive prefix for this thread and writes the block total to *total.
__device__ __forceinline__ int block_inclusive_scan(int v, int* scratch, int* total)
{
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int incl = wave_inclusive_scan(v);
    __syncthreads();
    if (lane_id() == 63) scratch[w] = incl;
    __syncthreads();
    if (w == 0) {
        int s = (lane_id() < nw) ? scratch[lane_id()] : 0;
        s = wave_inclusive_scan(s);
        if (lane_id() < nw) scratch[lane_id()] = s;
    }
    __syncthreads();
    const int base = (w == 0) ? 0 : scratch[w - 1];
    *total = scratch[nw - 1];
    return incl + base;
}

__device__ __forceinline__ uint32_t cand_key(float v)
{
    return (v == 0.0f) ? 0u : __float_as_uint(v);   // v >= 0 here
}

constexpr int SEL_THREADS = 1024;
constexpr int SEL_BINS = 2048;

// LDS_BITS: the image's candidate bitmask (P / 32 words) is staged in LDS once, with coalesced
// loads, 4 in flight per thread; the count / radix / compaction passes then walk LDS.  Without it
// every pass re-reads the thread's contiguous word range from L2, one dependent load per word
// (11.7 us at 640x480, 31.6 us at 1024x768 — five passes of 10 / 24 words per thread).
constexpr int SEL_LDS_WORDS = 36 * 1024;          // 144 KB: images up to 1.18 Mpx
constexpr int SEL_LIST_CAP = 2048;                // candidates kept as an LDS list (<= 32 per thread)

template <bool LDS_BITS>
__global__ __launch_bounds__(SEL_THREADS) void k_select_compact(
    const float* __restrict__ center, const uint8_t* __restrict__ fg,
    const uint32_t* __restrict__ cand_bits,
    int H, int W, int words_per_image, int topk, int apply_fg, int max_centers,
    int32_t* __restrict__ centers_yx, int32_t* __restrict__ n_centers,
    float* __restrict__ scores, uint8_t* __restrict__ center_mask, int list_cap)
{
    __shared__ int hist[SEL_BINS];
    __shared__ int scratch[32];
    __shared__ uint32_t s_prefix;
    __shared__ int s_krem;

    extern __shared__ uint32_t s_bits[];
    const int b = blockIdx.x;
    const int P = H * W;
    const float* img = center + (size_t)b * P;
    const uint32_t* gbits = cand_bits + (size_t)b * words_per_image;
    const uint8_t* fgb = fg ? fg + (size_t)b * P : nullptr;
    if (LDS_BITS) {
        // 16-byte pieces, 4 in flight per thread: the 38 KB mask of a 640x480 image is ONE round trip
        // for the 1024 threads (4-byte loads: three)
        const int nquads = ((((uintptr_t)gbits) & 15) == 0) ? (words_per_image >> 2) : 0;
        for (int i0 = threadIdx.x; i0 < nquads; i0 += 4 * SEL_THREADS) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * SEL_THREADS;
                v[u] = (i < nquads) ? ((const uint4*)gbits)[i] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * SEL_THREADS;
                if (i < nquads) {
                    s_bits[4 * i] = v[u].x; s_bits[4 * i + 1] = v[u].y;
                    s_bits[4 * i + 2] = v[u].z; s_bits[4 * i + 3] = v[u].w;
                }
            }
        }
        for (int i = 4 * nquads + threadIdx.x; i < words_per_image; i += SEL_THREADS) s_bits[i] = gbits[i];
        __syncthreads();
    }
    const uint32_t* bits = LDS_BITS ? (const uint32_t*)s_bits : gbits;

    // contiguous word range per thread (keeps raster order for the compaction)
    const int wpt = (words_per_image + SEL_THREADS - 1) / SEL_THREADS;
    const int w_begin = min((int)threadIdx.x * wpt, words_per_image);
    const int w_end = min(w_begin + wpt, words_per_image);

    // ---- number of candidates ------------------------------------------------
    int cnt = 0;
    for (int w = w_begin; w < w_end; ++w) cnt += __popc(bits[w]);
    int total;
    const int cnt_incl = block_inclusive_scan(cnt, scratch, &total);

    // ---- the candidates as a LIST in LDS (raster order): position + key, gathered ONCE.  The
    //      radix passes and the compaction then walk a dense, evenly split list instead of every
    //      thread's mask words with one dependent global gather per pass (five in all) ----------
    uint32_t* s_lpos = s_bits + words_per_image;
    uint32_t* s_lkey = s_lpos + list_cap;
    const bool use_list = LDS_BITS && total <= list_cap;      // block-uniform
    if (use_list) {
        int at = cnt_incl - cnt;
        for (int w = w_begin; w < w_end; ++w) {
            uint32_t m = bits[w];
            while (m) {
                const int bit = __ffs((int)m) - 1;
                m &= m - 1;
                const int p = (w << 5) + bit;
                s_lpos[at] = (uint32_t)p;
                s_lkey[at] = cand_key(img[p]);
                ++at;
            }
        }
        __syncthreads();
    }

    // ---- k-th largest candidate key (torch.topk(...)[..., -1], clamp(min=0)) --
    uint32_t key_kth = 0;
    if (total >= topk) {
        if (threadIdx.x == 0) { s_prefix = 0; s_krem = topk; }
        const int shifts[3] = {21, 10, 0};
        const int nbits[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            const int shift = shifts[pass];
            const int nb = 1 << nbits[pass];
            for (int i = threadIdx.x; i < SEL_BINS; i += SEL_THREADS) hist[i] = 0;
            __syncthreads();
            const uint32_t prefix = s_prefix;
            // bits above (shift + nbits) must equal the prefix found so far
            const int hi_shift = shift + nbits[pass];
            if (use_list) {
                for (int e = threadIdx.x; e < total; e += SEL_THREADS) {
                    const uint32_t k = s_lkey[e];
                    const bool match = (hi_shift >= 32) || ((k >> hi_shift) == (prefix >> hi_shift));
                    if (match) atomicAdd(&hist[(k >> shift) & (nb - 1)], 1);
                }
            } else {
                for (int w = w_begin; w < w_end; ++w) {
                    uint32_t m = bits[w];
                    while (m) {
                        const int bit = __ffs((int)m) - 1;
                        m &= m - 1;
                        const uint32_t k = cand_key(img[(w << 5) + bit]);
                        const bool match = (hi_shift >= 32) || ((k >> hi_shift) == (prefix >> hi_shift));
                        if (match) atomicAdd(&hist[(k >> shift) & (nb - 1)], 1);
                    }
                }
            }
            __syncthreads();
            // suffix sums: thread t owns bins 2t, 2t+1 (reversed so that an
            // inclusive prefix scan yields "count of keys in bins >= mine")
            const int t = threadIdx.x;
            const int hi_bin = SEL_BINS - 1 - 2 * t, lo_bin = hi_bin - 1;
            const int c_hi = hist[hi_bin], c_lo = hist[lo_bin];
            int dummy;
            const int incl = block_inclusive_scan(c_hi + c_lo, scratch, &dummy);
            const int above = incl - (c_hi + c_lo);     // keys in bins > hi_bin
            const int krem = s_krem;
            __syncthreads();
            if (above < krem && krem <= incl) {
                int sel, new_k;
                if (above + c_hi >= krem) { sel = hi_bin; new_k = krem - above; }
                else { sel = lo_bin; new_k = krem - above - c_hi; }
                s_prefix = prefix | ((uint32_t)sel << shift);
                s_krem = new_k;
            }
            __syncthreads();
        }
        key_kth = s_prefix;
    }

    // ---- ordered compaction ----------------------------------------------------
    auto kept = [&](int p) -> bool {
        if (apply_fg && !fgb[p]) return false;
        return cand_key(img[p]) >= key_kth;
    };
    if (use_list) {
        // contiguous list ranges per thread keep the raster order; every entry is looked at once
        // (its keep flag stays in a register mask: at most 32 entries per thread with
        // list_cap <= 32 * SEL_THREADS)
        const int per = (total + SEL_THREADS - 1) / SEL_THREADS;
        const int e0 = min((int)threadIdx.x * per, total), e1 = min(e0 + per, total);
        uint32_t keep = 0u;
        for (int e = e0; e < e1; ++e) {
            const bool k = s_lkey[e] >= key_kth && (!apply_fg || fgb[s_lpos[e]]);
            keep |= k ? (1u << (e - e0)) : 0u;
        }
        const int mine_l = __popc(keep);
        int n_total_l;
        const int incl_l = block_inclusive_scan(mine_l, scratch, &n_total_l);
        int pos_l = incl_l - mine_l;
        for (int e = e0; e < e1; ++e) {
            if (!((keep >> (e - e0)) & 1u)) continue;
            const int p = (int)s_lpos[e];
            if (pos_l < max_centers) {
                const int y = p / W;
                centers_yx[((size_t)b * max_centers + pos_l) * 2 + 0] = y;
                centers_yx[((size_t)b * max_centers + pos_l) * 2 + 1] = p - y * W;
                scores[(size_t)b * max_centers + pos_l] = img[p];
            }
            if (center_mask) center_mask[(size_t)b * P + p] = 1;
            ++pos_l;
        }
        if (threadIdx.x == 0) n_centers[b] = n_total_l;
        return;
    }
    int mine = 0;
    for (int w = w_begin; w < w_end; ++w) {
        uint32_t m = bits[w];
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1;
            mine += kept((w << 5) + bit) ? 1 : 0;
        }
    }
    int n_total;
    const int incl = block_inclusive_scan(mine, scratch, &n_total);
    int pos = incl - mine;
    for (int w = w_begin; w < w_end; ++w) {
        uint32_t m = bits[w];
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1;
            const int p = (w << 5) + bit;
            if (!kept(p)) continue;
            if (pos < max_centers) {
                const int y = p / W;
                centers_yx[((size_t)b * max_centers + pos) * 2 + 0] = y;
                centers_yx[((size_t)b * max_centers + pos) * 2 + 1] = p - y * W;
                scores[(size_t)b * max_centers + pos] = img[p];
            }
            if (center_mask) center_mask[(size_t)b * P + p] = 1;
            ++pos;
        }
    }
    if (threadIdx.x == 0) n_centers[b] = n_total;
}

}  // namespace nmsa

using namespace nmsa;

extern "C" size_t nmsa_center_nms_workspace_bytes(int B, int H, int W)
{
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const size_t words = ((size_t)H * W + 31) / 32;
    return (size_t)B * words * sizeof(uint32_t);
}

extern "C" int nmsa_center_nms_topk(const float* center, const uint8_t* fg,
                                    int B, int H, int W,
                                    float threshold, int ksize, int topk, int apply_fg,
                                    int max_centers,
                                    int32_t* centers_yx, int32_t* n_centers, float* scores,
                                    uint8_t* center_mask,
                                    void* workspace, size_t workspace_bytes,
                                    nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!center || !centers_yx || !n_centers || !scores || !workspace) return NMSA_ERR_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || max_centers <= 0) return NMSA_ERR_ARG;
    if (ksize < 1 || (ksize & 1) == 0 || topk < 1) return NMSA_ERR_ARG;
    if ((int64_t)H * W > (int64_t)1 << 30) return NMSA_ERR_ARG;
    if ((int64_t)H * W < topk) return NMSA_ERR_ARG;            // torch.topk would raise
    if (apply_fg && !fg) return NMSA_ERR_ARG;
    if (workspace_bytes < nmsa_center_nms_workspace_bytes(B, H, W)) return NMSA_ERR_WORKSPACE;

    const int words = (int)(((size_t)H * W + 31) / 32);
    uint32_t* bits = (uint32_t*)workspace;
    int rc;
    if (center_mask) {
        rc = check_hip(hipMemsetAsync(center_mask, 0, (size_t)B * H * W, stream));
        if (rc) return rc;
    }
    const int pad = (ksize - 1) / 2;
    // strip fast path: whole candidate words per wave (needs W % 32 == 0)
    static const int rows_env = getenv("NMSA_NMS_ROWS") ? atoi(getenv("NMSA_NMS_ROWS")) : 0;
    int R = rows_env > 0 ? rows_env : 4;
    while (R > 1 && (size_t)(R + 2 * pad) * (W + 2 * pad) * sizeof(float) > 48 * 1024) R >>= 1;
    const size_t strip_lds = (size_t)(R + 2 * pad) * (W + 2 * pad) * sizeof(float);
    static const int rows3_env = getenv("NMSA_NMS_ROWS3") ? atoi(getenv("NMSA_NMS_ROWS3")) : 1;
    if (pad == 1 && (W % 32) == 0 && rows3_env) {
        // rows per wave: 4 (6 row loads in flight, 1.5x the bytes requested, mostly L2 hits on the
        // neighbour's halo) measured 13.8-14.0 us vs 15.8 (6 rows), 17.6-17.8 (8), 20.5 (12),
        // 22.8 (16) at B=32 640x480: the kernel is a one-shot bound by latency and wave count
        static const int nr = getenv("NMSA_NMS_NR") ? atoi(getenv("NMSA_NMS_NR")) : 4;
        static const int half_env = getenv("NMSA_NMS_HALF") ? atoi(getenv("NMSA_NMS_HALF")) : -1;
        const bool half = half_env >= 0 ? (half_env != 0 && W % 128 == 0) : (W % 256 != 0 && W % 128 == 0);
        static const int xcd = getenv("NMSA_NMS_XCD") ? atoi(getenv("NMSA_NMS_XCD")) : 1;
        const int nx = device_geometry().xcds;
#define NMSA_ROWS3(NR) do {                                                                                 \
        const int waves = half ? (W / 128) * ((H + 2 * NR - 1) / (2 * NR)) : ((W + 255) / 256) * ((H + NR - 1) / NR); \
        const int bpi = (waves + 3) / 4;                                                                    \
        const long long nb = (long long)bpi * B;                                                            \
        const bool remap = xcd && nx > 1 && nb < (1ll << 30);                                               \
        const dim3 grid_ = remap ? dim3((unsigned)(((nb + nx - 1) / nx) * nx)) : dim3(bpi, B);              \
        if (half) hipLaunchKernelGGL((k_nms_rows3<NR, true>), grid_, dim3(256), 0, stream, center, bits, H, W, \
                                     words, threshold, bpi, remap ? (int)nb : 0, nx);                        \
        else hipLaunchKernelGGL((k_nms_rows3<NR, false>), grid_, dim3(256), 0, stream, center, bits, H, W,   \
                                words, threshold, bpi, remap ? (int)nb : 0, nx); } while (0)
        if (nr == 8) NMSA_ROWS3(8); else NMSA_ROWS3(4);
#undef NMSA_ROWS3
    } else if ((W % 32) == 0 && pad <= NMS_PAD_MAX && strip_lds <= 64 * 1024) {
        dim3 grid((H + R - 1) / R, B);
        // one thread per column when the row fits a workgroup (every wave fully used)
        static const int thr_env = getenv("NMSA_NMS_THREADS") ? atoi(getenv("NMSA_NMS_THREADS")) : 0;
        const int strip_threads = thr_env > 0 ? thr_env : 256;      // W threads measured equal
#define NMSA_STRIP(PADV) hipLaunchKernelGGL(k_nms_strip<PADV>, grid, dim3(strip_threads), strip_lds, stream, \
                                            center, bits, H, W, R, words, threshold)
        switch (pad) {
            case 0: NMSA_STRIP(0); break;
            case 1: NMSA_STRIP(1); break;
            case 2: NMSA_STRIP(2); break;
            case 3: NMSA_STRIP(3); break;
            default: NMSA_STRIP(4); break;
        }
#undef NMSA_STRIP
    } else {
        rc = check_hip(hipMemsetAsync(bits, 0, (size_t)B * words * sizeof(uint32_t), stream));
        if (rc) return rc;
        dim3 grid((W + NMS_TW - 1) / NMS_TW, (H + NMS_TH - 1) / NMS_TH, B);
        if (pad <= NMS_PAD_MAX)
            hipLaunchKernelGGL(k_nms_candidates<true>, grid, dim3(256), 0, stream,
                               center, bits, H, W, words, threshold, pad);
        else
            hipLaunchKernelGGL(k_nms_candidates<false>, grid, dim3(256), 0, stream,
                               center, bits, H, W, words, threshold, pad);
    }
    rc = check_launch();
    if (rc) return rc;
    // the LDS copy of the candidate mask needs a grant above 64 KB for large images; without
    // it (denied, another device) the global-memory variant runs instead
    // + the candidate list (position, key) behind the mask when both fit: SEL_LIST_CAP entries
    const int list_cap = (words + 2 * SEL_LIST_CAP <= SEL_LDS_WORDS) ? SEL_LIST_CAP : 0;
    const size_t sel_lds = ((size_t)words + 2 * (size_t)list_cap) * sizeof(uint32_t);
    if (words <= SEL_LDS_WORDS &&
        (sel_lds <= 48 * 1024 ||
         allow_dynamic_lds(k_select_compact<true>, SEL_LDS_WORDS * sizeof(uint32_t)) == NMSA_OK)) {
        hipLaunchKernelGGL(k_select_compact<true>, dim3(B), dim3(SEL_THREADS), sel_lds,
                           stream, center, apply_fg ? fg : nullptr, bits, H, W, words, topk, apply_fg,
                           max_centers, centers_yx, n_centers, scores, center_mask, list_cap);
    } else {
        hipLaunchKernelGGL(k_select_compact<false>, dim3(B), dim3(SEL_THREADS), 0, stream,
                           center, apply_fg ? fg : nullptr, bits, H, W, words, topk, apply_fg,
                           max_centers, centers_yx, n_centers, scores, center_mask, 0);
    }
    return check_launch();
}
