// resize.hip — the full-resolution step on gfx950: crop to the valid region and
// resize to the dataset resolution.
//
// Replaces, behind the C ABI of include/nmsa.h,
//   DensePostprocessingBase._crop_to_valid_region_and_resize_prediction
//                                                    (dense_base.py:15-58)
//   SemanticPostprocessing fullres softmax/argmax    (semantic.py:61-80)
//
//   k_resize_nearest    id / label / score maps: out[y,x] = in[y0+sy(y), x0+sx(x)]
//   k_resize_bilinear   logits at full resolution (only when somebody reads them)
//   k_argmax_resized    bilinear resize + class argmax + max-softmax score in ONE pass:
//                       the [B,C,Ho,Wo] full-resolution logits are never written
//
// Exactness: the arithmetic is the one of ATen's CPU kernels (the reference's
// `.cpu()` evaluation path), reverse-engineered against F.interpolate and pinned by
// tests/golden/fullres_cases.npz:
//   nearest : src = min(int(floorf(dst * scale)), in - 1),        scale = float(in)/float(out)
//   bilinear: s = max(fmaf(scale, dst + 0.5f, -0.5f), 0);  i0 = min(int(s), in-1);
//             i1 = min(i0 + 1, in - 1);  w1 = clamp(s - i0, 0, 1);  w0 = 1 - w1
//             t0 = fmaf(a, wx0, b*wx1); t1 = fmaf(c, wx0, d*wx1);  (width first)
//             out = fmaf(t0, wy0, t1*wy1)
//   (upsample_generic_Nd_kernel_impl — what ATen runs unless Ho + Wo <= 128, where it
//   dispatches to its channels-last kernel: a 4-weight form whose rounding depends on the
//   host's SIMD width)
// Integer maps take the reference's float32 round trip (dense_base.py:38-40), which is
// the identity below 2^24 and reproduced above it.
#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include <cmath>
#include "nmsa_common.hpp"
#include "argmax_state.hpp"

namespace nmsa {
namespace {

struct CropResize {
    int Hs, Ws;        // source plane size
    int y0, x0, h, w;  // valid region inside the source plane
    int Ho, Wo;        // output plane size
    float sy, sx;      // float(h)/float(Ho), float(w)/float(Wo)
};

__device__ __forceinline__ int nearest_src(float scale, int dst, int in)
{
    return min((int)floorf(__fmul_rn((float)dst, scale)), in - 1);
}

__device__ __forceinline__ void bilinear_src(float scale, int dst, int in,
                                             int& i0, int& i1, float& w0, float& w1)
{
    float s = __fmaf_rn(scale, __fadd_rn((float)dst, 0.5f), -0.5f);
    s = (s < 0.f) ? 0.f : s;
    i0 = min((int)s, in - 1);
    i1 = min(i0 + 1, in - 1);
    w1 = fminf(fmaxf(__fsub_rn(s, (float)i0), 0.f), 1.f);
    w0 = __fsub_rn(1.0f, w1);
}

__device__ __forceinline__ float bilerp(float a, float b, float c, float d,
                                        float wx0, float wx1, float wy0, float wy1)
{
    const float t0 = __fmaf_rn(a, wx0, __fmul_rn(b, wx1));
    const float t1 = __fmaf_rn(c, wx0, __fmul_rn(d, wx1));
    return __fmaf_rn(t0, wy0, __fmul_rn(t1, wy1));
}

template <int DTYPE>
__device__ __forceinline__ float ld_elem(const void* p, size_t i)
{
    if (DTYPE == NMSA_F32) return ((const float*)p)[i];
    const uint16_t h = ((const uint16_t*)p)[i];
    return (DTYPE == NMSA_BF16) ? bf16_to_f32(h) : f16_to_f32(h);
}

// the two horizontal neighbours (a = row[ix0], b = row[ix1]) with ONE load: ix1 == ix0 + 1
// everywhere but on the clamped right edge, where both are the last pixel.  `xb` =
// min(ix0, w - 2) is the pair's base, `edge` = (ix0 == w - 1).  Needs w >= 2.
template <int DTYPE>
__device__ __forceinline__ void ld_pair(const void* p, size_t i, bool edge, float& a, float& b)
{
    float lo, hi;
    if (DTYPE == NMSA_F32) {
        typedef float f32x2_t __attribute__((ext_vector_type(2), aligned(4)));
        const f32x2_t v = *(const f32x2_t*)((const float*)p + i);
        lo = v.x; hi = v.y;
    } else {
        typedef unsigned short u16x2_t __attribute__((ext_vector_type(2), aligned(2)));
        const u16x2_t v = *(const u16x2_t*)((const uint16_t*)p + i);
        lo = (DTYPE == NMSA_BF16) ? bf16_to_f32(v.x) : f16_to_f32(v.x);
        hi = (DTYPE == NMSA_BF16) ? bf16_to_f32(v.y) : f16_to_f32(v.y);
    }
    a = edge ? hi : lo;
    b = hi;
}

// round-to-nearest-even to the storage type of the logits (what the reference's
// F.interpolate returns for a bf16 / f16 input), kept as float
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float v)
{
    uint32_t u = __float_as_uint(v);
    if (v != v) return 0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// f32 -> f16 bits, rounding the f32 VALUE (round-to-nearest-even).  The reference rounds an
// interpolated value to f32 first and to f16 afterwards; without the opaque copy the compiler
// folds the producing fma and the conversion into v_fma_mixlo_f16, which rounds once and lands
// on the other side of double-rounding ties.
__device__ __forceinline__ uint16_t f32_to_f16_bits(float v)
{
    float r = v;
    asm volatile("" : "+v"(r));
    return __builtin_bit_cast(uint16_t, (_Float16)r);
}

template <int DTYPE>
__device__ __forceinline__ float round_to_storage(float v)
{
    if (DTYPE == NMSA_F32) return v;
    if (DTYPE == NMSA_BF16) return bf16_to_f32(f32_to_bf16_bits(v));
    return f16_to_f32(f32_to_f16_bits(v));
}

// XCD-aware tile order: consecutive workgroup ids go round-robin over the 8 XCDs (each
// with its own L2); give every XCD one contiguous range of tiles so that neighbouring
// output tiles — which share source rows — hit the same L2.
__device__ __forceinline__ long long xcd_contiguous_tile(long long wg, long long n_tiles)
{
    const long long per_xcd = (n_tiles + 7) / 8;
    return (wg & 7) * per_xcd + (wg >> 3);
}

// =================================================================================
// nearest: 4 consecutive output pixels of one row per thread
// =================================================================================
template <typename T, bool VIA_F32>
__device__ __forceinline__ T through_f32(T v)
{
    if (VIA_F32) return (T)(float)v;      // dense_base.py:38-40 / :52
    return v;
}

template <typename T, bool VIA_F32, bool VEC>
__global__ __launch_bounds__(256) void k_resize_nearest(
    const T* __restrict__ src, T* __restrict__ dst, CropResize g, int planes)
{
    const int Q = (g.Wo + 3) >> 2;
    const long long total = (long long)planes * g.Ho * Q;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int q = (int)(t % Q);
    const long long row = t / Q;
    const int y = (int)(row % g.Ho);
    const long long plane = row / g.Ho;
    const int iy = g.y0 + nearest_src(g.sy, y, g.h);
    const T* s = src + ((size_t)plane * g.Hs + iy) * g.Ws + g.x0;
    T* d = dst + (size_t)row * g.Wo + q * 4;
    T v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = min(q * 4 + j, g.Wo - 1);
        v[j] = through_f32<T, VIA_F32>(s[nearest_src(g.sx, x, g.w)]);
    }
    if (VEC) {
        typedef T vec4_t __attribute__((ext_vector_type(4)));
        vec4_t o;
        o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
        *(vec4_t*)d = o;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (q * 4 + j < g.Wo) d[j] = v[j];
    }
}

// =================================================================================
// bilinear (align_corners=False), output in the input's storage type
// =================================================================================
// One thread = 4 consecutive output pixels of one row, for RB_PLANES consecutive planes:
// the index / weight arithmetic is paid once and reused for every plane.
constexpr int RB_PLANES = 8;

template <int DTYPE, bool VEC>
__device__ __forceinline__ void store_px4(void* dst, size_t o, const float v[4], int nvalid)
{
    if (DTYPE == NMSA_F32) {
        float* d = (float*)dst + o;
        if (VEC) {
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            f32x4_t ov;
            ov.x = v[0]; ov.y = v[1]; ov.z = v[2]; ov.w = v[3];
            __builtin_nontemporal_store(ov, (f32x4_t*)d);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < nvalid) d[j] = v[j];
        }
    } else {
        uint16_t* d = (uint16_t*)dst + o;
        uint16_t hbits[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            hbits[j] = (DTYPE == NMSA_BF16) ? f32_to_bf16_bits(v[j])
                                            : f32_to_f16_bits(v[j]);
        if (VEC) {
            typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
            u16x4_t ov;
            ov.x = hbits[0]; ov.y = hbits[1]; ov.z = hbits[2]; ov.w = hbits[3];
            __builtin_nontemporal_store(ov, (u16x4_t*)d);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < nvalid) d[j] = hbits[j];
        }
    }
}

template <int DTYPE, bool VEC, bool PAIR>
__global__ __launch_bounds__(256) void k_resize_bilinear(
    const void* __restrict__ src, void* __restrict__ dst, CropResize g, int planes)
{
    const int Q = (g.Wo + 3) >> 2;
    const int plane_groups = (planes + RB_PLANES - 1) / RB_PLANES;
    const long long total = (long long)plane_groups * g.Ho * Q;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int q = (int)(t % Q);
    const long long row = t / Q;
    const int y = (int)(row % g.Ho);
    const int p_begin = (int)(row / g.Ho) * RB_PLANES;
    const int p_end = min(p_begin + RB_PLANES, planes);
    int iy0, iy1;
    float wy0, wy1;
    bilinear_src(g.sy, y, g.h, iy0, iy1, wy0, wy1);
    int xo[4], xo1[4];
    bool edge[4];
    float wx0[4], wx1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = min(q * 4 + j, g.Wo - 1);
        int ix0, ix1;
        bilinear_src(g.sx, x, g.w, ix0, ix1, wx0[j], wx1[j]);
        edge[j] = PAIR && (ix0 == g.w - 1);
        xo[j] = PAIR ? min(ix0, g.w - 2) : ix0;
        xo1[j] = ix1;
    }
    const size_t plane_stride = (size_t)g.Hs * g.Ws;
    size_t r0 = (size_t)p_begin * plane_stride + (size_t)(g.y0 + iy0) * g.Ws + g.x0;
    size_t r1 = (size_t)p_begin * plane_stride + (size_t)(g.y0 + iy1) * g.Ws + g.x0;
    size_t o = ((size_t)p_begin * g.Ho + y) * g.Wo + q * 4;
    const int nvalid = min(4, g.Wo - q * 4);
    for (int p = p_begin; p < p_end; ++p) {
        float a[4], b[4], c[4], d[4], v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (PAIR) {
                ld_pair<DTYPE>(src, r0 + xo[j], edge[j], a[j], b[j]);
                ld_pair<DTYPE>(src, r1 + xo[j], edge[j], c[j], d[j]);
            } else {
                a[j] = ld_elem<DTYPE>(src, r0 + xo[j]);
                b[j] = ld_elem<DTYPE>(src, r0 + xo1[j]);
                c[j] = ld_elem<DTYPE>(src, r1 + xo[j]);
                d[j] = ld_elem<DTYPE>(src, r1 + xo1[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = bilerp(a[j], b[j], c[j], d[j], wx0[j], wx1[j], wy0, wy1);
        store_px4<DTYPE, VEC>(dst, o, v, nvalid);
        r0 += plane_stride;
        r1 += plane_stride;
        o += (size_t)g.Ho * g.Wo;
    }
}

// =================================================================================
// fused bilinear resize + argmax + score: one output pixel per lane, 64 x 4 px tiles
// =================================================================================
constexpr int RZ_TW = 64, RZ_TH = 4, RZ_UNROLL = 4;

template <int DTYPE>
__device__ float2 resized_column_exact(
    const void* logits, size_t o00, size_t o01, size_t o10, size_t o11, size_t plane_stride,
    int C, float wx0, float wx1, float wy0, float wy1);

template <int DTYPE, bool WITH_SCORE, bool PAIR>
__global__ __launch_bounds__(RZ_TW * RZ_TH) void k_argmax_resized(
    const void* __restrict__ logits, CropResize g, int C, int tiles_x, int tiles_y,
    long long n_tiles,
    uint8_t* __restrict__ idx_u8, int64_t* __restrict__ idx_i64, float* __restrict__ score)
{
    const long long tile = xcd_contiguous_tile(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;
    const int tx = (int)(tile % tiles_x);
    const long long r = tile / tiles_x;
    const int ty = (int)(r % tiles_y);
    const int b = (int)(r / tiles_y);
    const int x = tx * RZ_TW + (threadIdx.x & (RZ_TW - 1));
    const int y = ty * RZ_TH + (threadIdx.x / RZ_TW);
    const bool valid = x < g.Wo && y < g.Ho;

    int ix0, ix1, iy0, iy1;
    float wx0, wx1, wy0, wy1;
    bilinear_src(g.sx, min(x, g.Wo - 1), g.w, ix0, ix1, wx0, wx1);
    bilinear_src(g.sy, min(y, g.Ho - 1), g.h, iy0, iy1, wy0, wy1);
    const size_t plane_stride = (size_t)g.Hs * g.Ws;
    const size_t img = (size_t)b * C * plane_stride;
    const size_t o00 = img + (size_t)(g.y0 + iy0) * g.Ws + g.x0 + ix0;
    const size_t o01 = img + (size_t)(g.y0 + iy0) * g.Ws + g.x0 + ix1;
    const size_t o10 = img + (size_t)(g.y0 + iy1) * g.Ws + g.x0 + ix0;
    const size_t o11 = img + (size_t)(g.y0 + iy1) * g.Ws + g.x0 + ix1;
    // PAIR: both horizontal neighbours with one load (see ld_pair)
    const bool edge = PAIR && (ix0 == g.w - 1);
    const int xb = PAIR ? min(ix0, g.w - 2) : 0;
    const size_t q0 = img + (size_t)(g.y0 + iy0) * g.Ws + g.x0 + xb;
    const size_t q1 = img + (size_t)(g.y0 + iy1) * g.Ws + g.x0 + xb;

    ArgmaxState st;
    argmax_init(st);
    int c = 0;
    for (; c + RZ_UNROLL <= C; c += RZ_UNROLL) {
        float a[RZ_UNROLL], bb[RZ_UNROLL], cc[RZ_UNROLL], d[RZ_UNROLL];
#pragma unroll
        for (int u = 0; u < RZ_UNROLL; ++u) {
            const size_t pc = (size_t)(c + u) * plane_stride;
            if (PAIR) {
                ld_pair<DTYPE>(logits, pc + q0, edge, a[u], bb[u]);
                ld_pair<DTYPE>(logits, pc + q1, edge, cc[u], d[u]);
            } else {
                a[u] = ld_elem<DTYPE>(logits, pc + o00);
                bb[u] = ld_elem<DTYPE>(logits, pc + o01);
                cc[u] = ld_elem<DTYPE>(logits, pc + o10);
                d[u] = ld_elem<DTYPE>(logits, pc + o11);
            }
        }
#pragma unroll
        for (int u = 0; u < RZ_UNROLL; ++u)
            argmax_step<WITH_SCORE>(st, 0, round_to_storage<DTYPE>(
                bilerp(a[u], bb[u], cc[u], d[u], wx0, wx1, wy0, wy1)), c + u);
    }
    for (; c < C; ++c) {
        const size_t pc = (size_t)c * plane_stride;
        argmax_step<WITH_SCORE>(st, 0, round_to_storage<DTYPE>(bilerp(
            ld_elem<DTYPE>(logits, pc + o00), ld_elem<DTYPE>(logits, pc + o01),
            ld_elem<DTYPE>(logits, pc + o10), ld_elem<DTYPE>(logits, pc + o11),
            wx0, wx1, wy0, wy1)), c);
    }
    if (!valid) return;
    // non-finite column, or a maximum small enough for two classes to share its probability
    // (panoptic.hip): exact re-evaluation
    int cls = st.am[0];
    float sc = WITH_SCORE ? (1.0f / st.se[0]) : 0.f;
    if (st.nf[0] != st.nf[0] || may_tie_in_probability<DTYPE>(st.m[0])) {
        const float2 ex = resized_column_exact<DTYPE>(logits, o00, o01, o10, o11, plane_stride,
                                                      C, wx0, wx1, wy0, wy1);
        sc = ex.x;
        cls = __float_as_int(ex.y);
    }
    const size_t o = ((size_t)b * g.Ho + y) * g.Wo + x;
    if (idx_u8) idx_u8[o] = (uint8_t)cls;
    if (idx_i64) idx_i64[o] = cls;
    if (WITH_SCORE) score[o] = sc;
}

// =================================================================================
// LDS-staged tiles (upscaling, the common case: network 480x640 -> dataset resolution)
// =================================================================================
// The gather kernels above issue 4 scattered loads per output pixel and class and are bound
// by the texture-address path, not by HBM.  When upscaling, a 64 x 16 output tile reads a
// source footprint of at most ~(63*sx + 3) x (15*sy + 3) <= 67 x 19 pixels per class: stage
// it once per chunk of 4 classes, then interpolate out of LDS.
// Staging is LDS-DMA in 16-byte pieces (`global_load_lds_dwordx4`: no VGPR round trip, 1 KiB
// per wave-instruction, ONE instruction per thread and class): the footprint window is
// widened to whole pieces (P = 4 or 8 elements x ceil(SWt / ..)) and, at the right image
// border, shifted left so that it never leaves the source row; piece e = row * (P/EPP) + q
// lands at LDS byte 16*e — lane-linear per wave, as the DMA requires — so the LDS image is
// the window itself with pitch P.  Two LDS buffers: chunk i+1 is in flight while chunk i is
// interpolated, one barrier per chunk.  Lane = output x (conflict-free LDS reads), each
// thread owns 4 consecutive output rows.
#ifndef NMSA_LT_TH
#define NMSA_LT_TH 16
#endif
#ifndef NMSA_LT_TW
#define NMSA_LT_TW 64
#endif
// a wave = 64 output columns x 4 output rows; a workgroup = (LT_TW / 64) x (LT_TH / 4) waves
constexpr int LT_TW = NMSA_LT_TW, LT_TH = NMSA_LT_TH, LT_THREADS = NMSA_LT_TW * NMSA_LT_TH / 4;
static_assert(LT_TW % 64 == 0 && LT_TH % 4 == 0 && LT_THREADS <= 1024, "tile of whole waves");
// classes per chunk and LDS buffers of the ring (k_resized_tile): 4 x 2 for the score mode
// (groups of four classes), 2 x 4 otherwise — the same bytes of LDS either way
#ifndef NMSA_LT_CH
#define NMSA_LT_CH 4
#define NMSA_LT_NB 2
#endif
__host__ __device__ constexpr int lt_ch(int mode) { return mode == 1 ? 4 : NMSA_LT_CH; }
__host__ __device__ constexpr int lt_nb(int mode) { return mode == 1 ? 2 : NMSA_LT_NB; }
constexpr int LT_MODE_ARGMAX = 0, LT_MODE_ARGMAX_SCORE = 1, LT_MODE_MATERIALISE = 2;

// cache policy of the staging DMAs (aux operand: 0 default, 2 = nt, non-temporal)
#ifndef NMSA_LT_AUX
#define NMSA_LT_AUX 0
#endif
__device__ __forceinline__ void glds_piece(const void* gsrc, void* lds_wave_base)
{
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)lds_wave_base, 16, 0, NMSA_LT_AUX);
}

template <int DTYPE>
__device__ __forceinline__ float lds_elem(const void* plane, int i)
{
    if (DTYPE == NMSA_F32) return ((const float*)plane)[i];
    const uint16_t h = ((const uint16_t*)plane)[i];
    return (DTYPE == NMSA_BF16) ? bf16_to_f32(h) : f16_to_f32(h);
}

// Exact re-evaluation of one output column (rare: the fast path's denominator came out NaN):
// softmax-then-max semantics of semantic.py:74-75 incl. the degenerate columns.
// returns (score, class index as int bits) by value: no scratch slot
template <int DTYPE>
__device__ __noinline__ float2 resized_column_exact(
    const void* logits, size_t o00, size_t o01, size_t o10, size_t o11, size_t plane_stride,
    int C, float wx0, float wx1, float wy0, float wy1)
{
    bool nan_or_pinf = false, any_finite = false;
    float m = -INFINITY;
    int am = 0;
    for (int c = 0; c < C; ++c) {
        const size_t pc = (size_t)c * plane_stride;
        const float v = round_to_storage<DTYPE>(bilerp(
            ld_elem<DTYPE>(logits, pc + o00), ld_elem<DTYPE>(logits, pc + o01),
            ld_elem<DTYPE>(logits, pc + o10), ld_elem<DTYPE>(logits, pc + o11),
            wx0, wx1, wy0, wy1));
        if (v != v || v == INFINITY) nan_or_pinf = true;
        if (fabsf(v) < INFINITY) any_finite = true;
        if (v > m) { m = v; am = c; }
    }
    if (nan_or_pinf || !any_finite) return make_float2(__int_as_float(0x7fc00000), __int_as_float(0));
    auto ld = [&](int c) -> float {
        const size_t pc = (size_t)c * plane_stride;
        return round_to_storage<DTYPE>(bilerp(
            ld_elem<DTYPE>(logits, pc + o00), ld_elem<DTYPE>(logits, pc + o01),
            ld_elem<DTYPE>(logits, pc + o10), ld_elem<DTYPE>(logits, pc + o11),
            wx0, wx1, wy0, wy1));
    };
    float se = 0.f;
    int first = am;           // lowest earlier class that may share the maximum's probability
    for (int c = 0; c < C; ++c) {
        const float v = ld(c);
        se += (v == -INFINITY) ? 0.f : __expf(v - m);
        if (c < first && __fsub_rn(v, m) >= TIE_CANDIDATE_GAP) first = c;
    }
    if (first >= am) return make_float2(1.0f / se, __int_as_float(am));
    float pm;                 // decided with the reference's own arithmetic (argmax_state.hpp)
    const int cls = class_by_probability(ld, C, m, am, first, &pm);
    return make_float2(pm, __int_as_float(cls));
}

template <int DTYPE, int MODE, int K16>
__global__ __launch_bounds__(LT_THREADS) void k_resized_tile(
    const void* __restrict__ src, CropResize g, int planes, int group, int tiles_x, int tiles_y,
    long long n_tiles, int pra_lg,
    uint8_t* __restrict__ idx_u8, int64_t* __restrict__ idx_i64, float* __restrict__ score,
    void* __restrict__ dst)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // a ring of LT_NB LDS buffers of LT_CH classes each: the chunks i+1 .. i+LT_NB-1 are in
    // flight (LDS-DMA) while chunk i is interpolated.  Round 5: 2 classes x 4 buffers instead of
    // 4 x 2 (the same 32 KB: occupancy unchanged) — three chunks = 6 classes ahead instead of
    // one = 4, and a wave waits for the DMAs of chunk i only (counted vmcnt), not for everything
    // it has issued; the score mode keeps groups of four classes (argmax_group4_score)
    constexpr int LT_CH = lt_ch(MODE), LT_NB = lt_nb(MODE);
    constexpr int ESZ = (DTYPE == NMSA_F32) ? 4 : 2;   // bytes per element
    constexpr int EPP = 16 / ESZ;                      // elements per 16-byte piece
    constexpr int PLANE = K16 * LT_THREADS * EPP;      // elements per staged class plane
    const long long tile = xcd_contiguous_tile(blockIdx.x, n_tiles);
    if (tile >= n_tiles) return;                       // block-uniform
    const int tx = (int)(tile % tiles_x);
    const long long rr = tile / tiles_x;
    const int ty = (int)(rr % tiles_y);
    const int grp = (int)(rr / tiles_y);
    const int p_begin = grp * group;
    const int C = min(group, planes - p_begin);        // planes (classes) of this tile
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wvx = wv % (LT_TW / 64), wvy = wv / (LT_TW / 64);       // the wave's place in the tile

    // ---- tile footprint in the source (block-uniform) --------------------------------------
    int xs0, xs1, ys0, ys1, ti;
    float tf0, tf1;
    bilinear_src(g.sx, min(tx * LT_TW, g.Wo - 1), g.w, xs0, ti, tf0, tf1);
    bilinear_src(g.sx, min(tx * LT_TW + LT_TW - 1, g.Wo - 1), g.w, ti, xs1, tf0, tf1);
    bilinear_src(g.sy, min(ty * LT_TH, g.Ho - 1), g.h, ys0, ti, tf0, tf1);
    bilinear_src(g.sy, min(ty * LT_TH + LT_TH - 1, g.Ho - 1), g.h, ti, ys1, tf0, tf1);
    const int SHt = ys1 - ys0 + 1;
    // Two window shapes (plan_tiles decides, block-uniform):
    //  * row-aligned (pra_lg > 0): 2^pra_lg lanes per window row, the window starts on a 16-byte
    //    boundary of the source row — every wave DMA instruction is 64 / 2^pra_lg WHOLE row
    //    segments of aligned 16-byte pieces (no piece straddles a 16-byte boundary, no row break
    //    inside a segment of lanes), the LDS pitch is a power of two;
    //  * packed (pra_lg == 0): the window starts at the first source column, pieces in row-major
    //    order over ceil(width / EPP) pieces per row (sources whose rows are not 16-byte
    //    multiples, windows wider than 16 pieces)
    const int PR = pra_lg ? (1 << pra_lg) : (xs1 - xs0 + EPP) / EPP;      // pieces per window row
    const int P = PR * EPP;                            // window width = LDS pitch (elements)
    // window start relative to the crop's first column (negative in the aligned shape when the
    // crop itself starts off a 16-byte boundary); never past the row end
    const int xa = pra_lg ? min(((g.x0 + xs0) / EPP) * EPP, g.Ws - P) - g.x0
                          : min(xs0, g.Ws - g.x0 - P);
    const int n_pieces = PR * SHt;                     // <= K16 * 256 (host-checked bound)

    // ---- this thread's 4 output pixels ---------------------------------------------------------
    const int x = tx * LT_TW + wvx * 64 + lane;
    int ix0, ix1;
    float wx0, wx1;
    bilinear_src(g.sx, min(x, g.Wo - 1), g.w, ix0, ix1, wx0, wx1);
    const int cx0 = ix0 - xa, cx1 = ix1 - xa;
    int r0[4], r1[4], yy[4];
    float wy0[4], wy1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        yy[j] = ty * LT_TH + wvy * 4 + j;
        int iy0, iy1;
        bilinear_src(g.sy, min(yy[j], g.Ho - 1), g.h, iy0, iy1, wy0[j], wy1[j]);
        r0[j] = (iy0 - ys0) * P;
        r1[j] = (iy1 - ys0) * P;
    }
    // A wave owns 4 whole output rows, so the source rows of its pixels are WAVE-UNIFORM: the
    // horizontal interpolation of a source row (2 LDS reads + mul + fma per class) is shared by
    // every output row that touches it — when upscaling, the 8 (row0, row1) slots of the 4
    // output rows name only 3-5 distinct source rows.  The vertical step keeps ATen's order
    // (width first, then `fma(t0, wy0, t1 * wy1)`), so results are unchanged.
    int ur0[4], ur1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ur0[j] = __builtin_amdgcn_readfirstlane(r0[j]);
        ur1[j] = __builtin_amdgcn_readfirstlane(r1[j]);
    }
    auto hlerp = [&](const void* L, int cc, int row) -> float {
        // cc * PLANE folds into the ds_read immediate offset
        return __fmaf_rn(lds_elem<DTYPE>(L, cc * PLANE + row + cx0), wx0,
                         __fmul_rn(lds_elem<DTYPE>(L, cc * PLANE + row + cx1), wx1));
    };

    // ---- staging slots: piece e of the window -> source offset (the tail lanes of the last
    //      slot re-load the last piece into the padding of the LDS plane) -----------------------
    int goff[K16];
#pragma unroll
    for (int k = 0; k < K16; ++k) {
        const int e = min((int)threadIdx.x + k * LT_THREADS, n_pieces - 1);
        const int r = pra_lg ? (e >> pra_lg) : e / PR, q = e - r * PR;
        goff[k] = r * g.Ws + q * EPP;
    }
    const size_t plane_stride = (size_t)g.Hs * g.Ws;
    const size_t base = (size_t)p_begin * plane_stride + (size_t)(g.y0 + ys0) * g.Ws + (size_t)(g.x0 + xa);
    unsigned char* wave_lds = lds_raw + (size_t)wv * 64 * 16;
    const int k_used = (n_pieces + LT_THREADS - 1) / LT_THREADS;

    // every thread issues exactly LT_CH * K16 DMA pieces per chunk (a class beyond the tile's
    // planes or an unused slot re-loads a valid piece: the counted waits below rely on the count)
    (void)k_used;
    auto stage = [&](int c0, int buf) {
#pragma unroll
        for (int cc = 0; cc < LT_CH; ++cc) {
            const size_t pb = base + (size_t)min(c0 + cc, C - 1) * plane_stride;
            unsigned char* L = wave_lds + (size_t)(buf * LT_CH + cc) * PLANE * ESZ;
#pragma unroll
            for (int k = 0; k < K16; ++k)
                glds_piece((const unsigned char*)src + (pb + goff[k]) * ESZ,
                           L + (size_t)k * LT_THREADS * 16);
        }
    };
    ArgmaxState st;
    argmax_init(st);
    // (three buffers — two chunks in flight — were tried: 48 KB of LDS per workgroup leave 3
    // instead of 5 workgroups per CU and the kernel lost 3-11 %)
    const int n_chunks = (C + LT_CH - 1) / LT_CH;
#pragma unroll
    for (int i = 0; i < LT_NB - 1; ++i) if (i < n_chunks) stage(i * LT_CH, i);
    int buf = 0;
    for (int ci = 0; ci < n_chunks; ++ci, buf = (buf + 1 == LT_NB) ? 0 : buf + 1) {
        const int c0 = ci * LT_CH;
        const int nch = min(LT_CH, C - c0);
        // my DMAs of chunk ci have landed: all but the youngest (chunks in flight behind it) x
        // (pieces per chunk and thread) of my vector-memory operations are done ...
        const int behind = min(LT_NB - 2, n_chunks - 1 - ci);          // (wave-uniform)
        if (behind >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * LT_CH * K16) : "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * LT_CH * K16) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... and so have everybody's; the buffer chunk ci + LT_NB - 1 goes to (read LT_NB - 1
        // iterations ago... no: ONE iteration ago, chunk ci - 1) is free
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ci + LT_NB - 1 < n_chunks) stage((ci + LT_NB - 1) * LT_CH, (buf + LT_NB - 1) % LT_NB);
        const void* L = lds_raw + (size_t)buf * LT_CH * PLANE * ESZ;
        // FULL: all LT_CH classes of the chunk exist (every chunk but possibly the last) — no
        // per-class bounds checks in the unrolled body
        auto chunk = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            float h0[LT_CH], h1[LT_CH];   // horizontal interpolations of source rows have0 / have1
            int have0 = -1, have1 = -1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ur0[j] != have0) {                              // wave-uniform branches
                    if (ur0[j] == have1) {
#pragma unroll
                        for (int cc = 0; cc < LT_CH; ++cc) h0[cc] = h1[cc];
                    } else {
#pragma unroll
                        for (int cc = 0; cc < LT_CH; ++cc) h0[cc] = hlerp(L, cc, ur0[j]);
                    }
                    have0 = ur0[j];
                }
                if (ur1[j] != have1) {
                    if (ur1[j] == have0) {
#pragma unroll
                        for (int cc = 0; cc < LT_CH; ++cc) h1[cc] = h0[cc];
                    } else {
#pragma unroll
                        for (int cc = 0; cc < LT_CH; ++cc) h1[cc] = hlerp(L, cc, ur1[j]);
                    }
                    have1 = ur1[j];
                }
                float v[LT_CH];
#pragma unroll
                for (int cc = 0; cc < LT_CH; ++cc) {
                    const float val = round_to_storage<DTYPE>(
                        __fmaf_rn(h0[cc], wy0[j], __fmul_rn(h1[cc], wy1[j])));
                    v[cc] = (FULL || cc < nch) ? val : -INFINITY;  // tail chunk: stale LDS, ignored
                }
                if (MODE == LT_MODE_MATERIALISE) {
                    if (x < g.Wo && yy[j] < g.Ho) {
#pragma unroll
                        for (int cc = 0; cc < LT_CH; ++cc) {
                            if (!FULL && cc >= nch) break;
                            const size_t o = ((size_t)(p_begin + c0 + cc) * g.Ho + yy[j]) * g.Wo + x;
                            if (DTYPE == NMSA_F32) __builtin_nontemporal_store(v[cc], (float*)dst + o);
                            else if (DTYPE == NMSA_BF16) ((uint16_t*)dst)[o] = f32_to_bf16_bits(v[cc]);
                            else ((uint16_t*)dst)[o] = f32_to_f16_bits(v[cc]);
                        }
                    }
                } else if (MODE == LT_MODE_ARGMAX_SCORE) {
                    if constexpr (LT_CH == 4) argmax_group4_score(st, j, v, c0);
                } else {
#pragma unroll
                    for (int cc = 0; cc < LT_CH; ++cc)
                        if (FULL || cc < nch) argmax_step<false>(st, j, v[cc], c0 + cc);
                }
            }
        };
        if (nch == LT_CH) chunk(std::true_type{});
        else chunk(std::false_type{});
    }
    if (MODE == LT_MODE_MATERIALISE) return;
    if (x >= g.Wo) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (yy[j] >= g.Ho) continue;
        int cls = st.am[j];
        float sc = 0.f;
        const bool suspicious = ((MODE == LT_MODE_ARGMAX_SCORE) ? (st.se[j] != st.se[j])
                                                                : (st.nf[j] != st.nf[j])) ||
                                may_tie_in_probability<DTYPE>(st.m[j]);
        if (MODE == LT_MODE_ARGMAX_SCORE) sc = 1.0f / st.se[j];
        if (suspicious) {
            const size_t img = (size_t)p_begin * plane_stride;
            const size_t q0 = img + (size_t)(g.y0 + ys0 + r0[j] / P) * g.Ws + g.x0;
            const size_t q1 = img + (size_t)(g.y0 + ys0 + r1[j] / P) * g.Ws + g.x0;
            const float2 ex = resized_column_exact<DTYPE>(src, q0 + ix0, q0 + ix1, q1 + ix0, q1 + ix1,
                                                          plane_stride, C, wx0, wx1, wy0[j], wy1[j]);
            sc = ex.x;
            cls = __float_as_int(ex.y);
        }
        const size_t o = ((size_t)grp * g.Ho + yy[j]) * g.Wo + x;
        if (idx_u8) idx_u8[o] = (uint8_t)cls;
        if (idx_i64) idx_i64[o] = cls;
        if (MODE == LT_MODE_ARGMAX_SCORE) score[o] = sc;
    }
}

// bilinear_src's source indices on the host: the same IEEE operations (fmaf / single adds), so
// the host sees exactly the windows the kernel is going to compute
void bilinear_src_host(float scale, int dst, int in, int& i0, int& i1)
{
    float s = fmaf(scale, (float)dst + 0.5f, -0.5f);
    s = (s < 0.f) ? 0.f : s;
    i0 = std::min((int)s, in - 1);
    i1 = std::min(i0 + 1, in - 1);
}

// How k_resized_tile stages this resize: k16 = staging slots per thread and class (1 or 2 pieces
// of 16 bytes; 0: the gather kernels must be used — downscaling: the footprint of a tile does not
// fit the staging slots; or a source row narrower than one window), pra_lg = log2 of the lanes
// per window row of the row-aligned shape (0: the packed shape).
struct TilePlan { int k16, pra_lg; };

TilePlan plan_tiles(const CropResize& g, int elem_bytes, const void* src)
{
    if (getenv("NMSA_RESIZE_NO_LDS")) return {0, 0};
    const int epp = 16 / elem_bytes;
    // row-aligned shape: rows of whole 16-byte pieces from a 16-byte aligned base; the exact
    // largest window of any tile (columns counted from the 16-byte boundary at or below its first
    // source column) must fit 8 or 16 pieces, its rows the staging slots
    if (!getenv("NMSA_RESIZE_PACKED_STAGING") && ((size_t)g.Ws * elem_bytes) % 16 == 0 &&
        (uintptr_t)src % 16 == 0) {
        const int tiles_x = (g.Wo + LT_TW - 1) / LT_TW, tiles_y = (g.Ho + LT_TH - 1) / LT_TH;
        int wmax = 0, hmax = 0, a, b, t;
        for (int tx = 0; tx < tiles_x; ++tx) {
            bilinear_src_host(g.sx, std::min(tx * LT_TW, g.Wo - 1), g.w, a, t);
            bilinear_src_host(g.sx, std::min(tx * LT_TW + LT_TW - 1, g.Wo - 1), g.w, t, b);
            wmax = std::max(wmax, g.x0 + b - ((g.x0 + a) / epp) * epp + 1);
        }
        for (int ty = 0; ty < tiles_y; ++ty) {
            bilinear_src_host(g.sy, std::min(ty * LT_TH, g.Ho - 1), g.h, a, t);
            bilinear_src_host(g.sy, std::min(ty * LT_TH + LT_TH - 1, g.Ho - 1), g.h, t, b);
            hmax = std::max(hmax, b - a + 1);
        }
        for (int lg = 3; lg <= 4; ++lg) {
            const int pr = 1 << lg;
            if (wmax > pr * epp || g.Ws < pr * epp) continue;
            if (hmax * pr <= LT_THREADS) return {1, lg};
            if (hmax * pr <= 2 * LT_THREADS) return {2, lg};
        }
    }
    // packed shape: ix1(last) - ix0(first) + 1 <= 63*sx + 3 (+1 slack for the float rounding of the indices)
    const long long sw = (long long)((float)(LT_TW - 1) * g.sx) + 4, sh = (long long)((LT_TH - 1) * g.sy) + 4;
    const long long pr = (sw + epp - 1) / epp;
    if (pr * sh > 2 * LT_THREADS) return {0, 0};
    if ((long long)g.Ws - g.x0 < pr * epp) return {0, 0};
    return {pr * sh > LT_THREADS ? 2 : 1, 0};
}

template <int DTYPE, int MODE, int K16>
int launch_tile_k(const void* src, const CropResize& g, int pra_lg, int planes, int group,
                  uint8_t* idx_u8, int64_t* idx_i64, float* score, void* dst, hipStream_t stream)
{
    const int tiles_x = (g.Wo + LT_TW - 1) / LT_TW;
    const int tiles_y = (g.Ho + LT_TH - 1) / LT_TH;
    const int groups = (planes + group - 1) / group;
    const long long n_tiles = (long long)tiles_x * tiles_y * groups;
    const long long blocks = ((n_tiles + 7) / 8) * 8;       // see xcd_contiguous_tile
    if (blocks > 0x7fffffffLL) return NMSA_ERR_ARG;
    const size_t lds_bytes = (size_t)K16 * LT_THREADS * 16 * lt_ch(MODE) * lt_nb(MODE);
    hipLaunchKernelGGL((k_resized_tile<DTYPE, MODE, K16>), dim3((unsigned)blocks), dim3(LT_THREADS),
                       lds_bytes, stream, src, g, planes, group, tiles_x, tiles_y, n_tiles, pra_lg,
                       idx_u8, idx_i64, score, dst);
    return check_launch();
}

template <int DTYPE, int MODE>
int launch_tile(const void* src, const CropResize& g, TilePlan plan, int planes, int group,
                uint8_t* idx_u8, int64_t* idx_i64, float* score, void* dst, hipStream_t stream)
{
    if (plan.k16 == 1)
        return launch_tile_k<DTYPE, MODE, 1>(src, g, plan.pra_lg, planes, group, idx_u8, idx_i64, score, dst, stream);
    return launch_tile_k<DTYPE, MODE, 2>(src, g, plan.pra_lg, planes, group, idx_u8, idx_i64, score, dst, stream);
}

bool bad_geometry(int planes, int Hs, int Ws, int y0, int x0, int h, int w, int Ho, int Wo)
{
    if (planes <= 0 || Hs <= 0 || Ws <= 0 || h <= 0 || w <= 0 || Ho <= 0 || Wo <= 0) return true;
    if (y0 < 0 || x0 < 0 || (int64_t)y0 + h > Hs || (int64_t)x0 + w > Ws) return true;
    if ((int64_t)Hs * Ws > ((int64_t)1 << 30) || (int64_t)Ho * Wo > ((int64_t)1 << 30)) return true;
    return false;
}

CropResize make_geometry(int Hs, int Ws, int y0, int x0, int h, int w, int Ho, int Wo)
{
    CropResize g;
    g.Hs = Hs; g.Ws = Ws; g.y0 = y0; g.x0 = x0; g.h = h; g.w = w; g.Ho = Ho; g.Wo = Wo;
    g.sy = (float)h / (float)Ho;      // ATen compute_scales_value<float>
    g.sx = (float)w / (float)Wo;
    return g;
}

template <typename T, bool VIA_F32>
int launch_nearest(const void* src, void* dst, const CropResize& g, int planes, hipStream_t stream)
{
    const long long total = (long long)planes * g.Ho * ((g.Wo + 3) / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return NMSA_ERR_ARG;
    const bool vec = (g.Wo % 4 == 0) && ((uintptr_t)dst % (4 * sizeof(T)) == 0);
    if (vec) hipLaunchKernelGGL((k_resize_nearest<T, VIA_F32, true>), dim3((unsigned)blocks), dim3(256),
                                0, stream, (const T*)src, (T*)dst, g, planes);
    else hipLaunchKernelGGL((k_resize_nearest<T, VIA_F32, false>), dim3((unsigned)blocks), dim3(256),
                            0, stream, (const T*)src, (T*)dst, g, planes);
    return check_launch();
}

template <int DTYPE>
int launch_bilinear(const void* src, void* dst, const CropResize& g, int planes, hipStream_t stream)
{
    const TilePlan tile_k = plan_tiles(g, DTYPE == NMSA_F32 ? 4 : 2, src);
    if (tile_k.k16 > 0)
        return launch_tile<DTYPE, LT_MODE_MATERIALISE>(src, g, tile_k, planes, planes < 32 ? planes : 32,
                                                       nullptr, nullptr, nullptr, dst, stream);
    const long long groups = (planes + RB_PLANES - 1) / RB_PLANES;
    const long long total = groups * g.Ho * ((g.Wo + 3) / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return NMSA_ERR_ARG;
    const size_t esz = (DTYPE == NMSA_F32) ? 4 : 2;
    const bool vec = (g.Wo % 4 == 0) && ((uintptr_t)dst % (4 * esz) == 0);
    const bool pair = g.w >= 2;
#define NMSA_LAUNCH_RB(V, PR)                                                                \
    hipLaunchKernelGGL((k_resize_bilinear<DTYPE, V, PR>), dim3((unsigned)blocks), dim3(256), \
                       0, stream, src, dst, g, planes)
    if (vec) { if (pair) NMSA_LAUNCH_RB(true, true); else NMSA_LAUNCH_RB(true, false); }
    else { if (pair) NMSA_LAUNCH_RB(false, true); else NMSA_LAUNCH_RB(false, false); }
#undef NMSA_LAUNCH_RB
    return check_launch();
}

template <int DTYPE>
int launch_argmax_resized(const void* logits, const CropResize& g, int B, int C,
                          uint8_t* idx_u8, int64_t* idx_i64, float* score, hipStream_t stream)
{
    const TilePlan tile_k = plan_tiles(g, DTYPE == NMSA_F32 ? 4 : 2, logits);
    if (tile_k.k16 > 0) {
        if (score) return launch_tile<DTYPE, LT_MODE_ARGMAX_SCORE>(logits, g, tile_k, B * C, C, idx_u8,
                                                                    idx_i64, score, nullptr, stream);
        return launch_tile<DTYPE, LT_MODE_ARGMAX>(logits, g, tile_k, B * C, C, idx_u8, idx_i64, score,
                                                  nullptr, stream);
    }
    const int tiles_x = (g.Wo + RZ_TW - 1) / RZ_TW;
    const int tiles_y = (g.Ho + RZ_TH - 1) / RZ_TH;
    const long long n_tiles = (long long)tiles_x * tiles_y * B;
    const long long blocks = ((n_tiles + 7) / 8) * 8;       // see xcd_contiguous_tile
    if (blocks > 0x7fffffffLL) return NMSA_ERR_ARG;
    const bool pair = g.w >= 2 && g.sx > 1.0f;    // paired loads pay off only when downscaling
#define NMSA_LAUNCH_RZ(S, PR)                                                               \
    hipLaunchKernelGGL((k_argmax_resized<DTYPE, S, PR>), dim3((unsigned)blocks),            \
                       dim3(RZ_TW * RZ_TH), 0, stream, logits, g, C, tiles_x, tiles_y,      \
                       n_tiles, idx_u8, idx_i64, score)
    if (score) { if (pair) NMSA_LAUNCH_RZ(true, true); else NMSA_LAUNCH_RZ(true, false); }
    else { if (pair) NMSA_LAUNCH_RZ(false, true); else NMSA_LAUNCH_RZ(false, false); }
#undef NMSA_LAUNCH_RZ
    return check_launch();
}

}  // namespace
}  // namespace nmsa

using namespace nmsa;

extern "C" int nmsa_resize_nearest(const void* src, int elem_type, int planes, int Hs, int Ws,
                                   int y0, int x0, int h, int w, int Ho, int Wo, void* dst,
                                   nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst || bad_geometry(planes, Hs, Ws, y0, x0, h, w, Ho, Wo)) return NMSA_ERR_ARG;
    const CropResize g = make_geometry(Hs, Ws, y0, x0, h, w, Ho, Wo);
    switch (elem_type) {
        case NMSA_U8: return launch_nearest<uint8_t, false>(src, dst, g, planes, stream);
        case NMSA_I16: return launch_nearest<int16_t, false>(src, dst, g, planes, stream);
        case NMSA_I32: return launch_nearest<int32_t, true>(src, dst, g, planes, stream);
        case NMSA_I64: return launch_nearest<int64_t, true>(src, dst, g, planes, stream);
        case NMSA_ELEM_F32: return launch_nearest<float, false>(src, dst, g, planes, stream);
        default: return NMSA_ERR_ARG;
    }
}

extern "C" int nmsa_resize_bilinear(const void* src, int dtype, int planes, int Hs, int Ws,
                                    int y0, int x0, int h, int w, int Ho, int Wo, void* dst,
                                    nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst || bad_geometry(planes, Hs, Ws, y0, x0, h, w, Ho, Wo)) return NMSA_ERR_ARG;
    const CropResize g = make_geometry(Hs, Ws, y0, x0, h, w, Ho, Wo);
    switch (dtype) {
        case NMSA_F32: return launch_bilinear<NMSA_F32>(src, dst, g, planes, stream);
        case NMSA_BF16: return launch_bilinear<NMSA_BF16>(src, dst, g, planes, stream);
        case NMSA_F16: return launch_bilinear<NMSA_F16>(src, dst, g, planes, stream);
        default: return NMSA_ERR_ARG;
    }
}

extern "C" int nmsa_semantic_argmax_resized(const void* logits, int logits_dtype, int B, int C,
                                            int Hs, int Ws, int y0, int x0, int h, int w,
                                            int Ho, int Wo,
                                            uint8_t* idx_u8, int64_t* idx_i64, float* score,
                                            nmsa_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!logits || C <= 0 || B <= 0 || (int64_t)B * C > 0x7fffffffLL ||
        bad_geometry(B * C, Hs, Ws, y0, x0, h, w, Ho, Wo))
        return NMSA_ERR_ARG;
    if (idx_u8 && C > 256) return NMSA_ERR_ARG;
    const CropResize g = make_geometry(Hs, Ws, y0, x0, h, w, Ho, Wo);
    switch (logits_dtype) {
        case NMSA_F32: return launch_argmax_resized<NMSA_F32>(logits, g, B, C, idx_u8, idx_i64, score, stream);
        case NMSA_BF16: return launch_argmax_resized<NMSA_BF16>(logits, g, B, C, idx_u8, idx_i64, score, stream);
        case NMSA_F16: return launch_argmax_resized<NMSA_F16>(logits, g, B, C, idx_u8, idx_i64, score, stream);
        default: return NMSA_ERR_ARG;
    }
}
