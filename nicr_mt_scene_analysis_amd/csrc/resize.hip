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
//   (upsample_generic_Nd_kernel_impl — what ATen runs for every output of >= 4096 px;
//   for tiny outputs ATen switches to a 4-weight form that differs by <= 1 ulp.)
// Integer maps take the reference's float32 round trip (dense_base.py:38-40), which is
// the identity below 2^24 and reproduced above it.
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

// round-to-nearest-even to the storage type of the logits (what the reference's
// F.interpolate returns for a bf16 / f16 input), kept as float
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float v)
{
    uint32_t u = __float_as_uint(v);
    if (v != v) return 0x7fc0;
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

template <int DTYPE>
__device__ __forceinline__ float round_to_storage(float v)
{
    if (DTYPE == NMSA_F32) return v;
    if (DTYPE == NMSA_BF16) return bf16_to_f32(f32_to_bf16_bits(v));
    return (float)(_Float16)v;
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
template <int DTYPE, bool VEC>
__global__ __launch_bounds__(256) void k_resize_bilinear(
    const void* __restrict__ src, void* __restrict__ dst, CropResize g, int planes)
{
    const int Q = (g.Wo + 3) >> 2;
    const long long total = (long long)planes * g.Ho * Q;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int q = (int)(t % Q);
    const long long row = t / Q;
    const int y = (int)(row % g.Ho);
    const long long plane = row / g.Ho;
    int iy0, iy1;
    float wy0, wy1;
    bilinear_src(g.sy, y, g.h, iy0, iy1, wy0, wy1);
    const size_t r0 = ((size_t)plane * g.Hs + g.y0 + iy0) * g.Ws + g.x0;
    const size_t r1 = ((size_t)plane * g.Hs + g.y0 + iy1) * g.Ws + g.x0;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = min(q * 4 + j, g.Wo - 1);
        int ix0, ix1;
        float wx0, wx1;
        bilinear_src(g.sx, x, g.w, ix0, ix1, wx0, wx1);
        v[j] = bilerp(ld_elem<DTYPE>(src, r0 + ix0), ld_elem<DTYPE>(src, r0 + ix1),
                      ld_elem<DTYPE>(src, r1 + ix0), ld_elem<DTYPE>(src, r1 + ix1),
                      wx0, wx1, wy0, wy1);
    }
    const size_t o = (size_t)row * g.Wo + q * 4;
    if (DTYPE == NMSA_F32) {
        float* d = (float*)dst + o;
        if (VEC) {
            typedef float f32x4_t __attribute__((ext_vector_type(4)));
            f32x4_t ov;
            ov.x = v[0]; ov.y = v[1]; ov.z = v[2]; ov.w = v[3];
            __builtin_nontemporal_store(ov, (f32x4_t*)d);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (q * 4 + j < g.Wo) d[j] = v[j];
        }
    } else {
        uint16_t* d = (uint16_t*)dst + o;
        uint16_t hbits[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            hbits[j] = (DTYPE == NMSA_BF16) ? f32_to_bf16_bits(v[j])
                                            : __builtin_bit_cast(uint16_t, (_Float16)v[j]);
        if (VEC) {
            typedef unsigned short u16x4_t __attribute__((ext_vector_type(4)));
            u16x4_t ov;
            ov.x = hbits[0]; ov.y = hbits[1]; ov.z = hbits[2]; ov.w = hbits[3];
            *(u16x4_t*)d = ov;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (q * 4 + j < g.Wo) d[j] = hbits[j];
        }
    }
}

// =================================================================================
// fused bilinear resize + argmax + score: one output pixel per lane, 64 x 4 px tiles
// =================================================================================
constexpr int RZ_TW = 64, RZ_TH = 4, RZ_UNROLL = 4;

template <int DTYPE>
__device__ __noinline__ bool resized_column_degenerate(
    const void* logits, size_t o00, size_t o01, size_t o10, size_t o11, size_t plane_stride,
    int C, float wx0, float wx1, float wy0, float wy1)
{
    bool nan_or_pinf = false, any_finite = false;
    for (int c = 0; c < C; ++c) {
        const size_t pc = (size_t)c * plane_stride;
        const float v = round_to_storage<DTYPE>(bilerp(
            ld_elem<DTYPE>(logits, pc + o00), ld_elem<DTYPE>(logits, pc + o01),
            ld_elem<DTYPE>(logits, pc + o10), ld_elem<DTYPE>(logits, pc + o11),
            wx0, wx1, wy0, wy1));
        if (v != v || v == INFINITY) nan_or_pinf = true;
        if (fabsf(v) < INFINITY) any_finite = true;
    }
    return nan_or_pinf || !any_finite;
}

template <int DTYPE, bool WITH_SCORE>
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

    ArgmaxState st;
    argmax_init(st);
    int c = 0;
    for (; c + RZ_UNROLL <= C; c += RZ_UNROLL) {
        float a[RZ_UNROLL], bb[RZ_UNROLL], cc[RZ_UNROLL], d[RZ_UNROLL];
#pragma unroll
        for (int u = 0; u < RZ_UNROLL; ++u) {
            const size_t pc = (size_t)(c + u) * plane_stride;
            a[u] = ld_elem<DTYPE>(logits, pc + o00);
            bb[u] = ld_elem<DTYPE>(logits, pc + o01);
            cc[u] = ld_elem<DTYPE>(logits, pc + o10);
            d[u] = ld_elem<DTYPE>(logits, pc + o11);
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
    bool degenerate = false;
    if (st.nf[0] != st.nf[0])
        degenerate = resized_column_degenerate<DTYPE>(logits, o00, o01, o10, o11, plane_stride,
                                                      C, wx0, wx1, wy0, wy1);
    const int cls = degenerate ? 0 : st.am[0];
    const size_t o = ((size_t)b * g.Ho + y) * g.Wo + x;
    if (idx_u8) idx_u8[o] = (uint8_t)cls;
    if (idx_i64) idx_i64[o] = cls;
    if (WITH_SCORE) score[o] = degenerate ? __int_as_float(0x7fc00000) : (1.0f / st.se[0]);
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
    const long long total = (long long)planes * g.Ho * ((g.Wo + 3) / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return NMSA_ERR_ARG;
    const size_t esz = (DTYPE == NMSA_F32) ? 4 : 2;
    const bool vec = (g.Wo % 4 == 0) && ((uintptr_t)dst % (4 * esz) == 0);
    if (vec) hipLaunchKernelGGL((k_resize_bilinear<DTYPE, true>), dim3((unsigned)blocks), dim3(256),
                                0, stream, src, dst, g, planes);
    else hipLaunchKernelGGL((k_resize_bilinear<DTYPE, false>), dim3((unsigned)blocks), dim3(256),
                            0, stream, src, dst, g, planes);
    return check_launch();
}

template <int DTYPE>
int launch_argmax_resized(const void* logits, const CropResize& g, int B, int C,
                          uint8_t* idx_u8, int64_t* idx_i64, float* score, hipStream_t stream)
{
    const int tiles_x = (g.Wo + RZ_TW - 1) / RZ_TW;
    const int tiles_y = (g.Ho + RZ_TH - 1) / RZ_TH;
    const long long n_tiles = (long long)tiles_x * tiles_y * B;
    const long long blocks = ((n_tiles + 7) / 8) * 8;       // see xcd_contiguous_tile
    if (blocks > 0x7fffffffLL) return NMSA_ERR_ARG;
    if (score) hipLaunchKernelGGL((k_argmax_resized<DTYPE, true>), dim3((unsigned)blocks),
                                  dim3(RZ_TW * RZ_TH), 0, stream, logits, g, C, tiles_x, tiles_y,
                                  n_tiles, idx_u8, idx_i64, score);
    else hipLaunchKernelGGL((k_argmax_resized<DTYPE, false>), dim3((unsigned)blocks),
                            dim3(RZ_TW * RZ_TH), 0, stream, logits, g, C, tiles_x, tiles_y,
                            n_tiles, idx_u8, idx_i64, score);
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
