// The forms of the reference's losses that no task helper uses, on the device all the same:
//   MSELoss / L1Loss with reduction='none'      loss/mse.py:21-41 (else branch), loss/l1.py:21-41
//   CosineEmbeddingLoss on [N, D] rows with per-row labels (+1 similar / -1 dissimilar) and any
//   reduction                                    loss/cos_emb.py:21-56 (`target_similarity`)
//   VonMisesLossBiternion with reduction='none'  loss/vonmises.py:27-51
// Plain streaming kernels: every byte is read once (the row kernels read a row twice in the
// backward pass, the second time from L2).  The 2-D [N, C] rows of MSE / L1 need no kernel of
// their own: sum_n mean_c f = (1 / C) * sum over all elements, which the masked kernels already
// compute (loss/_elementwise.py).
#include "nmsa_common.hpp"
#include "loss_common.hpp"

namespace nmsa {
namespace {

template <int DTYPE>
__device__ __forceinline__ float ldf(const void* base, size_t i)
{
    if (DTYPE == NMSA_F32) return ((const float*)base)[i];
    const uint16_t h = ((const uint16_t*)base)[i];
    return (DTYPE == NMSA_BF16) ? bf16_to_f32(h) : f16_to_f32(h);
}
template <int DTYPE>
__device__ __forceinline__ void stf(void* base, size_t i, float v)
{
    if (DTYPE == NMSA_F32) ((float*)base)[i] = v;
    else ((uint16_t*)base)[i] = (DTYPE == NMSA_BF16) ? f32_to_bf16(v) : f32_to_f16(v);
}

// out[i] = f(pred[i] - target[i]) (BWD: grad[i] = upstream[i] * f'(pred[i] - target[i])),
// computed in fp32 and rounded once to the output's type, as ATen's op-math does
template <int DTYPE, int ODT, int KIND, bool BWD>
__global__ __launch_bounds__(256) void k_elem_none(
    const void* __restrict__ pred, const float* __restrict__ target, long long n,
    const void* __restrict__ upstream, void* __restrict__ out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float d = ldf<DTYPE>(pred, i) - target[i];
        if (!BWD) stf<ODT>(out, i, KIND == 0 ? d * d : fabsf(d));
        else {
            const float dd = KIND == 0 ? 2.0f * d : (float)((d > 0.f) - (d < 0.f));
            stf<DTYPE>(out, i, ldf<ODT>(upstream, i) * dd);
        }
    }
}

// One wave per row of [N, D]:  cos = x.y / sqrt((|x|^2 + eps)(|y|^2 + eps)), eps = 1e-12;
// label +1: 1 - cos;  label -1: max(0, cos - margin);  any other label: 0
// (ATen cosine_embedding_loss, which the reference wraps).  BWD: d loss / d x times the row's
// upstream gradient (`up_stride` 0: one scalar for every row).
template <int DTYPE, bool BWD>
__global__ __launch_bounds__(256) void k_cos_rows(
    const void* __restrict__ x, const float* __restrict__ y, const float* __restrict__ labels,
    long long n_rows, int D, float margin, const float* __restrict__ upstream, int up_stride,
    float* __restrict__ loss_rows, void* __restrict__ grad)
{
    const float EPS = 1e-12f;
    const int lane = threadIdx.x & 63;
    const long long waves = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long r = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < n_rows; r += waves) {
        const size_t base = (size_t)r * D;
        float xy = 0.f, xx = 0.f, yy = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float xv = ldf<DTYPE>(x, base + d), yv = y[base + d];
            xy = fmaf(xv, yv, xy); xx = fmaf(xv, xv, xx); yy = fmaf(yv, yv, yy);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            xy += __shfl_xor(xy, o); xx += __shfl_xor(xx, o); yy += __shfl_xor(yy, o);
        }
        const float den = sqrtf((xx + EPS) * (yy + EPS));
        const float c = xy / den;
        const float lab = labels ? labels[r] : 1.0f;
        if (!BWD) {
            if (lane == 0) loss_rows[r] = lab == 1.0f ? 1.0f - c : (lab == -1.0f ? fmaxf(c - margin, 0.f) : 0.f);
        } else {
            // d cos / d x = y / den - cos * x / (|x|^2 + eps)
            const float s = lab == 1.0f ? -1.0f : ((lab == -1.0f && c - margin >= 0.f) ? 1.0f : 0.f);
            const float g = s * upstream[(size_t)r * up_stride];
            const float k1 = g / den, k2 = -g * c / (xx + EPS);
            for (int d = lane; d < D; d += 64)
                stf<DTYPE>(grad, base + d, fmaf(k2, ldf<DTYPE>(x, base + d), k1 * y[base + d]));
        }
    }
}

// VonMisesLossBiternion per row of [N, 2] (loss/vonmises.py:27-51, reduction='none'):
// 1 - exp(kappa * (x . y - 1)); one row per lane.  BWD: upstream[row] * d / d x
template <int DTYPE, bool BWD>
__global__ __launch_bounds__(256) void k_vm_rows(
    const void* __restrict__ x, const float* __restrict__ y, long long n_rows, float kappa,
    const float* __restrict__ upstream, float* __restrict__ loss_rows, void* __restrict__ grad)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride) {
        const float x0 = ldf<DTYPE>(x, 2 * r), x1 = ldf<DTYPE>(x, 2 * r + 1);
        const float y0 = y[2 * r], y1 = y[2 * r + 1];
        const float e = __expf(kappa * (fmaf(x0, y0, x1 * y1) - 1.0f));
        if (!BWD) loss_rows[r] = 1.0f - e;
        else {
            const float g = -upstream[r] * kappa * e;
            stf<DTYPE>(grad, 2 * r, g * y0);
            stf<DTYPE>(grad, 2 * r + 1, g * y1);
        }
    }
}

int grid_for(long long work_items, int per_block)
{
    const long long want = (work_items + per_block - 1) / per_block;
    const long long cap = (long long)device_geometry().cus * 8;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

template <int KIND, bool BWD>
int launch_elem_none(const void* pred, int dtype, const float* target, long long n, int out_dtype,
                     const void* upstream, void* out, hipStream_t stream)
{
    const int grid = grid_for(n, 256 * 4);
#define NMSA_EN(DT, OD) hipLaunchKernelGGL((k_elem_none<DT, OD, KIND, BWD>), dim3(grid), dim3(256), 0, stream, \
                                           pred, target, n, upstream, out)
    const bool of32 = out_dtype == NMSA_F32;
    if (dtype == NMSA_F32) NMSA_EN(NMSA_F32, NMSA_F32);
    else if (dtype == NMSA_BF16) { if (of32) NMSA_EN(NMSA_BF16, NMSA_F32); else NMSA_EN(NMSA_BF16, NMSA_BF16); }
    else { if (of32) NMSA_EN(NMSA_F16, NMSA_F32); else NMSA_EN(NMSA_F16, NMSA_F16); }
#undef NMSA_EN
    return check_launch();
}

bool float_dtype(int dtype) { return dtype == NMSA_F32 || dtype == NMSA_BF16 || dtype == NMSA_F16; }

}  // namespace
}  // namespace nmsa

using namespace nmsa;

extern "C" int nmsa_loss_elementwise_none_fwd(const void* pred, int dtype, const float* target,
                                              int64_t n, int kind, int out_dtype, void* out,
                                              nmsa_stream_t stream)
{
    if (!pred || !target || !out || n < 0 || !float_dtype(dtype)) return NMSA_ERR_ARG;
    if (kind != 0 && kind != 1) return NMSA_ERR_ARG;
    if (out_dtype != NMSA_F32 && out_dtype != dtype) return NMSA_ERR_ARG;
    if (n == 0) return NMSA_OK;
    return kind == 0 ? launch_elem_none<0, false>(pred, dtype, target, n, out_dtype, nullptr, out, (hipStream_t)stream)
                     : launch_elem_none<1, false>(pred, dtype, target, n, out_dtype, nullptr, out, (hipStream_t)stream);
}

extern "C" int nmsa_loss_elementwise_none_bwd(const void* pred, int dtype, const float* target,
                                              int64_t n, int kind, int upstream_dtype,
                                              const void* upstream, void* grad_pred,
                                              nmsa_stream_t stream)
{
    if (!pred || !target || !upstream || !grad_pred || n < 0 || !float_dtype(dtype)) return NMSA_ERR_ARG;
    if (kind != 0 && kind != 1) return NMSA_ERR_ARG;
    if (upstream_dtype != NMSA_F32 && upstream_dtype != dtype) return NMSA_ERR_ARG;
    if (n == 0) return NMSA_OK;
    return kind == 0 ? launch_elem_none<0, true>(pred, dtype, target, n, upstream_dtype, upstream, grad_pred, (hipStream_t)stream)
                     : launch_elem_none<1, true>(pred, dtype, target, n, upstream_dtype, upstream, grad_pred, (hipStream_t)stream);
}

extern "C" int nmsa_loss_cos_rows_fwd(const void* input, int dtype, const float* target,
                                      const float* labels, int64_t n_rows, int D, float margin,
                                      float* loss_rows, nmsa_stream_t stream)
{
    if (!input || !target || !loss_rows || n_rows < 0 || D <= 0 || !float_dtype(dtype)) return NMSA_ERR_ARG;
    if (n_rows == 0) return NMSA_OK;
    const int grid = grid_for(n_rows, 4);
#define NMSA_CR(DT) hipLaunchKernelGGL((k_cos_rows<DT, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                                       input, target, labels, (long long)n_rows, D, margin, (const float*)nullptr, 0, \
                                       loss_rows, (void*)nullptr)
    if (dtype == NMSA_F32) NMSA_CR(NMSA_F32); else if (dtype == NMSA_BF16) NMSA_CR(NMSA_BF16); else NMSA_CR(NMSA_F16);
#undef NMSA_CR
    return check_launch();
}

extern "C" int nmsa_loss_cos_rows_bwd(const void* input, int dtype, const float* target,
                                      const float* labels, int64_t n_rows, int D, float margin,
                                      const float* upstream, int upstream_is_scalar,
                                      void* grad_input, nmsa_stream_t stream)
{
    if (!input || !target || !upstream || !grad_input || n_rows < 0 || D <= 0 || !float_dtype(dtype))
        return NMSA_ERR_ARG;
    if (n_rows == 0) return NMSA_OK;
    const int grid = grid_for(n_rows, 4);
    const int up_stride = upstream_is_scalar ? 0 : 1;
#define NMSA_CR(DT) hipLaunchKernelGGL((k_cos_rows<DT, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                                       input, target, labels, (long long)n_rows, D, margin, upstream, up_stride, \
                                       (float*)nullptr, grad_input)
    if (dtype == NMSA_F32) NMSA_CR(NMSA_F32); else if (dtype == NMSA_BF16) NMSA_CR(NMSA_BF16); else NMSA_CR(NMSA_F16);
#undef NMSA_CR
    return check_launch();
}

extern "C" int nmsa_loss_vonmises_rows_fwd(const void* input, int dtype, const float* target,
                                           int64_t n_rows, float kappa, float* loss_rows,
                                           nmsa_stream_t stream)
{
    if (!input || !target || !loss_rows || n_rows < 0 || !float_dtype(dtype)) return NMSA_ERR_ARG;
    if (n_rows == 0) return NMSA_OK;
    const int grid = grid_for(n_rows, 256);
#define NMSA_VR(DT) hipLaunchKernelGGL((k_vm_rows<DT, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                                       input, target, (long long)n_rows, kappa, (const float*)nullptr, loss_rows, (void*)nullptr)
    if (dtype == NMSA_F32) NMSA_VR(NMSA_F32); else if (dtype == NMSA_BF16) NMSA_VR(NMSA_BF16); else NMSA_VR(NMSA_F16);
#undef NMSA_VR
    return check_launch();
}

extern "C" int nmsa_loss_vonmises_rows_bwd(const void* input, int dtype, const float* target,
                                           int64_t n_rows, float kappa, const float* upstream,
                                           void* grad_input, nmsa_stream_t stream)
{
    if (!input || !target || !upstream || !grad_input || n_rows < 0 || !float_dtype(dtype)) return NMSA_ERR_ARG;
    if (n_rows == 0) return NMSA_OK;
    const int grid = grid_for(n_rows, 256);
#define NMSA_VR(DT) hipLaunchKernelGGL((k_vm_rows<DT, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, \
                                       input, target, (long long)n_rows, kappa, upstream, (float*)nullptr, grad_input)
    if (dtype == NMSA_F32) NMSA_VR(NMSA_F32); else if (dtype == NMSA_BF16) NMSA_VR(NMSA_BF16); else NMSA_VR(NMSA_F16);
#undef NMSA_VR
    return check_launch();
}
