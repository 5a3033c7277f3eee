// loss_common.hpp — pieces shared by the loss kernel files (losses.hip: one kernel per loss,
// losses_split.hip: wide-column cross entropy, losses_multi.hip: one launch for several losses).
#pragma once
#include "nmsa_common.hpp"

// run CALL(<dtype constant>) for the runtime dtype code (host side)
#define NMSA_DISPATCH_DTYPE(dtype, CALL)          \
    switch (dtype) {                              \
        case NMSA_F32: CALL(NMSA_F32); break;     \
        case NMSA_BF16: CALL(NMSA_BF16); break;   \
        case NMSA_F16: CALL(NMSA_F16); break;     \
        default: return NMSA_ERR_ARG;             \
    }

namespace nmsa {

constexpr int LOSS_THREADS = 256;

// one per workgroup; summed in a FIXED order by k_loss_finalize (deterministic results)
struct LossPartial { double sum; double aux; long long count; long long pad; };

typedef float f32x4_s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_s __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_s __attribute__((ext_vector_type(2)));

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ uint16_t f32_to_bf16(float f)
{
    // round-to-nearest-even via the hardware conversion (keeps NaN a NaN)
    return __builtin_bit_cast(uint16_t, (__bf16)f);
}
__device__ __forceinline__ uint16_t f32_to_f16(float f)
{
    return __builtin_bit_cast(uint16_t, (_Float16)f);
}

// element j (0..1) of a dword holding two 16-bit values / the dword itself for f32
template <int DTYPE>
__device__ __forceinline__ float unpack16(uint32_t w, int j)
{
    if (DTYPE == NMSA_BF16) return __uint_as_float(j ? (w & 0xFFFF0000u) : (w << 16));
    return f16_to_f32((uint16_t)(j ? (w >> 16) : (w & 0xFFFFu)));
}
template <int DTYPE>
__device__ __forceinline__ uint32_t pack16(float a, float b)
{
    if (DTYPE == NMSA_BF16) return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
    return (uint32_t)f32_to_f16(a) | ((uint32_t)f32_to_f16(b) << 16);
}

// max of two floats as ONE v_max_f32.  `fmaxf` costs three: LLVM canonicalises both operands first
// (`v_max_f32 x, x, x`) because either might be a signalling NaN — and the running maximum gets
// re-canonicalised in every basic block.  The instruction itself already returns the other
// operand for a NaN (IEEE mode), which is fmaxf's rule.
__device__ __forceinline__ float vmax(float a, float b)
{
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Speculative gradients (forward kernels write the gradient for an EXPECTED upstream scale).
// The backward kernels are launched with `computed_for` = that expected scale: when the real
// upstream gradient is bit-equal, the gradient buffer is already right and every workgroup
// returns at once; otherwise the kernel recomputes it.  counters[0] / [1] count the outcomes.
__device__ __forceinline__ bool grad_already_computed(const float* __restrict__ gscale,
                                                      const float* __restrict__ computed_for,
                                                      int* __restrict__ counters)
{
    if (!computed_for) return false;
    // a NaN in `computed_for` means "the forward pass wrote no gradient": never confirmed, even
    // when the real upstream gradient is that very NaN (it must come out as NaN gradients)
    const float e = *computed_for;
    const bool same = e == e && __float_as_uint(*gscale) == __float_as_uint(e);
    if (counters && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        atomicAdd(&counters[same ? 0 : 1], 1);
    return same;
}

// host side, defined in losses.hip
int loss_ce_fwd_grad_partials(const void* logits, int dtype, const uint8_t* target, const float* weights,
                              int B, int C, int P, float ls, const float* expected_gscale, void* grad,
                              LossPartial* partials, int32_t* status, hipStream_t stream);
// losses_split.hip: cross entropy for 49..256 classes (k_ce_split)
constexpr int CE_SPLIT_MAX_C = 256;
int ce_split_blocks(int P, int dtype);
int launch_ce_split(bool loss, const void* logits, int dtype, const uint8_t* target,
                    const float* weights, int B, int C, int P, float ls, const float* gscale,
                    const float* computed_for, int32_t* counters, void* grad,
                    LossPartial* partials, int32_t* status, hipStream_t stream);
// losses_cos.hip: cosine-embedding loss, forward + gradient in one pass (k_cos_split)
// (k_cos_split, or k_cos_parts for columns beyond one workgroup: `xch` = its granule exchange buffer)
int cos_split_blocks(int B, int D, int P, int L, int dtype);
size_t cos_split_xch_bytes(int B, int D, int P, int L, int dtype);
int launch_cos_split(bool loss, const void* pred, int dtype, const int32_t* indices, const float* lut,
                     int B, int D, int P, int L, const float* gscale, const float* computed_for,
                     int32_t* counters, void* grad, LossPartial* partials, int32_t* status,
                     void* xch, size_t xch_bytes, hipStream_t stream);
// losses.hip: the two-walk cosine kernels — shapes no one-pass kernel takes, and the device-gated
// fallback of k_cos_parts (`gate` non-null: the kernels return at once unless *gate != 0)
int cos_two_walk_blocks(int dtype, int B, int D, int P, int L);
int launch_cos_two_walk_fwd(const void* pred, int dtype, const int32_t* indices, const float* lut,
                            int B, int D, int P, int L, LossPartial* partials, int32_t* status,
                            float* dots_out, const int* gate, hipStream_t stream);
int launch_cos_two_walk_bwd(const void* pred, int dtype, const int32_t* indices, const float* lut,
                            int B, int D, int P, int L, const float* grad_scale, const float* dots,
                            void* grad_pred, const int* gate, int skip_nan_scale, hipStream_t stream);
int loss_finalize(const LossPartial* partials, int n, double* sum, double* aux, int64_t* count,
                  hipStream_t stream);
int loss_env_int(const char* name, int dflt);
bool loss_bad_shape(int B, int H, int W);

}  // namespace nmsa
