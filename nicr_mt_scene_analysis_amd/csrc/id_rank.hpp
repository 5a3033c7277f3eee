// id_rank.hpp — shared pieces of the kernels that rank sparse instance ids (0..65535) per
// image: presence bitmap -> exclusive popcount prefix -> dense rank (ascending id order).
// Used by merge_wide.hip (GT-side deeplab merge) and targets.hip (target generation).
#pragma once
#include "nmsa_common.hpp"

namespace nmsa {

constexpr int MW_MAX_ID = 65535;
constexpr int MW_WORDS = (MW_MAX_ID + 1) / 32;      // 2048 bitmap words per image
constexpr int MW_ST_ID_RANGE = 32;                  // status bit: instance id outside [0, 65535]
constexpr int MW_ST_OVERFLOW = 1;                   // more distinct ids than max_segments

__device__ __forceinline__ int64_t mw_load(const void* p, int dtype, size_t i)
{
    switch (dtype) {
        case NMSA_U8: return ((const uint8_t*)p)[i];
        case NMSA_I16: return ((const int16_t*)p)[i];
        case NMSA_I32: return ((const int32_t*)p)[i];
        default: return ((const int64_t*)p)[i];
    }
}

__device__ __forceinline__ int mw_wave_scan(int x)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(x, o);
        if (lane_id() >= o) x += t;
    }
    return x;
}

__device__ __forceinline__ int mw_block_scan(int v, int* scratch, int* total)
{
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int incl = mw_wave_scan(v);
    __syncthreads();
    if (lane_id() == 63) scratch[w] = incl;
    __syncthreads();
    if (w == 0) {
        int s = (lane_id() < nw) ? scratch[lane_id()] : 0;
        s = mw_wave_scan(s);
        if (lane_id() < nw) scratch[lane_id()] = s;
    }
    __syncthreads();
    *total = scratch[nw - 1];
    return incl + ((w == 0) ? 0 : scratch[w - 1]);
}

// dense rank of a present id: number of smaller present ids
__device__ __forceinline__ int id_rank_dense(const uint32_t* bitmap, const uint32_t* prefix, int id)
{
    const uint32_t w = bitmap[id >> 5];
    return (int)prefix[id >> 5] + __popc(w & ((1u << (id & 31)) - 1u));
}

}  // namespace nmsa
