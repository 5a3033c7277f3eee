// nmsa_common.hpp — shared device/host helpers of libnmsa_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nmsa.h"

namespace nmsa {

constexpr int kWave = 64;  // CDNA wavefront

extern thread_local int g_last_hip_error;

inline int check_launch()
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { g_last_hip_error = (int)e; return NMSA_ERR_LAUNCH; }
    return NMSA_OK;
}

inline int check_hip(hipError_t e)
{
    if (e != hipSuccess) { g_last_hip_error = (int)e; return NMSA_ERR_LAUNCH; }
    return NMSA_OK;
}

// Dynamic LDS above the 64 KB default is granted per kernel function AND per device: the grant
// is cached per (kernel, device) pair (api.hip).  NMSA_OK, or NMSA_ERR_LAUNCH when the runtime
// denies it — callers then take their smaller-LDS variant or report the error, they never
// launch into a denied grant.
int allow_dynamic_lds_impl(const void* kernel, size_t bytes);
template <typename K>
int allow_dynamic_lds(K kernel, size_t bytes)
{
    return allow_dynamic_lds_impl(reinterpret_cast<const void*>(kernel), bytes);
}

// What the launch geometry is sized from: read ONCE per device from the HIP runtime (api.hip) —
// never a literal 256 / 8: a partitioned MI355X (DPX / QPX / CPX modes) or a CU mask shows fewer
// CUs and XCDs, and the cooperating-workgroup cosine kernel's forward progress depends on the grid
// fitting the chip.  NMSA_ASSUME_CUS / NMSA_ASSUME_XCDS (read at every call) override the queried
// values: tests shrink the chip (results must not change) or overstate it (the cosine kernel's
// partners then time out and the call takes its fallback).  Without a device (CPU-only symbol
// checks, workspace queries) the MI355X's numbers are returned.
struct DeviceGeometry {
    int cus;                // compute units of the current device
    int xcds;               // XCDs (each with its own L2)
    size_t lds_per_cu;      // bytes
    size_t lds_per_block;   // largest dynamic + static LDS one workgroup may ask for
};
DeviceGeometry device_geometry();

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ float bf16_to_f32(uint16_t v)
{
    return __uint_as_float(((uint32_t)v) << 16);
}

__device__ __forceinline__ float f16_to_f32(uint16_t v)
{
    return (float)__builtin_bit_cast(_Float16, v);
}

// Wave-aggregated histogram increment: lanes holding the same key are counted
// with one ballot + popcount and ONE atomic by the group's leader.  Keys are
// spatially coherent on this path (neighbouring pixels share instance/class),
// so the loop runs 1-3 times per wave.  key < 0 = lane does not contribute.
template <typename AddFn>
__device__ __forceinline__ void wave_aggregate_add(int key, AddFn add)
{
    unsigned long long todo = __ballot(key >= 0);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader);
        const unsigned long long same = __ballot(key == k) & todo;
        if (lane_id() == leader) add(k, (uint32_t)__popcll(same));
        todo &= ~same;
    }
}

// Lanes hold consecutive pixel groups, so equal keys come in RUNS: every run head learns its
// run's length and last lane from two ballots — all heads then act in parallel (one atomic per
// run), with no leader-serial loop over the distinct keys.  key < 0 = gap.  Wave-uniform call.
__device__ __forceinline__ bool wave_run_head(int key, int& len, int& last)
{
    const int prev = __shfl_up(key, 1);
    const int l = lane_id();
    const bool head = key >= 0 && (l == 0 || prev != key);
    const unsigned long long heads = __ballot(head);
    const unsigned long long gaps = __ballot(key < 0);
    const unsigned long long stop = (heads | gaps) & ~((2ull << l) - 1ull);
    const int nxt = stop ? (__ffsll((long long)stop) - 1) : 64;
    len = nxt - l;
    last = nxt - 1;
    return head;
}

// Slot of `key` in a small LDS hash set / table `keys[n]` (n a power of two, empty = -1): linear
// probing, 4 tries, -1 when full there (callers then fall back to a global atomic).  Used to
// privatise per-workgroup accumulators whose keys (instance ids, (instance, class) pairs) are
// few per workgroup but shared by hundreds of workgroups — direct global atomics on them
// serialise on a handful of addresses.
__device__ __forceinline__ int lds_hash_slot(int* keys, int n, int key)
{
    int slot = (int)(((uint32_t)key * 2654435761u) >> 16) & (n - 1);
    for (int t = 0; t < 4; ++t) {
        const int old = atomicCAS(&keys[slot], -1, key);
        if (old == -1 || old == key) return slot;
        slot = (slot + 1) & (n - 1);
    }
    return -1;
}

__device__ __forceinline__ float wave_reduce_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

__device__ __forceinline__ double wave_reduce_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

}  // namespace nmsa
