// read + write streaming ceiling in the access pattern of the wide-column loss kernels (k_ce_split,
// k_cos_split): B images of C planes of P 16-bit pixels; a workgroup of NW waves covers 64 * PX
// consecutive pixels, wave w holds the planes [w CQ, (w + 1) CQ) of those pixels in registers (one
// PX * 2 byte piece per plane and lane), then writes them to the same place of a second tensor.
// Nothing is computed: what this reaches is what the memory system gives that pattern.
//   build: hipcc -O3 --offload-arch=gfx950 plane_copy.hip -o plane_copy
//   run:   ./plane_copy [B C H W]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned first(unsigned v) { return v; }
__device__ inline unsigned first(u2 v) { return v.x; }
__device__ inline unsigned first(u4 v) { return v.x; }
template <int N> struct piece;
template <> struct piece<1> { typedef unsigned int type; };
template <> struct piece<2> { typedef u2 type; };
template <> struct piece<4> { typedef u4 type; };

// DW: dwords per lane and plane (2: 8 B = 4 px, 4: 16 B = 8 px); NP: planes per wave (register tile);
// MODE 0 copy, 1 read only, 2 write only; RUN consecutive tiles per workgroup;
// PREFETCH: a plane's register takes the next tile's piece right after its store
template <int DW, int NP, int MODE, bool PREFETCH>
__global__ __launch_bounds__(512) void k_plane_copy(const uint16_t* __restrict__ x, uint16_t* __restrict__ y,
                                                    int C, int P, int run, unsigned* sink)
{
    typedef typename piece<DW>::type T;
    constexpr int PX = DW * 2;
    const int nw = blockDim.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int CQ = (C + nw - 1) / nw;
    const int c0 = w * CQ, nc = max(0, min(CQ, C - c0));
    const size_t img = (size_t)blockIdx.y * C * P;
    const int n_tiles = (P + 64 * PX - 1) / (64 * PX);
    const int t0 = blockIdx.x * run, t1 = min(n_tiles, t0 + run);
    T r[NP];
    unsigned acc = 0;
    auto off = [&](int tile) { const int q = (tile * 64 + l) * PX; return (size_t)(q < P ? q : 0); };
    if (MODE != 2 && t0 < t1) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            if (i < nc) r[i] = __builtin_nontemporal_load((const T*)(x + img + (size_t)(c0 + i) * P + off(t0)));
    }
    for (int tile = t0; tile < t1; ++tile) {
        const bool more = tile + 1 < t1;
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < NP; ++i) { r[i] = T(tile + i); }
        }
        if (MODE != 2) {
#pragma unroll
            for (int i = 0; i < NP; ++i) if (i < nc) acc ^= first(r[i]);       // (a wait for every plane, like a maximum walk)
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            if (i < nc) {
                if (MODE != 1) __builtin_nontemporal_store(r[i], (T*)(y + img + (size_t)(c0 + i) * P + off(tile)));
                if (MODE != 2 && PREFETCH && more)
                    r[i] = __builtin_nontemporal_load((const T*)(x + img + (size_t)(c0 + i) * P + off(tile + 1)));
            }
        }
        if (MODE != 2 && !PREFETCH && more) {
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if (i < nc) r[i] = __builtin_nontemporal_load((const T*)(x + img + (size_t)(c0 + i) * P + off(tile + 1)));
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int DW, int NP, int MODE, bool PF>
static void bench(const char* name, const uint16_t* x, uint16_t* y, int B, int C, int P, int nw, int run, unsigned* sink)
{
    const int PX = DW * 2;
    const int n_tiles = (P + 64 * PX - 1) / (64 * PX);
    if ((C + nw - 1) / nw > NP) { printf("%-44s skipped (planes per wave > %d)\n", name, NP); return; }
    dim3 grid((n_tiles + run - 1) / run, B), block(64 * nw);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_plane_copy<DW, NP, MODE, PF>), grid, block, 0, 0, x, y, C, P, run, sink);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_plane_copy<DW, NP, MODE, PF>), grid, block, 0, 0, x, y, C, P, run, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (double)B * C * P * 2 * (MODE == 0 ? 2 : 1);
    printf("%-44s waves %d run %2d: %7.3f ms  %6.2f TB/s\n", name, nw, run, ms, bytes / ms / 1e9);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv)
{
    const int B = argc > 4 ? atoi(argv[1]) : 16, C = argc > 4 ? atoi(argv[2]) : 150;
    const int H = argc > 4 ? atoi(argv[3]) : 768, W = argc > 4 ? atoi(argv[4]) : 1024;
    const int P = H * W;
    const size_t n = (size_t)B * C * P;
    uint16_t *x, *y;
    unsigned* sink;
    if (hipMalloc(&x, n * 2) != hipSuccess || hipMalloc(&y, n * 2) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) {
        fprintf(stderr, "allocation failed\n");
        return 1;
    }
    hipMemset(x, 1, n * 2);
    hipMemset(y, 0, n * 2);
    printf("B=%d C=%d %dx%d 16-bit: %.2f GB per tensor\n", B, C, W, H, n * 2 / 1e9);
    for (int run : {1, 4}) {
        bench<2, 40, 1, false>("read only, 8 B pieces", x, y, B, C, P, 4, run, sink);
        bench<2, 40, 2, false>("write only, 8 B pieces", x, y, B, C, P, 4, run, sink);
        bench<2, 40, 0, false>("copy, 8 B pieces", x, y, B, C, P, 4, run, sink);
        bench<2, 40, 0, true>("copy, 8 B pieces, next tile after each store", x, y, B, C, P, 4, run, sink);
        bench<4, 20, 1, false>("read only, 16 B pieces", x, y, B, C, P, 8, run, sink);
        bench<4, 20, 2, false>("write only, 16 B pieces", x, y, B, C, P, 8, run, sink);
        bench<4, 20, 0, false>("copy, 16 B pieces", x, y, B, C, P, 8, run, sink);
        bench<4, 20, 0, true>("copy, 16 B pieces, next tile after each store", x, y, B, C, P, 8, run, sink);
        bench<4, 40, 0, false>("copy, 16 B pieces, 4 waves", x, y, B, C, P, 4, run, sink);
        bench<1, 40, 1, false>("read only, 4 B pieces", x, y, B, C, P, 4, run, sink);
        bench<1, 40, 0, false>("copy, 4 B pieces", x, y, B, C, P, 4, run, sink);
        bench<1, 20, 0, false>("copy, 4 B pieces, 8 waves", x, y, B, C, P, 8, run, sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    hipFree(x); hipFree(y); hipFree(sink);
    return 0;
}
