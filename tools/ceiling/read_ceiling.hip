// read-only streaming ceiling on one MI355X: what a kernel that only LOADS can reach, in the
// access patterns of k_panoptic_fused (build: hipcc -O3 --offload-arch=gfx950 read_ceiling.hip -o read_ceiling)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void k_linear(const f4* __restrict__ x, size_t n4, float* out)
{
    float acc = 0.f;
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(x + i + (size_t)u * 256) : x[i + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

// plane pattern: image b, C planes of P pixels; a workgroup covers 1024 consecutive pixels (4 per lane),
// walks the C planes with U loads in flight
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_planes(const float* __restrict__ x, int C, size_t P, float* out)
{
    const size_t b = blockIdx.y;
    const size_t p0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const float* base = x + b * C * P + p0;
    float acc = 0.f;
    for (int c = 0; c < C; c += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f4* q = (const f4*)(base + (size_t)(c + u) * P);
            v[u] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fmaxf(acc, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

// the same walk + what the fused kernel stores: NS u8 maps (uchar4 per lane) and optionally two more
// 16-B loads per lane at the END of the walk (the offset planes)
template <int U, int NS, bool TAIL_LOADS>
__global__ __launch_bounds__(256) void k_planes_store(const float* __restrict__ x, int C, size_t P,
                                                      const float* __restrict__ off, unsigned char* __restrict__ o8)
{
    const size_t b = blockIdx.y;
    const size_t p0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const float* base = x + b * C * P + p0;
    float acc = 0.f;
    for (int c = 0; c < C; c += U) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const f4*)(base + (size_t)(c + u) * P));
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fmaxf(acc, fmaxf(fmaxf(v[u].x, v[u].y), fmaxf(v[u].z, v[u].w)));
    }
    if (TAIL_LOADS) {
        const f4 a = __builtin_nontemporal_load((const f4*)(off + b * 2 * P + p0));
        const f4 c2 = __builtin_nontemporal_load((const f4*)(off + b * 2 * P + P + p0));
        acc += a.x + c2.y;
    }
    typedef unsigned char u8x4 __attribute__((ext_vector_type(4)));
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    const unsigned char q = (unsigned char)acc;
    if (NS >= 30) {                   // NS - 30 maps, staged in LDS, ONE wave writes 1 KB per map (16 B per lane)
        __shared__ unsigned int stage[3][256];
#pragma unroll
        for (int k = 0; k < NS - 30; ++k) stage[k][threadIdx.x] = q * 0x01010101u;
        __syncthreads();
        if (threadIdx.x < 64) {
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int k = 0; k < NS - 30; ++k) {
                const u32x4 v = *(const u32x4*)&stage[k][threadIdx.x * 4];
                *(u32x4*)(o8 + ((size_t)k * gridDim.y + b) * P + (size_t)blockIdx.x * 1024 + threadIdx.x * 16) = v;
            }
        }
    } else if (NS == 12) {                   // ONE 8-byte store per lane: two u8 maps packed as u16
        *(u16x4*)(o8 + (b * P + p0) * 2) = u16x4{q, q, q, q};
    } else if (NS >= 20) {            // NS - 20 maps, non-temporal stores
#pragma unroll
        for (int k = 0; k < NS - 20; ++k)
            __builtin_nontemporal_store(u8x4{q, q, q, q}, (u8x4*)(o8 + ((size_t)k * gridDim.y + b) * P + p0));
    } else {
#pragma unroll
        for (int k = 0; k < NS; ++k) *(u8x4*)(o8 + ((size_t)k * gridDim.y + b) * P + p0) = u8x4{q, q, q, q};
    }
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)

template <typename F>
float timeit(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    float best = 1e9f, sum = 0;
    for (int i = 0; i < reps; ++i) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); best = ms < best ? ms : best; sum += ms;
    }
    printf("  avg %.4f ms  best %.4f ms", sum / reps, best);
    return best;
}

int main()
{
    const int B = 32, C = 40; const size_t P = 480 * 640;
    const size_t n = (size_t)B * C * P;
    float *x, *out;
    CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(x, 0, n * 4));
    const double gb = n * 4 / 1e9;
    printf("bytes %.3f GB\n", gb);
#define LIN(U, NT) { printf("linear U=%d nt=%d:", U, NT); float ms = timeit([&] { hipLaunchKernelGGL((k_linear<U, NT>), dim3(n / 4 / 256 / U), dim3(256), 0, 0, (const f4*)x, n / 4, out); }, 20); printf("  -> %.2f TB/s\n", gb / ms); }
    LIN(4, true) LIN(8, true) LIN(16, true) LIN(8, false)
#define PL(U, NT) { printf("planes U=%d nt=%d:", U, NT); float ms = timeit([&] { hipLaunchKernelGGL((k_planes<U, NT>), dim3(P / 1024, B), dim3(256), 0, 0, x, C, P, out); }, 20); printf("  -> %.2f TB/s\n", gb / ms); }
    PL(4, true) PL(8, true) PL(10, true) PL(8, false)
    float* off; unsigned char* o8;
    CK(hipMalloc(&off, (size_t)B * 2 * P * 4)); CK(hipMalloc(&o8, (size_t)3 * B * P));
    CK(hipMemset(off, 0, (size_t)B * 2 * P * 4));
#define PS(NS, TL) { printf("planes U=8 + %d u8 stores, tail loads %d:", NS, TL); float ms = timeit([&] { hipLaunchKernelGGL((k_planes_store<8, NS, TL>), dim3(P / 1024, B), dim3(256), 0, 0, x, C, P, off, o8); }, 20); printf("  -> %.2f TB/s of the logits\n", gb / ms); }
    PS(1, false) PS(2, false) PS(3, false) PS(31, false) PS(32, false) PS(33, false) PS(22, false)
    return 0;
}
