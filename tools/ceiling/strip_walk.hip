// read ceiling of a "strip walk" over class planes (f2 diagnosis, round 5): B images of C planes of
// H x W f32; a wave owns 64 source columns (start = strip * STEP rounded down to 4: the windows of
// neighbouring strips overlap like the source windows of 64 output columns when upscaling) and a
// range of ROWS source rows; per row it loads one dword per lane from each of the C planes (C
// coalesced 256-byte pieces), the next row is requested before the current one is consumed.
// Nothing is computed (an xor per value): what this reaches is what the memory system gives the
// pattern a register-resident bilinear walk would have.
//   build: hipcc -O3 --offload-arch=gfx950 strip_walk.hip -o strip_walk
//   run:   ./strip_walk [B C H W STEP ROWS NT WAVES_PER_WG]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

template <int C, bool NT>
__global__ __launch_bounds__(256) void k_strip_walk(const float* __restrict__ x, int H, int W, int step, int rows,
                                                    int strips, int ranges, unsigned* sink)
{
    const int wpw = blockDim.x >> 6;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    // workgroup -> (image, range, group of adjacent strips)
    const int groups = (strips + wpw - 1) / wpw;
    int id = blockIdx.x;
    const int sg = id % groups; id /= groups;
    const int rg = id % ranges; id /= ranges;
    const int b = id;
    const int strip = sg * wpw + w;
    if (strip >= strips) return;
    int x0 = (strip * step) & ~3;
    if (x0 + 64 > W) x0 = W - 64;
    const int y0 = rg * rows, y1 = min(H, y0 + rows + 1);          // one halo row
    const size_t plane = (size_t)H * W;
    const float* p = x + (size_t)b * C * plane + x0 + l;
    float cur[C], nxt[C];
    auto ld = [&](const float* q) { return NT ? __builtin_nontemporal_load(q) : *q; };
#pragma unroll
    for (int c = 0; c < C; ++c) cur[c] = ld(p + (size_t)c * plane + (size_t)y0 * W);
    unsigned acc = 0;
    for (int y = y0; y < y1; ++y) {
        const int yn = min(y + 1, y1 - 1);
#pragma unroll
        for (int c = 0; c < C; ++c) nxt[c] = ld(p + (size_t)c * plane + (size_t)yn * W);
#pragma unroll
        for (int c = 0; c < C; ++c) acc ^= __float_as_uint(cur[c]);
#pragma unroll
        for (int c = 0; c < C; ++c) cur[c] = nxt[c];
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char** argv)
{
    int B = 32, H = 480, W = 640, step = 56, rows = 60, nt = 0, wpw = 4;
    constexpr int C = 40;
    if (argc > 1) B = atoi(argv[1]);
    if (argc > 3) { H = atoi(argv[2]); W = atoi(argv[3]); }
    if (argc > 4) step = atoi(argv[4]);
    if (argc > 5) rows = atoi(argv[5]);
    if (argc > 6) nt = atoi(argv[6]);
    if (argc > 7) wpw = atoi(argv[7]);
    const size_t n = (size_t)B * C * H * W;
    float* x; unsigned* sink;
    hipMalloc(&x, n * 4); hipMalloc(&sink, 4);
    hipMemset(x, 1, n * 4);
    const int strips = (W - 64 + step - 1) / step + 1, ranges = (H + rows - 1) / rows;
    const int groups = (strips + wpw - 1) / wpw;
    const dim3 grid((unsigned)(B * ranges * groups)), block(64 * wpw);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        const int K = 20;
        for (int i = 0; i < K; ++i) {
            if (nt) hipLaunchKernelGGL((k_strip_walk<C, true>), grid, block, 0, 0, x, H, W, step, rows, strips, ranges, sink);
            else hipLaunchKernelGGL((k_strip_walk<C, false>), grid, block, 0, 0, x, H, W, step, rows, strips, ranges, sink);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms / K * 1e3;
        printf("B=%d %dx%d step=%d rows=%d nt=%d waves/wg=%d: %d strips x %d ranges, %u wgs: %.1f us = %.2f TB/s of the source (%.2f GB)\n",
               B, W, H, step, rows, nt, wpw, strips, ranges, grid.x, us, n * 4 / us * 1e-6, n * 4 * 1e-9);
    }
    if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
    return 0;
}
