"""Times the full-resolution step kernels (csrc/resize.hip) on one MI355X.
usage: python tools/microbench_fullres.py [B] [C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops   # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    C = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    Hs, Ws = 480, 640
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn((B, C, Hs, Ws), device='cuda', dtype=torch.float32).to(dt)
        pan = torch.randint(0, 40 << 16, (B, Hs, Ws), device='cuda', dtype=torch.int64)
        u8 = torch.randint(0, 255, (B, Hs, Ws), device='cuda', dtype=torch.uint8)
        for size in ((530, 730), (768, 1024), (240, 320)):
            in_bytes = x.numel() * x.element_size()
            npx = B * size[0] * size[1]
            t = timeit(lambda: ops.semantic_argmax_resized(x, size, None))
            tns = timeit(lambda: ops.semantic_argmax_resized(x, size, None, want_score=False))
            print(f"    (no score: {tns*1e3:8.1f} us)")
            alg = in_bytes + npx * 12
            print(f'{dt} {Hs}x{Ws}->{size}: argmax_resized {t*1e3:8.1f} us  '
                  f'{alg/t/1e9:7.2f} TB/s(alg)  {npx/t/1e6:8.2f} Gpx/s')
            t2 = timeit(lambda: ops.semantic_argmax(ops.resize_bilinear(x, size, None)), n=5)
            print(f'    materialise + argmax {t2*1e3:8.1f} us   ({t2/t:.2f}x)')
            t3 = timeit(lambda: ops.resize_bilinear(x, size, None), n=5)
            outb = npx * C * x.element_size()
            print(f'    resize_bilinear      {t3*1e3:8.1f} us   {(in_bytes+outb)/t3/1e9:6.2f} TB/s')
            t4 = timeit(lambda: ops.resize_nearest(pan, size, None))
            print(f'    nearest i64          {t4*1e3:8.1f} us   {(pan.numel()*8+npx*8)/t4/1e9:6.2f} TB/s')
            t5 = timeit(lambda: ops.resize_nearest(u8, size, None))
            print(f'    nearest u8           {t5*1e3:8.1f} us   {(u8.numel()+npx)/t5/1e9:6.2f} TB/s')
            if dt == torch.float32 and size == (530, 730):
                t6 = timeit(lambda: torch.nn.functional.interpolate(
                    x, size=size, mode='bilinear', align_corners=False).softmax(1).max(1), n=3)
                print(f'    ATen interpolate+softmax+max {t6*1e3:8.1f} us   ({t6/t:.1f}x)')
        del x


if __name__ == '__main__':
    main()
