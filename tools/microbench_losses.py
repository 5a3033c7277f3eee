#!/usr/bin/env python3
"""Loss kernels at BASELINE configs[2] (B=64, 640x480, C=40, bf16 predictions):
forward and forward+backward timings with HIP events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd.loss import (CrossEntropyLossSemantic, L1Loss, MSELoss,   # noqa: E402
                                             VonMisesLossBiternion, CosineEmbeddingLoss)


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    B, C, H, W = 64, 40, 480, 640
    dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
    es = 2 if dt == torch.bfloat16 else 4
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(0)
    px = B * H * W
    logits = (torch.randn((B, C, H, W), device=dev, generator=g) * 3).to(dt).requires_grad_(True)
    labels = torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8)
    w = torch.rand(C, device=dev, generator=g) + 0.5
    center = torch.rand((B, H, W), device=dev, generator=g).to(dt).requires_grad_(True)
    center_t = torch.rand((B, H, W), device=dev, generator=g)
    offset = torch.randn((B, 2, H, W), device=dev, generator=g).to(dt).requires_grad_(True)
    offset_t = torch.randn((B, 2, H, W), device=dev, generator=g)
    ori = torch.randn((B, 2, H, W), device=dev, generator=g).to(dt).requires_grad_(True)
    ori_t = torch.randn((B, 2, H, W), device=dev, generator=g)
    m1 = torch.rand((B, H, W), device=dev, generator=g) < 0.7
    m2 = torch.rand((B, H, W), device=dev, generator=g) < 0.5
    m3 = torch.rand((B, H, W), device=dev, generator=g) < 0.3
    ce = CrossEntropyLossSemantic(weights=w)
    mse, l1, vm = MSELoss(), L1Loss(), VonMisesLossBiternion()

    def fwd_all():
        (lc, n), = ce([logits], [labels])
        a = mse.masked_sum(center, center_t, m1)
        b = l1.masked_sum(offset, offset_t, m2)
        c = vm.masked_sum(ori, ori_t, m3)
        return lc / n + a[0] / a[1] + b[0] / b[1] + c[0] / c[1]

    def fwd_bwd_all():
        for t in (logits, center, offset, ori):
            t.grad = None
        fwd_all().backward()

    def ce_fwd():
        return ce([logits], [labels])

    def ce_fwd_bwd():
        logits.grad = None
        (l, n), = ce([logits], [labels])
        (l / n).backward()

    rows = [
        ('CE fwd', ce_fwd, es * C + 1),
        ('CE fwd+bwd', ce_fwd_bwd, 2 * (es * C + 1) + es * C + 8),      # + saved log-sum-exp w/r
        ('4 losses fwd', fwd_all, es * C + 1 + (es + 4 + 1) + 2 * (2 * es + 8 + 1)),
        ('4 losses fwd+bwd', fwd_bwd_all,
         2 * (es * C + 1 + (es + 4 + 1) + 2 * (2 * es + 8 + 1)) + es * C + es + 2 * es + 2 * es + 8),
    ]
    for name, fn, bpp in rows:
        us = timeit(fn)
        print(f'{name:18s} {us:9.1f} us  {px / us:9.1f} Mpix/s  {px * bpp / us / 1e6:6.2f} TB/s '
              f'(bytes moved by this implementation: {bpp} B/px)')
    if len(sys.argv) > 2 and sys.argv[2] == 'dve':
        B2, D, H2, W2 = 8, 512, 768, 1024
        pred = torch.randn((B2, D, H2, W2), device=dev, generator=g).to(dt).requires_grad_(True)
        idx = torch.randint(0, 65, (B2, H2, W2), device=dev, generator=g, dtype=torch.int32)
        lut = torch.nn.functional.normalize(torch.randn((B2, 64, D), device=dev, generator=g), dim=-1)
        cos = CosineEmbeddingLoss()

        def dve_fwd():
            return cos.lut_sum(pred, idx, lut)

        def dve_fwd_bwd():
            pred.grad = None
            l, n = cos.lut_sum(pred, idx, lut)
            (l / n).backward()
        px2 = B2 * H2 * W2
        for name, fn, bpp in (('DVE cos fwd', dve_fwd, es * D + 4), ('DVE cos fwd+bwd', dve_fwd_bwd, 3 * es * D + 8)):
            us = timeit(fn, reps=5, warm=1)
            print(f'{name:18s} {us:9.1f} us  {px2 / us:9.1f} Mpix/s  {px2 * bpp / us / 1e6:6.2f} TB/s ({bpp} B/px)')


if __name__ == '__main__':
    main()
