#!/usr/bin/env python3
"""Does the time of the one-pass cosine kernel depend on WHERE its tensors live?  The same launch
on freshly allocated prediction / gradient tensors, six times inside one process, with spacer
allocations of different sizes in between (the caching allocator is emptied each time):
python tools/diag_cos_placement.py [D]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                    # noqa: E402
from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss                 # noqa: E402

dev = torch.device('cuda', 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B, H, W, L = 16, 768, 1024, 64
g = torch.Generator(device=dev).manual_seed(11)
idx = torch.randint(0, L + 1, (B, H // 16, W // 16), device=dev, generator=g, dtype=torch.int32)
idx = idx.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
lut = torch.nn.functional.normalize(torch.randn((B, L, D), device=dev, generator=g), dim=-1)
cos = CosineEmbeddingLoss()
spacers = [0, 1 << 20, 3 << 20, 257 << 20, 1 << 30, 5 << 30] + [(2 * k + 1) << 21 for k in range(10)]
for trial, sp in enumerate(spacers):
    torch.cuda.empty_cache()
    keep = torch.empty(sp, device=dev, dtype=torch.uint8) if sp else None
    pred = torch.empty((B, D, H, W), device=dev, dtype=torch.bfloat16)
    for b in range(B):
        pred[b] = torch.randn((D, H, W), device=dev, generator=g).to(torch.bfloat16)
    pred.requires_grad_(True)

    def fwd_bwd():
        pred.grad = None
        l, n = cos.lut_sum(pred, idx, lut)
        (l / n).backward()
    ms = [bench.hip_timed(fwd_bwd, reps=3, warm=1) for _ in range(3)]
    gp = pred.grad.data_ptr()
    print(f'trial {trial}: spacer {sp >> 20:5d} MiB, pred at {pred.data_ptr():#x}, gradient at {gp:#x} '
          f'(+{(gp - pred.data_ptr()) >> 21} x 2 MiB): ' + ' '.join(f'{m:.3f}' for m in ms) + ' ms', flush=True)
    del pred, keep
