#!/usr/bin/env python3
"""mIoU update on uint8 maps at the bench shape (SemanticTaskHelper's form), HIP-event timed:
   python tools/diag_confmat.py      (NMSA_CM_NO_U8=1: the general kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                         # noqa: E402
from nicr_mt_scene_analysis_amd import metric                        # noqa: E402

g = torch.Generator(device='cuda').manual_seed(1)
B, H, W, n = 32, 480, 640, 41
cell = torch.randint(0, n - 1, (B, H // 32, W // 32), device='cuda', generator=g)
pred = cell.repeat_interleave(32, 1).repeat_interleave(32, 2).to(torch.uint8).contiguous()
out = []
for name, tgt in (('coherent target', torch.roll(pred, (3, 3), (1, 2)).contiguous()),
                  ('random target', torch.randint(0, n, (B, H, W), device='cuda', generator=g).to(torch.uint8))):
    m = metric.MeanIntersectionOverUnion(n - 1)
    ms = bench.hip_timed(lambda: m.update_masked_void(pred, tgt), reps=50, warm=5)
    m._status.zero_()
    out.append(f'{name}: {ms * 1e3:6.1f} us')
print(' | '.join(out))
