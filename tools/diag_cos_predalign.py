#!/usr/bin/env python3
"""Does the one-pass cosine kernel's time depend on where the PREDICTION lies inside one
allocation (same physical arena, shifted by k x STEP)?  python tools/diag_cos_predalign.py [D] [step_KiB] [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                    # noqa: E402
from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss                 # noqa: E402

dev = torch.device('cuda', 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEP = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 10
N = int(sys.argv[3]) if len(sys.argv) > 3 else 20
B, H, W, L = 16, 768, 1024, 64
g = torch.Generator(device=dev).manual_seed(11)
idx = torch.randint(0, L + 1, (B, H // 16, W // 16), device=dev, generator=g, dtype=torch.int32)
idx = idx.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
lut = torch.nn.functional.normalize(torch.randn((B, L, D), device=dev, generator=g), dim=-1)
cos = CosineEmbeddingLoss()
src = torch.empty((B, D, H, W), device=dev, dtype=torch.bfloat16)
for b in range(B):
    src[b] = torch.randn((D, H, W), device=dev, generator=g).to(torch.bfloat16)
nbytes = src.numel() * 2
arena = torch.empty(nbytes + N * STEP + (1 << 21), device=dev, dtype=torch.uint8)
base = (-arena.data_ptr()) % (1 << 21)
for k in range(N):
    o = base + k * STEP
    pred = arena[o:o + nbytes].view(torch.bfloat16).view(src.shape)
    pred.copy_(src)
    pred.requires_grad_(True)

    def fwd_bwd():
        pred.grad = None
        l, n = cos.lut_sum(pred, idx, lut)
        (l / n).backward()
    ms = [bench.hip_timed(fwd_bwd, reps=3, warm=1) for _ in range(2)]
    print(f'k {k:3d}: prediction at {pred.data_ptr():#x} (mod 16 MiB {(pred.data_ptr() % (1 << 24)) / (1 << 20):6.2f}): '
          + ' '.join(f'{m:.3f}' for m in ms) + ' ms', flush=True)
    del pred
