#!/usr/bin/env python3
"""Does the one-pass cosine kernel's time depend on where the GRADIENT it writes lies relative to
the prediction it reads?  One prediction tensor, the gradient as a view into one arena at offsets
of k x STEP: python tools/diag_cos_relalign.py [D] [step_KiB] [n]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                    # noqa: E402
from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss, _multi         # noqa: E402

dev = torch.device('cuda', 0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEP = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024) << 10
N = int(sys.argv[3]) if len(sys.argv) > 3 else 33
B, H, W, L = 16, 768, 1024, 64
g = torch.Generator(device=dev).manual_seed(11)
idx = torch.randint(0, L + 1, (B, H // 16, W // 16), device=dev, generator=g, dtype=torch.int32)
idx = idx.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
lut = torch.nn.functional.normalize(torch.randn((B, L, D), device=dev, generator=g), dim=-1)
cos = CosineEmbeddingLoss()
pred = torch.empty((B, D, H, W), device=dev, dtype=torch.bfloat16)
for b in range(B):
    pred[b] = torch.randn((D, H, W), device=dev, generator=g).to(torch.bfloat16)
pred.requires_grad_(True)
nbytes = pred.numel() * 2
arena = torch.empty(nbytes + N * STEP + (1 << 21), device=dev, dtype=torch.uint8)
base = (-arena.data_ptr()) % (1 << 21)                       # 2 MiB aligned start inside the arena
state = {'off': 0}


def alloc(x):
    if x.numel() * x.element_size() != nbytes:
        return torch.empty_like(x)
    o = base + state['off']
    return arena[o:o + nbytes].view(torch.bfloat16).view(x.shape)


_multi._alloc_grad = alloc


def fwd_bwd():
    pred.grad = None
    l, n = cos.lut_sum(pred, idx, lut)
    (l / n).backward()


for k in range(N):
    state['off'] = k * STEP
    ms = [bench.hip_timed(fwd_bwd, reps=3, warm=1) for _ in range(2)]
    gp = arena.data_ptr() + base + state['off']
    rel = gp - pred.data_ptr()
    print(f'k {k:3d}: gradient - prediction = {rel / (1 << 20):10.2f} MiB (mod 16 MiB {(rel % (1 << 24)) / (1 << 20):6.2f}, '
          f'pred mod 16 MiB {(pred.data_ptr() % (1 << 24)) / (1 << 20):5.2f}): ' + ' '.join(f'{m:.3f}' for m in ms) + ' ms', flush=True)
