#!/usr/bin/env python3
"""Wide cross entropy (C = 150, bf16, B = 16, 1024 x 768) with per-pixel random labels (bench.py's leg:
every class plane is some pixel's target in every wave) and with piecewise-constant labels (label maps
of real scenes: most planes of a wave's 256 pixels have no target pixel): python tools/diag_ce_split.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                                    # noqa: E402
from nicr_mt_scene_analysis_amd.loss import CrossEntropyLossSemantic, reset_speculation_state   # noqa: E402

dev = torch.device('cuda', 0)
B, C, H, W = 16, 150, 768, 1024
g = torch.Generator(device=dev).manual_seed(13)
logits = (torch.randn((B, C, H, W), device=dev, generator=g) * 3).to(torch.bfloat16).requires_grad_(True)
ce = CrossEntropyLossSemantic(weights=torch.rand(C, device=dev, generator=g) + 0.5)
random_labels = torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8)
coarse = torch.randint(0, C + 1, (B, H // 32, W // 32), device=dev, generator=g).to(torch.uint8)
blocky_labels = coarse.repeat_interleave(32, 1).repeat_interleave(32, 2).contiguous()
for name, labels in (('random', random_labels), ('32x32 blocks', blocky_labels)):
    reset_speculation_state()

    def fwd():
        (lc, n), = ce([logits], [labels])
        return lc / n

    def fwd_bwd():
        logits.grad = None
        fwd().backward()
    with torch.no_grad():
        ms_f = bench.hip_timed(fwd, reps=8, warm=2)
    ms_fb = bench.hip_timed(fwd_bwd, reps=8, warm=2)
    print(f'{name:14s} labels: fwd {ms_f:.4f} ms, fwd+bwd {ms_fb:.4f} ms')
