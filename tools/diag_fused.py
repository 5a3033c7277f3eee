#!/usr/bin/env python3
"""Fused-kernel timing on one shape with switches (diagnosis):
   python tools/diag_fused.py B C H W K dtype [nocenters] [uniform]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops                           # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn      # noqa: E402

B, C, H, W, K = map(int, sys.argv[1:6])
dt = {'f32': None, 'bf16': torch.bfloat16, 'f16': torch.float16}[sys.argv[6]]
flags = sys.argv[7:]
dev = torch.device('cuda')
inp = syn.make_panoptic_inputs_torch(B, C, H, W, n_centers=K, seed=4321, device=dev, logits_dtype=dt)
center = inp['instance_center']
logits = inp['semantic_logits']
if 'nocenters' in flags:
    center = torch.zeros_like(center)
if 'uniform' in flags:                  # one class everywhere
    logits = torch.zeros_like(logits)
    logits[:, C - 1] = 5
if 'stuff' in flags:                    # no thing pixel at all
    logits = torch.zeros_like(logits)
    logits[:, 0] = 5
a = (logits, center, inp['instance_offset'], inp['semantic_classes_is_thing'])
ev = []
want_score = 'score' in flags
for _ in range(25):
    r = ops.panoptic_pipeline(*a, fused_kernel_events=ev, want_score=want_score)
torch.cuda.synchronize()
ms = float(np.mean([x.elapsed_time(y) for x, y in ev[5:]]))
es = logits.element_size()
print(f'{sys.argv[1:]} TILED={os.environ.get("NMSA_FUSED_TILED", "default")}: fused {ms * 1e3:8.1f} us '
      f'{B * H * W * (es * C + 9) / ms / 1e6:7.1f} GB/s  n_centers {r["n_centers"].tolist()[:4]} '
      f'fg {float(r["foreground"].float().mean()):.2f}')
