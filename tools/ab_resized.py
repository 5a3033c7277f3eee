#!/usr/bin/env python3
"""f2 fused crop + bilinear + softmax + max at the bench's two output sizes, HIP-event timed:
   python tools/ab_resized.py   (NMSA_LIB_PATH selects the library build)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                         # noqa: E402
from nicr_mt_scene_analysis_amd import ops                           # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn      # noqa: E402

dev = torch.device('cuda')
x = syn.make_panoptic_inputs_torch(32, 40, 480, 640, n_centers=24, seed=99, device=dev)['semantic_logits']
out = []
for size in ((530, 730), (768, 1024)):
    for score in (False, True):
        ms = bench.hip_timed(lambda: ops.semantic_argmax_resized(x, size, None, want_score=score), reps=30, warm=5)
        out.append(f'{size[1]}x{size[0]}{" +score" if score else ""}: {ms * 1e3:6.1f} us')
print(' | '.join(out))
