"""k_resized_tile in a loop for rocprofv3: python tools/diag_resized.py [Ho Wo] [score]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops   # noqa: E402

nums = [int(a) for a in sys.argv[1:] if a.isdigit()]
size = tuple(nums[:2]) if len(nums) >= 2 else (530, 730)
score = 'score' in sys.argv
dt = torch.bfloat16 if 'bf16' in sys.argv else torch.float32
if 'randn' in sys.argv:       # unit-variance noise: ~every wave meets a column whose maximum is <= 1 -> the exact tie path
    x = torch.randn((32, 40, 480, 640), device='cuda').to(dt)
else:                         # the bench's blobby logits (bench.py secondary_next_rows)
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402
    x = syn.make_panoptic_inputs_torch(32, 40, 480, 640, n_centers=24, seed=99,
                                       device=torch.device('cuda'))['semantic_logits'].to(dt)
for _ in range(12):
    ops.semantic_argmax_resized(x, size, None, want_score=score)
torch.cuda.synchronize()
