#!/usr/bin/env python3
"""PQ + confusion-matrix update timing at the bench shapes (diagnosis): python tools/diag_pq.py [B C H W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops                           # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn      # noqa: E402
from tools import bench_support                                      # noqa: E402
import bench                                                         # noqa: E402

_nums = [a for a in sys.argv[1:] if a.isdigit()]
B, C, H, W = (int(x) for x in (_nums[:4] if len(_nums) >= 4 else (32, 40, 480, 640)))
dev = torch.device('cuda')
inp = syn.make_panoptic_inputs_torch(B, C, H, W, n_centers=24, seed=4321, device=dev)
m = bench_support.MetricAccumulators(C + 1, dev, inp, 0, side_stream=False)
r = ops.panoptic_pipeline(inp['semantic_logits'], inp['instance_center'], inp['instance_offset'],
                          inp['semantic_classes_is_thing'])
pan = r['panoptic']
if 'coherent' in sys.argv:            # semantic target = the prediction's classes; big uniform regions
    m.target_semantic = (pan // 65536).to(torch.uint8)
if 'flat' in sys.argv:                # one segment per image: every lane of every wave hits one slot
    pan = torch.full_like(pan, 3 * 65536)
    m.target_panoptic = pan.clone()
    m.target_semantic = torch.full_like(m.target_semantic, 3)
what = pan if ('flat' in sys.argv or 'map' in sys.argv) else r      # the pipeline's result: the update reads its parts
ms = bench.hip_timed(lambda: m.update_and_reduce(what), reps=30, warm=5)
m.pq._check_status()
print(f'PXB={os.environ.get("NMSA_PQ_PXB", "default")}: metric update {ms * 1e3:7.1f} us '
      f'({B * H * W * 17 / ms / 1e6:6.1f} GB/s of 17 B/px)')
