"""f4 target generation in a loop (for rocprofv3 --kernel-trace --stats) + HIP-event timing:
   python tools/prof_f4.py [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                         # noqa: E402
from nicr_mt_scene_analysis_amd import ops                           # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda', 0)
B, C, H, W = 32, 40, 480, 640
import numpy as np   # noqa: E402
m = syn.make_label_maps(2, C + 1, H, W, 30, seed=2)
r2 = (B + 1) // 2
sem = torch.from_numpy(np.tile(m['semantic'], (r2, 1, 1))[:B]).to(dev)
ins = torch.from_numpy(np.tile(m['instance'], (r2, 1, 1))[:B]).to(dev)
th = torch.from_numpy(m['semantic_classes_is_thing'].astype(np.uint8)).to(dev)
st = torch.from_numpy((~m['semantic_classes_is_thing']).astype(np.uint8)).to(dev)
ops.instance_clear_stuff(sem, ins, st)
t1 = bench.hip_timed(lambda: ops.instance_targets(sem, ins, C + 1, th, st, 8, True), reps=reps, warm=3)
t2 = bench.hip_timed(lambda: ops.panoptic_targets(sem, ins, C + 1, th, 1 << 16), reps=reps, warm=3)
print(f'instance_targets {1e3 * t1:.1f} us   panoptic_targets {1e3 * t2:.1f} us')
