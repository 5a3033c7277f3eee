#!/usr/bin/env python3
"""One leg of bench.py's `secondary` object in a loop, for rocprofv3:

    rocprofv3 --kernel-trace --stats -d out -o leg -- python3 tools/profile_leg.py cfg5_bf16 [reps]

legs: cfg2_f32 cfg2_bf16 cfg5_f32 cfg5_bf16 (pipeline + metric updates), cfg3_losses,
ce150 (cross entropy at 150 classes), cos512, cos768, next_rows (f2 full resolution, f3 scores,
f4 targets), api (postprocess / validation / training step through the reference-shaped API;
NMSA_BENCH_API_PROFILE=1 adds a cProfile of the training step on stderr)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                         # noqa: E402
from nicr_mt_scene_analysis_amd import ops                           # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn      # noqa: E402

leg = sys.argv[1] if len(sys.argv) > 1 else 'cfg5_bf16'
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
shapes = {'cfg2': (32, 40, 480, 640, 24), 'cfg5': (16, 150, 768, 1024, 48)}
if leg[:4] in shapes and leg[5:] in ('f32', 'bf16', 'f16'):
    dt = {'f32': None, 'bf16': torch.bfloat16, 'f16': torch.float16}[leg[5:]]
    out = bench.secondary_pipeline(ops, syn, dev, *shapes[leg[:4]], dt, overlap=False)
elif leg == 'cfg3_losses':
    out = bench.secondary_losses(dev)
elif leg == 'next_rows':
    out = bench.secondary_next_rows(ops, syn, dev)
elif leg == 'cfg5_full':
    out = bench.secondary_cfg5_full(ops, syn, dev)
elif leg == 'api':
    out = bench.secondary_api(syn, dev)
elif leg == 'ce150':
    out = bench.secondary_ce(dev)
elif leg in ('cos512', 'cos768'):
    out = bench.secondary_cos_emb(dev, B=16, D=int(leg[3:]))
else:
    raise SystemExit(f'unknown leg {leg}')
print(out)
