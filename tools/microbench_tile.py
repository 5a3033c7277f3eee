#!/usr/bin/env python3
"""The LDS-tile loss kernels at the configs[4] shapes: cosine embedding D=512 / 768 and cross
entropy at 150 classes, forward + backward through the loss classes (HIP events), with the
algorithmic-bytes fraction of the 8 TB/s peak.
  python tools/microbench_tile.py [cos512 cos768 ce150 ce64]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                         # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
legs = sys.argv[1:] or ['cos512', 'cos768', 'ce150']
for leg in legs:
    if leg.startswith('cos'):
        out = bench.secondary_cos_emb(dev, B=16, D=int(leg[3:]))
    elif leg.startswith('ce'):
        out = bench.secondary_ce(dev, C=int(leg[2:]))
    else:
        raise SystemExit(leg)
    print(leg, {k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ('ms', 'frac') if kk in v})
                for k, v in out.items()}, flush=True)
    torch.cuda.empty_cache()
