#!/usr/bin/env python3
"""Where k_pq_count spends its life (diagnosis build with phase stamps, -DNMSA_PQ_STAMPS):
   NMSA_LIB_PATH=tools/ab/libnmsa_pqstamps.so python tools/diag_pq_stamps.py [map]
stamps per workgroup, thread 0 (100 MHz wall clock): 0 start, 1 LDS tables cleared, 2 first tile
counted, 3 all tiles counted (wave 0), 4 all waves there, 5 slab written + table flushed"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import _lib as L, ops                 # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn      # noqa: E402
from tools import bench_support                                      # noqa: E402

B, Cc, H, W = 32, 40, 480, 640
dev = torch.device('cuda')
inp = syn.make_panoptic_inputs_torch(B, Cc, H, W, n_centers=24, seed=4321, device=dev)
m = bench_support.MetricAccumulators(Cc + 1, dev, inp, 0, side_stream=False)
r = ops.panoptic_pipeline(inp['semantic_logits'], inp['instance_center'], inp['instance_offset'],
                          inp['semantic_classes_is_thing'])
what = r['panoptic'] if 'map' in sys.argv else r
for _ in range(5):
    m.update_and_reduce(what)
torch.cuda.synchronize()
h = C.CDLL(L.LIB_PATH)
n = 4096 * 8
buf = (C.c_ulonglong * n)()
assert h.nmsa_debug_pq_stamps(buf, n, 1) == 0
m.update_and_reduce(what)
torch.cuda.synchronize()
assert h.nmsa_debug_pq_stamps(buf, n, 0) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
print(f'{len(s)} workgroups; us after the first workgroup started')
for name, col in (('start', 0), ('tables cleared', 1), ('first tile done', 2), ('tiles done (wave 0)', 3),
                  ('all waves there', 4), ('end', 5)):
    v = (s[:, col] - t0) / 100.0
    print(f'  {name:20s} min {v.min():6.1f}  p10 {np.percentile(v, 10):6.1f}  median {np.median(v):6.1f}  '
          f'p90 {np.percentile(v, 90):6.1f}  max {v.max():6.1f}')
for name, a, b in (('clear', 0, 1), ('first tile', 1, 2), ('other tiles', 2, 3), ('wait for the waves', 3, 4),
                   ('slab + flush', 4, 5), ('life', 0, 5)):
    d = (s[:, b] - s[:, a]) / 100.0
    print(f'  {name:20s} per workgroup: median {np.median(d):5.1f}  p90 {np.percentile(d, 90):5.1f}  max {d.max():5.1f}')
print(f'  slow-path entries of wave 0 per workgroup: median {np.median(s[:, 6]):.0f} max {s[:, 6].max()} (of 16 steps); '
      f'wave 0 counting (loads landed -> tile counted), all tiles: median {np.median(s[:, 7]) / 100:.1f} us max {s[:, 7].max() / 100:.1f} us')
