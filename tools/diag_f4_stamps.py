#!/usr/bin/env python3
"""Where k_tg_scan spends its life (diagnosis build with phase stamps):
   NMSA_LIB_PATH=tools/ab/libnmsa_stamps.so python tools/diag_f4_stamps.py [panoptic]
stamps per workgroup (100 MHz wall clock): 0 start, 1 scan loop done, 2 flush drained, 3 ticket
drawn, 4 tail done (the image's last workgroup only)"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import _lib as L, ops                 # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn      # noqa: E402

dev = torch.device('cuda', 0)
B, Cc, H, W = 32, 40, 480, 640
m = syn.make_label_maps(2, Cc + 1, H, W, 30, seed=2)
r2 = (B + 1) // 2
sem = torch.from_numpy(np.tile(m['semantic'], (r2, 1, 1))[:B]).to(dev)
ins = torch.from_numpy(np.tile(m['instance'], (r2, 1, 1))[:B]).to(dev)
th = torch.from_numpy(m['semantic_classes_is_thing'].astype(np.uint8)).to(dev)
st = torch.from_numpy((~m['semantic_classes_is_thing']).astype(np.uint8)).to(dev)
ops.instance_clear_stuff(sem, ins, st)
pan = 'panoptic' in sys.argv
call = (lambda: ops.panoptic_targets(sem, ins, Cc + 1, th, 1 << 16)) if pan else \
    (lambda: ops.instance_targets(sem, ins, Cc + 1, th, st, 8, True))
for _ in range(5):
    call()
torch.cuda.synchronize()
h = C.CDLL(L.LIB_PATH)
h.nmsa_debug_tg_stamps_clear()
call()
torch.cuda.synchronize()
n = 8192 * 8
buf = (C.c_ulonglong * n)()
assert h.nmsa_debug_tg_stamps(buf, n) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
us = lambda x: (x - t0) / 100.0
print(f'{len(s)} workgroups; all times in us after the first workgroup started')
for name, col in (('start', 0), ('scan loop done', 1), ('flush drained', 2), ('ticket drawn', 3)):
    v = us(s[:, col])
    print(f'  {name:16s} min {v.min():6.1f}  median {np.median(v):6.1f}  max {v.max():6.1f}')
for name, a, b in (('scan loop', 0, 1), ('flush + drain', 1, 2), ('ticket', 2, 3)):
    d = (s[:, b] - s[:, a]) / 100.0
    print(f'  {name:16s} per workgroup: median {np.median(d):5.1f}  p90 {np.percentile(d, 90):5.1f}  max {d.max():5.1f}')
tails = s[s[:, 4] > 0]
d = (tails[:, 4] - tails[:, 3]) / 100.0
for name, a, b_ in (('keys, bitmap, prefix, ranks', 3, 5), ('per-instance rows', 5, 6), ('lists / naive ranks + dict', 6 if not pan else 5, 7), ('cleaning + end', 7, 4)):
    if (tails[:, a] > 0).all() and (tails[:, b_] > 0).all():
        dd = (tails[:, b_] - tails[:, a]) / 100.0
        print(f'    tail phase {name:28s} median {np.median(dd):5.1f}  max {dd.max():5.1f}')
print(f'  tails: {len(tails)}; start median {np.median(us(tails[:, 3])):6.1f} max {us(tails[:, 3]).max():6.1f}; '
      f'duration median {np.median(d):5.1f} max {d.max():5.1f}; last tail ends at {us(tails[:, 4]).max():6.1f}')
