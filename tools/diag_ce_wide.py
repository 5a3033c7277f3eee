#!/usr/bin/env python3
"""time nmsa_loss_ce_fwd_grad alone (HIP events) — knobs through the environment (NMSA_CE_SPLIT,
NMSA_CE_SPLIT_RUN, NMSA_CE_FORCE_SPLIT are read once per process):
  python tools/diag_tile.py ce C [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import _lib as L        # noqa: E402

kind = sys.argv[1]
R = int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
H, W = 768, 1024
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.empty((B, R, H, W), device=dev, dtype=torch.bfloat16)
for b in range(B):
    x[b] = torch.randn((R, H, W), device=dev, generator=g).to(torch.bfloat16)
grad = torch.empty_like(x)
s = torch.zeros(1, dtype=torch.float64, device=dev)
n = torch.zeros(1, dtype=torch.int64, device=dev)
wsum = torch.zeros(1, dtype=torch.float64, device=dev)
status = torch.zeros(4, dtype=torch.int32, device=dev)
exp = torch.full((1,), 1e-6, device=dev)
lib = L.lib()
if kind != 'ce':
    raise SystemExit('ce only')
else:
    t = torch.randint(0, R + 1, (B, H, W), device=dev, generator=g).to(torch.uint8)
    w = torch.rand(R, device=dev, generator=g) + 0.5
    nb = lib.nmsa_loss_workspace_bytes(B, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)

    def run():
        L.check(lib.nmsa_loss_ce_fwd_grad(L.ptr(x), 1, L.ptr(t), L.ptr(w), B, R, H, W, 0.0, L.ptr(exp),
                                          L.ptr(s), L.ptr(n), L.ptr(wsum), L.ptr(grad), L.ptr(status),
                                          L.ptr(ws), nb, L.stream_ptr(dev)), 'ce')
    bytes_px = 4 * R + 1
for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
px = B * H * W
knobs = {k: v for k, v in os.environ.items() if k.startswith('NMSA_')}
print(f'{kind} R={R} B={B} {knobs}: {ms:.3f} ms  {px * bytes_px / ms / 1e9:.2f} TB/s algorithmic', flush=True)
