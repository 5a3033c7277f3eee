"""configs[2] loss step (four losses, forward + backward) in a loop, for rocprofv3:
   python tools/diag_losses.py [B] [reps]      (NMSA_SPECULATIVE_GRAD=0: two-kernel path)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd.loss import (CrossEntropyLossSemantic, L1Loss, MSELoss,  # noqa: E402
                                             VonMisesLossBiternion, speculation_stats)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
C, H, W = 40, 480, 640
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(7)
dt = torch.bfloat16


def rnd(*s):
    return torch.randn(s, device=dev, generator=g)


logits = (rnd(B, C, H, W) * 3).to(dt).requires_grad_(True)
labels = torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8)
w = torch.rand(C, device=dev, generator=g) + 0.5
center = torch.rand((B, H, W), device=dev, generator=g).to(dt).requires_grad_(True)
center_t = torch.rand((B, H, W), device=dev, generator=g)
offset = rnd(B, 2, H, W).to(dt).requires_grad_(True)
offset_t = rnd(B, 2, H, W)
ori = rnd(B, 2, H, W).to(dt).requires_grad_(True)
ori_t = torch.nn.functional.normalize(rnd(B, 2, H, W), dim=1)
m1 = torch.rand((B, H, W), device=dev, generator=g) < 0.7
m2 = torch.rand((B, H, W), device=dev, generator=g) < 0.5
m3 = torch.rand((B, H, W), device=dev, generator=g) < 0.3
ce = CrossEntropyLossSemantic(weights=w)
mse, l1, vm = MSELoss(), L1Loss(), VonMisesLossBiternion()


def step():
    for t in (logits, center, offset, ori):
        t.grad = None
    (lc, n), = ce([logits], [labels])
    a = mse.masked_sum(center, center_t, m1)
    b = l1.masked_sum(offset, offset_t, m2)
    c = vm.masked_sum(ori, ori_t, m3)
    (lc / n + a[0] / a[1] + b[0] / b[1] + c[0] / c[1]).backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
n_px = B * H * W
ms = 1e3 * (t2 - t0) / reps
print(f'B={B}: host issue {1e3 * (t1 - t0) / reps:.3f} ms/step, step {ms:.3f} ms '
      f'({n_px * 204 / ms / 1e6:.0f} GB/s of the 204 B/px algorithmic), {speculation_stats()}',
      flush=True)
