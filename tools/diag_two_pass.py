"""Does the second read of a tile come from the caches?  k_ce_bwd with the saved log-sum-exp
(logits read once) vs without (pass 1 + pass 2 over the same 164 KB tile per workgroup):
   python tools/diag_two_pass.py [lse|nolse]   (under rocprofv3 for FETCH_SIZE)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import _lib as L   # noqa: E402

dev = torch.device('cuda:0')
B, C, H, W = 64, 40, 480, 640
g = torch.Generator(device=dev).manual_seed(0)
x = (torch.randn((B, C, H, W), device=dev, generator=g) * 3).to(torch.bfloat16)
t = torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8)
lse = torch.empty((B, H, W), device=dev)
s = torch.empty(1, dtype=torch.float64, device=dev)
n = torch.empty(1, dtype=torch.int64, device=dev)
st = torch.zeros(4, dtype=torch.int32, device=dev)
nb = L.lib().nmsa_loss_workspace_bytes(B, H, W)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
one = torch.ones(1, device=dev)
grad = torch.empty_like(x)
L.check(L.lib().nmsa_loss_ce_fwd(L.ptr(x), 1, L.ptr(t), None, B, C, H, W, 0.0, L.ptr(s), L.ptr(n), None,
                                 L.ptr(lse), L.ptr(st), L.ptr(ws), nb, L.stream_ptr(dev)), 'fwd')
modes = sys.argv[1:] or ['lse', 'nolse']
for mode in modes:
    def run():
        L.check(L.lib().nmsa_loss_ce_bwd(L.ptr(x), 1, L.ptr(t), None, B, C, H, W, 0.0, L.ptr(one),
                                         L.ptr(lse) if mode == 'lse' else None, L.ptr(grad),
                                         L.stream_ptr(dev)), 'bwd')
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    b.record()
    torch.cuda.synchronize()
    print(f'{mode}: {a.elapsed_time(b) / 10 * 1e3:.1f} us', flush=True)
