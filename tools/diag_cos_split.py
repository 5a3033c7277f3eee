#!/usr/bin/env python3
"""k_cos_split (one-pass cosine loss: forward + gradient) against the two-kernel path: values and
time.   python3 tools/diag_cos_split.py [D] [B]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nicr_mt_scene_analysis_amd import _lib as L                    # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H, W, NL = 768, 1024, 64
dev = torch.device('cuda', 0)
g = torch.Generator(device=dev).manual_seed(11)
pred = torch.empty((B, D, H, W), device=dev, dtype=torch.bfloat16)
for b in range(B):
    pred[b] = torch.randn((D, H, W), device=dev, generator=g).to(torch.bfloat16)
idx = torch.randint(0, NL + 1, (B, H // 16, W // 16), device=dev, generator=g, dtype=torch.int32)
idx = idx.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
lut = torch.nn.functional.normalize(torch.randn((B, NL, D), device=dev, generator=g), dim=-1)
lib = L.lib()
assert lib.nmsa_loss_cos_emb_fwd_grad_supported(1, D, H, W, NL), 'shape not supported'
n_valid = int((idx != 0).sum())
gscale = torch.tensor([1.0 / n_valid], dtype=torch.float32, device=dev)
loss = torch.zeros((1,), dtype=torch.float64, device=dev)
n = torch.zeros((1,), dtype=torch.int64, device=dev)
grad = torch.empty_like(pred)
status = torch.zeros((4,), dtype=torch.int32, device=dev)
nb = lib.nmsa_loss_cos_emb_fwd_grad_workspace_bytes(B, D, H, W, NL)
ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
st = L.stream_ptr(dev)


def run():
    L.check(lib.nmsa_loss_cos_emb_fwd_grad(L.ptr(pred), 1, L.ptr(idx), L.ptr(lut), B, D, H, W, NL, L.ptr(gscale),
                                           L.ptr(loss), L.ptr(n), L.ptr(grad), L.ptr(status), L.ptr(ws), nb, st),
            'nmsa_loss_cos_emb_fwd_grad')


for _ in range(2):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
byts = B * H * W * (4 * D + 4)
print(f'k_cos_split D={D} B={B}: {ms:.3f} ms  {byts / ms / 1e9:.2f} TB/s algorithmic  frac {byts / ms / 1e9 / 8:.3f}')
# reference: the two-kernel path of the package (forward kernel, then the backward kernel)
from nicr_mt_scene_analysis_amd.loss import _functional as F_         # noqa: E402
p2 = pred.detach().clone().requires_grad_(True)
l2, n2 = F_.cosine_embedding_lut_sum(p2, idx, lut)
(l2 / n2).backward()
torch.cuda.synchronize()
print('loss', float(loss), float(l2), 'n', int(n), int(n2), 'status', status.tolist())
d = (grad.float() - p2.grad.float()).abs()
print('grad max abs diff', float(d.max()), 'max |grad|', float(p2.grad.float().abs().max()),
      'mismatching elements', int((grad != p2.grad).sum()), 'of', grad.numel())
