"""same-process A/B of the one-pass cosine kernels (forward + backward of bench.py's leg):
   python tools/diag_cos_parts.py B D "ENV=V,ENV=V" "ENV=V" ...   (configs alternate, 3 rounds)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch   # noqa: E402
import bench   # noqa: E402
from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss   # noqa: E402

B, D = int(sys.argv[1]), int(sys.argv[2])
configs = sys.argv[3:] or ['']
dev = torch.device('cuda', 0)
H, W, L = 768, 1024, 64
g = torch.Generator(device=dev).manual_seed(11)
pred = torch.empty((B, D, H, W), device=dev, dtype=torch.bfloat16)
for b in range(B):
    pred[b] = torch.randn((D, H, W), device=dev, generator=g).to(torch.bfloat16)
pred.requires_grad_(True)
idx = torch.randint(0, L + 1, (B, H // 16, W // 16), device=dev, generator=g, dtype=torch.int32)
idx = idx.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
lut = torch.nn.functional.normalize(torch.randn((B, L, D), device=dev, generator=g), dim=-1)
cos = CosineEmbeddingLoss()


def fwd_bwd():
    pred.grad = None
    l, n = cos.lut_sum(pred, idx, lut)
    (l / n).backward()


res = {c: [] for c in configs}
for rnd in range(3):
    for c in configs:
        keys = []
        for kv in filter(None, c.split(',')):
            k, v = kv.split('=')
            os.environ[k] = v
            keys.append(k)
        res[c].append(round(bench.hip_timed(fwd_bwd, reps=8, warm=2), 4))
        for k in keys:
            del os.environ[k]
n_px = B * H * W
for c, ms in res.items():
    best = min(ms)
    print(json.dumps({'B': B, 'D': D, 'cfg': c, 'ms': ms, 'frac_best': round((4 * D + 4) * n_px / (best * 1e-3) / 8e12, 4)}))
