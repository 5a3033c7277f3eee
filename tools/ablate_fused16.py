#!/usr/bin/env python3
"""Where does the 16-bit fused kernel lose bandwidth?  argmax only / fused with 0 centers /
fused with the bench's 24 centers, f32 vs bf16 vs f16 (B=32 640x480 C=40)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import _lib as L                  # noqa: E402
from nicr_mt_scene_analysis_amd import ops                        # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402
from microbench import timeit                                     # noqa: E402

B, C, H, W = 32, 40, 480, 640
dev = torch.device('cuda')
NCEN = int(os.environ.get('ABLATE_CENTERS', '24'))
inp = syn.make_panoptic_inputs_torch(B, C, H, W, device=dev, seed=1, n_centers=NCEN)
cen = ops.center_nms_topk(inp['instance_center'])
thing = inp['semantic_classes_is_thing'].view(torch.uint8)
off = inp['instance_offset']
sem_u8 = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
inst = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
votes = torch.zeros((B, 256, C + 1), dtype=torch.int32, device=dev)
n0 = torch.zeros_like(cen['n_centers'])
px = B * H * W
lib = L.lib()
LIBS = {'new': lib}
_old = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ab', 'libnmsa_old.so')
if os.path.exists(_old):                      # A/B on the same box: an older build of the library
    import ctypes
    old = ctypes.CDLL(_old)
    for name, (res, args) in L._SIGNATURES.items():
        if hasattr(old, name):
            getattr(old, name).restype, getattr(old, name).argtypes = res, args
    LIBS['old'] = old
for dt in (torch.float32, torch.bfloat16, torch.float16):
    x = inp['semantic_logits'].to(dt)
    eb = x.element_size()

    def fused(n, lib=lib):
        L.check(lib.nmsa_panoptic_fused(
            L.ptr(x), L.float_dtype_code(x), L.ptr(off), L.ptr(cen['centers_yx']), L.ptr(n),
            L.ptr(thing), B, C, H, W, 256, float(H), float(W), 0, 0.0,
            L.ptr(sem_u8), L.ptr(inst), None, None, L.ptr(votes), 1, 65, L.stream_ptr(dev)), 'fused')

    def argmax(lib, score):
        sc = torch.empty((B, H, W), dtype=torch.float32, device=dev) if score else None
        L.check(lib.nmsa_semantic_argmax(L.ptr(x), L.float_dtype_code(x), B, C, H, W, L.ptr(sem_u8),
                                         None, L.ptr(sc), L.stream_ptr(dev)), 'argmax')
    for rep in range(2):
        for tag, lb in LIBS.items():
            rows = [('argmax u8', lambda: argmax(lb, False), eb * C + 1),
                    ('argmax u8+score', lambda: argmax(lb, True), eb * C + 5),
                    ('fused, 0 centers', lambda: fused(n0, lb), eb * C + 9),
                    (f'fused, {NCEN} centers', lambda: fused(cen['n_centers'], lb), eb * C + 9)]
            for name, fn, bpp in rows:
                us = timeit(fn)
                print(f'{str(dt):16s} {tag} {name:20s} {us:8.1f} us  {px * bpp / us / 1e6:6.2f} TB/s ({bpp} B/px)')
