#!/usr/bin/env python3
"""Pipeline timings at other BASELINE configs: bf16 logits at cfg2 shapes and the cfg5 shape
(1024x768, 150 classes).  python tools/microbench_cfg.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops                      # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


dev = torch.device('cuda')
for name, (B, C, H, W, K, dt) in {
    'cfg2 f32 ': (32, 40, 480, 640, 24, None),
    'cfg2 bf16': (32, 40, 480, 640, 24, torch.bfloat16),
    'cfg5 f32 ': (8, 150, 768, 1024, 48, None),
    'cfg5 bf16': (8, 150, 768, 1024, 48, torch.bfloat16),
}.items():
    inp = syn.make_panoptic_inputs_torch(B, C, H, W, n_centers=K, device=dev, seed=1, logits_dtype=dt)
    es = 2 if dt is not None else 4
    a = (inp['semantic_logits'], inp['instance_center'], inp['instance_offset'],
         inp['semantic_classes_is_thing'])
    ev = []
    us = timeit(lambda: ops.panoptic_pipeline(*a, fused_kernel_events=ev))
    fused = sum(x.elapsed_time(y) for x, y in ev[5:]) / len(ev[5:]) * 1e3
    px = B * H * W
    bpp = es * C + 21
    print(f'{name}: pipeline {us:8.1f} us  {px / us:9.1f} Mpix/s  {px * bpp / us / 1e6:5.2f} TB/s ({bpp} B/px) | '
          f'fused kernel {fused:8.1f} us {px * (es * C + 9) / fused / 1e6:5.2f} TB/s')
    del inp, a
    torch.cuda.empty_cache()
