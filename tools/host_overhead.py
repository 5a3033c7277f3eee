#!/usr/bin/env python3
"""How long does the HOST need to enqueue one bench step (vs the GPU time of the step)?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops                         # noqa: E402
from nicr_mt_scene_analysis_amd.metric import bench_support      # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402

dev = torch.device('cuda')
inp = syn.make_panoptic_inputs_torch(32, 40, 480, 640, device=dev, seed=1)
m = bench_support.MetricAccumulators(41, dev, inp, 0)
a = (inp['semantic_logits'], inp['instance_center'], inp['instance_offset'], inp['semantic_classes_is_thing'])
for _ in range(10):
    r = ops.panoptic_pipeline(*a)
    m.update_and_reduce(r['panoptic'])
torch.cuda.synchronize()
N = 100
t0 = time.perf_counter()
for _ in range(N):
    r = ops.panoptic_pipeline(*a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'pipeline only : host enqueue {1e6*(t1-t0)/N:7.1f} us/step, total {1e6*(t2-t0)/N:7.1f} us/step')
t0 = time.perf_counter()
for _ in range(N):
    r = ops.panoptic_pipeline(*a)
    m.update_and_reduce(r['panoptic'])
t1 = time.perf_counter()
m.wait()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'with metrics  : host enqueue {1e6*(t1-t0)/N:7.1f} us/step, total {1e6*(t2-t0)/N:7.1f} us/step')
