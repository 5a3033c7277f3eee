"""where the K = 20 headline loses time: host wall clock of the timed region vs the GPU-side span
(events at the head of the first step's stream and behind the last metric kernel), per schedule
   python tools/diag_k20.py [K]"""
import gc
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops                       # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402
from tools import bench_support                                   # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
B, C, H, W = 32, 40, 480, 640
inp = syn.make_panoptic_inputs_torch(B, C, H, W, n_centers=24, seed=1234, device=dev)
logits, center, offset = inp['semantic_logits'], inp['instance_center'], inp['instance_offset']
is_thing = inp['semantic_classes_is_thing']
metrics = bench_support.MetricAccumulators(C + 1, dev, inp, 0, side_stream=True, sync_every_step=False)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]


EVENTS = len(sys.argv) > 2 and sys.argv[2] == 'events'
events = []


def step(i):
    with torch.cuda.stream(streams[i % 2]):
        r = ops.panoptic_pipeline(logits, center, offset, is_thing, want_foreground=not os.environ.get('DIAG_NO_FG'),
                                  fused_kernel_events=events if EVENTS else None)
        metrics.update_and_reduce(r['panoptic'], None)
    return r


for rep in range(4):
    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    with torch.cuda.stream(streams[0]):
        e0.record()
    for i in range(K):
        step(i)
    th = time.perf_counter()
    metrics.finalize(None)
    with torch.cuda.stream(metrics.stream):
        e1.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gc.enable()
    span = e0.elapsed_time(e1)
    print(f'K={K} events={EVENTS}: wall {1e3 * (t1 - t0):.3f} ms = {1e3 * (t1 - t0) / K:.4f}/step -> '
          f'{B * H * W * K / (t1 - t0) / 1e9:.2f} Gpix/s; host issue {1e3 * (th - t0):.3f}; GPU span (first '
          f'record -> last metric kernel) {span:.3f} ms; wall - span {1e3 * (t1 - t0) - span:.3f}')
