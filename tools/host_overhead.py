#!/usr/bin/env python3
"""How long does the HOST need to enqueue one bench step (vs the GPU time of the step)?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops                         # noqa: E402
from tools import bench_support                                   # noqa: E402
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

# ---- the reference-shaped API: PanopticPostprocessing.postprocess (one host sync per call for
#      the center counts, Python dicts for ids / instance meta) --------------------------------
from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY   # noqa: E402
from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class  # noqa: E402

is_thing = tuple(bool(x) for x in inp['semantic_classes_is_thing'].cpu().tolist())
post = get_postprocessing_class('panoptic')(
    semantic_postprocessing=get_postprocessing_class('semantic')(),
    instance_postprocessing=get_postprocessing_class('instance')(),
    semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing)
FH, FW = (int(v) for v in os.environ.get('FULLRES', '480x640').split('x'))   # dataset resolution
batch = {'rgb_fullres': torch.zeros((32, 3, FH, FW)),
         APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize', 'valid_region_slice_y': slice(0, 480),
                                       'valid_region_slice_x': slice(0, 640)}]] * 32}
data = ((a[0], (a[1], a[2])), (None, None))
for _ in range(5):
    post.postprocess(data, batch, is_training=False)
torch.cuda.synchronize()
per_call = []
for _ in range(100):
    t0 = time.perf_counter()
    rr = post.postprocess(data, batch, is_training=False)
    if (FH, FW) != (480, 640):                  # what a validation step reads
        _ = rr['semantic_segmentation_idx_fullres'], rr['panoptic_segmentation_deeplab_fullres']
    per_call.append(time.perf_counter() - t0)
torch.cuda.synchronize()
per_call.sort()
dt = per_call[len(per_call) // 2]
print(f'postprocess() : median {1e6*dt:7.1f} us/call (min {1e6*per_call[0]:.1f}, max {1e6*per_call[-1]:.1f}; '
      f'{32*480*640/dt/1e6:8.1f} Mpix/s; id dicts / meta lazy)')
per_call = []
for _ in range(100):
    t0 = time.perf_counter()
    rr = post.postprocess(data, batch, is_training=False)
    ids, meta = rr['panoptic_segmentation_deeplab_ids'], rr['panoptic_segmentation_deeplab_instance_meta']
    per_call.append(time.perf_counter() - t0)
torch.cuda.synchronize()
per_call.sort()
dt = per_call[len(per_call) // 2]
print(f'  + dicts/meta: median {1e6*dt:7.1f} us/call (min {1e6*per_call[0]:.1f}, max {1e6*per_call[-1]:.1f}; '
      f'{32*480*640/dt/1e6:8.1f} Mpix/s)')

# ---- defer_host_sync=True: no wait inside postprocess(); consecutive calls pipeline -------------
post_d = get_postprocessing_class('panoptic')(
    semantic_postprocessing=get_postprocessing_class('semantic')(),
    instance_postprocessing=get_postprocessing_class('instance')(),
    semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing,
    defer_host_sync=True)
for _ in range(5):
    post_d.postprocess(data, batch, is_training=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    rr = post_d.postprocess(data, batch, is_training=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'postprocess(defer_host_sync=True): host {1e6*(t1-t0)/100:7.1f} us/call, '
      f'pipelined total {1e6*(t2-t0)/100:7.1f} us/call ({32*480*640/((t2-t0)/100)/1e6:8.1f} Mpix/s)')

if '--profile' in sys.argv:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        rr = post.postprocess(data, batch, is_training=False)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
