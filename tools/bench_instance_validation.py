#!/usr/bin/env python3
"""Instance branch validation step (task_helper/instance.py:289-357):
`InstancePostprocessing.postprocess` with the ground-truth foreground -> `InstanceTaskHelper.
validation_step` (center / offset losses, GT-semantics merge, PQ), B=32 640x480, 41 classes."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops                                            # noqa: E402
from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY   # noqa: E402
from nicr_mt_scene_analysis_amd.model.postprocessing import InstancePostprocessing    # noqa: E402
from nicr_mt_scene_analysis_amd.task_helper import InstanceTaskHelper                 # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn                       # noqa: E402

B, NC, H, W = 32, 41, 480, 640
dev = torch.device('cuda')
lab = syn.make_label_maps(B, NC, H, W, n_instances=24, seed=2, mixed_fraction=0.0)
is_thing = lab['semantic_classes_is_thing']
sem = torch.from_numpy(lab['semantic']).to(dev)
ins = torch.from_numpy(lab['instance']).to(dev)
stuff = torch.from_numpy((~is_thing).astype(np.uint8)).to(dev)
ops.instance_clear_stuff(sem, ins, stuff)
tg = ops.instance_targets(sem, ins, NC, torch.from_numpy(is_thing.astype(np.uint8)).to(dev), stuff, 8, True)
pan_gt = sem.long() * 65536 + (ins.long() % 65536)
batch = {'instance_foreground': tg['foreground'].bool(), 'instance_fullres': ins, 'semantic_fullres': sem,
         'panoptic_fullres': pan_gt, 'panoptic_ids_to_instance_dict': [{} for _ in range(B)],
         'instance_center': tg['center'], 'instance_center_mask': tg['center_mask'].bool(),
         'instance_offset': tg['offset'].float(),
         APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize', 'valid_region_slice_y': slice(0, H),
                                       'valid_region_slice_x': slice(0, W)}]] * B}
post = InstancePostprocessing()
helper = InstanceTaskHelper(NC, tuple(bool(x) for x in is_thing), disable_multiscale_supervision=True)
helper.initialize(dev)
data = ((tg['center'].unsqueeze(1), tg['offset'].float()), (None,))


def step(i):
    preds = post.postprocess(data, batch, is_training=False)
    helper.validation_step(batch, i, preds)


for i in range(5):
    step(i)
torch.cuda.synchronize()
N = 30
t0 = time.perf_counter()
for i in range(N):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
_, _, logs = helper.validation_epoch_end()
print(f'instance validation step: {1e3*dt:.3f} ms/step ({B*H*W/dt/1e6:.0f} Mpix/s), '
      f'things pq {float(logs["instance_things_deeplab_pq"]):.3f}')
