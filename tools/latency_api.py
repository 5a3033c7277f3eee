#!/usr/bin/env python3
"""Single-image latency of the reference-shaped API (robot use case): `postprocess()` incl. the
read of the panoptic map's host-side id dict, eager vs `defer_host_sync=True`, B = 1 / 4."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY   # noqa: E402
from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class  # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402

dev = torch.device('cuda')
for B in (1, 4):
    inp = syn.make_panoptic_inputs_torch(B, 40, 480, 640, device=dev, seed=1)
    is_thing = tuple(bool(x) for x in inp['semantic_classes_is_thing'].cpu().tolist())
    batch = {'rgb_fullres': torch.zeros((B, 3, 480, 640)),
             APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize', 'valid_region_slice_y': slice(0, 480),
                                           'valid_region_slice_x': slice(0, 640)}]] * B}
    data = ((inp['semantic_logits'], (inp['instance_center'], inp['instance_offset'])), (None, None))
    for defer in (False, True):
        post = get_postprocessing_class('panoptic')(
            semantic_postprocessing=get_postprocessing_class('semantic')(),
            instance_postprocessing=get_postprocessing_class('instance')(),
            semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing,
            defer_host_sync=defer)
        for _ in range(20):
            post.postprocess(data, batch, is_training=False)
        torch.cuda.synchronize()
        lat = []
        for _ in range(200):
            t0 = time.perf_counter()
            r = post.postprocess(data, batch, is_training=False)
            ids = r['panoptic_segmentation_deeplab_ids']          # host object: waits for the GPU
            lat.append(time.perf_counter() - t0)
        lat.sort()
        print(f'B={B} defer_host_sync={defer}: call + id dicts  median {1e6*lat[100]:.0f} us  '
              f'p90 {1e6*lat[180]:.0f} us')
