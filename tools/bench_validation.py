#!/usr/bin/env python3
"""One validation step through the reference-shaped API, end to end (SURVEY §8 a4 + a14):
`PanopticPostprocessing.postprocess` -> `PanopticTaskHelper.validation_step` (PQ + mIoU of the
merged map, fused kernel) -> `SemanticTaskHelper.validation_step` (loss + mIoU of the semantic
map), B=32, 40 classes, 640x480 network resolution.
  python tools/bench_validation.py [FHxFW]      dataset resolution (default 480x640 = no resize)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY   # noqa: E402
from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class  # noqa: E402
from nicr_mt_scene_analysis_amd.task_helper import PanopticTaskHelper, SemanticTaskHelper   # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402

B, C, H, W = 32, 40, 480, 640
FH, FW = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '480x640').split('x'))
dev = torch.device('cuda')
inp = syn.make_panoptic_inputs_torch(B, C, H, W, device=dev, seed=1)
is_thing = tuple(bool(x) for x in inp['semantic_classes_is_thing'].cpu().tolist())
g = torch.Generator(device=dev).manual_seed(3)
batch = {'rgb_fullres': torch.zeros((B, 3, FH, FW)),
         'semantic': torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8),
         'semantic_fullres': torch.randint(0, C + 1, (B, FH, FW), device=dev, generator=g).to(torch.uint8),
         'panoptic_fullres': torch.randint(0, C + 1, (B, FH, FW), device=dev, generator=g) * 65536,
         'panoptic_ids_to_instance_dict': [{} for _ in range(B)],
         APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize', 'valid_region_slice_y': slice(0, H),
                                       'valid_region_slice_x': slice(0, W)}]] * B}
data = ((inp['semantic_logits'], (inp['instance_center'], inp['instance_offset'])), (None, None))
# ground-truth panoptic map like bench.py's: the prediction shifted by 3 px with a void
# band — spatially coherent segments, not per-pixel noise
_post = get_postprocessing_class('panoptic')(
    semantic_postprocessing=get_postprocessing_class('semantic')(),
    instance_postprocessing=get_postprocessing_class('instance')(),
    semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing)
_pan = _post.postprocess(data, batch, is_training=False)['panoptic_segmentation_deeplab_fullres']
_tgt = torch.roll(_pan, shifts=(3, 3), dims=(1, 2)).contiguous()
_tgt[:, :3] = 0
batch['panoptic_fullres'] = _tgt
del _post, _pan
for defer in (False, True):
    post = get_postprocessing_class('panoptic')(
        semantic_postprocessing=get_postprocessing_class('semantic')(),
        instance_postprocessing=get_postprocessing_class('instance')(),
        semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing,
        defer_host_sync=defer)
    ph = PanopticTaskHelper(C + 1, (False,) + is_thing)
    sh = SemanticTaskHelper(n_classes=C, disable_multiscale_supervision=True)
    ph.initialize(dev)
    sh.initialize(dev)

    def step(i):
        r = post.postprocess(data, batch, is_training=False)
        ph.validation_step(batch, i, r)
        sh.validation_step(batch, i, r)

    for i in range(5):
        step(i)
    torch.cuda.synchronize()
    N = 40
    best = None
    for rep in range(4):
        t0 = time.perf_counter()
        for i in range(N):
            step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if best is None or t2 - t0 < best[2] - best[0]:
            best = (t0, t1, t2)
        print(f'   rep {rep}: {1e3*(t2-t0)/N:.3f} ms/step (host {1e3*(t1-t0)/N:.3f} ms)')
    t0, t1, t2 = best
    _, _, logs = ph.validation_epoch_end()
    print(f'validation step ({FH}x{FW}, defer_host_sync={defer}): {1e3*(t2-t0)/N:.3f} ms/step '
          f'(host {1e3*(t1-t0)/N:.3f} ms), {B*H*W/((t2-t0)/N)/1e6:.0f} Mpix/s network pixels; '
          f'pq={float(logs["panoptic_all_deeplab_pq"]):.4f}')
    if '--profile' in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(18)
