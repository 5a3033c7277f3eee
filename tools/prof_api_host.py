"""cProfile of the HOST side of PanopticPostprocessing.postprocess (B=32 640x480 C=40), deferred mode
so that nothing waits for the GPU: where do the ~0.27 ms of Python per call go?"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY   # noqa: E402
from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class  # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn                        # noqa: E402

dev = torch.device('cuda')
inp = syn.make_panoptic_inputs_torch(32, 40, 480, 640, device=dev, seed=1)
is_thing = tuple(bool(x) for x in inp['semantic_classes_is_thing'].cpu().tolist())
post = get_postprocessing_class('panoptic')(
    semantic_postprocessing=get_postprocessing_class('semantic')(),
    instance_postprocessing=get_postprocessing_class('instance')(),
    semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing,
    defer_host_sync=bool(int(os.environ.get('DEFER', '1'))))
batch = {'rgb_fullres': torch.zeros((32, 3, 480, 640)),
         APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize', 'valid_region_slice_y': slice(0, 480),
                                       'valid_region_slice_x': slice(0, 640)}]] * 32}
data = ((inp['semantic_logits'], (inp['instance_center'], inp['instance_offset'])), (None, None))
for _ in range(10):
    post.postprocess(data, batch, is_training=False)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    post.postprocess(data, batch, is_training=False)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(28)
