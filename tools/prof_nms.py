"""center NMS + top-k of the headline batch in a loop (for kernel stats / FETCH_SIZE passes)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nicr_mt_scene_analysis_amd import ops                           # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn      # noqa: E402
dev = torch.device('cuda', 0)
inp = syn.make_panoptic_inputs_torch(32, 40, 480, 640, n_centers=24, seed=1234, device=dev)
for _ in range(30):
    cen = ops.center_nms_topk(inp['instance_center'])
torch.cuda.synchronize()
