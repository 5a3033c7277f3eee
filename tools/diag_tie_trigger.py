import sys, torch
sys.path.insert(0, '.')
from nicr_mt_scene_analysis_amd import ops
from nicr_mt_scene_analysis_amd.testing import synthetic as syn
dev = torch.device('cuda:0')
inp = syn.make_panoptic_inputs_torch(32, 40, 480, 640, n_centers=24, seed=1, device=dev)
for scale in (1.0, 0.01):
    lg = inp['semantic_logits'] * scale
    f = lambda: ops.panoptic_pipeline(lg, inp['instance_center'], inp['instance_offset'], inp['semantic_classes_is_thing'])
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): f()
    e.record(); torch.cuda.synchronize()
    frac = float((lg.amax(dim=1).abs() < 0.5).float().mean())
    print(f'logits x{scale}: pipeline {s.elapsed_time(e)/20:.3f} ms/step, {100*frac:.1f}% of the pixels behind the trigger')
