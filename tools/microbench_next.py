"""Times the SURVEY §8(f) kernels f3 (compute_scores) and f4 (target generation) on one MI355X.
usage: python tools/microbench_next.py [B]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops   # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3        # us


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    C, H, W = 40, 480, 640
    px = B * H * W
    inp = syn.make_panoptic_inputs_torch(B, C, H, W, device='cuda', seed=1)
    p = ops.panoptic_pipeline(inp['semantic_logits'], inp['instance_center'], inp['instance_offset'],
                              inp['semantic_classes_is_thing'], want_score=True)
    tab = torch.zeros((B, 256), dtype=torch.float32, device='cuda')
    tab[:, 1:] = p['center_scores'][:, :255]
    t = timeit(lambda: ops.panoptic_scores(inp['semantic_logits'], p['semantic_idx_u8'],
                                           p['semantic_score'], p['instance'], p['panoptic'],
                                           p['pan_of_inst'], tab, 1 << 16))
    alg = px * (4 + 1 + 1 + 8 + 4 + 4 + 1 + 8 + 8)
    print(f'f3 panoptic_scores        {t:8.1f} us  {alg/t/1e6:6.2f} TB/s  {px/t/1e3:7.2f} Gpx/s')

    for n_inst in (30, 200):
        m = syn.make_label_maps(2, C + 1, H, W, n_inst, seed=2, max_radius=None if n_inst < 100 else 40)
        reps = (B + 1) // 2
        sem = torch.from_numpy(np.tile(m['semantic'], (reps, 1, 1))[:B]).cuda()
        ins = torch.from_numpy(np.tile(m['instance'], (reps, 1, 1))[:B]).cuda()
        th = torch.from_numpy(m['semantic_classes_is_thing'].astype(np.uint8)).cuda()
        st = torch.from_numpy((~m['semantic_classes_is_thing']).astype(np.uint8)).cuda()
        ops.instance_clear_stuff(sem, ins, st)
        t = timeit(lambda: ops.instance_clear_stuff(sem, ins, st))
        print(f'f4 clear_stuff  n={n_inst:4d}     {t:8.1f} us  {px*(1+4+4)/t/1e6:6.2f} TB/s')
        t = timeit(lambda: ops.instance_targets(sem, ins, C + 1, th, st, 8, True))
        print(f'f4 instance_targets n={n_inst:4d} {t:8.1f} us  {px*(3*(1+4)+4+8+1+1)/t/1e6:6.2f} TB/s  '
              f'{px/t/1e3:7.2f} Gpx/s')
        t = timeit(lambda: ops.panoptic_targets(sem, ins, C + 1, th, 1 << 16))
        print(f'f4 panoptic_targets n={n_inst:4d} {t:8.1f} us  {px*(3*(1+4)+8)/t/1e6:6.2f} TB/s  '
              f'{px/t/1e3:7.2f} Gpx/s')
        r = ops.panoptic_targets(sem, ins, C + 1, th, 1 << 16)
        keys = r['ids_pan'][:, :64].contiguous()
        nk = torch.clamp(r['n_ids'], max=64)
        t = timeit(lambda: ops.dve_targets(r['panoptic'], keys, nk))
        print(f'f4 dve_indices  K=64         {t:8.1f} us  {px*(8+4)/t/1e6:6.2f} TB/s')


if __name__ == '__main__':
    main()
