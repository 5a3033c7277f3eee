#!/usr/bin/env python3
"""Per-op timings of the C-ABI entry points in isolation (HIP events, B=32 640x480 C=40).
Run on the GPU box:  python tools/microbench.py [op ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops                      # noqa: E402
from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality   # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402


def timeit(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


def main():
    B, C, H, W = 32, 40, 480, 640
    dev = torch.device('cuda')
    inp = syn.make_panoptic_inputs_torch(B, C, H, W, device=dev, seed=1)
    r = ops.panoptic_pipeline(inp['semantic_logits'], inp['instance_center'], inp['instance_offset'],
                              inp['semantic_classes_is_thing'])
    pan = r['panoptic']
    tgt = torch.roll(pan, (3, 3), (1, 2)).contiguous()
    tgt[:, :3] = 0
    sem_t = torch.randint(0, C + 1, pan.shape, device=dev).to(torch.uint8)
    is_thing = [False] + inp['semantic_classes_is_thing'].cpu().tolist()
    pq = PanopticQuality(C + 1, 0, 1 << 16, 256 ** 3, is_thing, device=dev)
    miou = MeanIntersectionOverUnion(C + 1, True, device=dev)
    px = B * H * W
    from nicr_mt_scene_analysis_amd import _lib as L
    pan_out = torch.empty_like(pan)
    thing_u8 = inp['semantic_classes_is_thing'].view(torch.uint8)

    def paint():
        L.check(L.lib().nmsa_panoptic_paint(
            L.ptr(r['semantic_idx_u8']), L.ptr(r['instance']), L.ptr(r['pan_of_inst']),
            L.ptr(thing_u8), B, C, H, W, 1 << 16, 0, L.ptr(pan_out), None, L.stream_ptr(dev)), 'paint')
    tests = {
        'pipeline': (lambda: ops.panoptic_pipeline(inp['semantic_logits'], inp['instance_center'],
                                                   inp['instance_offset'],
                                                   inp['semantic_classes_is_thing']), 181),
        'center_nms': (lambda: ops.center_nms_topk(inp['instance_center']), 4),
        'paint': (lambda: paint(), 10),
        'pq_update': (lambda: pq.update(pan, tgt), 16),
        'confmat_pan_u8': (lambda: miou.update_from_panoptic(pan, sem_t, 65536), 9),
        'confmat_u8_u8': (lambda: miou.update(r['semantic_idx_u8'], sem_t), 2),
        'argmax': (lambda: ops.semantic_argmax(inp['semantic_logits'], want_u8=True, want_i64=False,
                                               want_score=True), 4 * C + 5),
        'softmax': (lambda: ops.semantic_softmax(inp['semantic_logits']), 8 * C),
        'group_offsets': (lambda: ops.group_offsets(inp['instance_offset'], r['foreground'],
                                                    r['centers_yx'], r['n_centers'], H, W), 10),
    }
    sem_i64 = (r['semantic_idx_u8'].long() + 1)
    lut = torch.zeros((C + 1,), dtype=torch.uint8, device=dev)
    lut[1:] = inp['semantic_classes_is_thing'].to(torch.uint8)
    ins_i32 = r['instance'].int() * 37          # sparse "ground-truth" ids
    tests['merge_u8'] = (lambda: ops.panoptic_merge(sem_i64, r['instance'], r['foreground'], lut,
                                                    1 << 16, 0), 8 + 1 + 1 + 8)
    tests['merge_wide'] = (lambda: ops.panoptic_merge_wide(sem_i64, ins_i32, r['foreground'], lut,
                                                           1 << 16, 0), 8 + 4 + 1 + 8)
    tests['orientation'] = (lambda: ops.instance_orientation_sums(
        inp['instance_offset'], r['instance'], r['foreground']), 8 + 1 + 1)
    want = sys.argv[1:] or list(tests)
    for name in want:
        fn, bpp = tests[name]
        us = timeit(fn)
        print(f'{name:18s} {us:9.1f} us   {px * bpp / us / 1e6:8.2f} TB/s (algorithmic {bpp} B/px)')


if __name__ == '__main__':
    main()
