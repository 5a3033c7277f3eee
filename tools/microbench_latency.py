"""Small-batch latency of the panoptic pipeline (robot-style B=1 inference) on one MI355X.
usage: python tools/microbench_latency.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nicr_mt_scene_analysis_amd import ops   # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402


def main():
    for B in (1, 2, 4, 8, 32):
        inp = syn.make_panoptic_inputs_torch(B, 40, 480, 640, device='cuda', seed=1)

        def run():
            return ops.panoptic_pipeline(inp['semantic_logits'], inp['instance_center'],
                                         inp['instance_offset'], inp['semantic_classes_is_thing'])
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        n = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        gpu_us = e0.elapsed_time(e1) / n * 1e3
        t0 = time.perf_counter()
        for _ in range(n):
            run()
        host_us = (time.perf_counter() - t0) / n * 1e6          # enqueue only
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            run()
            torch.cuda.synchronize()
        sync_us = (time.perf_counter() - t0) / 50 * 1e6          # one call, waited for
        print(f'B={B:2d}: back-to-back {gpu_us:7.1f} us/step  host enqueue {host_us:6.1f} us  '
              f'call+sync {sync_us:7.1f} us  ({B * 480 * 640 / gpu_us:8.1f} Mpix/s)')


if __name__ == '__main__':
    main()
