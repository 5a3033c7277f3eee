import sys, time, gc, torch
sys.path.insert(0, '.')
from nicr_mt_scene_analysis_amd import ops
from nicr_mt_scene_analysis_amd.testing import synthetic as syn
from tools import bench_support
dev = torch.device('cuda:0')
B, C, H, W = 32, 40, 480, 640
inp = syn.make_panoptic_inputs_torch(B, C, H, W, n_centers=24, seed=1234, device=dev)
logits, center, offset = inp['semantic_logits'], inp['instance_center'], inp['instance_offset']
is_thing = inp['semantic_classes_is_thing']
metrics = bench_support.MetricAccumulators(C + 1, dev, inp, 0, side_stream=True, sync_every_step=False)
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
def step(i):
    with torch.cuda.stream(streams[i % 2]):
        r = ops.panoptic_pipeline(logits, center, offset, is_thing)
        metrics.update_and_reduce(r['panoptic'], None)
        e = torch.cuda.Event(enable_timing=True); e.record()
    return e
for K in (20, 20, 200):
    for i in range(5): step(i)
    torch.cuda.synchronize(); gc.collect(); gc.disable()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev = [step(i) for i in range(K)]
    th = time.perf_counter()
    metrics.finalize(None)
    torch.cuda.synchronize()
    t1 = time.perf_counter(); gc.enable()
    done = [e0.elapsed_time(e) for e in ev]
    print(f'K={K}: total {1e3*(t1-t0):.3f} ms ({1e3*(t1-t0)/K:.4f}/step), host issue {1e3*(th-t0):.3f} ms; step completion times (ms after start): '
          + ' '.join(f'{d:.2f}' for d in done[:6]) + ' ... ' + ' '.join(f'{d:.2f}' for d in done[-3:]))
