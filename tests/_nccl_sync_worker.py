"""Worker of tests/test_bench_launch.py::test_metric_sync_on_rccl (run under torch.distributed.run):
`Metric.sync()` — one in-place all-reduce per state dtype (int64 confusion matrix, float64 PQ
vectors) — on the 'nccl' (= RCCL) backend with device-resident states."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality   # noqa: E402

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
miou = MeanIntersectionOverUnion(5, device=dev)
pq = PanopticQuality(5, 0, 65536, 256 ** 3, [False, False, True, True, True], device=dev)
g = torch.Generator(device=dev).manual_seed(3 + rank)
pred = torch.randint(0, 5, (2, 32, 48), device=dev, generator=g)
tgt = torch.randint(0, 5, (2, 32, 48), device=dev, generator=g)
miou.update(pred, tgt)
pq.update(pred * 65536, tgt * 65536)
before = (miou.confmat.clone(), pq.tp_per_class.clone(), pq.iou_per_class.clone())
miou.sync()
pq.sync()
torch.cuda.synchronize()
# every rank holds the sum over the ranks; with one rank: unchanged
gathered = [torch.zeros_like(before[0]) for _ in range(world)]
dist.all_gather(gathered, before[0])
assert torch.equal(miou.confmat, torch.stack(gathered).sum(0)), 'confmat != sum over ranks'
assert miou.confmat.dtype == torch.int64 and pq.tp_per_class.dtype == torch.float64
if world == 1:
    assert torch.equal(pq.tp_per_class, before[1]) and torch.equal(pq.iou_per_class, before[2])
print(f'RCCL_SYNC_OK rank {rank} of {world} confmat_sum {int(miou.confmat.sum())}', flush=True)
dist.barrier()
dist.destroy_process_group()
