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
# the same collective through the task helpers: `validation_epoch_end` -> `compute()` sums the
# states over the ranks (torchmetrics' sync_context), logs the global metric and leaves the
# rank-local states behind for the reset
from nicr_mt_scene_analysis_amd.task_helper import PanopticTaskHelper, SemanticTaskHelper   # noqa: E402
pan = PanopticTaskHelper(semantic_n_classes=5, semantic_classes_is_thing=[False, False, True, True, True])
pan.initialize(dev)
sem = SemanticTaskHelper(n_classes=4)
sem.initialize(dev)
pan._mae_pq_deeplab.update(pred * 65536, None, None, tgt * 65536, None, None,
                           miou=pan._metric_iou, semantic_target=tgt.to(torch.uint8), pred_div=65536)
sem._metric_iou.update_masked_void((pred - 1).clamp_(min=0), tgt.to(torch.uint8))
local_tp = pan._mae_pq_deeplab.tp_per_class.clone()
local_cm = pan._metric_iou.confmat.clone()
tp_all = [torch.zeros_like(local_tp) for _ in range(world)]
dist.all_gather(tp_all, local_tp)
res = pan._mae_pq_deeplab.compute(suffix='_deeplab')          # synced inside, local again behind it
assert torch.equal(pan._mae_pq_deeplab.tp_per_class, local_tp) and not pan._mae_pq_deeplab._is_synced
pan._mae_pq_deeplab.sync()
assert torch.equal(pan._mae_pq_deeplab.tp_per_class, torch.stack(tp_all).sum(0))
pan._mae_pq_deeplab.unsync()
artifacts, _, logs = pan.validation_epoch_end()
assert torch.equal(artifacts['panoptic_deeplab_semantic_cm'], local_cm)
assert torch.equal(logs['panoptic_all_deeplab_pq'], res['all_deeplab_pq'])
_, _, sem_logs = sem.validation_epoch_end()
assert 0.0 <= float(sem_logs['semantic_miou']) <= 1.0 and 0.0 <= float(logs['panoptic_deeplab_semantic_miou']) <= 1.0
assert int(pan._metric_iou.confmat.sum()) == 0
print(f'RCCL_SYNC_OK rank {rank} of {world} confmat_sum {int(miou.confmat.sum())}', flush=True)
dist.barrier()
dist.destroy_process_group()
