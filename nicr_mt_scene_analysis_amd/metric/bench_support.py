"""Streaming metric accumulation for bench.py (BASELINE.json configs[3]).

Every step each rank updates step-local accumulators from its shard of the
batch (HIP kernels), sums them over the ranks with ONE all-reduce per dtype
(RCCL over xGMI: the int64 confusion matrix, the float64 PQ vectors — ~15 KB,
the only collective on the path) and adds the result to the replicated running
totals.  Mirrors `dist_reduce_fx='sum'` of reference metric/miou.py:21-25 and
metric/pq.py:228-246.
"""
import torch

from .. import ops
from .miou import MeanIntersectionOverUnion
from .pq import PanopticQuality


class MetricAccumulators:
    def __init__(self, n_classes_with_void: int, device, inputs, rank: int = 0,
                 max_instances_per_category: int = 1 << 16) -> None:
        self.max_inst = max_instances_per_category
        n = n_classes_with_void
        is_thing = [False] + [bool(x) for x in inputs['semantic_classes_is_thing'].cpu().tolist()]
        self.step_miou = MeanIntersectionOverUnion(n, ignore_first_class=True, device=device)
        self.step_pq = PanopticQuality(n, 0, self.max_inst, 256 ** 3, is_thing, device=device)
        # running totals: one flat buffer per step-metric (same packing as Metric._pack)
        self._step_flat = [next(iter(self.step_miou._pack().values())),
                           next(iter(self.step_pq._pack().values()))]
        self._total_flat = [torch.zeros_like(f) for f in self._step_flat]
        # synthetic ground truth: the prediction shifted by 3 px with a
        # void band, and uniformly random semantic labels
        r = ops.panoptic_pipeline(inputs['semantic_logits'], inputs['instance_center'],
                                  inputs['instance_offset'], inputs['semantic_classes_is_thing'])
        pan = r['panoptic']
        tgt = torch.roll(pan, shifts=(3, 3), dims=(1, 2)).contiguous()
        tgt[:, :3, :] = 0
        self.target_panoptic = tgt
        g = torch.Generator(device=device).manual_seed(99 + rank)
        self.target_semantic = torch.randint(0, n, pan.shape, device=device, generator=g,
                                             dtype=torch.int64).to(torch.uint8)

    def update(self, panoptic_pred: torch.Tensor) -> None:
        self.step_miou.zero_()
        self.step_pq.zero_()
        # miou.update(pan // max_inst, semantic target)   (task_helper/panoptic.py:123-126)
        self.step_miou.update_from_panoptic(panoptic_pred, self.target_semantic, self.max_inst)
        # pq.update(pan, panoptic target)                  (task_helper/panoptic.py:111-118)
        self.step_pq.update(panoptic_pred, self.target_panoptic)

    def all_reduce(self, dist=None) -> None:
        if dist is not None:
            self.step_miou.sync()
            self.step_pq.sync()
        for tot, stp in zip(self._total_flat, self._step_flat):
            tot += stp

    @property
    def total_confmat(self) -> torch.Tensor:
        return self._total_flat[0].view_as(self.step_miou.confmat)

    @property
    def total_pq(self) -> torch.Tensor:
        return self._total_flat[1].view(4, -1)
