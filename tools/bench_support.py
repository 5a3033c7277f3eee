"""Streaming metric accumulation for bench.py (BASELINE.json configs[3]).

Every step each rank updates accumulators from its shard of the batch (HIP kernels).
With more than one rank the step-local accumulators are summed over the ranks with ONE
all-reduce (RCCL over xGMI): the int64 confusion matrix (step-local counts < 2^53, exact in
float64) and the float64 PQ vectors travel in one packed float64 buffer of ~15 KB — the only
collective on the path — and are added to the replicated running totals; with a single rank
the kernels accumulate into the totals directly.  Mirrors
`dist_reduce_fx='sum'` of reference metric/miou.py:21-25 and metric/pq.py:228-246.

The metric kernels are enqueued on a side HIP stream: they only depend on the panoptic
map of their own step, so the latency-bound per-image kernels (matching, accumulation,
all-reduce) overlap with the next step's streaming kernels.
"""
import os

import torch

from nicr_mt_scene_analysis_amd import ops
from nicr_mt_scene_analysis_amd.metric.miou import MeanIntersectionOverUnion
from nicr_mt_scene_analysis_amd.metric.pq import PanopticQuality


class MetricAccumulators:
    def __init__(self, n_classes_with_void: int, device, inputs, rank: int = 0,
                 max_instances_per_category: int = 1 << 16, world_size: int = 1,
                 side_stream: bool = True, sync_every_step: bool = True,
                 exercise_collective: bool = False) -> None:
        """sync_every_step=False: ranks accumulate locally and `finalize()` sums the totals
        over the ranks ONCE — the torchmetrics behaviour of the reference (`dist_reduce_fx`
        is applied by `compute()`, metric/miou.py:21-25)."""
        self.max_inst = max_instances_per_category
        self.fused_metrics = not __import__('os').environ.get('NMSA_BENCH_SEPARATE_METRICS')
        # exercise_collective: take the multi-rank code path (pack, all-reduce, unpack) also in
        # a 1-rank process group — how a 1-GPU box rehearses the RCCL leg of `bench.py --gpus N`
        if exercise_collective:
            world_size = max(world_size, 2)
        self.reduce_world = world_size
        self.world_size = world_size if sync_every_step else 1      # per-step behaviour
        n = n_classes_with_void
        is_thing = [False] + [bool(x) for x in inputs['semantic_classes_is_thing'].cpu().tolist()]
        self.miou = MeanIntersectionOverUnion(n, ignore_first_class=True, device=device)
        self.pq = PanopticQuality(n, 0, self.max_inst, 256 ** 3, is_thing, device=device)
        # flat state buffers (same packing as Metric._pack); totals only when all-reducing
        self._step_flat = [next(iter(self.miou._pack().values())),
                           next(iter(self.pq._pack().values()))]
        # per-step reduction keeps separate running totals; local accumulation adds straight
        # into the metric states and `finalize()` reduces those in place
        self._total_flat = [torch.zeros_like(f) for f in self._step_flat] \
            if self.world_size > 1 else self._step_flat
        n_conf = self._step_flat[0].numel()
        self._packed = torch.zeros((n_conf + self._step_flat[1].numel(),), dtype=torch.float64,
                                   device=device) if world_size > 1 else None
        self._finalized = False
        self._n_conf = n_conf
        # NMSA_BENCH_METRIC_PRIORITY=-1: a high-priority side stream (the metric chain of a batch
        # then does not queue behind the next batch's streaming kernels)
        prio = int(os.environ.get('NMSA_BENCH_METRIC_PRIORITY', '0'))
        self.stream = torch.cuda.Stream(device=device, priority=prio) if side_stream else None
        self._ready = torch.cuda.Event()
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
        torch.cuda.synchronize(device)

    def update_and_reduce(self, panoptic_pred: torch.Tensor, dist=None) -> None:
        cur = torch.cuda.current_stream(panoptic_pred.device)
        stream = self.stream if self.stream is not None else cur
        if stream is not cur:
            self._ready.record(cur)
            stream.wait_event(self._ready)
            panoptic_pred.record_stream(stream)       # caching allocator: used on the side stream
        with torch.cuda.stream(stream):
            if self.world_size > 1:
                self.miou.zero_()
                self.pq.zero_()
            # miou.update(pan // max_inst, semantic target)   (task_helper/panoptic.py:123-126)
            # pq.update(pan, panoptic target)                  (task_helper/panoptic.py:111-118)
            # -> one pass over the prediction for both accumulators
            if self.fused_metrics:
                self.pq.update_with_miou(panoptic_pred, self.target_panoptic, self.miou,
                                         self.target_semantic, self.max_inst)
            else:
                self.miou.update_from_panoptic(panoptic_pred, self.target_semantic, self.max_inst)
                self.pq.update(panoptic_pred, self.target_panoptic)
            if self.world_size > 1:
                n = self._n_conf
                self._packed[:n].copy_(self._step_flat[0])            # i64 -> f64, exact
                self._packed[n:].copy_(self._step_flat[1])
                if dist is not None:
                    self._all_reduce(dist, self._packed)
                self._total_flat[0] += self._packed[:n].to(torch.int64)
                self._total_flat[1] += self._packed[n:]

    def warm_collective(self, dist=None) -> None:
        """untimed: run the packed all-reduce once on scratch data so that communicator set-up
        and the first-use costs of the collective do not land in the timed region"""
        if dist is None or self._packed is None:
            return
        cur = torch.cuda.current_stream(self._packed.device)
        stream = self.stream if self.stream is not None else cur
        with torch.cuda.stream(stream):
            self._all_reduce(dist, torch.zeros_like(self._packed))

    def finalize(self, dist=None) -> None:
        """end of the epoch: with local accumulation, sum the totals over the ranks (one
        all-reduce); a no-op when every step was already reduced or there is one rank"""
        if self.world_size > 1 or self.reduce_world == 1 or self._finalized or dist is None:
            return
        cur = torch.cuda.current_stream(self._packed.device)
        stream = self.stream if self.stream is not None else cur
        with torch.cuda.stream(stream):
            n = self._n_conf
            self._packed[:n].copy_(self._total_flat[0])               # < 2^53: exact in f64
            self._packed[n:].copy_(self._total_flat[1])
            self._all_reduce(dist, self._packed)
            self._total_flat[0].copy_(self._packed[:n].to(torch.int64))
            self._total_flat[1].copy_(self._packed[n:])
        self._finalized = True

    @staticmethod
    def _all_reduce(dist, buf: torch.Tensor) -> None:
        if dist.get_backend() == 'nccl':
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        else:                                   # gloo rehearsal with device states
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            buf.copy_(host)

    def wait(self) -> None:
        """make the current stream wait for everything enqueued on the side stream"""
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    @property
    def total_confmat(self) -> torch.Tensor:
        return self._total_flat[0].view_as(self.miou.confmat)

    @property
    def total_pq(self) -> torch.Tensor:
        return self._total_flat[1].view(4, -1)
