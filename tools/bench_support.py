"""Streaming metric accumulation for bench.py (BASELINE.json configs[3]).

Every step each rank updates accumulators from its shard of the batch (HIP kernels).
Under a process group the accumulators are summed over the ranks by the package's own
`Metric.sync()` — one in-place all-reduce per state dtype (RCCL over xGMI): the int64 confusion
matrix and the float64 PQ vectors, ~15 KB together, the only collective on the path — once after
the last step (default) or after every step.  Mirrors `dist_reduce_fx='sum'` of reference
metric/miou.py:21-25 and metric/pq.py:228-246.

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
                 side_stream: bool = True, sync_every_step: bool = True) -> None:
        """sync_every_step=False: ranks accumulate locally and `finalize()` sums the states
        over the ranks ONCE — the torchmetrics behaviour of the reference (`dist_reduce_fx`
        is applied by `compute()`, metric/miou.py:21-25).  The collective is the package's own
        `Metric.sync()` (one in-place all-reduce per state dtype: int64 confusion matrix,
        float64 PQ vectors) — the call `compute()` makes under an initialised process group."""
        self.max_inst = max_instances_per_category
        self.fused_metrics = not os.environ.get('NMSA_BENCH_SEPARATE_METRICS')
        import torch.distributed as dist_
        # without a process group the kernels accumulate straight into the totals
        self.sync_every_step = sync_every_step and dist_.is_available() and dist_.is_initialized()
        n = n_classes_with_void
        is_thing = [False] + [bool(x) for x in inputs['semantic_classes_is_thing'].cpu().tolist()]
        self._thing_lut = inputs['semantic_classes_is_thing'].to(device=device, dtype=torch.uint8).contiguous()
        # the bench reads the summed states themselves, so the metrics do not sum again in compute()
        self.miou = MeanIntersectionOverUnion(n, ignore_first_class=True, device=device,
                                              sync_on_compute=False)
        self.pq = PanopticQuality(n, 0, self.max_inst, 256 ** 3, is_thing, device=device,
                                  sync_on_compute=False)
        # NMSA_BENCH_METRIC_STREAMS=2 (experiment, default 1; needs the side stream and local
        # accumulation): consecutive steps alternate over TWO accumulator sets on two side streams
        # — two "virtual shards" whose states are added once at the end, exactly what the rank sum
        # does — so that the latency-bound chain of one step overlaps with the next step's instead
        # of queueing behind it.  Measured SLOWER: 25.3 vs 28.0 Gpix/s at K = 20 on one box — every
        # additional concurrent small kernel takes wave slots from the streaming kernel.
        n_sets = int(os.environ.get('NMSA_BENCH_METRIC_STREAMS', '1'))
        self._sets = [(self.miou, self.pq)]
        if side_stream and not self.sync_every_step and n_sets > 1:
            for _ in range(n_sets - 1):
                self._sets.append((MeanIntersectionOverUnion(n, ignore_first_class=True, device=device,
                                                             sync_on_compute=False),
                                   PanopticQuality(n, 0, self.max_inst, 256 ** 3, is_thing, device=device,
                                                   sync_on_compute=False)))
        self._turn = 0
        self._merged = len(self._sets) == 1
        flat = [next(iter(self.miou._pack().values())), next(iter(self.pq._pack().values()))]
        self.payload_bytes = sum(f.numel() * f.element_size() for f in flat)
        # per-step reduction: the states hold ONE step, summed over the ranks, and are added to
        # replicated running totals; otherwise the states are the (local, then summed) totals
        self._step_flat = flat
        self._total_flat = [torch.zeros_like(f) for f in flat] if self.sync_every_step else flat
        self._finalized = False
        # NMSA_BENCH_METRIC_PRIORITY=-1: a high-priority side stream (the metric chain of a batch
        # then does not queue behind the next batch's streaming kernels)
        prio = int(os.environ.get('NMSA_BENCH_METRIC_PRIORITY', '0'))
        self.stream = torch.cuda.Stream(device=device, priority=prio) if side_stream else None
        self._streams = [self.stream] + [torch.cuda.Stream(device=device, priority=prio)
                                         for _ in self._sets[1:]]
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

    def _side(self, device):
        cur = torch.cuda.current_stream(device)
        return cur, (self.stream if self.stream is not None else cur)

    def _merge_sets(self) -> None:
        """states of the extra accumulator sets -> the first one (on its stream, behind theirs)"""
        if self._merged:
            return
        for (miou, pq), st in zip(self._sets[1:], self._streams[1:]):
            self.stream.wait_stream(st)
            with torch.cuda.stream(self.stream):
                for total, part in zip(self._step_flat, (next(iter(miou._pack().values())),
                                                         next(iter(pq._pack().values())))):
                    total += part
                miou.zero_()
                pq.zero_()
        self._merged = True

    def _parts(self, r):
        """the pipeline's result as the `parts` of PanopticQuality.update_with_miou_parts (the
        metric pass then reads 2 B/px of labels instead of the 8 B/px painted map);
        NMSA_BENCH_METRIC_PARTS=0: always the map"""
        if not isinstance(r, dict) or os.environ.get('NMSA_BENCH_METRIC_PARTS', '1') == '0':
            return None
        return {'panoptic': r['panoptic'], 'semantic_idx_u8': r['semantic_idx_u8'], 'instance': r['instance'],
                'pan_of_inst': r['pan_of_inst'], 'is_thing': self._thing_lut, 'void_label': 0,
                'max_instances_per_category': self.max_inst}

    def update_and_reduce(self, panoptic_pred, dist=None) -> None:
        """panoptic_pred: the painted map, or the whole result dict of ops.panoptic_pipeline (the
        update then reads the parts the map was painted from)"""
        parts = self._parts(panoptic_pred)
        if isinstance(panoptic_pred, dict):
            panoptic_pred = panoptic_pred['panoptic']
        cur, stream = self._side(panoptic_pred.device)
        k = self._turn % len(self._sets)
        self._turn += 1
        miou, pq = self._sets[k]
        if len(self._sets) > 1:
            stream = self._streams[k]
            self._merged = False
        if stream is not cur:
            self._ready.record(cur)
            stream.wait_event(self._ready)
            panoptic_pred.record_stream(stream)       # caching allocator: used on the side stream
            if parts is not None:
                for k in ('semantic_idx_u8', 'instance', 'pan_of_inst'):
                    parts[k].record_stream(stream)
        with torch.cuda.stream(stream):
            # miou.update(pan // max_inst, semantic target)   (task_helper/panoptic.py:123-126)
            # pq.update(pan, panoptic target)                  (task_helper/panoptic.py:111-118)
            # -> one pass over the prediction for both accumulators
            if self.fused_metrics and parts is not None:
                pq.update_with_miou_parts(parts, self.target_panoptic, miou, self.target_semantic, self.max_inst)
            elif self.fused_metrics:
                pq.update_with_miou(panoptic_pred, self.target_panoptic, miou,
                                    self.target_semantic, self.max_inst)
            else:
                miou.update_from_panoptic(panoptic_pred, self.target_semantic, self.max_inst)
                pq.update(panoptic_pred, self.target_panoptic)
            if self.sync_every_step:
                self.miou.sync()
                self.pq.sync()
                for total, step in zip(self._total_flat, self._step_flat):
                    total += step
                self.miou.reset()                   # next step starts from zero, not synced
                self.pq.reset()

    def enqueue(self, panoptic_pred) -> None:
        """the update kernels of the first accumulator set on the CURRENT stream, nothing else
        (what a hipGraph of the metric chain captures)"""
        miou, pq = self._sets[0]
        parts = self._parts(panoptic_pred)
        if isinstance(panoptic_pred, dict):
            panoptic_pred = panoptic_pred['panoptic']
        if self.fused_metrics and parts is not None:
            pq.update_with_miou_parts(parts, self.target_panoptic, miou, self.target_semantic, self.max_inst)
        elif self.fused_metrics:
            pq.update_with_miou(panoptic_pred, self.target_panoptic, miou, self.target_semantic, self.max_inst)
        else:
            miou.update_from_panoptic(panoptic_pred, self.target_semantic, self.max_inst)
            pq.update(panoptic_pred, self.target_panoptic)

    def warm_collective(self, dist=None) -> None:
        """untimed: one all-reduce per state dtype on scratch buffers so that communicator set-up
        and the first-use costs of the collective do not land in the timed region"""
        if dist is None:
            return
        _, stream = self._side(self._step_flat[0].device)
        with torch.cuda.stream(stream):
            for f in self._step_flat:
                scratch = torch.zeros_like(f)
                if dist.get_backend() == 'nccl':
                    dist.all_reduce(scratch, op=dist.ReduceOp.SUM)
                else:
                    dist.all_reduce(scratch.cpu(), op=dist.ReduceOp.SUM)

    def finalize(self, dist=None) -> None:
        """end of the epoch: with local accumulation, sum the states over the ranks
        (`Metric.sync()`); a no-op when every step was already reduced or without a group"""
        self._merge_sets()
        if self.sync_every_step or self._finalized or dist is None:
            return
        _, stream = self._side(self._step_flat[0].device)
        with torch.cuda.stream(stream):
            self.miou.sync()
            self.pq.sync()
        self._finalized = True

    def wait(self) -> None:
        """make the current stream wait for everything enqueued on the side stream(s)"""
        if self.stream is not None:
            self._merge_sets()
            torch.cuda.current_stream().wait_stream(self.stream)

    @property
    def total_confmat(self) -> torch.Tensor:
        self._merge_sets()
        return self._total_flat[0].view_as(self.miou.confmat)

    @property
    def total_pq(self) -> torch.Tensor:
        self._merge_sets()
        return self._total_flat[1].view(4, -1)
