"""Minimal metric-state holder.

The reference subclasses `torchmetrics.Metric` only for `add_state` / `reset`
and the `dist_reduce_fx='sum'` declaration (metric/miou.py:21-25,
metric/pq.py:228-246).  torchmetrics is not a dependency here: states are views
into ONE flat device buffer per dtype, so that

  * `sync()` is one in-place all-reduce per dtype (RCCL when the process group
    is 'nccl', gloo in the CPU tests) — no packing kernels, no copies,
  * `zero_()` / `reset()` is one memset per dtype.

Distributed semantics are torchmetrics' (what `dist_reduce_fx='sum'` means under an initialised
process group): every subclass's `compute()` is wrapped — under a process group it caches the rank-local
states, sums them over the ranks, computes and restores the local states (`sync_context`), so every
rank reports the GLOBAL metric while `update()` keeps accumulating locally; a state read outside
`compute()` (the confusion-matrix artifacts of the task helpers) is rank-local, as in the reference.
`Metric(sync_on_compute=False)` switches that off; `sync()` / `unsync()` do it by hand.
"""
import contextlib
import functools
from typing import Dict, Iterator, List, Optional

import torch


class Metric(torch.nn.Module):
    full_state_update = False

    def __init__(self, device: Optional[torch.device] = None, sync_on_compute: bool = True,
                 process_group=None, **kwargs) -> None:
        super().__init__()
        self.sync_on_compute = sync_on_compute
        self.process_group = process_group
        self._is_synced = False
        self._cache: Optional[Dict[torch.dtype, torch.Tensor]] = None
        self._compute_depth = 0
        self._state_defaults: Dict[str, torch.Tensor] = {}
        self._state_reduce: Dict[str, Optional[str]] = {}
        self._flat: Optional[Dict[torch.dtype, torch.Tensor]] = None
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device()) \
                if torch.cuda.is_available() else torch.device('cpu')
        self._device = torch.device(device)

    @property
    def device(self) -> torch.device:
        return self._device

    def add_state(self, name: str, default: torch.Tensor,
                  dist_reduce_fx: Optional[str] = None) -> None:
        self._state_defaults[name] = default.detach().clone()
        self._state_reduce[name] = dist_reduce_fx
        setattr(self, name, default.detach().clone().to(self._device))
        self._flat = None                       # re-pack on next use

    def state_names(self) -> List[str]:
        return list(self._state_defaults.keys())

    def _pack(self) -> Dict[torch.dtype, torch.Tensor]:
        """Move the states into one flat buffer per dtype (states become views)."""
        if self._flat is not None:
            return self._flat
        by_dtype: Dict[torch.dtype, List[str]] = {}
        for n in self._state_defaults:
            by_dtype.setdefault(getattr(self, n).dtype, []).append(n)
        flat: Dict[torch.dtype, torch.Tensor] = {}
        for dtype, names in by_dtype.items():
            total = sum(getattr(self, n).numel() for n in names)
            buf = torch.zeros((total,), dtype=dtype, device=self._device)
            at = 0
            for n in names:
                t = getattr(self, n)
                k = t.numel()
                buf[at:at + k].copy_(t.reshape(-1))
                setattr(self, n, buf[at:at + k].view(t.shape))
                at += k
            flat[dtype] = buf
        self._flat = flat
        return flat

    def zero_(self) -> None:
        """In-place reset of all states (one memset per dtype)."""
        for buf in self._pack().values():
            buf.zero_()

    def reset(self) -> None:
        self._pack()
        self._cache = None
        self._is_synced = False
        all_zero = all(not bool(d.any()) for d in self._state_defaults.values())
        if all_zero:
            self.zero_()
        else:
            for name, default in self._state_defaults.items():
                getattr(self, name).copy_(default)

    def to(self, device, *args, **kwargs):          # keeps `.to(device)` of the reference API
        self._device = torch.device(device)
        for name in self._state_defaults:
            setattr(self, name, getattr(self, name).to(self._device).clone())
        self._flat = None
        return self

    # ------------------------------------------------------------------ distributed
    def compute(self, *args, **kwargs):
        raise NotImplementedError

    def __init_subclass__(cls, **kwargs):
        """every subclass's own `compute` runs inside a `sync_context` — at class level (copies,
        pickles and `deepcopy` of a metric keep working: nothing is bound to an instance), the
        OUTERMOST call only: `super().compute()` inside a subclass must not sum a second time"""
        super().__init_subclass__(**kwargs)
        for name, fn in list(cls.__dict__.items()):         # update*(): never on synced states
            if name.startswith('update') and callable(fn) and not isinstance(fn, (staticmethod, classmethod)) \
                    and not getattr(fn, '_nmsa_guarded', False):
                def guarded(self, *args, _fn=fn, **kw):
                    self._require_unsynced()
                    return _fn(self, *args, **kw)
                functools.update_wrapper(guarded, fn)
                guarded._nmsa_guarded = True
                setattr(cls, name, guarded)
        compute = cls.__dict__.get('compute')
        if compute is None or getattr(compute, '_nmsa_sync_wrapped', False):
            return

        @functools.wraps(compute)
        def wrapped(self, *args, **kw):
            if self._compute_depth:
                return compute(self, *args, **kw)
            self._compute_depth += 1
            try:
                with self.sync_context(should_sync=self.sync_on_compute):
                    return compute(self, *args, **kw)
            finally:
                self._compute_depth -= 1
        wrapped._nmsa_sync_wrapped = True
        cls.compute = wrapped

    def _require_unsynced(self) -> None:
        """an update between `sync()` and `unsync()` would land in the rank-SUMMED states and be
        thrown away by `unsync()` (torchmetrics raises here too)"""
        if self._is_synced:
            raise RuntimeError(f"{type(self).__name__}: the states are synced over the ranks (sync() without "
                               "unsync()); call unsync() or reset() before updating again")

    def _world_size(self, process_group=None) -> int:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 0                                # no process group: nothing to sum over
        return dist.get_world_size(process_group)

    def sync(self, process_group=None, should_sync: bool = True) -> None:
        """Sum the states over the ranks: one in-place all-reduce per dtype.  The rank-local
        states are kept aside for `unsync()`.  A no-op without an initialised process group or
        when the states are summed already (like torchmetrics, a one-rank group does run the
        collective: the single-GPU rehearsal of the multi-rank path)."""
        import torch.distributed as dist
        process_group = process_group if process_group is not None else self.process_group
        if not should_sync or self._is_synced or self._world_size(process_group) == 0:
            return
        if any(fx not in ('sum', None) for fx in self._state_reduce.values()):
            raise NotImplementedError('only dist_reduce_fx="sum" states exist on this path')
        backend = dist.get_backend(process_group)
        flat = self._pack()
        self._cache = {dtype: buf.clone() for dtype, buf in flat.items()}
        for buf in flat.values():
            if backend == 'nccl' or buf.device.type == 'cpu':
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=process_group)
            else:                                   # gloo with device states (tests)
                host = buf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=process_group)
                buf.copy_(host)
        self._is_synced = True

    def unsync(self, should_unsync: bool = True) -> None:
        """Back to the rank-local states `sync()` put aside (accumulation goes on per rank)."""
        if not should_unsync or not self._is_synced:
            return
        for dtype, buf in self._pack().items():
            buf.copy_(self._cache[dtype])
        self._cache = None
        self._is_synced = False

    @contextlib.contextmanager
    def sync_context(self, process_group=None, should_sync: bool = True,
                     should_unsync: bool = True) -> Iterator[None]:
        """States summed over the ranks inside the block, rank-local again behind it.  States
        that were summed by hand before (`sync()`) stay as they are."""
        mine = should_sync and not self._is_synced
        self.sync(process_group=process_group, should_sync=should_sync)
        try:
            yield
        finally:
            self.unsync(should_unsync=mine and should_unsync)
