"""Minimal metric-state holder.

The reference subclasses `torchmetrics.Metric` only for `add_state` / `reset`
and the `dist_reduce_fx='sum'` declaration (metric/miou.py:21-25,
metric/pq.py:228-246).  torchmetrics is not a dependency here: states are plain
device tensors, `sync()` is ONE explicit all-reduce (RCCL when the process
group is 'nccl', gloo in the CPU tests) of all states packed per dtype.
"""
from typing import Dict, List, Optional

import torch


class Metric(torch.nn.Module):
    full_state_update = False

    def __init__(self, device: Optional[torch.device] = None, **kwargs) -> None:
        super().__init__()
        self._state_defaults: Dict[str, torch.Tensor] = {}
        self._state_reduce: Dict[str, Optional[str]] = {}
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

    def state_names(self) -> List[str]:
        return list(self._state_defaults.keys())

    def reset(self) -> None:
        for name, default in self._state_defaults.items():
            setattr(self, name, default.clone().to(self._device))

    def to(self, device, *args, **kwargs):          # keeps `.to(device)` of the reference API
        self._device = torch.device(device)
        for name in self._state_defaults:
            setattr(self, name, getattr(self, name).to(self._device))
        return self

    def sync(self, process_group=None) -> None:
        """Sum every 'sum' state over the ranks: one all-reduce per dtype."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size(process_group) == 1:
            return
        names = [n for n, fx in self._state_reduce.items() if fx == 'sum']
        backend = dist.get_backend(process_group)
        comm_device = self._device if backend == 'nccl' else torch.device('cpu')
        by_dtype: Dict[torch.dtype, List[str]] = {}
        for n in names:
            by_dtype.setdefault(getattr(self, n).dtype, []).append(n)
        for dtype, group in by_dtype.items():
            flat = torch.cat([getattr(self, n).reshape(-1) for n in group]).to(comm_device)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)
            flat = flat.to(self._device)
            at = 0
            for n in group:
                t = getattr(self, n)
                setattr(self, n, flat[at:at + t.numel()].reshape(t.shape).clone())
                at += t.numel()
