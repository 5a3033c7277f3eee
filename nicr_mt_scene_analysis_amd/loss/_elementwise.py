"""Shared implementation of `MSELoss` / `L1Loss` (reference loss/mse.py, loss/l1.py).

reduction='sum' (what every task helper uses) on [B,H,W] / [B,C,H,W] inputs runs
in the HIP kernels k_elem_fwd / k_elem_bwd, optionally with the task helpers'
`pred*mask` folded in (`masked_sum`); 'mean' is the same kernel's sum divided by the pixel
count.  Inputs of any other rank go through the same kernels as ONE plane (2-D [N, C] rows:
sum_n mean_c f = the plane's sum / C); reduction='none' is k_elem_none (csrc/losses_forms.hip).
Only an empty input or a target that itself asks for a gradient is left to torch ops on the
device.  Host tensors raise: there is no CPU path.
"""
from typing import Optional, Tuple

import torch

from .. import _lib as L
from . import _functional as F_
from .base import LossBase


class _ElementwiseLoss(LossBase):
    _kind = 'mse'

    def __init__(self, reduction: str = 'sum') -> None:
        super().__init__()
        assert reduction in ('sum', 'mean', 'none')
        self._reduction = reduction

    def _pointwise(self, input_, target):
        d = input_ - target
        return d * d if self._kind == 'mse' else d.abs()

    def masked_sum(self, input_: torch.Tensor, target: torch.Tensor,
                   mask: Optional[torch.Tensor], expected_scale=None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
        """(sum_px mean_c f(input_*mask - target), sum(mask)) — the masking of
        task_helper/instance.py:129-139,154-167 without materialising input_*mask.
        `expected_scale`: see LossBase.forward; default = 1 / sum(mask)."""
        if expected_scale is None and self._can_speculate(input_) and input_.ndim in (3, 4) and \
                not target.requires_grad:
            loss, n, _ = self._speculative_single(self._kind, input_, target, mask=mask)
            return loss, n
        return F_.masked_elementwise_sum(input_, target, mask, self._kind, expected_scale)

    def _compute_loss(self, input_: torch.Tensor, target: torch.Tensor, expected_scale=None):
        L.require_device_tensor(input_, 'input_')        # no CPU path: raises for host tensors
        kernel_ok = input_.ndim in (3, 4) and input_.numel() > 0 and not target.requires_grad
        if self._reduction in ('sum', 'mean') and kernel_ok:
            n_px = input_.numel() // (input_.shape[1] if input_.ndim == 4 else 1)
            if expected_scale is None and self._kind != 'focal' and self._can_speculate(input_):
                loss, _, _ = self._speculative_single(self._kind, input_, target)
                if self._reduction == 'mean':
                    return loss / n_px, 1
                return loss, n_px
            loss, _ = F_.masked_elementwise_sum(input_, target, None, self._kind, expected_scale)
            if self._reduction == 'mean':
                return loss / n_px, 1           # mean over all elements == sum_px mean_c / n_px
            return loss, n_px
        plain = input_.numel() > 0 and not target.requires_grad and self._kind in ('mse', 'l1')
        if plain and self._reduction in ('sum', 'mean'):
            # any other rank: ONE plane of numel elements through the same kernels.  The reference
            # averages a 2-D input over its feature axis first (mse.py:30-34): sum_n mean_c f =
            # (sum over all elements) / C
            numel = input_.numel()
            flat = input_.reshape(1, numel, 1)
            total, _ = F_.masked_elementwise_sum(flat, target.expand_as(input_).reshape(1, numel, 1), None,
                                                 self._kind, None)
            if self._reduction == 'mean':
                return total / numel, 1
            if input_.ndim == 2:
                return total / input_.shape[1], input_.shape[0]
            return total, numel
        if plain and self._reduction == 'none':
            return F_.elementwise_none(input_, target, self._kind), input_.numel()
        # an empty input, or a target that itself asks for a gradient: plain torch ops ON THE DEVICE
        loss = self._pointwise(input_, target.to(input_.device))
        if self._reduction == 'sum':
            if loss.ndim in (2, 4):
                loss = loss.mean(dim=1)           # channel / feature axis
            return loss.sum(), loss.numel()
        if self._reduction == 'mean':
            return loss.mean(), 1
        return loss, input_.numel()
