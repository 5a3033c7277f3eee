"""`VonMisesLossBiternion` (reference loss/vonmises.py:18-51; Beyer et al., Biternion
Nets, GCPR 2015) on the HIP kernels k_vm_fwd / k_vm_bwd."""
from typing import Optional, Tuple

import torch

from .. import _lib as L
from . import _functional as F_
from .base import LossBase


class VonMisesLossBiternion(LossBase):
    def __init__(self, reduction: str = 'sum', kappa: float = 1.0) -> None:
        super().__init__()
        assert reduction in ('sum', 'none')
        self._kappa = kappa
        self._reduction = reduction

    def masked_sum(self, input_: torch.Tensor, target: torch.Tensor,
                   mask: Optional[torch.Tensor], expected_scale=None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
        """planar [B,2,H,W] prediction/target + [B,H,W] mask: the permute + boolean
        gather of task_helper/instance.py:186-216 folded into the kernel.
        `expected_scale`: see LossBase.forward; default = 1 / sum(mask)."""
        if expected_scale is None and self._can_speculate(input_) and not target.requires_grad:
            loss, n, _ = self._speculative_single('vonmises', input_, target, mask=mask, param=self._kappa)
            return loss, n
        return F_.vonmises_sum(input_, target, mask, self._kappa, expected_scale)

    def _compute_loss(self, input_: torch.Tensor, target: torch.Tensor, expected_scale=None):
        if input_.ndim != 2 or target.ndim != 2:
            raise ValueError(
                f"expected biternion rows of shape (n, 2), got {tuple(input_.shape)} / "
                f"{tuple(target.shape)}; permute (b, 2, h, w) to (b, h, w, 2) and flatten to "
                "(b*h*w, 2) first, or call masked_sum() with the planar tensors")
        L.require_device_tensor(input_, 'input_')        # no CPU path: raises for host tensors
        n = input_.shape[0]
        if n == 0 or target.requires_grad:
            # an empty input, or a target that itself asks for a gradient: torch ops ON THE DEVICE
            cos = (input_ * target.to(input_.device)).sum(dim=1, keepdim=True)
            score = 1 - torch.exp(self._kappa * (cos - 1))
            return (score.sum() if self._reduction == 'sum' else score), score.numel()
        if self._reduction == 'none':
            if input_.shape[1] != 2:
                raise ValueError(f'expected biternion rows of shape (n, 2), got {tuple(input_.shape)}')
            rows = F_.vonmises_rows(input_, target, self._kappa)          # k_vm_rows: [n, 1] like the reference's
            out_dtype = torch.result_type(input_, target)
            return (rows.to(out_dtype) if out_dtype in (torch.bfloat16, torch.float16) else rows), n
        # rows (n, 2) -> planar (1, 2, n, 1) for the kernel (autograd carries the transpose)
        x = input_.t().contiguous().view(1, 2, n, 1)
        y = target.t().contiguous().view(1, 2, n, 1)
        if expected_scale is None and self._can_speculate(x):
            loss, _, _ = self._speculative_single('vonmises', x, y, param=self._kappa)
            return loss, n
        loss, _ = F_.vonmises_sum(x, y, None, self._kappa, expected_scale)
        return loss, n
