"""`CenterFocalLoss` — EXTENSION, not in the reference (its center heat-map loss is MSE or L1,
task_helper/instance.py:60-67).  The penalty-reduced focal loss of CenterNet (Zhou et al., 2019)
with alpha = 2, beta = 4 on p = clamp(pred, 1e-4, 1 - 1e-4):

    -(1 - p)^2 log(p)                    where target == 1
    -(1 - target)^4 p^2 log(1 - p)       elsewhere

summed over the masked pixels; `n_elements` is the number of positive masked pixels (the usual
normaliser).  Kernel: k_elem_fwd / k_elem_bwd with KIND 2.  Parity is unpinned (no reference
function): the tests compare with a plain PyTorch fp32 implementation of the formula above.
"""
from typing import Optional, Tuple

import torch

from . import _functional as F_
from .base import LossBase


class CenterFocalLoss(LossBase):
    def __init__(self, reduction: str = 'sum') -> None:
        super().__init__()
        assert reduction == 'sum'

    def masked_sum(self, input_: torch.Tensor, target: torch.Tensor,
                   mask: Optional[torch.Tensor], expected_scale=None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
        """(loss sum over the masked pixels, max(#positive masked pixels, 1))"""
        # the divisor (number of heat-map peaks) is no count of mask bytes: no expectation, the
        # two-kernel path (forward, then backward) — also inside the multi-loss call
        loss, n_pos = F_.masked_elementwise_sum(input_, target, mask, 'focal', expected_scale)
        return loss, n_pos.clamp(min=1)

    def _compute_loss(self, input_: torch.Tensor, target: torch.Tensor, expected_scale=None):
        return self.masked_sum(input_, target, None, expected_scale)
