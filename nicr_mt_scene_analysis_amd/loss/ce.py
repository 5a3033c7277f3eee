"""`CrossEntropyLossSemantic` (reference loss/ce.py:13-68) on the HIP kernels
k_ce_fwd / k_ce_bwd.  `n_elements` is returned as a device int64 scalar (the
reference syncs with `.item()`); it compares / divides like the int it replaces."""
from typing import Optional, Tuple

import torch

from . import _functional as F_
from .base import LossBase


class CrossEntropyLossSemantic(LossBase):
    def __init__(
        self,
        weights: Optional[torch.Tensor] = None,
        label_smoothing: float = 0.0,
        weighted_reduction: bool = False    # the reduction used in ESANet
    ) -> None:
        super().__init__()
        self._weights = weights
        self._label_smoothing = label_smoothing
        self._weighted_reduction = weighted_reduction
        if weighted_reduction:
            assert self._weights is not None

    def _compute_loss(self, input_: torch.Tensor, target: torch.Tensor, expected_scale=None
                      ) -> Tuple[torch.Tensor, torch.Tensor]:
        # (the ESANet reduction divides by the weight sum, not by a count: nothing to expect)
        if expected_scale is None and not self._weighted_reduction and self._can_speculate(input_) \
                and input_.ndim == 4 and input_.shape[1] <= 255:
            # forward sum + gradient in one pass for the upstream gradient this instance has
            # learned to expect (callers divide the sum by a count of the targets)
            loss, n_elements, weight_sum = self._speculative_single(
                'ce', input_, None, mask=target, weights=self._weights, param=self._label_smoothing)
            return loss, n_elements
        loss, n_elements, weight_sum = F_.cross_entropy_sum(
            input_, target, self._weights, self._label_smoothing, expected_scale)
        if self._weighted_reduction:
            # sum(loss) / sum_c n_c * w_c  (ce.py:57-68): the divisor is the sum of the
            # label weights over the non-void pixels
            loss = loss / weight_sum.to(loss.dtype)
        return loss, n_elements
