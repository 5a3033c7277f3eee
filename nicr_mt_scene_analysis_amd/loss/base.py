"""`LossBase` (reference loss/base.py:12-33): one (loss, n_elements) pair per scale."""
import abc
from typing import Sequence, Tuple

import torch


class LossBase(abc.ABC, torch.nn.Module):
    def __init__(self) -> None:
        super().__init__()

    @abc.abstractmethod
    def _compute_loss(self, input_: torch.Tensor, target: torch.Tensor):
        """-> (loss, number of loss elements)"""

    def forward(
        self,
        input_tensors: Sequence[torch.Tensor],
        target_tensors: Sequence[torch.Tensor]
    ) -> Tuple[Tuple[torch.Tensor, int], ...]:
        return tuple(self._compute_loss(i, t) for i, t in zip(input_tensors, target_tensors))
