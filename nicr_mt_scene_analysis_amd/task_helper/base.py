"""`TaskHelperBase` and the step decorators (reference task_helper/base.py:17-210).

Same interface (initialize / training_step / validation_step / validation_epoch_end,
collect_* helpers, `accumulate_losses`, `mark_as_total`).  Element counts may be device
scalars here (the HIP losses return them without a host sync); `accumulate_losses` then
stays on the device as well.
"""
import abc
import functools
import time
import warnings
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from ..data.preprocessing.multiscale_supervision import get_downscale
from ..types import BatchType

TOTAL_LOSS_SUFFIX = '_total_loss'


def get_total_loss_key(key: str) -> str:
    return f'{key}{TOTAL_LOSS_SUFFIX}'


def append_detached_losses_to_logs(disabled: bool = False):
    def decorator(f):
        @functools.wraps(f)
        def wrapper(*args, **kwargs):
            losses, logs = f(*args, **kwargs)
            if not disabled:
                logs.update({k: v.detach().clone() for k, v in losses.items()})
            return losses, logs
        return wrapper
    return decorator


def append_profile_to_logs(key: str, disabled: bool = False):
    def decorator(f):
        @functools.wraps(f)
        def wrapper(*args, **kwargs):
            if disabled:
                return f(*args, **kwargs)
            t0 = time.perf_counter()
            results = f(*args, **kwargs)
            assert isinstance(results[-1], dict)       # last element: dict of logs
            results[-1][key] = time.perf_counter() - t0
            return results
        return wrapper
    return decorator


class TaskHelperBase(abc.ABC, torch.nn.Module):
    def __init__(self) -> None:
        super().__init__()

    def initialize(self, device: torch.device):
        pass

    def collect_predictions_and_targets_for_loss(
        self, batch: BatchType, batch_key: str, predictions_post: BatchType,
        predictions_post_key: str, side_outputs_key: Optional[str] = None
    ) -> Tuple[List[torch.Tensor], List[torch.Tensor], List[str]]:
        inputs, keys, downscales = self.collect_predictions_for_loss(
            predictions_post, predictions_post_key, side_outputs_key)
        targets = self.collect_targets_for_loss(batch, batch_key, downscales)
        return inputs, targets, keys

    def collect_predictions_for_loss(
        self, predictions_post: BatchType, predictions_post_key: str,
        side_outputs_key: Optional[str] = None
    ) -> Tuple[List[Any], List[str], List[int]]:
        def width(output):
            if isinstance(output, torch.Tensor):
                return output.shape[-1]
            if isinstance(output, tuple):                  # instance: tuple of tensors
                return output[0].shape[-1]
            raise Exception("Error while determining downscale")

        tensors = [predictions_post[predictions_post_key]]
        keys = ['main']
        downscales: List[int] = []
        if side_outputs_key is not None:
            main_width = width(predictions_post[predictions_post_key])
            for side in predictions_post[side_outputs_key] or ():
                if side is None:                           # no side outputs / eval mode
                    continue
                tensors.append(side)
                downscales.append(main_width // width(side))
                keys.append(f'down_{downscales[-1]}')
        return tensors, keys, downscales

    def collect_targets_for_loss(
        self, batch: BatchType, batch_key: str, downscales: Optional[List[int]] = None
    ) -> List[torch.Tensor]:
        targets = [batch[batch_key]]
        for d in downscales or ():
            sub = get_downscale(batch, d)
            if sub is None:                                # multiscale disabled
                continue
            targets.append(sub[batch_key])
        return targets

    def accumulate_losses(self, losses: Sequence[torch.Tensor], n_elements: Sequence) -> torch.Tensor:
        """sum(losses) / sum(n_elements) over main + side outputs (base.py:161-182)."""
        total_loss = torch.sum(torch.stack(list(losses)))
        total_n = sum(n_elements)
        if isinstance(total_n, torch.Tensor):
            # device count: no host sync; n == 0 returns the (zero) loss sum like the reference
            safe = total_n.clamp(min=1).to(total_loss.dtype)
            return torch.where(total_n == 0, total_loss, total_loss / safe)
        if total_n == 0:
            warnings.warn("Total number of loss elements is 0. Returning 0 as  "
                          "loss to avoid division by zero.")
            return total_loss
        return total_loss / total_n

    def mark_as_total(self, key: str) -> str:
        return get_total_loss_key(key)

    @abc.abstractmethod
    def training_step(self, batch: BatchType, batch_idx: int, predictions_post: BatchType
                      ) -> Tuple[Dict[str, torch.Tensor], Dict[str, Any]]:
        ...

    @abc.abstractmethod
    def validation_step(self, batch: BatchType, batch_idx: int, predictions_post: BatchType
                        ) -> Tuple[Dict[str, torch.Tensor], Dict[str, Any]]:
        ...

    @abc.abstractmethod
    def validation_epoch_end(self):
        ...
