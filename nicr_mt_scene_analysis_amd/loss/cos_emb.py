"""`CosineEmbeddingLoss` (reference loss/cos_emb.py:13-56) on the HIP kernel k_cos_emb."""
from typing import Optional, Tuple

import torch

from .. import _lib as L
from . import _functional as F_
from .base import LossBase


class CosineEmbeddingLoss(LossBase):
    def __init__(self, reduction: str = 'sum') -> None:
        super().__init__()
        assert reduction in ('sum', 'mean', 'none')
        self._reduction = reduction

    def lut_sum(self, input_: torch.Tensor, indices: torch.Tensor, lut: torch.Tensor
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        """planar [B,D,H,W] prediction, int indices [B,H,W] (0 = no target) and the
        per-image LUT [B,L,D]: the gathers of task_helper/dense_visual_embedding.py:110-171
        folded into the kernel.  -> (sum, number of valid px)"""
        from . import _multi
        if self._can_speculate(input_) and _multi.cos_supported(input_, lut):
            # forward + gradient in ONE pass over the prediction (csrc/losses_cos.hip), the gradient
            # written for this instance's learned expectation of the upstream factor and
            # confirmed or recomputed in backward
            loss, n, _ = self._speculative_single('cos', input_, lut, indices)
            return loss, n
        return F_.cosine_embedding_lut_sum(input_, indices, lut)

    def _compute_loss(self, input_: torch.Tensor, target: torch.Tensor,
                      target_similarity: Optional[torch.Tensor] = None, expected_scale=None):
        # expected_scale: accepted for LossBase.forward, unused (the embedding column does not
        # fit the registers; the backward kernel reads the saved dot products instead)
        L.require_device_tensor(input_, 'input_')        # no CPU path: raises for host tensors
        n, d = input_.shape
        # similar pairs (label +1) with 'sum' / 'mean' — what the task helper uses — run the LUT
        # kernel below.  Explicit labels (dissimilar pairs) and per-row losses ('none') run the
        # row kernel k_cos_rows (csrc/losses_forms.hip).  Only an empty input or a target that
        # itself asks for a gradient is left to ATen's op ON THE DEVICE.
        labelled = target_similarity is not None
        if n == 0 or target.requires_grad:
            labels = target_similarity if labelled else torch.ones(n, device=input_.device)
            loss = torch.nn.functional.cosine_embedding_loss(
                input_, target.to(input_.device), labels.to(input_.device), reduction='none')
            if self._reduction == 'sum':
                return loss.sum(), loss.numel()
            if self._reduction == 'mean':
                return loss.mean(), 1
            return loss, input_.numel()
        if labelled or self._reduction == 'none':
            rows = F_.cosine_embedding_rows(input_, target, target_similarity)
            out_dtype = torch.result_type(input_, target)
            if rows.dtype != out_dtype and out_dtype in (torch.bfloat16, torch.float16):
                rows = rows.to(out_dtype)
            if self._reduction == 'sum':
                return rows.sum(), rows.numel()
            if self._reduction == 'mean':
                return rows.mean(), 1
            return rows, input_.numel()
        # rows (n, d): prediction planar (1, d, n, 1); the targets are their own LUT
        x = input_.t().contiguous().view(1, d, n, 1)
        idx = torch.arange(1, n + 1, dtype=torch.int32, device=input_.device).view(1, n, 1)
        loss, _ = F_.cosine_embedding_lut_sum(x, idx, target.view(1, n, d))
        if self._reduction == 'mean':
            return loss / n, 1
        return loss, n
