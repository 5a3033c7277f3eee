"""`DenseVisualEmbeddingTaskHelper` (reference task_helper/dense_visual_embedding.py:33-345):
cosine-embedding (or MSE / L1) loss between the predicted embedding map and the per-image
LUT of target embeddings, addressed by an index map (0 = no target).  The index masking,
the per-image LUT gather and the NCHW->rows permute of the reference are folded into the
HIP kernel k_cos_emb.  Text / visual-mean mIoUs at validation time; examples out of scope."""
from typing import Any, Dict, List, Sequence, Tuple, Union

import numpy as np
import torch

from ..loss import _functional as F_

from ..data.preprocessing.multiscale_supervision import get_downscale
from ..data.preprocessing.resize import get_fullres
from ..data.preprocessing.resize import get_fullres_key
from ..loss import check_loss_status
from ..loss import CosineEmbeddingLoss
from ..loss import L1Loss
from ..loss import MSELoss
from ..metric import MeanIntersectionOverUnion
from ..types import BatchType
from .base import TaskHelperBase
from .base import append_detached_losses_to_logs
from .base import append_profile_to_logs

KNOWN_DENSE_VISUAL_EMBEDDING_LOSS_FUNCTIONS = ('mse', 'l1', 'cos_emb')


def _stack_luts(luts: Sequence[torch.Tensor], device) -> torch.Tensor:
    """list of [n_b, D] (not stackable in the reference) -> zero-padded [B, L, D]"""
    L_ = max(int(l.shape[0]) for l in luts)
    L_ = max(L_, 1)
    D = int(luts[0].shape[1])
    out = torch.zeros((len(luts), L_, D), dtype=torch.float32, device=device)
    for b, l in enumerate(luts):
        if l.shape[0]:
            out[b, :l.shape[0]] = l.to(device, torch.float32)
    return out


class DenseVisualEmbeddingTaskHelper(TaskHelperBase):
    def __init__(
        self,
        n_classes: int,
        loss_name: str = 'cos_emb',
        disable_multiscale_supervision: bool = False,
        examples_cmap: Union[Sequence[Tuple[int, int, int]], np.ndarray, None] = None
    ) -> None:
        super().__init__()
        self._loss_name = loss_name.lower()
        self._disable_multiscale_supervision = disable_multiscale_supervision
        self._examples: Dict[str, Any] = {}
        self._n_classes = n_classes
        self._examples_cmap = examples_cmap

    def initialize(self, device: torch.device):
        assert self._loss_name in KNOWN_DENSE_VISUAL_EMBEDDING_LOSS_FUNCTIONS
        self._loss = {'mse': MSELoss, 'l1': L1Loss, 'cos_emb': CosineEmbeddingLoss}[self._loss_name]()
        self._text_metric_iou = MeanIntersectionOverUnion(n_classes=self._n_classes, device=device)
        self._visual_mean_metric_iou = MeanIntersectionOverUnion(n_classes=self._n_classes,
                                                                 device=device)

    def _get_spatial_target_for_prediction(self, batch, batch_key, prediction):
        target = batch[batch_key]
        h_t, w_t = target.shape[-2:]
        h_p, w_p = prediction.shape[-2:]
        if (h_p, w_p) == (h_t, w_t):
            return target
        assert h_t % h_p == 0 and w_t % w_p == 0, (
            f"Prediction and target resolutions are incompatible: {(h_p, w_p)} vs {(h_t, w_t)}")
        assert h_t // h_p == w_t // w_p, "Non-uniform scaling is not supported"
        sub = get_downscale(batch, h_t // h_p)
        assert sub is not None and batch_key in sub, (
            f"Required downscale '{h_t // h_p}' for key '{batch_key}' is missing in batch.")
        return sub[batch_key]

    def _compute_losses(self, batch, batch_idx, predictions_post) -> Dict[str, torch.Tensor]:
        no_multiscale = self._disable_multiscale_supervision
        preds, keys, downscales = self.collect_predictions_for_loss(
            predictions_post=predictions_post,
            predictions_post_key='dense_visual_embedding_output',
            side_outputs_key=None if no_multiscale else 'dense_visual_embedding_side_outputs')
        luts = self.collect_targets_for_loss(batch=batch, batch_key='dense_visual_embedding_lut',
                                             downscales=downscales)
        outs = []
        if self._loss_name == 'cos_emb' and F_.speculation_enabled() and F_.wants_gradient(preds[0]):
            # every scale in ONE call: forward + gradient in one pass over each prediction, the
            # gradient written for the learned upstream factor of the total (loss/_multi.py)
            from ..loss import _multi
            items = []
            for pred, lut in zip(preds, luts):
                indices = self._get_spatial_target_for_prediction(
                    batch, 'dense_visual_embedding_indices', pred)
                lut_t = lut if isinstance(lut, torch.Tensor) and lut.ndim == 3 \
                    else _stack_luts(lut, pred.device)
                # per scale loss / max(n, 1); the total divides by max(sum of n, 1)
                items.append({'kind': 'cos', 'pred': pred.contiguous(), 'target': lut_t, 'mask': indices,
                              'total': 0, 'clamp': 2})
            if _multi.supported(items):
                return self.multi_losses(items, [f'dense_visual_embedding_loss_{k}' for k in keys],
                                         ('dense_visual_embedding',))
        for pred, lut in zip(preds, luts):
            indices = self._get_spatial_target_for_prediction(
                batch, 'dense_visual_embedding_indices', pred)
            lut_t = lut if isinstance(lut, torch.Tensor) and lut.ndim == 3 \
                else _stack_luts(lut, pred.device)
            if self._loss_name == 'cos_emb':
                outs.append(self._loss.lut_sum(pred.contiguous(), indices, lut_t))
            else:
                # mse / l1 on the gathered rows (reference :110-175); not the headline path
                valid = indices != 0
                rows = pred.permute(0, 2, 3, 1)[valid]
                b_idx = torch.where(valid)[0]
                tgt = lut_t[b_idx, (indices[valid] - 1).long()]
                outs.extend(self._loss([rows], [tgt]))
        loss_dict = {f'dense_visual_embedding_loss_{k}': l / (n.clamp(min=1) if
                     isinstance(n, torch.Tensor) else max(n, 1)) for k, (l, n) in zip(keys, outs)}
        loss_dict[self.mark_as_total('dense_visual_embedding')] = self.accumulate_losses(
            [l for l, _ in outs], [n for _, n in outs])
        return loss_dict

    @append_profile_to_logs('semantic_embedding_step_time')
    @append_detached_losses_to_logs()
    def training_step(self, batch, batch_idx, predictions_post):
        return self._compute_losses(batch, batch_idx, predictions_post), {}

    @append_profile_to_logs('dense_visual_embedding_step_time')
    @append_detached_losses_to_logs()
    def validation_step(self, batch, batch_idx, predictions_post):
        loss_dict = self._compute_losses(batch, batch_idx, predictions_post)
        target = get_fullres(batch, 'semantic')
        for key, metric in (('dense_visual_embedding_text_based_semantic_idx', self._text_metric_iou),
                            ('dense_visual_embedding_visual_mean_based_semantic_idx',
                             self._visual_mean_metric_iou)):
            k = get_fullres_key(key)
            if k in predictions_post:
                metric.update_masked_void(predictions_post[k], target)
        return loss_dict, {}

    @append_profile_to_logs('semantic_epoch_end_time')
    def validation_epoch_end(self):
        check_loss_status()        # out-of-range labels seen by the loss kernels (one host sync)
        miou, ious = self._text_metric_iou.compute(return_ious=True)
        vmiou, vious = self._visual_mean_metric_iou.compute(return_ious=True)
        logs = {'dense_visual_embedding_text_based_miou': miou,
                'dense_visual_embedding_visual_mean_based_miou': vmiou}
        artifacts = {
            'dense_visual_embedding_text_based_semantic_cm': self._text_metric_iou.confmat.clone(),
            'dense_visual_embedding_text_based_semantic_ious_per_class': ious.clone(),
            'dense_visual_embedding_visual_mean_based_semantic_cm':
                self._visual_mean_metric_iou.confmat.clone(),
            'dense_visual_embedding_visual_mean_based_semantic_ious_per_class': vious.clone(),
        }
        self._text_metric_iou.reset()
        self._visual_mean_metric_iou.reset()
        return artifacts, self._examples, logs
