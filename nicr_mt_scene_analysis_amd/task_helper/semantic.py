"""`SemanticTaskHelper` (reference task_helper/semantic.py:22-165): weighted CE over the main
and side outputs, masked mIoU on full resolution.  Visualisation examples are out of scope
(`_examples` stays empty)."""
from typing import Any, Dict, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ..data.preprocessing.resize import get_fullres
from ..data.preprocessing.resize import get_fullres_key
from ..loss import CrossEntropyLossSemantic
from ..metric import MeanIntersectionOverUnion
from ..types import BatchType
from .base import TaskHelperBase
from .base import append_detached_losses_to_logs
from .base import append_profile_to_logs


class SemanticTaskHelper(TaskHelperBase):
    def __init__(
        self,
        n_classes: int,
        class_weights: Optional[np.ndarray] = None,
        label_smoothing: float = 0.0,
        disable_multiscale_supervision: bool = False,
        examples_cmap: Union[Sequence[Tuple[int, int, int]], np.ndarray, None] = None
    ) -> None:
        super().__init__()
        self._n_classes = n_classes
        self._class_weights = class_weights
        self._label_smoothing = label_smoothing
        self._disable_multiscale_supervision = disable_multiscale_supervision
        self._examples: Dict[str, Any] = {}
        self._examples_cmap = examples_cmap

    def initialize(self, device: torch.device):
        if self._class_weights is not None:
            self._class_weights = torch.as_tensor(self._class_weights, dtype=torch.float,
                                                  device=device)
        self._loss = CrossEntropyLossSemantic(weights=self._class_weights,
                                              label_smoothing=self._label_smoothing)
        # the reference keeps this metric on the CPU "because it is faster"; here the
        # confusion matrix lives on the GPU next to the predictions
        self._metric_iou = MeanIntersectionOverUnion(n_classes=self._n_classes, device=device)
        self._metric_iou.reset()

    def _compute_losses(self, batch, batch_idx, predictions_post) -> Dict[str, torch.Tensor]:
        no_multiscale = self._disable_multiscale_supervision
        preds, targets, keys = self.collect_predictions_and_targets_for_loss(
            batch=batch, batch_key='semantic', predictions_post=predictions_post,
            predictions_post_key='semantic_output',
            side_outputs_key=None if no_multiscale else 'semantic_side_outputs')
        outs = self._loss(input_tensors=preds, target_tensors=targets)
        loss_dict = {f'semantic_loss_{key}': loss / n for key, (loss, n) in zip(keys, outs)}
        loss_dict[self.mark_as_total('semantic')] = self.accumulate_losses(
            losses=[loss for loss, _ in outs], n_elements=[n for _, n in outs])
        return loss_dict

    @append_profile_to_logs('semantic_step_time')
    @append_detached_losses_to_logs()
    def training_step(self, batch, batch_idx, predictions_post):
        return self._compute_losses(batch, batch_idx, predictions_post), {}

    @append_profile_to_logs('semantic_step_time')
    @append_detached_losses_to_logs()
    def validation_step(self, batch, batch_idx, predictions_post):
        loss_dict = self._compute_losses(batch, batch_idx, predictions_post)
        # preds[target != 0] vs target[target != 0] - 1 (semantic.py:124-128), fused
        target = get_fullres(batch, 'semantic')
        preds = predictions_post[get_fullres_key('semantic_segmentation_idx')]
        self._metric_iou.update_masked_void(preds, target)
        return loss_dict, {}

    @append_profile_to_logs('semantic_epoch_end_time')
    def validation_epoch_end(self):
        miou, ious = self._metric_iou.compute(return_ious=True)
        logs = {'semantic_miou': miou}
        artifacts = {'semantic_cm': self._metric_iou.confmat.clone(),
                     'semantic_ious_per_class': ious.clone()}
        self._metric_iou.reset()
        return artifacts, self._examples, logs
