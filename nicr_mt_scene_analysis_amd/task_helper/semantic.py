"""Semantic segmentation task (interface of reference task_helper/semantic.py:22-165).

Training: (weighted, label-smoothed) cross entropy over the main output and the side outputs,
normalised by the total number of non-void pixels.  Validation additionally feeds the
full-resolution argmax into the mIoU confusion matrix with void pixels masked out — on the GPU
next to the predictions (the reference keeps that metric on the CPU).  Visualisation examples
are out of scope: the example dict stays empty.
"""
import torch

from ..data.preprocessing.resize import get_fullres
from ..data.preprocessing.resize import get_fullres_key
from ..loss import _functional as F_
from ..loss import check_loss_status
from ..loss import CrossEntropyLossSemantic
from ..metric import MeanIntersectionOverUnion
from .base import TaskHelperBase
from .base import append_detached_losses_to_logs
from .base import append_profile_to_logs

_TASK = 'semantic'


class SemanticTaskHelper(TaskHelperBase):
    def __init__(self, n_classes, class_weights=None, label_smoothing=0.0,
                 disable_multiscale_supervision=False, examples_cmap=None):
        super().__init__()
        self._n_classes = n_classes
        self._class_weights = class_weights
        self._label_smoothing = label_smoothing
        self._disable_multiscale_supervision = disable_multiscale_supervision
        self._examples_cmap = examples_cmap
        self._examples = {}

    def initialize(self, device):
        weights = self._class_weights
        if weights is not None:
            weights = torch.as_tensor(weights, dtype=torch.float, device=device)
            self._class_weights = weights
        self._loss = CrossEntropyLossSemantic(weights=weights,
                                              label_smoothing=self._label_smoothing)
        self._metric_iou = MeanIntersectionOverUnion(n_classes=self._n_classes, device=device)
        self._metric_iou.reset()

    # ---- losses ------------------------------------------------------------------------------
    def _compute_losses(self, batch, batch_idx, predictions_post):
        side_key = None if self._disable_multiscale_supervision else f'{_TASK}_side_outputs'
        predictions, targets, scale_names = self.collect_predictions_and_targets_for_loss(
            batch=batch, batch_key=_TASK, predictions_post=predictions_post,
            predictions_post_key=f'{_TASK}_output', side_outputs_key=side_key)
        from ..loss import _multi
        items = [{'kind': 'ce', 'pred': p, 'mask': t, 'weights': self._class_weights,
                  'param': self._label_smoothing, 'total': 0} for p, t in zip(predictions, targets)]
        if F_.speculation_enabled() and _multi.supported(items) and \
                all(p.ndim == 4 for p in predictions):
            # all scales in ONE forward call (count, expectation, forward + gradient, finalize)
            return self.multi_losses(items, [f'{_TASK}_loss_{name}' for name in scale_names], (_TASK,))
        per_scale = self._loss(input_tensors=predictions, target_tensors=targets)
        sums = [loss_sum for loss_sum, _ in per_scale]
        counts = [count for _, count in per_scale]
        losses = {f'{_TASK}_loss_{name}': s / n for name, s, n in zip(scale_names, sums, counts)}
        losses[self.mark_as_total(_TASK)] = self.accumulate_losses(losses=sums, n_elements=counts)
        return losses

    @append_profile_to_logs(f'{_TASK}_step_time')
    @append_detached_losses_to_logs()
    def training_step(self, batch, batch_idx, predictions_post):
        return self._compute_losses(batch, batch_idx, predictions_post), {}

    # ---- validation --------------------------------------------------------------------------
    @append_profile_to_logs(f'{_TASK}_step_time')
    @append_detached_losses_to_logs()
    def validation_step(self, batch, batch_idx, predictions_post):
        losses = self._compute_losses(batch, batch_idx, predictions_post)
        # mIoU on preds[target != 0] vs target[target != 0] - 1 (semantic.py:124-128): the void
        # masking happens inside the confusion-matrix kernel
        key = get_fullres_key(f'{_TASK}_segmentation_idx')
        aux = getattr(predictions_post, 'aux', {})      # uint8 twin of the int64 class map
        self._metric_iou.update_masked_void(aux[key] if key in aux else predictions_post[key],
                                            get_fullres(batch, _TASK))
        return losses, {}

    @append_profile_to_logs(f'{_TASK}_epoch_end_time')
    def validation_epoch_end(self):
        check_loss_status()        # out-of-range labels seen by the loss kernels (one host sync)
        miou, ious = self._metric_iou.compute(return_ious=True)
        artifacts = {f'{_TASK}_cm': self._metric_iou.confmat.clone(),
                     f'{_TASK}_ious_per_class': ious.clone()}
        self._metric_iou.reset()
        return artifacts, self._examples, {f'{_TASK}_miou': miou}
