"""`PanopticTaskHelper` (reference task_helper/panoptic.py:28-212): PQ (+ orientation MAE)
and the post-merge mIoU, accumulated on the GPU from the device-resident predictions
(no `.cpu()` of the panoptic maps).  Visualisation examples are out of scope."""
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from ..data.preprocessing.resize import get_fullres
from ..data.preprocessing.resize import get_fullres_key
from ..metric import MeanIntersectionOverUnion
from ..metric.mae import PanopticQualityWithOrientationMAE
from ..types import BatchType
from .base import TaskHelperBase
from .base import append_profile_to_logs


class PanopticTaskHelper(TaskHelperBase):
    def __init__(
        self,
        semantic_n_classes: int,                 # with void!
        semantic_classes_is_thing: Tuple[bool],
        semantic_label_list: Optional[Any] = None    # colours only (visualisation)
    ) -> None:
        super().__init__()
        self._semantic_n_classes = semantic_n_classes
        self._semantic_classes_is_thing = semantic_classes_is_thing
        self._semantic_label_list = semantic_label_list
        self._max_instances_per_category = 1 << 16
        self._thing_ids = np.where(self._semantic_classes_is_thing)[0]
        self._with_orientation = False
        self._examples: Dict[str, Any] = {}

    def initialize(self, device: torch.device):
        self._mae_pq_deeplab = PanopticQualityWithOrientationMAE(
            num_categories=self._semantic_n_classes, ignored_label=0,
            max_instances_per_category=self._max_instances_per_category,
            offset=256 ** 3, is_thing=self._semantic_classes_is_thing, device=device)
        self._metric_iou = MeanIntersectionOverUnion(
            n_classes=self._semantic_n_classes,      # with void!
            ignore_first_class=True, device=device)
        self._metric_iou.reset()

    @append_profile_to_logs('panoptic_step_time')
    def training_step(self, batch, batch_idx, predictions_post):
        # nothing to train: the helper only merges and evaluates
        return {}, {}

    @append_profile_to_logs('panoptic_step_time')
    def validation_step(self, batch, batch_idx, predictions_post):
        self._with_orientation = 'orientations_present' in batch
        if self._with_orientation:
            orientations_results = \
                predictions_post['orientations_panoptic_segmentation_deeplab_instance']
            orientations_targets = batch['orientations_present']
            # only the orientation matching walks the id dicts (lazy entry: built when read)
            pred_id_dicts = predictions_post['panoptic_segmentation_deeplab_ids']
        else:
            orientations_results = None
            orientations_targets = None
            pred_id_dicts = None

        panoptic_targets = get_fullres(batch, 'panoptic')
        panoptic_preds = predictions_post[get_fullres_key('panoptic_segmentation_deeplab')]
        self._mae_pq_deeplab.update(
            panoptic_preds=panoptic_preds,
            orientation_preds=orientations_results,
            panoptic_preds_id_dicts=pred_id_dicts,
            panoptic_target=panoptic_targets,
            orientation_target=orientations_targets,
            panoptic_target_id_dicts=batch.get('panoptic_ids_to_instance_dict'),
            # merging may change classes: mIoU of panoptic // max_instances (panoptic.py:120-126)
            # rides on the PQ pass over the prediction (one read of the i64 map for both metrics)
            miou=self._metric_iou, semantic_target=get_fullres(batch, 'semantic'),
            pred_div=self._max_instances_per_category,
            # (used only when `panoptic_preds` is the very map these parts were painted into:
            # network resolution = dataset resolution)
            panoptic_pred_parts=getattr(predictions_post, 'aux', {}).get('panoptic_parts'))
        return {}, {}

    @append_profile_to_logs('panoptic_epoch_end_time')
    def validation_epoch_end(self):
        artifacts, logs = {}, {}
        for key, value in self._mae_pq_deeplab.compute(suffix="_deeplab").items():
            (logs if value.numel() == 1 else artifacts)[f'panoptic_{key}'] = value
        self._mae_pq_deeplab.reset()
        artifacts['panoptic_deeplab_semantic_cm'] = self._metric_iou.confmat.clone()
        miou, ious = self._metric_iou.compute(return_ious=True)
        logs['panoptic_deeplab_semantic_miou'] = miou
        artifacts['panoptic_deeplab_semantic_ious_per_class'] = ious
        self._metric_iou.reset()
        return artifacts, self._examples, logs
