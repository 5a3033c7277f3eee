"""`InstanceTaskHelper` (reference task_helper/instance.py:35-446): center (MSE | L1),
offset (L1) and orientation (von Mises) losses with the reference's masking conventions —
folded into the HIP loss kernels instead of `pred*mask` / boolean gathers — and the
instance-quality PQ with ground-truth semantics at validation time.  Visualisation
examples are out of scope (`_examples` stays empty)."""
from typing import Any, Dict, Tuple

import numpy as np
import torch

from ..data.preprocessing.resize import get_fullres
from ..data.preprocessing.resize import get_fullres_key
from ..loss import _functional as F_
from ..loss import check_loss_status
from ..loss import CenterFocalLoss
from ..loss import L1Loss
from ..loss import MSELoss
from ..loss import VonMisesLossBiternion
from ..metric.mae import MeanAbsoluteAngularError
from ..metric.mae import PanopticQualityWithOrientationMAE
from ..types import BatchType
from ..utils.panoptic_merge import _ids_to_dicts
from ..utils.panoptic_merge import _merge_on_device
from .base import TaskHelperBase
from .base import append_detached_losses_to_logs
from .base import append_profile_to_logs

KNOWN_INSTANCE_CENTER_LOSS_FUNCTIONS = ('mse', 'l1', 'focal')    # 'focal': extension, loss/focal.py


class InstanceTaskHelper(TaskHelperBase):
    def __init__(
        self,
        semantic_n_classes: int,
        semantic_classes_is_thing: Tuple[bool],
        loss_name_instance_center: str = 'mse',
        disable_multiscale_supervision: bool = False
    ) -> None:
        super().__init__()
        self._loss_name_instance_center = loss_name_instance_center
        self._disable_multiscale_supervision = disable_multiscale_supervision
        self._semantic_n_classes = semantic_n_classes
        self._semantic_classes_is_thing = semantic_classes_is_thing
        self._with_orientation = False           # detected on the fly
        self._examples: Dict[str, Any] = {}
        self._max_instances_per_category = 1 << 16     # hypersim: > 256 instances / image
        self._thing_ids = np.where(self._semantic_classes_is_thing)[0]

    def initialize(self, device: torch.device):
        assert self._loss_name_instance_center in KNOWN_INSTANCE_CENTER_LOSS_FUNCTIONS
        self._loss_center = {'mse': MSELoss, 'l1': L1Loss, 'focal': CenterFocalLoss}[
            self._loss_name_instance_center](reduction='sum')
        self._loss_offset = L1Loss(reduction='sum')
        self._loss_orientation = VonMisesLossBiternion()
        self._mae_pq_deeplab = PanopticQualityWithOrientationMAE(
            num_categories=self._semantic_n_classes, ignored_label=0,
            max_instances_per_category=self._max_instances_per_category,
            offset=256 ** 3, is_thing=self._semantic_classes_is_thing, device=device)
        self._mae_gt = MeanAbsoluteAngularError(device=device)

    def _compute_losses(self, batch, batch_idx, predictions_post) -> Dict[str, torch.Tensor]:
        no_multiscale = self._disable_multiscale_supervision
        preds, keys, downscales = self.collect_predictions_for_loss(
            predictions_post=predictions_post, predictions_post_key='instance_output',
            side_outputs_key=None if no_multiscale else 'instance_side_outputs')
        preds_center = [p[0][:, 0] for p in preds]               # drop the channel axis
        preds_offset = [p[1] for p in preds]
        preds_orientation = [p[2] for p in preds if len(p) == 3]
        self._with_orientation = len(preds_orientation) > 0

        def targets(key):
            return self.collect_targets_for_loss(batch, batch_key=key, downscales=downscales)

        def expected(name, preds_, masks, clamp=False):
            """per scale: the gradient the total over the scales sends back (None: not asked)"""
            if not (torch.is_grad_enabled() and F_.speculation_enabled()) or \
                    not any(p.requires_grad for p in preds_):
                return [None] * len(preds_), masks
            masks = [F_._u8(m.to(p.device)) for m, p in zip(masks, preds_)]
            counts = [F_.count_u8(m) for m in masks]
            if clamp:
                counts = [c.clamp(min=1) for c in counts]
            return [self.expected_scale_for_total(counts, preds_, name)] * len(preds_), masks

        # center: pred*mask vs target, n = sum(mask)            (instance.py:115-139)
        center_focal = self._loss_name_instance_center == 'focal'     # n = #positives: no count
        exp, masks = ([None] * len(preds_center), targets('instance_center_mask')) if center_focal \
            else expected('instance_center', preds_center, targets('instance_center_mask'))
        out_center = [self._loss_center.masked_sum(p.contiguous(), t, m, expected_scale=e)
                      for p, t, m, e in zip(preds_center, targets('instance_center'), masks, exp)]
        # offset: pred*foreground vs target, n = sum(foreground)  (instance.py:141-167)
        exp, masks = expected('instance_offset', preds_offset, targets('instance_foreground'))
        out_offset = [self._loss_offset.masked_sum(p.contiguous(), t, m, expected_scale=e)
                      for p, t, m, e in zip(preds_offset, targets('instance_offset'), masks, exp)]
        loss_dict = {}
        loss_dict.update({f'instance_center_loss_{k}': l / n
                          for k, (l, n) in zip(keys, out_center)})
        loss_dict.update({f'instance_offset_loss_{k}': l / n
                          for k, (l, n) in zip(keys, out_offset)})
        total = {
            'instance_center': self.accumulate_losses([l for l, _ in out_center],
                                                      [n for _, n in out_center]),
            'instance_offset': self.accumulate_losses([l for l, _ in out_offset],
                                                      [n for _, n in out_offset]),
        }
        if self._with_orientation:
            # masked rows, n = max(sum(mask), 1)                 (instance.py:170-216)
            out_ori = []
            exp, masks = expected('instance_orientation', preds_orientation,
                                  targets('orientation_foreground'), clamp=True)
            for p, t, m, e in zip(preds_orientation, targets('orientation'), masks, exp):
                l, n = self._loss_orientation.masked_sum(p.contiguous(), t, m, expected_scale=e)
                out_ori.append((l, n.clamp(min=1)))
            loss_dict.update({f'instance_orientation_loss_{k}': l / n
                              for k, (l, n) in zip(keys, out_ori)})
            total['instance_orientation'] = self.accumulate_losses(
                [l for l, _ in out_ori], [n for _, n in out_ori])
        for k, v in total.items():
            loss_dict[self.mark_as_total(k)] = v
        return loss_dict

    @append_profile_to_logs('instance_step_time')
    @append_detached_losses_to_logs()
    def training_step(self, batch, batch_idx, predictions_post):
        return self._compute_losses(batch, batch_idx, predictions_post), {}

    @append_profile_to_logs('instance_step_time')
    @append_detached_losses_to_logs()
    def validation_step(self, batch, batch_idx, predictions_post):
        loss_dict = self._compute_losses(batch, batch_idx, predictions_post)
        if self._with_orientation:
            orientations_results = \
                predictions_post['orientations_instance_segmentation_gt_orientation_foreground']
            orientations_full_gt = \
                predictions_post['orientations_gt_instance_gt_orientation_foreground']
            orientations_targets = batch['orientations_present']
            self._mae_gt.update(orientations_full_gt, orientations_targets)
        else:
            orientations_results = None
            orientations_targets = None

        # instance quality with GT semantics and GT foreground (instance.py:313-357)
        dev = self._mae_pq_deeplab.device
        semantic_batch = get_fullres(batch, 'semantic').to(dev)
        instance_batch = get_fullres(batch, 'instance').to(dev)
        instance_result = predictions_post[
            get_fullres_key('instance_segmentation_gt_foreground')].to(dev)
        instance_foreground = instance_batch != 0
        panoptic_targets = get_fullres(batch, 'panoptic').to(dev)
        panoptic_targets_id_dicts = batch['panoptic_ids_to_instance_dict']
        # same kernels as deeplab_merge_batch; the {panoptic id: instance id} dicts (one
        # device->host copy) are only built when the orientation matching walks them
        merged = _merge_on_device(semantic_batch, instance_result, instance_foreground,
                                  self._max_instances_per_category, self._thing_ids, 0,
                                  n_classes=self._semantic_n_classes)
        panoptic_preds = merged['panoptic']
        panoptic_id_dicts = _ids_to_dicts(merged['ids_pan'], merged['ids_ins'], merged['n_ids']) \
            if self._with_orientation else None
        self._mae_pq_deeplab.update(panoptic_preds, orientations_results, panoptic_id_dicts,
                                    panoptic_targets, orientations_targets,
                                    panoptic_targets_id_dicts)
        return loss_dict, {}

    @append_profile_to_logs('instance_epoch_end_time')
    def validation_epoch_end(self):
        check_loss_status()        # out-of-range labels seen by the loss kernels (one host sync)
        artifacts, logs = {}, {}
        for key, value in self._mae_pq_deeplab.compute(suffix="_deeplab").items():
            (logs if value.numel() == 1 else artifacts)[f'instance_{key}'] = value
        self._mae_pq_deeplab.reset()
        if self._with_orientation:
            mae_rad, mae_deg = self._mae_gt.compute()
            logs['orientation_mae_gt_rad'] = mae_rad
            logs['orientation_mae_gt_deg'] = mae_deg
            self._mae_gt.reset()
        return artifacts, self._examples, logs
