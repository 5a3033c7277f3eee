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
        # the center map keeps its channel axis of size 1 (the kernels take [B,1,H,W] as C = 1): a
        # `[:, 0]` here costs autograd a zero fill and a copy of every center gradient in backward
        # (select_backward: 43 us per training step at B = 64, three scales)
        preds_center = [p[0] if p[0].ndim == 4 and p[0].shape[1] == 1 else p[0][:, 0] for p in preds]
        preds_offset = [p[1] for p in preds]
        preds_orientation = [p[2] for p in preds if len(p) == 3]
        self._with_orientation = len(preds_orientation) > 0

        def targets(key):
            return self.collect_targets_for_loss(batch, batch_key=key, downscales=downscales)

        center_kind = self._loss_name_instance_center
        items, names = [], []
        # center: pred*mask vs target, n = sum(mask)            (instance.py:115-139)
        for k, p, t, m in zip(keys, preds_center, targets('instance_center'), targets('instance_center_mask')):
            items.append({'kind': center_kind, 'pred': p.contiguous(), 'target': t, 'mask': m, 'total': 0,
                          'clamp': center_kind == 'focal'})      # focal: n = max(#peaks, 1) per scale
            names.append(f'instance_center_loss_{k}')
        # offset: pred*foreground vs target, n = sum(foreground)  (instance.py:141-167)
        for k, p, t, m in zip(keys, preds_offset, targets('instance_offset'), targets('instance_foreground')):
            items.append({'kind': 'l1', 'pred': p.contiguous(), 'target': t, 'mask': m, 'total': 1})
            names.append(f'instance_offset_loss_{k}')
        total_names = ['instance_center', 'instance_offset']
        if self._with_orientation:
            # masked rows, n = max(sum(mask), 1)                 (instance.py:170-216)
            for k, p, t, m in zip(keys, preds_orientation, targets('orientation'),
                                  targets('orientation_foreground')):
                items.append({'kind': 'vonmises', 'pred': p.contiguous(), 'target': t, 'mask': m,
                              'param': self._loss_orientation._kappa, 'total': 2, 'clamp': True})
                names.append(f'instance_orientation_loss_{k}')
            total_names.append('instance_orientation')
        from ..loss import _multi
        if F_.speculation_enabled() and _multi.supported(items):
            # every loss of every scale in ONE forward call
            return self.multi_losses(items, names, tuple(total_names))
        # host tensors, > 16 items: loss by loss
        out = []
        for it in items:
            if it['kind'] == 'vonmises':
                l, n = self._loss_orientation.masked_sum(it['pred'], it['target'], it['mask'])
                n = n.clamp(min=1)
            elif it['total'] == 0:
                l, n = self._loss_center.masked_sum(it['pred'], it['target'], it['mask'])
            else:
                l, n = self._loss_offset.masked_sum(it['pred'], it['target'], it['mask'])
            out.append((l, n))
        loss_dict = {name: l / n for name, (l, n) in zip(names, out)}
        for ti, k in enumerate(total_names):
            sel = [o for it, o in zip(items, out) if it['total'] == ti]
            loss_dict[self.mark_as_total(k)] = self.accumulate_losses([l for l, _ in sel], [n for _, n in sel])
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
