"""`MeanIntersectionOverUnion` with a device-resident confusion matrix
(reference metric/miou.py:9-94).  `update` is the HIP kernel k_confmat
(csrc/metrics.hip); `compute` is O(n^2) tensor math like the reference's."""
from typing import Optional

import torch

from .. import _lib as L
from .base import Metric


def confmat_update(confmat: torch.Tensor, status: torch.Tensor, preds: torch.Tensor,
                   target: torch.Tensor, n_classes: int, pred_div: int = 1,
                   mode: int = 0) -> None:
    """confmat[t, p] += 1 on the device (see include/nmsa.h nmsa_confmat_update)."""
    dev = confmat.device
    p = preds.to(dev).contiguous()
    t = target.to(dev).contiguous()
    if p.dtype == torch.bool:
        p = p.view(torch.uint8)
    if t.dtype == torch.bool:
        t = t.view(torch.uint8)
    if p.numel() != t.numel():
        raise ValueError('preds and target must have the same number of elements')
    ws_bytes = L.lib().nmsa_confmat_workspace_bytes(int(n_classes))
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    L.check(L.lib().nmsa_confmat_update(
        L.ptr(p), L.int_dtype_code(p), int(pred_div), L.ptr(t), L.int_dtype_code(t),
        p.numel(), int(n_classes), int(mode), L.ptr(confmat), L.ptr(status),
        L.ptr(ws), ws_bytes, L.stream_ptr(dev)), 'nmsa_confmat_update')


class MeanIntersectionOverUnion(Metric):
    def __init__(self, n_classes: int, ignore_first_class: bool = False,
                 device: Optional[torch.device] = None, **kwargs) -> None:
        super().__init__(device=device, **kwargs)
        self.add_state('confmat', torch.zeros((n_classes, n_classes), dtype=torch.int64),
                       dist_reduce_fx='sum')
        self._n_classes = n_classes
        self._ignore_first_class = ignore_first_class
        self._status = torch.zeros((1,), dtype=torch.int32, device=self.device)

    def to(self, device, *args, **kwargs):
        super().to(device)
        self._status = self._status.to(self.device)
        return self

    def _require_gpu(self):
        self._require_unsynced()                # (also reached through PanopticQuality.update_with_miou)
        if self.device.type != 'cuda':
            raise L.NmsaError('MeanIntersectionOverUnion.update needs the MI355X '
                              '(states live on the GPU; no CPU fallback)')

    def update(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        self._require_gpu()
        confmat_update(self.confmat, self._status, preds, target, self._n_classes)

    def update_from_panoptic(self, panoptic_preds: torch.Tensor, target: torch.Tensor,
                             max_instances_per_category: int) -> None:
        """`update(panoptic // max_instances, target)` of task_helper/panoptic.py:123-126
        without materialising the divided map."""
        self._require_gpu()
        confmat_update(self.confmat, self._status, panoptic_preds, target, self._n_classes,
                       pred_div=max_instances_per_category)

    def update_masked_void(self, preds: torch.Tensor, target: torch.Tensor) -> None:
        """`update(preds[target != 0], target[target != 0] - 1)` of
        task_helper/semantic.py:124-128 without the boolean gathers."""
        self._require_gpu()
        confmat_update(self.confmat, self._status, preds, target, self._n_classes, mode=1)

    def _check_status(self):
        st = int(self._status.item())
        if st:
            self._status.zero_()
            raise ValueError('MeanIntersectionOverUnion: label outside [0, n_classes) '
                             '(the reference raises in bincount/reshape)')

    def compute(self, return_ious: bool = False):
        self._check_status()
        cm = self.confmat
        tp = torch.diag(cm).float()
        sum_pred = cm.sum(dim=0).float()
        sum_gt = cm.sum(dim=1).float()
        if self._ignore_first_class:                       # void is row/col 0
            tp, sum_pred, sum_gt = tp[1:], sum_pred[1:], sum_gt[1:]
            sum_pred = sum_pred - cm[0, 1:].float()
        has_gt = sum_gt != 0                               # classes without GT do not count
        iou = tp[has_gt] / (sum_pred[has_gt] + sum_gt[has_gt] - tp[has_gt])
        miou = iou.mean()
        if not return_ious:
            return miou
        ious = torch.full((self._n_classes,), float('nan'), dtype=torch.float32,
                          device=iou.device)
        idx = has_gt.nonzero(as_tuple=True)[0] + (1 if self._ignore_first_class else 0)
        ious[idx] = iou
        return miou, ious

    def reset(self) -> None:
        super().reset()
        if hasattr(self, '_status'):
            self._status = torch.zeros((1,), dtype=torch.int32, device=self.device)
