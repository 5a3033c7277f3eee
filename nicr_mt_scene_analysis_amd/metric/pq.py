"""`PanopticQuality` with device-resident accumulators
(reference metric/pq.py:190-361; per-image matching pq.py:60-179).

`update` runs k_pq_init / k_pq_count / k_pq_match / k_pq_accumulate
(csrc/metrics.hip) through `nmsa_pq_update`; no process pool, no `.cpu()` of
the panoptic maps.  The fp64 IoU sums are accumulated in the reference's order
(ascending intersection id per image, images in batch order), so the states
are bit-identical to the reference's for the same inputs.
"""
from typing import Dict, List, Optional, Tuple, Union

import torch

from .. import _lib as L
from .base import Metric

_EPSILON = 1e-10
_STATUS_MESSAGES = {
    1: 'more distinct segments / intersections per image than the device tables hold '
       '(2048 ids per side; intersections: H*W/48 rounded up to a power of two, at least 2048)',
    2: 'segment category outside [0, num_categories) (the reference raises IndexError)',
    4: 'inconsistent segment ids: intersection id decodes to an unknown segment '
       '(offset too small? the reference raises KeyError)',
    16: 'segment id equal to INT64_MIN is reserved',
}


def realdiv_maybe_zero(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    return torch.where(torch.abs(y) < _EPSILON, torch.zeros_like(x), x / y)


class PanopticQuality(Metric):
    def __init__(
        self,
        num_categories: int,
        ignored_label: int,
        max_instances_per_category: int,
        offset: int,
        is_thing: Union[torch.Tensor, List[bool]],
        num_workers=None,        # reference: size of the spawn pool; unused on the GPU
        device: Optional[torch.device] = None,
        **kwargs,                # Metric: sync_on_compute, process_group
    ) -> None:
        super().__init__(device=device, **kwargs)
        self.num_categories = num_categories
        self.ignored_label = ignored_label
        self.max_instances_per_category = max_instances_per_category
        self.offset = offset
        self.is_thing = torch.as_tensor(is_thing, dtype=torch.bool).clone()
        self.is_stuff = torch.logical_not(self.is_thing)
        assert len(self.is_thing) == self.num_categories
        # one void segment with instance id 0 (pq.py:220-222)
        self.void_segment_id = self.ignored_label * self.max_instances_per_category
        for name in ('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class'):
            self.add_state(name, torch.zeros(self.num_categories, dtype=torch.float64),
                           dist_reduce_fx='sum')
        self._status = torch.zeros((1,), dtype=torch.int32, device=self.device)
        self._match_capacity = 1024
        self._workspaces = {}

    def to(self, device, *args, **kwargs):
        super().to(device)
        self._status = self._status.to(self.device)
        self._workspaces = {}
        return self

    def reset(self) -> None:
        super().reset()
        if hasattr(self, '_status'):
            self._status = torch.zeros((1,), dtype=torch.int32, device=self.device)

    # ------------------------------------------------------------------ update
    def _device_update(self, preds: torch.Tensor, targets: torch.Tensor, want_matches: bool,
                       miou=None, target_semantic: Optional[torch.Tensor] = None,
                       pred_div: int = 1, parts: Optional[dict] = None
                       ) -> Optional[Tuple[torch.Tensor, torch.Tensor]]:
        if self.device.type != 'cuda':
            raise L.NmsaError('PanopticQuality.update needs the MI355X '
                              '(states live on the GPU; no CPU fallback)')
        assert preds.ndim == 3
        assert targets.shape == preds.shape
        dev = self.device
        p = None if parts is not None else preds.to(dev, dtype=torch.int64).contiguous()
        t = targets.to(dev, dtype=torch.int64).contiguous()
        B, H, W = t.shape
        lib = L.lib()
        ws_bytes = lib.nmsa_pq_workspace_bytes(B, H, W, self.num_categories)
        # persistent workspace per (batch size, stream): a completed update leaves the hash
        # tables empty, so only the first use pays for the initialisation
        ws_key = (B, H, W, torch.cuda.current_stream(dev).cuda_stream)
        ws = self._workspaces.get(ws_key)
        clean = ws is not None
        if ws is None:
            ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
            self._workspaces[ws_key] = ws
        matches = n_matches = None
        if want_matches:
            matches = torch.empty((B, self._match_capacity, 2), dtype=torch.int64, device=dev)
            n_matches = torch.empty((B,), dtype=torch.int32, device=dev)
        common = (B, H, W, self.num_categories, int(self.ignored_label),
                  int(self.max_instances_per_category), int(self.offset), int(self.void_segment_id),
                  L.ptr(self.iou_per_class), L.ptr(self.tp_per_class), L.ptr(self.fn_per_class),
                  L.ptr(self.fp_per_class), L.ptr(matches), self._match_capacity, L.ptr(n_matches),
                  L.ptr(self._status), L.ptr(ws), ws_bytes, int(clean))
        if miou is None and parts is not None:
            # PQ alone from the parts (more classes than the fused confusion matrix holds)
            th = parts['is_thing']
            L.check(lib.nmsa_pq_update_with_confmat_parts(
                L.ptr(parts['semantic_idx_u8']), L.ptr(parts['instance']), L.ptr(parts['pan_of_inst']),
                L.ptr(th), int(th.numel()), int(parts.get('void_label', 0)), L.ptr(t), None,
                B, H, W, self.num_categories, int(self.ignored_label),
                int(self.max_instances_per_category), int(self.offset), int(self.void_segment_id),
                L.ptr(self.iou_per_class), L.ptr(self.tp_per_class), L.ptr(self.fn_per_class),
                L.ptr(self.fp_per_class), L.ptr(self._status), L.ptr(ws), ws_bytes, int(clean),
                0, 1, None, None, None, 0, L.stream_ptr(dev)), 'nmsa_pq_update_with_confmat_parts')
            return None
        if miou is None:
            L.check(lib.nmsa_pq_update(L.ptr(p), L.ptr(t), *common, L.stream_ptr(dev)),
                    'nmsa_pq_update')
        else:
            ts = target_semantic.to(dev).contiguous()
            n_cm = miou._n_classes
            cm_bytes = lib.nmsa_pq_confmat_workspace_bytes(B, H, W, n_cm)
            cm_ws = torch.empty((cm_bytes,), dtype=torch.uint8, device=dev)
            if parts is not None:
                # the prediction as the parts it was painted from (2 B/px instead of 8 B/px)
                th = parts['is_thing']
                L.check(lib.nmsa_pq_update_with_confmat_parts(
                    L.ptr(parts['semantic_idx_u8']), L.ptr(parts['instance']), L.ptr(parts['pan_of_inst']),
                    L.ptr(th), int(th.numel()), int(parts.get('void_label', 0)), L.ptr(t), L.ptr(ts),
                    B, H, W, self.num_categories, int(self.ignored_label),
                    int(self.max_instances_per_category), int(self.offset), int(self.void_segment_id),
                    L.ptr(self.iou_per_class), L.ptr(self.tp_per_class), L.ptr(self.fn_per_class),
                    L.ptr(self.fp_per_class), L.ptr(self._status), L.ptr(ws), ws_bytes, int(clean),
                    n_cm, int(pred_div), L.ptr(miou.confmat), L.ptr(miou._status), L.ptr(cm_ws), cm_bytes,
                    L.stream_ptr(dev)), 'nmsa_pq_update_with_confmat_parts')
                return None
            L.check(lib.nmsa_pq_update_with_confmat(
                L.ptr(p), L.ptr(t), L.ptr(ts), *common, n_cm, int(pred_div), L.ptr(miou.confmat),
                L.ptr(miou._status), L.ptr(cm_ws), cm_bytes, L.stream_ptr(dev)),
                'nmsa_pq_update_with_confmat')
        if want_matches:
            return matches, n_matches
        return None

    def _can_fuse(self, preds: torch.Tensor, miou, target_semantic: torch.Tensor) -> bool:
        return (target_semantic.dtype == torch.uint8 and miou._n_classes <= 64
                and miou.device == self.device and preds.ndim == 3
                and target_semantic.shape == preds.shape)

    def update_with_miou(self, preds: torch.Tensor, targets: torch.Tensor, miou,
                         target_semantic: torch.Tensor, pred_div: int) -> None:
        """`self.update(preds, targets)` and `miou.update(preds // pred_div, target_semantic)`
        (the two metric updates of task_helper/panoptic.py:104-126) with ONE pass over the
        prediction.  Falls back to the two separate kernels when the fused form does not apply
        (semantic target not uint8, more than 64 classes, metrics on different devices)."""
        if not self._can_fuse(preds, miou, target_semantic):
            miou.update_from_panoptic(preds, target_semantic, pred_div)
            self.update(preds, targets)
            return
        miou._require_gpu()
        self._device_update(preds, targets, want_matches=False, miou=miou,
                            target_semantic=target_semantic, pred_div=pred_div)

    @staticmethod
    def parts_usable(parts: Optional[dict], preds: torch.Tensor, max_instances_per_category: int) -> bool:
        """`parts` (ops.panoptic_pipeline's semantic_idx_u8 / instance / pan_of_inst + the thing
        LUT, what the map `preds` was painted from) can stand in for `preds` in
        `update_with_miou`: `preds` IS the painted map of these parts"""
        painted = parts.get('panoptic') if parts else None
        # the very map these parts were painted into: the tensor itself, or a view of ALL of it
        # (the full-resolution entry of a prediction at dataset resolution is a full slice of it)
        if not isinstance(painted, torch.Tensor) or not (
                painted is preds or (preds.data_ptr() == painted.data_ptr() and preds.shape == painted.shape
                                     and preds.stride() == painted.stride() and preds.dtype == painted.dtype
                                     and preds.is_contiguous())):
            return False
        s, i, t, th = (parts.get(k) for k in ('semantic_idx_u8', 'instance', 'pan_of_inst', 'is_thing'))
        return (all(isinstance(x, torch.Tensor) and x.is_cuda and x.is_contiguous() for x in (s, i, t, th))
                and s.dtype == torch.uint8 and i.dtype == torch.uint8 and t.dtype == torch.int64
                and th.dtype == torch.uint8 and s.shape == preds.shape and i.shape == preds.shape
                and t.shape == (preds.shape[0], 256) and th.numel() <= 255
                and int(parts.get('max_instances_per_category', -1)) == int(max_instances_per_category))

    def update_with_miou_parts(self, parts: dict, targets: torch.Tensor, miou,
                               target_semantic: torch.Tensor, pred_div: int) -> None:
        """`update_with_miou(parts['panoptic'], ...)` without reading the painted int64 map: the
        predicted id of every pixel is formed in registers from the parts the merge painted it
        from (csrc/metrics.hip k_pq_count_parts: compact keys): bit-identical states, 11 instead of 17
        bytes per pixel.  Falls back to `update_with_miou` when the parts do not apply."""
        preds = parts['panoptic']
        if not self.parts_usable(parts, preds, self.max_instances_per_category):
            self.update_with_miou(preds, targets, miou, target_semantic, pred_div)
            return
        if not self._can_fuse(preds, miou, target_semantic):
            # e.g. more than 64 classes: the confusion matrix in its own pass over the map, the PQ
            # count still from the parts
            miou.update_from_panoptic(preds, target_semantic, pred_div)
            self._device_update(preds, targets, want_matches=False, parts=parts)
            return
        miou._require_gpu()
        self._device_update(preds, targets, want_matches=False, miou=miou,
                            target_semantic=target_semantic, pred_div=pred_div, parts=parts)

    def update(self, preds: torch.Tensor, targets: torch.Tensor) -> None:
        self._device_update(preds, targets, want_matches=False)

    def _check_status(self) -> None:
        st = int(self._status.item())
        if st:
            self._status.zero_()
            msgs = [m for bit, m in _STATUS_MESSAGES.items() if st & bit]
            raise ValueError('PanopticQuality: ' + '; '.join(msgs))

    # ----------------------------------------------------------------- compute
    def _valid_categories(self) -> torch.Tensor:
        valid = (self.tp_per_class + self.fn_per_class + self.fp_per_class) != 0
        if 0 <= self.ignored_label < self.num_categories:
            valid[self.ignored_label] = False
        return valid

    def _valid_categories_with_gt(self) -> torch.Tensor:
        valid = (self.tp_per_class + self.fn_per_class) != 0
        if 0 <= self.ignored_label < self.num_categories:
            valid[self.ignored_label] = False
        return valid

    def result_per_category(self) -> Dict[str, torch.Tensor]:
        sq = realdiv_maybe_zero(self.iou_per_class, self.tp_per_class)
        rq = realdiv_maybe_zero(
            self.tp_per_class,
            self.tp_per_class + 0.5 * self.fn_per_class + 0.5 * self.fp_per_class)
        return {'sq_per_class': sq, 'rq_per_class': rq, 'pq_per_class': sq * rq}

    def compute(self, suffix: str = '') -> Dict[str, torch.Tensor]:
        self._check_status()
        results = self.result_per_category()
        valid = self._valid_categories()
        valid_gt = self._valid_categories_with_gt()
        thing = self.is_thing.to(valid.device)
        stuff = self.is_stuff.to(valid.device)
        sets = {
            f'all{suffix}': valid,
            f'things{suffix}': valid & thing,
            f'stuff{suffix}': valid & stuff,
            # variants that ignore FPs of classes without GT (fixed #categories)
            f'all_with_gt{suffix}': valid_gt,
            f'things_with_gt{suffix}': valid_gt & thing,
            f'stuff_with_gt{suffix}': valid_gt & stuff,
        }
        for name, sel in sets.items():
            if bool(sel.any()):
                results[f'{name}_pq'] = results['pq_per_class'][sel].mean()
                results[f'{name}_sq'] = results['sq_per_class'][sel].mean()
                results[f'{name}_rq'] = results['rq_per_class'][sel].mean()
                results[f'{name}_num_categories'] = sel.int().sum()
            else:
                for k in ('pq', 'sq', 'rq', 'num_categories'):
                    results[f'{name}_{k}'] = torch.tensor(0)
        return results


def compare_and_accumulate(
    pred: torch.Tensor,
    target: torch.Tensor,
    num_categories: int,
    ignored_label: int,
    max_instances_per_category,
    offset: int,
    void_segment_id: int
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, set]:
    """The reference's per-image function (metric/pq.py:60-179) with its signature and return
    value — (iou, tp, fn, fp per class as float64 [num_categories], set of matched
    (target segment id, predicted segment id)) — computed by the HIP kernels for one
    [H,W] pair.  `PanopticQuality.update` batches the same kernels over all images."""
    if pred.ndim != 2 or target.shape != pred.shape:
        raise ValueError('compare_and_accumulate expects one [H,W] prediction / target pair')
    dev = pred.device if pred.is_cuda else torch.device('cuda', torch.cuda.current_device())
    one = PanopticQuality(num_categories, ignored_label, int(max_instances_per_category), offset,
                          [False] * num_categories, device=dev)
    one.void_segment_id = int(void_segment_id)
    matches, n_matches = one._device_update(pred.unsqueeze(0), target.unsqueeze(0), want_matches=True)
    n = int(n_matches.cpu()[0])
    one._check_status()
    if n > one._match_capacity:
        raise ValueError('more matched segments than the match table holds')
    matched = {(int(t), int(p)) for t, p in matches[0, :n].cpu().tolist()}
    return (one.iou_per_class.clone(), one.tp_per_class.clone(), one.fn_per_class.clone(),
            one.fp_per_class.clone(), matched)
