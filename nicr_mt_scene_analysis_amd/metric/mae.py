"""Orientation MAE on matched instances (reference metric/mae.py:15-172).

The PQ part runs on the device; the per-matched-pair angle bookkeeping works on
Python dicts of <= a few dozen floats per image and stays on the host
("MAAE scalars stay host-side")."""
import math
from typing import Dict, List, Optional, Tuple

import torch

from .base import Metric
from .pq import PanopticQuality

OrientationDict = Dict[int, float]


def abs_angle_error_rad(pred_angle: torch.Tensor, target_angle: torch.Tensor) -> torch.Tensor:
    two_pi = 2 * math.pi
    diff = (pred_angle % two_pi) - (target_angle % two_pi)
    return torch.abs((diff + math.pi) % two_pi - math.pi)        # in [0, pi]


def _abs_angle_error_f32(pred_angle: float, target_angle: float) -> float:
    """same fp32 arithmetic as `abs_angle_error_rad(torch.tensor(a), torch.tensor(b))`"""
    return float(abs_angle_error_rad(torch.tensor(pred_angle), torch.tensor(target_angle)))


class _AngularErrorStates:
    """the two accumulators both metrics share: sum of |angle error| (rad) and its count"""

    def _add_angular_states(self) -> None:
        self.add_state('sum_angular_error', torch.tensor(0, dtype=torch.float64),
                       dist_reduce_fx='sum')
        self.add_state('n_elements', torch.tensor(0, dtype=torch.int64), dist_reduce_fx='sum')

    def _accumulate(self, pairs) -> None:
        """pairs: iterable of (predicted angle, target angle) in rad"""
        total, n = 0.0, 0
        for pred_angle, target_angle in pairs:
            total += _abs_angle_error_f32(float(pred_angle), float(target_angle))
            n += 1
        self.sum_angular_error += total
        self.n_elements += n

    def _mean_rad(self) -> torch.Tensor:
        return self.sum_angular_error / self.n_elements


class MeanAbsoluteAngularError(_AngularErrorStates, Metric):
    def __init__(self, **kwargs) -> None:
        super().__init__(**kwargs)
        self._add_angular_states()

    def update(self, orientation_preds: List[OrientationDict],
               orientation_target: List[OrientationDict]) -> None:
        self._accumulate((angle, targets[key])
                         for preds, targets in zip(orientation_preds, orientation_target)
                         for key, angle in preds.items())

    def compute(self) -> Tuple[torch.Tensor, torch.Tensor]:
        rad = self._mean_rad()
        return rad, torch.rad2deg(rad)


class PanopticQualityWithOrientationMAE(_AngularErrorStates, PanopticQuality):
    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._add_angular_states()

    def update(self,
               panoptic_preds: torch.Tensor,
               orientation_preds: Optional[List[OrientationDict]],
               panoptic_preds_id_dicts: Optional[List[Dict]],
               panoptic_target: torch.Tensor,
               orientation_target: Optional[List[OrientationDict]],
               panoptic_target_id_dicts: Optional[List[Dict]],
               miou=None, semantic_target: Optional[torch.Tensor] = None,
               pred_div: int = 1, panoptic_pred_parts: Optional[dict] = None) -> None:
        """`miou` / `semantic_target` / `pred_div` (extension): also do
        `miou.update(panoptic_preds // pred_div, semantic_target)` — in the same pass over the
        prediction when the fused kernel applies (see PanopticQuality.update_with_miou)."""
        assert panoptic_preds.ndim == 3
        assert len(panoptic_target) == len(panoptic_preds)
        with_mae = orientation_preds is not None and orientation_target is not None
        fuse = {}
        if miou is not None:
            if self._can_fuse(panoptic_preds, miou, semantic_target):
                miou._require_gpu()
                fuse = dict(miou=miou, target_semantic=semantic_target, pred_div=pred_div)
                # `panoptic_pred_parts` (extension): what the prediction was painted from — read
                # instead of the int64 map when no match list is needed (2 B/px instead of 8 B/px)
                if not with_mae and self.parts_usable(panoptic_pred_parts, panoptic_preds,
                                                      self.max_instances_per_category):
                    fuse['parts'] = panoptic_pred_parts
            else:
                miou.update_from_panoptic(panoptic_preds, semantic_target, pred_div)
        res = self._device_update(panoptic_preds, panoptic_target, want_matches=with_mae, **fuse)
        if not with_mae:
            return
        matches, n_matches = res
        # match counts and the status word in one copy (the sync of this update)
        head = torch.cat([n_matches, self._status]).cpu().tolist()
        n_host, status = head[:-1], head[-1]
        if status:
            self._check_status()
        if max(n_host, default=0) > self._match_capacity:
            raise ValueError('more matched segments per image than the match table holds')
        # only the filled rows travel (the table holds 1024 pairs per image)
        m_host = matches[:, :max(max(n_host, default=0), 1)].cpu().tolist()
        for b, n in enumerate(n_host):
            self.update_mae(orientation_preds[b], panoptic_preds_id_dicts[b],
                            orientation_target[b], panoptic_target_id_dicts[b],
                            [tuple(m_host[b][i]) for i in range(n)])

    def update_mae(self, orientation_preds: OrientationDict, panoptic_preds_id_dicts: Dict,
                   orientation_target: OrientationDict, panoptic_target_id_dicts: Dict,
                   matching: List[Tuple[int, int]]) -> None:
        """matching: (target panoptic id, predicted panoptic id) of the true positives; a pair
        counts when both ids map to instances that carry an orientation (mae.py:129-162)"""
        def angle_of(pan_id, id_dict, angles):
            instance = id_dict.get(pan_id)
            return None if instance is None else angles.get(instance)

        pairs = []
        for target_id, pred_id in matching:
            if target_id == 0:                                  # stuff / void / background
                continue
            target_angle = angle_of(target_id, panoptic_target_id_dicts, orientation_target)
            pred_angle = angle_of(pred_id, panoptic_preds_id_dicts, orientation_preds)
            if target_angle is not None and pred_angle is not None:
                pairs.append((pred_angle, target_angle))
        self._accumulate(pairs)

    def compute(self, suffix: str = '') -> Dict[str, torch.Tensor]:
        r = super().compute(suffix=suffix)
        rad = self._mean_rad()
        r[f'mae{suffix}_rad'] = rad
        r[f'mae{suffix}_deg'] = torch.rad2deg(rad)
        return r
