"""Batch-level, on-device twin of `PanopticTargetGenerator`
(reference data/preprocessing/panoptic.py:16-85 -> naive_merge_semantic_and_instance_np,
utils/panoptic_merge.py:43-107)."""
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from ... import ops
from ...utils.panoptic_merge import _ids_to_dicts
from .instance import _device_lut


class PanopticTargetGenerator:
    def __init__(self, semantic_classes_is_thing: Optional[Tuple[bool]] = None,
                 max_instances: int = 1024, **kwargs) -> None:
        self._is_thing = None if semantic_classes_is_thing is None else \
            np.asarray(semantic_classes_is_thing, dtype=bool)
        # hypersim has more than 256 instances per image (panoptic.py:36-38)
        self._max_instances_per_category = 1 << 16
        self._void_label = 0
        self._max_instances = max_instances
        self._max_segments = 2048
        self._luts: Dict = {}

    def __call__(self, batch: Dict[str, Any], n_classes: Optional[int] = None) -> Dict[str, Any]:
        if 'instance' not in batch or 'semantic' not in batch:
            return batch
        sem, ins = batch['semantic'], batch['instance']
        th = None if self._is_thing is None else _device_lut(self._is_thing, sem.device, self._luts)
        nc = len(self._is_thing) if self._is_thing is not None else (n_classes or 256)
        while True:
            r = ops.panoptic_targets(sem, ins, nc, th, self._max_instances_per_category,
                                     self._void_label, self._max_instances, self._max_segments)
            status = int(r['status'].item())
            if status & 1 and self._max_instances < 4096:
                self._max_instances = min(4096, self._max_instances * 4)
                continue
            if status & 128 and self._max_segments < (1 << 16):
                self._max_segments *= 4
                continue
            break
        if status & 32:
            raise ValueError('instance ids outside [0, 65535]')
        if status & 64:
            raise ValueError(f'semantic labels outside [0, {nc})')
        if status & (1 | 128):
            raise NotImplementedError('too many instances / segments in one image')
        batch['panoptic'] = r['panoptic']
        batch['panoptic_ids_to_instance_dict'] = _ids_to_dicts(r['ids_pan'], r['ids_ins'], r['n_ids'])
        return batch
