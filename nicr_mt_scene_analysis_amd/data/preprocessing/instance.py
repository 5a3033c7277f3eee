"""Batch-level, on-device twins of the reference's instance target preprocessing
(reference data/preprocessing/instance.py:19-286).

The reference runs these per SAMPLE on numpy arrays inside the dataloader workers; here they
run per BATCH on device tensors in the on-wire dtypes (`semantic` uint8 [B,H,W], `instance`
int32 [B,H,W] holding uint16 ids) and write the same keys a collated reference batch holds:
`instance_center` f32 [B,H,W], `instance_offset` [B,2,H,W], `instance_foreground` /
`instance_center_mask` bool [B,H,W].  The arithmetic is the HIP library's
(`nmsa_instance_clear_stuff`, `nmsa_instance_targets`); constructor kwargs are the
reference's (the `*_from_meta` / multiscale plumbing of the numpy pipeline is not mirrored).
"""
from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from ... import ops


def _device_lut(flags: np.ndarray, dev: torch.device, cache: Dict) -> torch.Tensor:
    if dev not in cache:
        cache[dev] = torch.from_numpy(np.ascontiguousarray(flags, dtype=np.uint8)).to(dev)
    return cache[dev]


class InstanceClearStuffIDs:
    def __init__(self, semantic_classes_is_thing: Tuple[bool], **kwargs) -> None:
        # stuff classes INCLUDING void (instance.py:31-32)
        self._is_stuff = np.logical_not(np.asarray(semantic_classes_is_thing, dtype=bool))
        self._luts: Dict = {}

    def __call__(self, batch: Dict[str, Any]) -> Dict[str, Any]:
        if 'instance' not in batch or 'semantic' not in batch:
            return batch
        lut = _device_lut(self._is_stuff, batch['semantic'].device, self._luts)
        ops.instance_clear_stuff(batch['semantic'], batch['instance'], lut)      # in place
        return batch


class InstanceTargetGenerator:
    def __init__(
        self,
        sigma: int,
        semantic_classes_is_thing: Optional[Tuple[bool]] = None,      # with void
        normalized_offset: bool = True,
        max_instances: int = 1024,
        **kwargs
    ) -> None:
        self._sigma = int(sigma)
        self._normalized_offset = normalized_offset
        self._max_instances = max_instances
        if semantic_classes_is_thing is not None:
            is_thing = np.asarray(semantic_classes_is_thing, dtype=bool)
            self._is_thing = is_thing
            # stuff ids without the first one = void (instance.py:126-129)
            stuff_ids = np.where(~is_thing)[0][1:]
            self._is_stuff = np.zeros_like(is_thing)
            self._is_stuff[stuff_ids] = True
            self._n_classes = len(is_thing)
        else:
            self._is_thing = self._is_stuff = None
            self._n_classes = None
        self._luts_thing: Dict = {}
        self._luts_stuff: Dict = {}
        self.last_dynamic_parameters: Dict[str, Any] = {}

    def __call__(self, batch: Dict[str, Any], n_classes: Optional[int] = None) -> Dict[str, Any]:
        if 'instance' not in batch:
            return batch                                      # inference (instance.py:162-165)
        sem, ins = batch['semantic'], batch['instance']
        dev = sem.device
        th = st = None
        if self._is_thing is not None:
            th = _device_lut(self._is_thing, dev, self._luts_thing)
            st = _device_lut(self._is_stuff, dev, self._luts_stuff)
        nc = self._n_classes or n_classes or 256
        while True:
            r = ops.instance_targets(sem, ins, nc, th, st, self._sigma, self._normalized_offset,
                                     self._max_instances)
            host = torch.cat([r['status'], r['n_encoded'], r['n_skipped']]).cpu().tolist()
            status = host[0]
            if status & 1 and self._max_instances < 4096:
                self._max_instances = min(4096, self._max_instances * 4)
                continue
            break
        B = sem.shape[0]
        if status & 32:
            raise ValueError('instance ids outside [0, 65535]')
        if status & 64:
            raise ValueError(f'semantic labels outside [0, {nc})')
        if status & 1:
            raise NotImplementedError('more than 4096 distinct instance ids in one image')
        n_enc, n_skp = host[1:1 + B], host[1 + B:1 + 2 * B]
        # the reference asserts that every non-foreground pixel has id 0 (instance.py:255-257):
        # a skipped (stuff-majority) instance always violates it
        assert sum(n_skp) == 0, \
            'instances with a stuff majority class: apply InstanceClearStuffIDs first'
        batch['instance_center'] = r['center']
        batch['instance_offset'] = r['offset']
        batch['instance_foreground'] = r['foreground']
        batch['instance_center_mask'] = r['center_mask']
        enc = r['encoded_ids'].cpu()
        self.last_dynamic_parameters = {
            'encoded_instances': [enc[b, :n_enc[b]].tolist() for b in range(B)],
            'skipped_instances_due_to_stuff': [[] for _ in range(B)],
        }
        return batch
