"""
Semantic + instance -> panoptic merge on the MI355X.

Drop-in for the reference's `utils/panoptic_merge.py` entry points that sit on
the hot path (`deeplab_merge_batch` :18-40, `deeplab_merge_semantic_and_instance`
:172-225).  The arithmetic runs in the HIP kernels k_merge_votes / k_assign /
k_merge_paint (csrc/panoptic.hip) through `nmsa_panoptic_merge`.

The numpy twins of the reference (`*_np`, used by the CPU dataloader's
PanopticTargetGenerator) keep their signatures at the end of this module and run on the same
HIP kernels (upload, merge, download).
"""
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import ops

_MAX_INSTANCE_ID = 255


def _ids_to_dicts(ids_pan: torch.Tensor, ids_ins: torch.Tensor, n_ids: torch.Tensor,
                  limit: Optional[int] = None) -> List[Dict[int, int]]:
    """ONE device->host copy of the small per-image id tables.  `limit`: a known upper bound of
    n_ids (e.g. the largest number of instances of an image), so that only that many columns
    travel."""
    k = ids_pan.shape[1] if limit is None else max(1, min(int(limit), ids_pan.shape[1]))
    packed = torch.cat([ids_pan[:, :k], ids_ins[:, :k], n_ids.to(torch.int64).unsqueeze(1)],
                       dim=1).cpu()
    out = []
    for row in packed.tolist():
        n = min(row[2 * k], k)
        out.append(dict(zip(row[:n], row[k:k + n])))           # insertion order = ascending id
    return out


_THING_LUTS: Dict[tuple, torch.Tensor] = {}


def _merge_on_device(sem, ins, fg, max_instances_per_category, thing_ids, void_label,
                     n_classes=None) -> Dict[str, torch.Tensor]:
    """the kernels behind `deeplab_merge_batch` on device tensors, no host objects: the result
    dict holds 'panoptic' and the id tables (`_ids_to_dicts` turns those into the reference's
    dicts).  In-package callers that do not need the dicts stop here — no device->host copy."""
    dev = sem.device
    if n_classes is None:
        n_classes = int(sem.max()) + 1
    n_classes = max(int(n_classes), 1)
    thing_list = tuple(int(t) for t in thing_ids)
    key = (dev, n_classes, thing_list)
    lut = _THING_LUTS.get(key)
    if lut is None:                             # uploaded once per (device, classes, thing ids)
        lut = torch.zeros((n_classes,), dtype=torch.uint8)
        for t in thing_list:
            if 0 <= t < n_classes:
                lut[t] = 1
        lut = _THING_LUTS[key] = lut.to(dev)
    if ins.dtype in (torch.uint8, torch.bool):
        # prediction path: uint8 ids (reference instance.py:236), direct-indexed kernels
        return ops.panoptic_merge(sem, ins, fg, lut, int(max_instances_per_category),
                                  int(void_label))
    # ground-truth maps: ids 0..65535, ranked per image on the device
    for max_segments in (1024, 4096, 65536):        # (65536: every id a uint16 map can hold)
        r = ops.panoptic_merge_wide(sem, ins, fg, lut, int(max_instances_per_category),
                                    int(void_label), max_segments=max_segments)
        st = int(r['status'].item())
        if st & 32:
            raise NotImplementedError('instance ids outside [0, 65535] are not supported '
                                      '(dataset instance maps are uint16)')
        if not (st & 1):
            return r
    raise AssertionError('unreachable: 65536 segments hold every id of [0, 65535]')


def deeplab_merge_batch(
    semantic_batch: torch.Tensor,
    instance_batch: torch.Tensor,
    instance_fg_batch: torch.Tensor,
    max_instances_per_category: int,
    thing_ids: Sequence[int],
    void_label: int,
    n_classes: Optional[int] = None,
) -> Tuple[torch.Tensor, List[Dict[int, int]]]:
    """Same signature/returns as the reference (panoptic_merge.py:18-40) plus the
    optional `n_classes` (number of class VALUES incl. void; saves one device
    reduction).  Tensors on the GPU stay there; CPU tensors (what the reference's
    callers pass after `.cpu()`) are uploaded, merged by the HIP kernels and the
    panoptic map is returned on the CPU like the reference does."""
    if not torch.cuda.is_available():
        raise ops.L.NmsaError('deeplab_merge_batch needs the MI355X HIP path '
                              '(no CPU fallback in this package)')
    in_device = semantic_batch.device
    dev = in_device if in_device.type == 'cuda' else torch.device('cuda', torch.cuda.current_device())
    sem = semantic_batch.to(dev)
    ins = instance_batch.to(dev)
    fg = instance_fg_batch.to(dev)
    if sem.is_floating_point() or ins.is_floating_point():
        raise TypeError('semantic / instance maps must be integer tensors')
    if sem.ndim != 3 or ins.shape != sem.shape or fg.shape != sem.shape:
        raise ValueError('expected three tensors of shape (B, H, W)')

    r = _merge_on_device(sem, ins, fg, max_instances_per_category, thing_ids, void_label, n_classes)
    dicts = _ids_to_dicts(r['ids_pan'], r['ids_ins'], r['n_ids'])
    pan = r['panoptic']
    if in_device.type != 'cuda':
        pan = pan.to(in_device)
    return pan, dicts


def deeplab_merge_semantic_and_instance(
    sem_seg: torch.Tensor,
    ins_seg: torch.Tensor,
    semantic_thing_seg: torch.Tensor,
    max_instances_per_category: int,
    thing_ids: Sequence[int],
    void_label: int,
) -> Tuple[torch.Tensor, Dict[int, int]]:
    """Single-image form (panoptic_merge.py:172-225)."""
    pan, dicts = deeplab_merge_batch(sem_seg.unsqueeze(0), ins_seg.unsqueeze(0),
                                     semantic_thing_seg.unsqueeze(0),
                                     max_instances_per_category, thing_ids, void_label)
    return pan[0], dicts[0]


# ---- numpy entry points (reference utils/panoptic_merge.py:43-169) -------------------------------
# The reference's dataloader-side twins take and return numpy arrays (semantic uint8 / uint16,
# instance uint16 -> panoptic uint32 + id dict).  Same signatures here; the arrays are uploaded,
# merged by the HIP kernels and downloaded again — for batches use the tensor forms
# (`PanopticTargetGenerator`, `deeplab_merge_batch`), which stay on the device.
def _np_inputs(sem_seg, ins_seg, void_label):
    import numpy as np
    assert sem_seg.dtype in (np.uint8, np.uint16)
    assert ins_seg.dtype == np.uint16
    assert void_label >= 0
    if not torch.cuda.is_available():
        raise ops.L.NmsaError('the panoptic merge needs the MI355X HIP path (no CPU fallback)')
    dev = torch.device('cuda', torch.cuda.current_device())
    sem = torch.from_numpy(sem_seg.astype(np.int32)).to(dev).unsqueeze(0)
    ins = torch.from_numpy(ins_seg.astype(np.int32)).to(dev).unsqueeze(0)
    return np, dev, sem, ins


def naive_merge_semantic_and_instance_np(sem_seg, ins_seg, max_instances_per_category: int,
                                         thing_ids: Sequence[int], void_label: int):
    """reference utils/panoptic_merge.py:43-107 (every (instance, class) pair is a segment)."""
    np, dev, sem, ins = _np_inputs(sem_seg, ins_seg, void_label)
    n_classes = int(sem_seg.max()) + 1 if sem_seg.size else 1
    lut = torch.zeros((n_classes,), dtype=torch.uint8)
    for t in thing_ids:
        if 0 <= int(t) < n_classes:
            lut[int(t)] = 1
    max_instances, max_segments = 1024, 2048
    while True:
        r = ops.panoptic_targets(sem, ins, n_classes, lut.to(dev), int(max_instances_per_category),
                                 int(void_label), max_instances, max_segments)
        st = int(r['status'].item())
        if st & 1 and max_instances < 4096:
            max_instances = 4096
            continue
        if st & 128 and max_segments < (1 << 16):
            max_segments *= 4
            continue
        break
    if st:
        raise NotImplementedError(f'naive merge: unsupported input (status {st})')
    dicts = _ids_to_dicts(r['ids_pan'], r['ids_ins'], r['n_ids'])
    return r['panoptic'][0].cpu().numpy().astype(np.uint32), dicts[0]


def deeplab_merge_semantic_and_instance_np(sem_seg, ins_seg, semantic_thing_seg,
                                           max_instances_per_category: int,
                                           thing_ids: Sequence[int], void_label: int):
    """reference utils/panoptic_merge.py:108-169 (majority vote per instance)."""
    np, dev, sem, ins = _np_inputs(sem_seg, ins_seg, void_label)
    thing = torch.from_numpy(np.ascontiguousarray(semantic_thing_seg > 0)).to(dev).unsqueeze(0)
    pan, dicts = deeplab_merge_batch(sem, ins, thing, max_instances_per_category, thing_ids,
                                     void_label)
    return pan[0].cpu().numpy().astype(np.uint32), dicts[0]
