"""
Functional front-end of the HIP hot path: torch device tensors in, torch device
tensors out, every call stream-ordered on torch's current HIP stream, no host
synchronisation.  These are thin argument marshallers over include/nmsa.h; the
reference-shaped classes (model/postprocessing, utils/panoptic_merge, metric,
loss) are built on top of them.
"""
from typing import Dict, Optional, Tuple

import torch

from . import _lib as L

DEFAULT_MAX_CENTERS = 256

# persistent, always-zero vote tables (one per device / shape): nmsa_panoptic_assign clears
# the rows it read, so the next step needs no memset.  Keyed by stream as well, so that
# concurrent pipelines on different streams never share a table.
# At most _VOTE_TABLES_MAX tables are kept (least recently used first out): a ragged last batch
# or short-lived streams must not leave a table behind each.  A dropped table is freed through
# torch's allocator, stream-ordered behind the kernels that used it; a table that a captured
# hipGraph holds stays alive through the graph's private pool.
_VOTE_TABLES: 'collections.OrderedDict[tuple, torch.Tensor]' = __import__('collections').OrderedDict()
_VOTE_TABLES_MAX = 8


def _vote_table(dev: torch.device, B: int, n_cols: int) -> torch.Tensor:
    key = (dev, B, n_cols, torch.cuda.current_stream(dev).cuda_stream)
    t = _VOTE_TABLES.get(key)
    if t is None:
        t = torch.zeros((B, 256, n_cols), dtype=torch.int32, device=dev)
        _VOTE_TABLES[key] = t
        while len(_VOTE_TABLES) > _VOTE_TABLES_MAX:
            _VOTE_TABLES.popitem(last=False)
    else:
        _VOTE_TABLES.move_to_end(key)
    return t


def _u8(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """bool / uint8 tensor viewed as uint8 (torch.bool storage is one byte)."""
    if t is None:
        return None
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)
    if t.dtype == torch.uint8:
        return t.contiguous()
    return (t != 0).contiguous().view(torch.uint8)


# ----------------------------------------------------------------------------- a2
def center_nms_topk(
    center_heatmap: torch.Tensor,
    foreground_mask: Optional[torch.Tensor] = None,
    threshold: float = 0.1,
    kernel_size: int = 3,
    top_k: int = 64,
    apply_foreground_mask: bool = False,
    max_centers: int = DEFAULT_MAX_CENTERS,
    want_mask: bool = False,
) -> Dict[str, torch.Tensor]:
    """reference: InstancePostprocessing._get_instance_centers (instance.py:79-168)"""
    c = L.require_device_tensor(center_heatmap, 'center_heatmap')
    if c.dtype != torch.float32:
        c = c.float()
    if c.ndim == 4:
        assert c.shape[1] == 1
        c = c[:, 0]
    c = c.contiguous()
    B, H, W = c.shape
    dev = c.device
    fg = _u8(foreground_mask) if apply_foreground_mask else None
    cyx = torch.empty((B, max_centers, 2), dtype=torch.int32, device=dev)
    n = torch.empty((B,), dtype=torch.int32, device=dev)
    scores = torch.empty((B, max_centers), dtype=torch.float32, device=dev)
    mask = torch.empty((B, H, W), dtype=torch.uint8, device=dev) if want_mask else None
    ws_bytes = L.lib().nmsa_center_nms_workspace_bytes(B, H, W)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    L.check(L.lib().nmsa_center_nms_topk(
        L.ptr(c), L.ptr(fg), B, H, W, float(threshold), int(kernel_size), int(top_k),
        int(bool(apply_foreground_mask)), int(max_centers),
        L.ptr(cyx), L.ptr(n), L.ptr(scores), L.ptr(mask), L.ptr(ws), ws_bytes,
        L.stream_ptr(dev)), 'nmsa_center_nms_topk')
    out = {'centers_yx': cyx, 'n_centers': n, 'scores': scores}
    if want_mask:
        out['center_mask'] = mask.view(torch.bool)
    return out


# ----------------------------------------------------------------------------- a3
def group_offsets(
    center_offset: torch.Tensor,
    foreground_mask: torch.Tensor,
    centers_yx: torch.Tensor,
    n_centers: torch.Tensor,
    scale_y: float = 1.0,
    scale_x: float = 1.0,
    distance_threshold: Optional[float] = None,
    want_area: bool = True,
) -> Dict[str, torch.Tensor]:
    """reference: InstancePostprocessing._get_instance_segmentation (instance.py:187-253)"""
    off = L.require_device_tensor(center_offset, 'center_offset')
    if off.dtype != torch.float32:
        off = off.float()
    B, two, H, W = off.shape
    assert two == 2
    dev = off.device
    fg = _u8(foreground_mask)
    inst = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    area = torch.empty((B, 256), dtype=torch.int32, device=dev) if want_area else None
    L.check(L.lib().nmsa_group_offsets(
        L.ptr(off), L.ptr(fg), L.ptr(centers_yx), L.ptr(n_centers), B, H, W,
        int(centers_yx.shape[1]), float(scale_y), float(scale_x),
        0 if distance_threshold is None else 1,
        0.0 if distance_threshold is None else float(distance_threshold),
        L.ptr(inst), L.ptr(area), L.stream_ptr(dev)), 'nmsa_group_offsets')
    return {'instance': inst, 'area': area}


# ----------------------------------------------------------------------------- f2
NMSA_ELEM_F32 = 8


def _crop_geometry(t: torch.Tensor, crop) -> Tuple[int, int, int, int, int, int]:
    Hs, Ws = int(t.shape[-2]), int(t.shape[-1])
    if crop is None:
        return Hs, Ws, 0, 0, Hs, Ws
    sl_h, sl_w = crop
    y0, y1, st_y = sl_h.indices(Hs)
    x0, x1, st_x = sl_w.indices(Ws)
    if st_y != 1 or st_x != 1 or y1 <= y0 or x1 <= x0:
        raise ValueError(f'unsupported valid-region slices {crop}')
    return Hs, Ws, y0, x0, y1 - y0, x1 - x0


def resize_nearest(maps: torch.Tensor, size: Tuple[int, int], crop=None) -> torch.Tensor:
    """crop `maps[..., crop]` and F.interpolate(mode='nearest') to `size`, bit-identical to the
    reference's `_crop_to_valid_region_and_resize_prediction` (dense_base.py:15-58),
    including its float32 round trip for int32 / int64 maps."""
    x = L.require_device_tensor(maps, 'maps')
    code = NMSA_ELEM_F32 if x.dtype == torch.float32 else L.int_dtype_code(x)
    Hs, Ws, y0, x0, h, w = _crop_geometry(x, crop)
    Ho, Wo = int(size[0]), int(size[1])
    planes = x.numel() // (Hs * Ws)
    out = torch.empty(tuple(x.shape[:-2]) + (Ho, Wo), dtype=x.dtype, device=x.device)
    L.check(L.lib().nmsa_resize_nearest(
        L.ptr(x), code, planes, Hs, Ws, y0, x0, h, w, Ho, Wo, L.ptr(out),
        L.stream_ptr(x.device)), 'nmsa_resize_nearest')
    return out


def resize_bilinear(maps: torch.Tensor, size: Tuple[int, int], crop=None) -> torch.Tensor:
    """crop + F.interpolate(mode='bilinear', align_corners=False) (dense_base.py:15-58)."""
    x = L.require_device_tensor(maps, 'maps')
    Hs, Ws, y0, x0, h, w = _crop_geometry(x, crop)
    Ho, Wo = int(size[0]), int(size[1])
    planes = x.numel() // (Hs * Ws)
    out = torch.empty(tuple(x.shape[:-2]) + (Ho, Wo), dtype=x.dtype, device=x.device)
    L.check(L.lib().nmsa_resize_bilinear(
        L.ptr(x), L.float_dtype_code(x), planes, Hs, Ws, y0, x0, h, w, Ho, Wo, L.ptr(out),
        L.stream_ptr(x.device)), 'nmsa_resize_bilinear')
    return out


def semantic_argmax_resized(
    logits: torch.Tensor,
    size: Tuple[int, int],
    crop=None,
    want_u8: bool = False,
    want_i64: bool = True,
    want_score: bool = True,
) -> Dict[str, torch.Tensor]:
    """reference: semantic.py:61-80 — crop + bilinear resize + softmax + max at the dataset
    resolution, in one pass over the network-resolution logits."""
    x = L.require_device_tensor(logits, 'logits')
    B, Cn = int(x.shape[0]), int(x.shape[1])
    Hs, Ws, y0, x0, h, w = _crop_geometry(x, crop)
    Ho, Wo = int(size[0]), int(size[1])
    dev = x.device
    u8 = torch.empty((B, Ho, Wo), dtype=torch.uint8, device=dev) if want_u8 else None
    i64 = torch.empty((B, Ho, Wo), dtype=torch.int64, device=dev) if want_i64 else None
    sc = torch.empty((B, Ho, Wo), dtype=torch.float32, device=dev) if want_score else None
    L.check(L.lib().nmsa_semantic_argmax_resized(
        L.ptr(x), L.float_dtype_code(x), B, Cn, Hs, Ws, y0, x0, h, w, Ho, Wo,
        L.ptr(u8), L.ptr(i64), L.ptr(sc), L.stream_ptr(dev)), 'nmsa_semantic_argmax_resized')
    return {'idx_u8': u8, 'idx': i64, 'score': sc}


# ----------------------------------------------------------------------------- a1
def semantic_argmax(
    logits: torch.Tensor,
    want_u8: bool = False,
    want_i64: bool = True,
    want_score: bool = True,
) -> Dict[str, torch.Tensor]:
    """reference: SemanticPostprocessing._postprocess_inference (semantic.py:52-53)"""
    x = L.require_device_tensor(logits, 'logits')
    B, Cn, H, W = x.shape
    dev = x.device
    u8 = torch.empty((B, H, W), dtype=torch.uint8, device=dev) if want_u8 else None
    i64 = torch.empty((B, H, W), dtype=torch.int64, device=dev) if want_i64 else None
    sc = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_score else None
    L.check(L.lib().nmsa_semantic_argmax(
        L.ptr(x), L.float_dtype_code(x), B, Cn, H, W, L.ptr(u8), L.ptr(i64), L.ptr(sc),
        L.stream_ptr(dev)), 'nmsa_semantic_argmax')
    return {'idx_u8': u8, 'idx': i64, 'score': sc}


def semantic_softmax(logits: torch.Tensor) -> torch.Tensor:
    """reference: F.softmax(output, dim=1) (semantic.py:52)"""
    x = L.require_device_tensor(logits, 'logits')
    B, Cn, H, W = x.shape
    probs = torch.empty((B, Cn, H, W), dtype=torch.float32, device=x.device)
    L.check(L.lib().nmsa_semantic_softmax(
        L.ptr(x), L.float_dtype_code(x), B, Cn, H, W, L.ptr(probs),
        L.stream_ptr(x.device)), 'nmsa_semantic_softmax')
    return probs


# ------------------------------------------------------------------ a1+a3+a4+a5
def panoptic_pipeline(
    semantic_logits: torch.Tensor,
    center_heatmap: torch.Tensor,
    center_offset: torch.Tensor,
    is_thing: torch.Tensor,                 # u8/bool [C], on device
    threshold: float = 0.1,
    kernel_size: int = 3,
    top_k: int = 64,
    apply_foreground_mask: bool = False,
    normalized_offset: bool = True,
    distance_threshold: Optional[float] = None,
    max_instances_per_category: int = 1 << 16,
    void_label: int = 0,
    max_centers: int = DEFAULT_MAX_CENTERS,
    want_score: bool = False,
    want_foreground: bool = True,
    want_panoptic_semantic: bool = False,
    fused_kernel_events: Optional[list] = None,
    on_centers=None,
) -> Dict[str, torch.Tensor]:
    """center-NMS -> fused argmax/grouping/votes -> assign -> paint.

    `on_centers`: called with the center tables right after the top-k selection is queued (the
    postprocessing API starts the asynchronous copy of the center counts there, so that its
    overflow check never waits for the streaming kernels behind it).

    `fused_kernel_events`: if a list is given, a (start, end) pair of HIP events
    recorded on the launch stream around the dominant kernel is appended
    (bench.py's live roofline measurement).

    reference: PanopticPostprocessing._postprocess_inference (panoptic.py:77-168).
    When the foreground-masked heatmap option is on, the foreground depends on
    the semantic argmax, so the argmax runs first as its own kernel.
    """
    lib = L.lib()
    x = L.require_device_tensor(semantic_logits, 'semantic_logits')
    off = L.require_device_tensor(center_offset, 'center_offset')
    if off.dtype != torch.float32:
        off = off.float()
    B, Cn, H, W = x.shape
    dev = x.device
    st = L.stream_ptr(dev)
    thing = _u8(is_thing)

    fg_for_nms = None
    if apply_foreground_mask:
        pre = semantic_argmax(x, want_u8=True, want_i64=False, want_score=False)
        fg_for_nms = thing[pre['idx_u8'].long()]
    cen = center_nms_topk(center_heatmap, fg_for_nms, threshold, kernel_size, top_k,
                          apply_foreground_mask, max_centers)
    if on_centers is not None:
        on_centers(cen)

    sem_u8 = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    inst = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    fg = torch.empty((B, H, W), dtype=torch.uint8, device=dev) if want_foreground else None
    score = torch.empty((B, H, W), dtype=torch.float32, device=dev) if want_score else None
    votes = _vote_table(dev, B, Cn + 1)
    vote_key = (dev, B, Cn + 1, torch.cuda.current_stream(dev).cuda_stream)
    sy, sx = (float(H), float(W)) if normalized_offset else (1.0, 1.0)
    if fused_kernel_events is not None:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(dev))
    L.check(lib.nmsa_panoptic_fused(
        L.ptr(x), L.float_dtype_code(x), L.ptr(off), L.ptr(cen['centers_yx']),
        L.ptr(cen['n_centers']), L.ptr(thing), B, Cn, H, W, int(max_centers), sy, sx,
        0 if distance_threshold is None else 1,
        0.0 if distance_threshold is None else float(distance_threshold),
        L.ptr(sem_u8), L.ptr(inst), L.ptr(fg), L.ptr(score), L.ptr(votes), 1,
        int(top_k) + 1, st), 'nmsa_panoptic_fused')
    if fused_kernel_events is not None:
        ev1.record(torch.cuda.current_stream(dev))
        fused_kernel_events.append((ev0, ev1))

    pan_of_inst = torch.empty((B, 256), dtype=torch.int64, device=dev)
    area = torch.empty((B, 256), dtype=torch.int32, device=dev)
    ids_pan = torch.empty((B, 256), dtype=torch.int64, device=dev)
    ids_ins = torch.empty((B, 256), dtype=torch.int64, device=dev)
    n_ids = torch.empty((B,), dtype=torch.int32, device=dev)
    try:
        L.check(lib.nmsa_panoptic_assign(
            L.ptr(votes), B, Cn + 1, 1, int(max_instances_per_category), int(void_label),
            L.ptr(pan_of_inst), L.ptr(area), L.ptr(ids_pan), L.ptr(ids_ins), L.ptr(n_ids), st),
            'nmsa_panoptic_assign')
    except Exception:
        _VOTE_TABLES.pop(vote_key, None)          # the table was written but not cleared
        raise

    pan = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    pan_sem = torch.empty((B, H, W), dtype=torch.int64, device=dev) \
        if want_panoptic_semantic else None
    L.check(lib.nmsa_panoptic_paint(
        L.ptr(sem_u8), L.ptr(inst), L.ptr(pan_of_inst), L.ptr(thing), B, Cn, H, W,
        int(max_instances_per_category), int(void_label), L.ptr(pan), L.ptr(pan_sem), st),
        'nmsa_panoptic_paint')

    return {
        'semantic_idx_u8': sem_u8, 'semantic_score': score,
        'foreground': None if fg is None else fg.view(torch.bool),
        'instance': inst, 'panoptic': pan, 'panoptic_semantic': pan_sem,
        'centers_yx': cen['centers_yx'], 'n_centers': cen['n_centers'],
        'center_scores': cen['scores'], 'area': area,
        'ids_pan': ids_pan, 'ids_ins': ids_ins, 'n_ids': n_ids,
        'pan_of_inst': pan_of_inst,
    }


# ----------------------------------------------------------------------------- f3
def panoptic_scores(
    logits: torch.Tensor,
    semantic_idx_u8: torch.Tensor,
    semantic_prob: torch.Tensor,
    instance: torch.Tensor,
    panoptic: torch.Tensor,
    pan_of_inst: torch.Tensor,
    instance_score_table: torch.Tensor,
    max_instances_per_category: int,
) -> Dict[str, torch.Tensor]:
    """reference: the `compute_scores` branch of PanopticPostprocessing (panoptic.py:171-239)"""
    x = L.require_device_tensor(logits, 'logits')
    B, Cn, H, W = x.shape
    dev = x.device
    sem = L.require_device_tensor(semantic_idx_u8, 'semantic_idx_u8')
    prob = L.require_device_tensor(semantic_prob, 'semantic_prob')
    ins = L.require_device_tensor(instance, 'instance')
    pan = L.require_device_tensor(panoptic, 'panoptic')
    poi = L.require_device_tensor(pan_of_inst, 'pan_of_inst')
    tab = L.require_device_tensor(instance_score_table, 'instance_score_table')
    assert sem.dtype == torch.uint8 and ins.dtype == torch.uint8 and pan.dtype == torch.int64
    assert prob.dtype == torch.float32 and tab.dtype == torch.float32 and poi.dtype == torch.int64
    assert tuple(tab.shape) == (B, 256) and tuple(poi.shape) == (B, 256)
    out = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(3)]
    mean = torch.empty((B, 256), dtype=torch.float32, device=dev)
    ws_bytes = L.lib().nmsa_panoptic_scores_workspace_bytes(B)
    ws = torch.empty((ws_bytes // 8,), dtype=torch.float64, device=dev)
    L.check(L.lib().nmsa_panoptic_scores(
        L.ptr(x), L.float_dtype_code(x), L.ptr(sem), L.ptr(prob), L.ptr(ins), L.ptr(pan),
        L.ptr(poi), L.ptr(tab), B, Cn, H, W, int(max_instances_per_category),
        L.ptr(out[0]), L.ptr(out[1]), L.ptr(out[2]), L.ptr(mean), L.ptr(ws), ws_bytes,
        L.stream_ptr(dev)), 'nmsa_panoptic_scores')
    return {'semantic_score': out[0], 'instance_score': out[1], 'panoptic_score': out[2],
            'mean_semantic_score': mean}


# ----------------------------------------------------------------------------- f4
_GAUSS_LUTS: Dict[tuple, torch.Tensor] = {}


def _gauss_lut(sigma: int, dev: torch.device) -> torch.Tensor:
    """heat-map value by integer squared distance: the entries of the reference's precomputed
    (6s+3)^2 patch (data/preprocessing/instance.py:147-154), same numpy float64 exp, rounded to
    float32 exactly like `np.maximum(center_img, gauss)` stored into the float32 image does."""
    key = (int(sigma), dev)
    if key not in _GAUSS_LUTS:
        import numpy as np
        r = 3 * int(sigma) + 1
        d2 = np.arange(2 * r * r + 1, dtype=np.float64)
        lut = np.exp(-d2 / (2 * int(sigma) ** 2)).astype(np.float32)
        _GAUSS_LUTS[key] = torch.from_numpy(lut).to(dev)
    return _GAUSS_LUTS[key]


# Persistent workspaces of the target generators, one per (device, stream, B, classes,
# max_instances): on the on-wire layout a call leaves its hash tables zeroed, so the next call on
# the same workspace skips the memset (`workspace_is_clean`).  LRU of 4.
_TARGET_WORKSPACES: 'collections.OrderedDict[tuple, list]' = __import__('collections').OrderedDict()


def _targets_workspace(B: int, n_classes: int, max_instances: int, dev, reusable: bool = False):
    """(workspace, bytes, entry): entry = [tensor, clean]; `clean` = this workspace was last used
    by a target-generator call on the on-wire layout that was enqueued successfully, and may skip
    its memset (the caller sets entry[1] = 1 after its own call went through)"""
    nbytes = L.lib().nmsa_targets_workspace_bytes(B, n_classes, max_instances)
    if nbytes == 0:
        raise ValueError('max_instances must be in [1, 4096]')
    if not reusable:
        t = torch.empty(((nbytes + 7) // 8,), dtype=torch.int64, device=dev)
        return t, nbytes, [t, 0]
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, B, n_classes, max_instances)
    entry = _TARGET_WORKSPACES.get(key)
    if entry is None:
        entry = _TARGET_WORKSPACES[key] = [torch.empty(((nbytes + 7) // 8,), dtype=torch.int64, device=dev), 0]
        while len(_TARGET_WORKSPACES) > 4:
            _TARGET_WORKSPACES.popitem(last=False)
    else:
        _TARGET_WORKSPACES.move_to_end(key)
    return entry[0], nbytes, entry


def _targets_on_wire(sem: torch.Tensor, ins: torch.Tensor, H: int, W: int, n_classes: int) -> bool:
    """the layouts the one-launch front end of csrc/targets.hip takes (it SETS the status word and
    leaves its workspace clean): the on-wire dtypes, rows of 4 pixels"""
    import os
    return (os.environ.get('NMSA_TG_FUSED', '1') != '0' and sem.dtype == torch.uint8 and
            ins.dtype == torch.int32 and W % 4 == 0 and n_classes <= 16384 and
            sem.data_ptr() % 4 == 0 and ins.data_ptr() % 16 == 0)


def instance_clear_stuff(semantic: torch.Tensor, instance: torch.Tensor,
                         is_stuff_class: torch.Tensor) -> torch.Tensor:
    """reference: InstanceClearStuffIDs (data/preprocessing/instance.py:46-93); in place."""
    sem = L.require_device_tensor(semantic, 'semantic')
    if not instance.is_cuda or not instance.is_contiguous():
        raise L.NmsaError('instance must be a contiguous device tensor (modified in place)')
    lut = _u8(is_stuff_class)
    L.check(L.lib().nmsa_instance_clear_stuff(
        L.ptr(sem), L.int_dtype_code(sem), L.ptr(instance), L.int_dtype_code(instance),
        L.ptr(lut), int(lut.numel()), int(sem.numel()), L.stream_ptr(sem.device)),
        'nmsa_instance_clear_stuff')
    return instance


def instance_targets(
    semantic: torch.Tensor,
    instance: torch.Tensor,
    n_classes: int,
    is_thing_class: Optional[torch.Tensor],
    is_stuff_class: Optional[torch.Tensor],
    sigma: int,
    normalized_offset: bool = True,
    max_instances: int = 1024,
) -> Dict[str, torch.Tensor]:
    """reference: InstanceTargetGenerator._preprocess (data/preprocessing/instance.py:157-286)
    for a whole batch [B,H,W]."""
    sem = L.require_device_tensor(semantic, 'semantic')
    ins = L.require_device_tensor(instance, 'instance')
    B, H, W = sem.shape
    dev = sem.device
    th = None if is_thing_class is None else _u8(is_thing_class)
    st = None if is_stuff_class is None else _u8(is_stuff_class)
    cap = ((int(max_instances) + 1023) // 1024) * 1024
    center = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    offset = torch.empty((B, 2, H, W), dtype=torch.float32 if normalized_offset else torch.int16,
                         device=dev)
    fg = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    cm = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    enc = torch.empty((B, cap), dtype=torch.int32, device=dev)
    skp = torch.empty((B, cap), dtype=torch.int32, device=dev)
    n_enc = torch.empty((B,), dtype=torch.int32, device=dev)
    n_skp = torch.empty((B,), dtype=torch.int32, device=dev)
    on_wire = _targets_on_wire(sem, ins, H, W, int(n_classes))
    ws, ws_bytes, ws_entry = _targets_workspace(B, int(n_classes), int(max_instances), dev, reusable=on_wire)
    clean, ws_entry[1] = ws_entry[1], 0
    status = torch.empty((1,), dtype=torch.int32, device=dev) if on_wire else \
        torch.zeros((1,), dtype=torch.int32, device=dev)
    L.check(L.lib().nmsa_instance_targets(
        L.ptr(sem), L.int_dtype_code(sem), L.ptr(ins), L.int_dtype_code(ins), L.ptr(th), L.ptr(st),
        B, int(n_classes), H, W, int(sigma), L.ptr(_gauss_lut(sigma, dev)),
        int(bool(normalized_offset)), int(max_instances),
        L.ptr(center), L.ptr(offset), L.ptr(fg), L.ptr(cm), L.ptr(enc), L.ptr(n_enc),
        L.ptr(skp), L.ptr(n_skp), L.ptr(status), L.ptr(ws), ws_bytes, int(clean), L.stream_ptr(dev)),
        'nmsa_instance_targets')
    ws_entry[1] = int(on_wire)
    return {'center': center, 'offset': offset, 'foreground': fg.view(torch.bool),
            'center_mask': cm.view(torch.bool), 'encoded_ids': enc, 'n_encoded': n_enc,
            'skipped_ids': skp, 'n_skipped': n_skp, 'status': status}


def panoptic_targets(
    semantic: torch.Tensor,
    instance: torch.Tensor,
    n_classes: int,
    is_thing_class: Optional[torch.Tensor],
    max_instances_per_category: int,
    void_label: int = 0,
    max_instances: int = 1024,
    max_segments: int = 2048,
) -> Dict[str, torch.Tensor]:
    """reference: naive_merge_semantic_and_instance_np (utils/panoptic_merge.py:43-107) as
    called by PanopticTargetGenerator (data/preprocessing/panoptic.py:48-85), per batch."""
    sem = L.require_device_tensor(semantic, 'semantic')
    ins = L.require_device_tensor(instance, 'instance')
    B, H, W = sem.shape
    dev = sem.device
    th = None if is_thing_class is None else _u8(is_thing_class)
    pan = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    ids_pan = torch.empty((B, int(max_segments)), dtype=torch.int64, device=dev)
    ids_ins = torch.empty((B, int(max_segments)), dtype=torch.int64, device=dev)
    n_ids = torch.empty((B,), dtype=torch.int32, device=dev)
    on_wire = _targets_on_wire(sem, ins, H, W, int(n_classes))
    ws, ws_bytes, ws_entry = _targets_workspace(B, int(n_classes), int(max_instances), dev, reusable=on_wire)
    clean, ws_entry[1] = ws_entry[1], 0
    status = torch.empty((1,), dtype=torch.int32, device=dev) if on_wire else \
        torch.zeros((1,), dtype=torch.int32, device=dev)
    L.check(L.lib().nmsa_panoptic_targets(
        L.ptr(sem), L.int_dtype_code(sem), L.ptr(ins), L.int_dtype_code(ins), L.ptr(th),
        B, int(n_classes), H, W, int(max_instances_per_category), int(void_label),
        int(max_instances), int(max_segments), L.ptr(pan), L.ptr(ids_pan), L.ptr(ids_ins),
        L.ptr(n_ids), L.ptr(status), L.ptr(ws), ws_bytes, int(clean), L.stream_ptr(dev)),
        'nmsa_panoptic_targets')
    ws_entry[1] = int(on_wire)
    return {'panoptic': pan, 'ids_pan': ids_pan, 'ids_ins': ids_ins, 'n_ids': n_ids,
            'status': status}


def dve_targets(
    panoptic: torch.Tensor,
    keys: torch.Tensor,
    n_keys: torch.Tensor,
    embeddings: Optional[torch.Tensor] = None,
    image_embedding: Optional[torch.Tensor] = None,
    diff_factor: float = 0.65,
) -> Dict[str, torch.Tensor]:
    """reference: DenseVisualEmbeddingTargetGenerator (dense_visual_embedding.py:22-93)."""
    pan = L.require_device_tensor(panoptic, 'panoptic')
    k = L.require_device_tensor(keys, 'keys')
    nk = L.require_device_tensor(n_keys, 'n_keys')
    assert pan.dtype == torch.int64 and k.dtype == torch.int64 and nk.dtype == torch.int32
    B, H, W = pan.shape
    K = int(k.shape[1])
    dev = pan.device
    idx = torch.empty((B, H, W), dtype=torch.int32, device=dev)
    lut = None
    D = 0
    emb = img = None
    if embeddings is not None:
        emb = L.require_device_tensor(embeddings, 'embeddings').float()
        img = L.require_device_tensor(image_embedding, 'image_embedding').float()
        D = int(emb.shape[2])
        lut = torch.empty((B, K, D), dtype=torch.float32, device=dev)
    L.check(L.lib().nmsa_dve_targets(
        L.ptr(pan), L.ptr(k), L.ptr(nk), L.ptr(emb), L.ptr(img), float(diff_factor),
        B, K, D, H, W, L.ptr(lut), L.ptr(idx), L.stream_ptr(dev)), 'nmsa_dve_targets')
    return {'indices': idx, 'lut': lut}


# ----------------------------------------------------------------------------- a5
def panoptic_merge(
    semantic: torch.Tensor,
    instance: torch.Tensor,
    thing_seg: torch.Tensor,
    is_thing_class: torch.Tensor,           # u8/bool [n_classes] (class VALUE domain)
    max_instances_per_category: int,
    void_label: int = 0,
) -> Dict[str, torch.Tensor]:
    """reference: deeplab_merge_batch (panoptic_merge.py:18-40,172-225)"""
    sem = L.require_device_tensor(semantic, 'semantic')
    ins = L.require_device_tensor(instance, 'instance')
    if sem.dtype == torch.bool:
        sem = sem.view(torch.uint8)
    B, H, W = sem.shape
    dev = sem.device
    thing = _u8(thing_seg)
    lut = _u8(is_thing_class)
    n_classes = int(lut.numel())
    votes = torch.empty((B, 256, n_classes), dtype=torch.int32, device=dev)
    pan_of_inst = torch.empty((B, 256), dtype=torch.int64, device=dev)
    pan = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    ids_pan = torch.empty((B, 256), dtype=torch.int64, device=dev)
    ids_ins = torch.empty((B, 256), dtype=torch.int64, device=dev)
    n_ids = torch.empty((B,), dtype=torch.int32, device=dev)
    L.check(L.lib().nmsa_panoptic_merge(
        L.ptr(sem), L.int_dtype_code(sem), L.ptr(ins), L.int_dtype_code(ins), L.ptr(thing),
        L.ptr(lut), B, n_classes, H, W, int(max_instances_per_category), int(void_label),
        L.ptr(votes), L.ptr(pan_of_inst), L.ptr(pan), L.ptr(ids_pan), L.ptr(ids_ins),
        L.ptr(n_ids), L.stream_ptr(dev)), 'nmsa_panoptic_merge')
    return {'panoptic': pan, 'ids_pan': ids_pan, 'ids_ins': ids_ins, 'n_ids': n_ids}


def panoptic_merge_wide(
    semantic: torch.Tensor,
    instance: torch.Tensor,
    thing_seg: torch.Tensor,
    is_thing_class: torch.Tensor,
    max_instances_per_category: int,
    void_label: int = 0,
    max_segments: int = 1024,
) -> Dict[str, torch.Tensor]:
    """deeplab_merge_batch for instance ids 0..65535 (ground-truth maps)."""
    sem = L.require_device_tensor(semantic, 'semantic')
    ins = L.require_device_tensor(instance, 'instance')
    B, H, W = sem.shape
    dev = sem.device
    thing = _u8(thing_seg)
    lut = _u8(is_thing_class)
    n_classes = int(lut.numel())
    cap = ((max_segments + 1023) // 1024) * 1024
    pan = torch.empty((B, H, W), dtype=torch.int64, device=dev)
    ids_pan = torch.empty((B, cap), dtype=torch.int64, device=dev)
    ids_ins = torch.empty((B, cap), dtype=torch.int64, device=dev)
    n_ids = torch.empty((B,), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws_bytes = L.lib().nmsa_panoptic_merge_wide_workspace_bytes(B, n_classes, max_segments)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    L.check(L.lib().nmsa_panoptic_merge_wide(
        L.ptr(sem), L.int_dtype_code(sem), L.ptr(ins), L.int_dtype_code(ins), L.ptr(thing),
        L.ptr(lut), B, n_classes, H, W, int(max_instances_per_category), int(void_label),
        int(max_segments), L.ptr(pan), L.ptr(ids_pan), L.ptr(ids_ins), L.ptr(n_ids), L.ptr(status),
        L.ptr(ws), ws_bytes, L.stream_ptr(dev)), 'nmsa_panoptic_merge_wide')
    return {'panoptic': pan, 'ids_pan': ids_pan, 'ids_ins': ids_ins, 'n_ids': n_ids,
            'status': status}


# ------------------------------------------------------------------------- next-1
def instance_orientation_sums(
    orientation: torch.Tensor,
    instance: torch.Tensor,
    mask: Optional[torch.Tensor] = None,
) -> Dict[str, torch.Tensor]:
    """reference: InstancePostprocessing._get_instance_orientation (instance.py:271-319)"""
    o = L.require_device_tensor(orientation, 'orientation')
    if o.dtype != torch.float32:
        o = o.float()
    ins = L.require_device_tensor(instance, 'instance')
    assert ins.dtype == torch.uint8
    B, two, H, W = o.shape
    dev = o.device
    sums = torch.empty((B, 256, 2), dtype=torch.float64, device=dev)
    count = torch.empty((B, 256), dtype=torch.int32, device=dev)
    L.check(L.lib().nmsa_instance_orientation(
        L.ptr(o), L.ptr(ins), L.ptr(_u8(mask)), B, H, W, L.ptr(sums), L.ptr(count),
        L.stream_ptr(dev)), 'nmsa_instance_orientation')
    return {'sums': sums, 'count': count}


def instance_orientation_sums_wide(
    orientation: torch.Tensor,
    instance: torch.Tensor,
    mask: Optional[torch.Tensor] = None,
    max_instances: int = 1024,
) -> Dict[str, torch.Tensor]:
    """`instance_orientation_sums` for ground-truth instance maps (ids 0..65535, any integer
    dtype): sums / counts by position in the ascending `ids` list."""
    o = L.require_device_tensor(orientation, 'orientation')
    if o.dtype != torch.float32:
        o = o.float()
    ins = L.require_device_tensor(instance, 'instance')
    B, two, H, W = o.shape
    dev = o.device
    cap = ((int(max_instances) + 1023) // 1024) * 1024
    ids = torch.empty((B, cap), dtype=torch.int32, device=dev)
    n_ids = torch.empty((B,), dtype=torch.int32, device=dev)
    sums = torch.empty((B, cap, 2), dtype=torch.float64, device=dev)
    count = torch.empty((B, cap), dtype=torch.int32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws, ws_bytes, _ = _targets_workspace(B, 1, int(max_instances), dev)
    L.check(L.lib().nmsa_instance_orientation_wide(
        L.ptr(o), L.ptr(ins), L.int_dtype_code(ins), L.ptr(_u8(mask)), B, H, W, int(max_instances),
        L.ptr(ids), L.ptr(n_ids), L.ptr(sums), L.ptr(count), L.ptr(status), L.ptr(ws), ws_bytes,
        L.stream_ptr(dev)), 'nmsa_instance_orientation_wide')
    return {'ids': ids, 'n_ids': n_ids, 'sums': sums, 'count': count, 'status': status}
