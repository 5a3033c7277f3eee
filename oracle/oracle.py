"""
TEST INFRASTRUCTURE — numpy front-end of the C oracle (oracle/nmsa_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (the timed
CPU baseline and, behind the timed regions, the checks of GPU results against it:
`gpu_matches_oracle_bit_exact`, `oracle_check` of the cosine legs) import this
module — as the checker, never as the thing measured; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libnmsa_oracle.so')


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, 'nmsa_oracle.c')
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libnmsa_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_miou_compute.restype = C.c_float
    return _lib


def _p(a, t):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(t))


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class OracleError(RuntimeError):
    pass


def _chk(rc, what):
    if rc != 0:
        raise OracleError(f'{what}: oracle error {rc}')


# -- a1 -----------------------------------------------------------------------
def semantic_argmax(logits):
    logits = _c(logits, np.float32)
    B, Cn, H, W = logits.shape
    idx = np.empty((B, H, W), np.int64)
    score = np.empty((B, H, W), np.float32)
    _chk(lib().orc_semantic_argmax(_p(logits, C.c_float), B, Cn, H, W,
                                   _p(idx, C.c_int64), _p(score, C.c_float)),
         'semantic_argmax')
    return idx, score


# -- a2 -----------------------------------------------------------------------
def center_nms_topk(center, fg=None, threshold=0.1, ksize=3, topk=64,
                    apply_fg=False, max_centers=256):
    center = _c(center, np.float32)
    if center.ndim == 4:
        center = center[:, 0]
    center = np.ascontiguousarray(center)
    B, H, W = center.shape
    fg_ = None if fg is None else _c(fg, np.uint8)
    cyx = np.zeros((B, max_centers, 2), np.int32)
    n = np.zeros((B,), np.int32)
    scores = np.zeros((B, max_centers), np.float32)
    mask = np.zeros((B, H, W), np.uint8)
    rc = lib().orc_center_nms_topk(
        _p(center, C.c_float), _p(fg_, C.c_uint8), B, H, W,
        C.c_float(threshold), ksize, topk, int(bool(apply_fg)), max_centers,
        _p(cyx, C.c_int32), _p(n, C.c_int32), _p(scores, C.c_float),
        _p(mask, C.c_uint8))
    _chk(rc, 'center_nms_topk')
    return cyx, n, scores, mask.astype(bool)


# -- a3 -----------------------------------------------------------------------
def group_offsets(offset, fg, centers_yx, n_centers, scale_y=1.0, scale_x=1.0,
                  dist_thr=None):
    offset = _c(offset, np.float32)
    B, two, H, W = offset.shape
    assert two == 2
    fg = _c(fg, np.uint8)
    centers_yx = _c(centers_yx, np.int32)
    n_centers = _c(n_centers, np.int32)
    max_centers = centers_yx.shape[1]
    inst = np.zeros((B, H, W), np.uint8)
    area = np.zeros((B, 256), np.int32)
    rc = lib().orc_group_offsets(
        _p(offset, C.c_float), _p(fg, C.c_uint8), _p(centers_yx, C.c_int32),
        _p(n_centers, C.c_int32), B, H, W, max_centers,
        C.c_float(scale_y), C.c_float(scale_x),
        0 if dist_thr is None else 1,
        C.c_float(0.0 if dist_thr is None else dist_thr),
        _p(inst, C.c_uint8), _p(area, C.c_int32))
    _chk(rc, 'group_offsets')
    return inst, area


# -- a5 -----------------------------------------------------------------------
def _merge(fn, sem, ins, thing_seg, max_inst, thing_ids, void_label, cap=1024):
    sem = _c(sem, np.int64)
    ins = _c(ins, np.int64)
    B, H, W = sem.shape
    thing_ids = _c(list(thing_ids), np.int64)
    pan = np.empty((B, H, W), np.int64)
    id_pan = np.zeros((B, cap), np.int64)
    id_ins = np.zeros((B, cap), np.int64)
    n_ids = np.zeros((B,), np.int32)
    if thing_seg is not None:
        thing_seg = _c(thing_seg, np.uint8)
        rc = fn(_p(sem, C.c_int64), _p(ins, C.c_int64), _p(thing_seg, C.c_uint8),
                B, H, W, C.c_int64(max_inst), _p(thing_ids, C.c_int64),
                len(thing_ids), C.c_int64(void_label), _p(pan, C.c_int64), cap,
                _p(id_pan, C.c_int64), _p(id_ins, C.c_int64), _p(n_ids, C.c_int32))
    else:
        rc = fn(_p(sem, C.c_int64), _p(ins, C.c_int64),
                B, H, W, C.c_int64(max_inst), _p(thing_ids, C.c_int64),
                len(thing_ids), C.c_int64(void_label), _p(pan, C.c_int64), cap,
                _p(id_pan, C.c_int64), _p(id_ins, C.c_int64), _p(n_ids, C.c_int32))
    _chk(rc, 'merge')
    dicts = [{int(id_pan[b, i]): int(id_ins[b, i]) for i in range(n_ids[b])}
             for b in range(B)]
    return pan, dicts


def deeplab_merge(sem, ins, thing_seg, max_inst, thing_ids, void_label=0, cap=1024):
    return _merge(lib().orc_deeplab_merge, sem, ins, thing_seg, max_inst,
                  thing_ids, void_label, cap=cap)


def naive_merge(sem, ins, max_inst, thing_ids, void_label=0, cap=1024):
    return _merge(lib().orc_naive_merge, sem, ins, None, max_inst,
                  thing_ids, void_label, cap=cap)


# -- orientation ----------------------------------------------------------------
def instance_orientation(orientation, inst, mask=None):
    orientation = _c(orientation, np.float32)
    inst = _c(inst, np.uint8)
    B, H, W = inst.shape
    mask_ = None if mask is None else _c(mask, np.uint8)
    angle = np.zeros((B, 256), np.float32)
    present = np.zeros((B, 256), np.uint8)
    _chk(lib().orc_instance_orientation(
        _p(orientation, C.c_float), _p(inst, C.c_uint8), _p(mask_, C.c_uint8),
        B, H, W, _p(angle, C.c_float), _p(present, C.c_uint8)), 'orientation')
    return [{i: float(angle[b, i]) for i in range(256) if present[b, i]}
            for b in range(B)]


# -- a11 ----------------------------------------------------------------------
def confmat_update(preds, target, n_classes, confmat=None):
    preds = _c(preds, np.int64).reshape(-1)
    target = _c(target, np.int64).reshape(-1)
    if confmat is None:
        confmat = np.zeros((n_classes, n_classes), np.int64)
    _chk(lib().orc_confmat_update(_p(preds, C.c_int64), _p(target, C.c_int64),
                                  C.c_int64(preds.size), n_classes,
                                  _p(confmat, C.c_int64)), 'confmat_update')
    return confmat


def miou_compute(confmat, ignore_first_class=False):
    confmat = _c(confmat, np.int64)
    n = confmat.shape[0]
    ious = np.empty((n,), np.float32)
    miou = lib().orc_miou_compute(_p(confmat, C.c_int64), n,
                                  int(bool(ignore_first_class)), _p(ious, C.c_float))
    return float(miou), ious


# -- a12 ----------------------------------------------------------------------
def pq_compare_and_accumulate(pred, target, num_categories, ignored_label,
                              max_instances_per_category, offset,
                              void_segment_id=None, state=None, match_cap=4096):
    """pred/target: one image [H,W] int64.  Returns (iou, tp, fn, fp, matches)."""
    pred = _c(pred, np.int64).reshape(-1)
    target = _c(target, np.int64).reshape(-1)
    if void_segment_id is None:
        void_segment_id = ignored_label * max_instances_per_category
    # the per-image function starts from zeros (pq.py:76-79); `PanopticQuality.update` then adds
    # the image's vectors to the states (pq.py:291-296) — same rounding order here
    img = [np.zeros((num_categories,), np.float64) for _ in range(4)]
    matches = np.zeros((match_cap, 2), np.int64)
    nm = C.c_int32(0)
    _chk(lib().orc_pq_compare_and_accumulate(
        _p(pred, C.c_int64), _p(target, C.c_int64), C.c_int64(pred.size),
        num_categories, C.c_int64(ignored_label),
        C.c_int64(max_instances_per_category), C.c_int64(offset),
        C.c_int64(void_segment_id),
        _p(img[0], C.c_double), _p(img[1], C.c_double),
        _p(img[2], C.c_double), _p(img[3], C.c_double),
        match_cap, _p(matches, C.c_int64), C.byref(nm)), 'pq')
    m = [(int(a), int(b)) for a, b in matches[:nm.value]]
    if state is None:
        state = img
    else:
        state = list(state)
        for i in range(4):
            state[i] = state[i] + img[i]
    return state[0], state[1], state[2], state[3], m


# -- losses ---------------------------------------------------------------------
def loss_ce(logits, target, weights=None, label_smoothing=0.0, want_grad=False):
    logits = _c(logits, np.float32)
    target = _c(target, np.uint8)
    B, Cn, H, W = logits.shape
    w = None if weights is None else _c(weights, np.float32)
    s = C.c_double(0)
    n = C.c_int64(0)
    wd = C.c_double(0)
    grad = np.empty_like(logits) if want_grad else None
    _chk(lib().orc_loss_ce(_p(logits, C.c_float), _p(target, C.c_uint8),
                           _p(w, C.c_float), B, Cn, H, W,
                           C.c_float(label_smoothing), C.byref(s), C.byref(n),
                           C.byref(wd), _p(grad, C.c_float)), 'loss_ce')
    return s.value, n.value, wd.value, grad


def loss_masked_elementwise(pred, target, mask, kind, want_grad=False):
    """kind: 'mse' | 'l1'.  pred/target [B,H,W] or [B,C,H,W]; mask [B,H,W]|None."""
    pred = _c(pred, np.float32)
    target = _c(target, np.float32)
    if pred.ndim == 3:
        B, H, W = pred.shape
        Cn = 1
    else:
        B, Cn, H, W = pred.shape
    m = None if mask is None else _c(mask, np.uint8)
    s = C.c_double(0)
    n = C.c_int64(0)
    grad = np.empty_like(pred) if want_grad else None
    _chk(lib().orc_loss_masked_elementwise(
        _p(pred, C.c_float), _p(target, C.c_float), _p(m, C.c_uint8),
        B, Cn, H, W, 0 if kind == 'mse' else 1, C.byref(s), C.byref(n),
        _p(grad, C.c_float)), 'loss_elementwise')
    return s.value, n.value, grad


def loss_vonmises(pred, target, mask, kappa=1.0, want_grad=False):
    pred = _c(pred, np.float32)
    target = _c(target, np.float32)
    B, two, H, W = pred.shape
    m = None if mask is None else _c(mask, np.uint8)
    s = C.c_double(0)
    n = C.c_int64(0)
    grad = np.empty_like(pred) if want_grad else None
    _chk(lib().orc_loss_vonmises(_p(pred, C.c_float), _p(target, C.c_float),
                                 _p(m, C.c_uint8), B, H, W, C.c_float(kappa),
                                 C.byref(s), C.byref(n), _p(grad, C.c_float)),
         'loss_vonmises')
    return s.value, n.value, grad


def loss_cosine_embedding(pred, indices, lut, want_grad=False):
    pred = _c(pred, np.float32)
    indices = _c(indices, np.int32)
    lut = _c(lut, np.float32)
    B, D, H, W = pred.shape
    L = lut.shape[1]
    s = C.c_double(0)
    n = C.c_int64(0)
    grad = np.empty_like(pred) if want_grad else None
    _chk(lib().orc_loss_cosine_embedding(
        _p(pred, C.c_float), _p(indices, C.c_int32), _p(lut, C.c_float),
        B, D, H, W, L, C.byref(s), C.byref(n), _p(grad, C.c_float)),
        'loss_cos_emb')
    return s.value, n.value, grad


def loss_elementwise_form(pred, target, kind, reduction, upstream=None):
    """MSELoss / L1Loss for every reduction and rank (reference loss/mse.py:21-41, l1.py:21-41):
    per-element f = (pred - target)^2 | |pred - target|; 'sum': 2-D / 4-D inputs are averaged over
    axis 1 first (mse.py:30-34), n = elements left; 'mean': mean of f, n = 1; 'none': f itself,
    n = pred.size.  -> (loss, n, d(sum(loss * upstream)) / d pred); plain numpy in fp64."""
    x = np.asarray(pred, np.float64)
    d = x - np.asarray(target, np.float64)
    f = d * d if kind == 'mse' else np.abs(d)
    df = 2.0 * d if kind == 'mse' else np.sign(d)
    if reduction == 'none':
        return f, x.size, df * (1.0 if upstream is None else np.asarray(upstream, np.float64))
    if reduction == 'mean':
        return f.mean(), 1, df / x.size
    if x.ndim in (2, 4):
        return f.mean(axis=1).sum(), x.size // x.shape[1], df / x.shape[1]
    return f.sum(), x.size, df


def loss_cosine_rows(input_, target, labels, reduction, upstream=None):
    """CosineEmbeddingLoss on [N, D] rows with labels (reference loss/cos_emb.py:21-56: the wrapped
    torch.nn.CosineEmbeddingLoss(reduction='none'), margin 0): cos = x.y / sqrt((|x|^2 + 1e-12)
    (|y|^2 + 1e-12)); +1: 1 - cos; -1: max(0, cos); labels None = all +1.
    -> (loss, n, gradient w.r.t. input_ of sum(loss * upstream)); plain numpy in fp64."""
    x = np.asarray(input_, np.float64)
    y = np.asarray(target, np.float64)
    lab = np.ones(x.shape[0]) if labels is None else np.asarray(labels, np.float64)
    m1 = (x * x).sum(1) + 1e-12
    m2 = (y * y).sum(1) + 1e-12
    den = np.sqrt(m1 * m2)
    c = (x * y).sum(1) / den
    rows = np.where(lab == 1, 1.0 - c, 0.0) + np.where(lab == -1, np.maximum(c, 0.0), 0.0)
    dc = y / den[:, None] - (c / m1)[:, None] * x
    sgn = np.where(lab == 1, -1.0, np.where((lab == -1) & (c >= 0), 1.0, 0.0))
    if reduction == 'none':
        up = np.ones_like(rows) if upstream is None else np.asarray(upstream, np.float64)
        return rows, x.size, (sgn * up)[:, None] * dc
    if reduction == 'mean':
        return rows.mean(), 1, (sgn / rows.size)[:, None] * dc
    return rows.sum(), rows.size, sgn[:, None] * dc


# -- f2: crop + resize (dense_base.py:15-58) -------------------------------------
def _crop_args(shape, crop):
    Hs, Ws = shape[-2:]
    (ys, xs) = crop if crop is not None else (slice(0, Hs), slice(0, Ws))
    y0, y1, _ = ys.indices(Hs)
    x0, x1, _ = xs.indices(Ws)
    return Hs, Ws, y0, x0, y1 - y0, x1 - x0


def resize_nearest(maps, size, crop=None):
    """maps [..., Hs, Ws] (uint8/bool/int16/int32/int64/float32) -> [..., Ho, Wo]."""
    a = np.ascontiguousarray(maps)
    code = {np.dtype(np.uint8): 1, np.dtype(np.bool_): 1, np.dtype(np.int16): 2,
            np.dtype(np.int32): 4, np.dtype(np.int64): 8, np.dtype(np.float32): -4}[a.dtype]
    Hs, Ws, y0, x0, h, w = _crop_args(a.shape, crop)
    Ho, Wo = int(size[0]), int(size[1])
    planes = int(np.prod(a.shape[:-2], dtype=np.int64))
    out = np.empty(a.shape[:-2] + (Ho, Wo), a.dtype)
    _chk(lib().orc_resize_nearest(a.ctypes.data_as(C.c_void_p), code, planes, Hs, Ws,
                                  y0, x0, h, w, Ho, Wo, out.ctypes.data_as(C.c_void_p)),
         'resize_nearest')
    return out


def resize_bilinear(maps, size, crop=None):
    """float32 [..., Hs, Ws] -> [..., Ho, Wo], align_corners=False."""
    a = _c(maps, np.float32)
    Hs, Ws, y0, x0, h, w = _crop_args(a.shape, crop)
    Ho, Wo = int(size[0]), int(size[1])
    planes = int(np.prod(a.shape[:-2], dtype=np.int64))
    out = np.empty(a.shape[:-2] + (Ho, Wo), np.float32)
    _chk(lib().orc_resize_bilinear(_p(a, C.c_float), planes, Hs, Ws, y0, x0, h, w, Ho, Wo,
                                   _p(out, C.c_float)), 'resize_bilinear')
    return out


# -- f3: compute_scores (panoptic.py:171-239) -------------------------------------
def panoptic_scores(logits, pan_sem, pan, ids, inst_score_by_id):
    """ids: list (per image) of dict pan_id -> ins_id in insertion order."""
    logits = _c(logits, np.float32)
    B, Cn, H, W = logits.shape
    pan_sem = _c(pan_sem, np.int64)
    pan = _c(pan, np.int64)
    cap = max(1, max(len(d) for d in ids))
    ids_pan = np.zeros((B, cap), np.int64)
    ids_ins = np.zeros((B, cap), np.int64)
    n_ids = np.zeros((B,), np.int32)
    for b, d in enumerate(ids):
        n_ids[b] = len(d)
        for j, (k, v) in enumerate(d.items()):
            ids_pan[b, j], ids_ins[b, j] = k, v
    tab = _c(inst_score_by_id, np.float32)
    assert tab.shape == (B, 256)
    sem = np.empty((B, H, W), np.float32)
    ins = np.empty((B, H, W), np.float32)
    pns = np.empty((B, H, W), np.float32)
    mean = np.empty((B, 256), np.float32)
    _chk(lib().orc_panoptic_scores(
        _p(logits, C.c_float), _p(pan_sem, C.c_int64), _p(pan, C.c_int64),
        _p(ids_pan, C.c_int64), _p(ids_ins, C.c_int64), _p(n_ids, C.c_int32), cap,
        _p(tab, C.c_float), B, Cn, H, W, _p(sem, C.c_float), _p(ins, C.c_float),
        _p(pns, C.c_float), _p(mean, C.c_float)), 'panoptic_scores')
    return sem, ins, pns, mean


# -- f4: target generation ----------------------------------------------------------
def instance_targets(sem, ins, n_classes, is_thing=None, is_stuff=None, sigma=8,
                     normalized=True, cap=4096):
    """sem u8 [B,H,W], ins int32 [B,H,W] -> dict of batched targets (offset [B,2,H,W])."""
    sem = _c(sem, np.uint8)
    ins = _c(ins, np.int32)
    B, H, W = sem.shape
    th = None if is_thing is None else _c(is_thing, np.uint8)
    st = None if is_stuff is None else _c(is_stuff, np.uint8)
    center = np.empty((B, H, W), np.float32)
    off_f = np.empty((B, 2, H, W), np.float32)
    off_i = np.empty((B, 2, H, W), np.int16)
    fg = np.empty((B, H, W), np.uint8)
    cm = np.empty((B, H, W), np.uint8)
    enc, skp = [], []
    for b in range(B):
        e = np.zeros((cap,), np.int32)
        s = np.zeros((cap,), np.int32)
        ne, ns = C.c_int32(0), C.c_int32(0)
        _chk(lib().orc_instance_targets(
            _p(sem[b], C.c_uint8), _p(ins[b], C.c_int32), H, W, int(n_classes),
            _p(th, C.c_uint8), _p(st, C.c_uint8), int(sigma), int(bool(normalized)),
            _p(center[b], C.c_float), _p(off_f[b], C.c_float), _p(off_i[b], C.c_int16),
            _p(fg[b], C.c_uint8), _p(cm[b], C.c_uint8),
            _p(e, C.c_int32), C.byref(ne), _p(s, C.c_int32), C.byref(ns), cap), 'instance_targets')
        enc.append(e[:ne.value].tolist())
        skp.append(s[:ns.value].tolist())
    return dict(center=center, offset=off_f if normalized else off_i, foreground=fg.astype(bool),
                center_mask=cm.astype(bool), encoded=enc, skipped=skp)


def dve_indices(pan, keys):
    """pan i64 [B,H,W], keys: list (per image) of panoptic ids -> int32 indices [B,H,W]."""
    pan = _c(pan, np.int64)
    out = np.empty(pan.shape, np.int32)
    for b in range(pan.shape[0]):
        k = _c(list(keys[b]), np.int64)
        _chk(lib().orc_dve_indices(_p(pan[b], C.c_int64), pan[b].size, _p(k, C.c_int64), len(k),
                                   _p(out[b], C.c_int32)), 'dve_indices')
    return out
