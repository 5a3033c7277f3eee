"""
Deterministic synthetic inputs of the shapes in SURVEY.md §8d.

Everything is generated with numpy (PCG64) in float64 using only + - * / and
one `exp` per heatmap pixel, then rounded once to float32, so that the same
seed gives bit-identical arrays in the build container and on the GPU box
(`input_digest` lets a test verify that before trusting a committed golden).

Layouts/dtypes follow what the reference's decoders/targets produce:
  semantic logits  [B,C,H,W] f32          (model/decoder: raw, pre-softmax)
  instance center  [B,1,H,W] f32 in [0,1] (sigmoid head; targets are Gaussians,
                                            data/preprocessing/instance.py:143-150)
  instance offset  [B,2,H,W] f32, (dy,dx) normalised by (H,W)
                                           (data/preprocessing/instance.py:252-256)
  orientation      [B,2,H,W] f32 unit biternion (cos, sin)
"""
import hashlib
from typing import Dict

import numpy as np


def _bilinear_up(coarse: np.ndarray, H: int, W: int) -> np.ndarray:
    """align_corners=False bilinear upsample of [..., h, w] -> [..., H, W] (f64)."""
    h, w = coarse.shape[-2:]
    ys = (np.arange(H, dtype=np.float64) + 0.5) * (h / H) - 0.5
    xs = (np.arange(W, dtype=np.float64) + 0.5) * (w / W) - 0.5
    ys = np.clip(ys, 0, h - 1)
    xs = np.clip(xs, 0, w - 1)
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    wy = (ys - y0)[:, None]
    wx = (xs - x0)[None, :]
    a = coarse[..., y0[:, None], x0[None, :]]
    b = coarse[..., y0[:, None], x1[None, :]]
    c = coarse[..., y1[:, None], x0[None, :]]
    d = coarse[..., y1[:, None], x1[None, :]]
    return (a * (1 - wx) + b * wx) * (1 - wy) + (c * (1 - wx) + d * wx) * wy


def make_panoptic_inputs(
    batch_size: int,
    n_classes: int = 40,
    height: int = 480,
    width: int = 640,
    n_centers: int = 24,
    seed: int = 0,
    quantize_offsets: bool = True,
    sigma: float = 8.0,
    offset_noise_px: float = 2.0,
    with_orientation: bool = False,
) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    B, Cn, H, W = batch_size, n_classes, height, width
    gh, gw = max(H // 32, 2), max(W // 32, 2)

    logits = np.empty((B, Cn, H, W), np.float32)
    center = np.empty((B, 1, H, W), np.float32)
    offset = np.empty((B, 2, H, W), np.float32)
    planted = np.empty((B, n_centers, 2), np.int32)
    orientation = np.empty((B, 2, H, W), np.float32) if with_orientation else None

    yy = np.arange(H, dtype=np.float64)[:, None]
    xx = np.arange(W, dtype=np.float64)[None, :]
    border = int(min(8, H // 4, W // 4))

    for b in range(B):
        coarse = rng.standard_normal((Cn, gh, gw))
        logits[b] = (4.0 * _bilinear_up(coarse, H, W)).astype(np.float32)

        cy = rng.integers(border, H - border, size=n_centers)
        cx = rng.integers(border, W - border, size=n_centers)
        planted[b, :, 0] = cy
        planted[b, :, 1] = cx

        heat = np.zeros((H, W), np.float64)
        best_d2 = np.full((H, W), np.inf)
        near_y = np.zeros((H, W), np.float64)
        near_x = np.zeros((H, W), np.float64)
        for k in range(n_centers):
            dy = cy[k] - yy
            dx = cx[k] - xx
            d2 = dy * dy + dx * dx
            heat = np.maximum(heat, np.exp(-d2 / (2.0 * sigma * sigma)))
            closer = d2 < best_d2
            best_d2 = np.where(closer, d2, best_d2)
            near_y = np.where(closer, dy, near_y)
            near_x = np.where(closer, dx, near_x)
        center[b, 0] = heat.astype(np.float32)

        oy = near_y + offset_noise_px * rng.standard_normal((H, W))
        ox = near_x + offset_noise_px * rng.standard_normal((H, W))
        if quantize_offsets:
            oy = np.round(oy * 2.0) / 2.0
            ox = np.round(ox * 2.0) / 2.0
        offset[b, 0] = (oy / H).astype(np.float32)
        offset[b, 1] = (ox / W).astype(np.float32)

        if with_orientation:
            # smooth angle field -> unit biternion (cos, sin) via rational
            # parametrisation (no trig: keeps generation platform-stable)
            t = _bilinear_up(rng.standard_normal((1, gh, gw)), H, W)[0] * 2.0
            den = 1.0 + t * t
            orientation[b, 0] = ((1.0 - t * t) / den).astype(np.float32)
            orientation[b, 1] = ((2.0 * t) / den).astype(np.float32)

    is_thing = tuple(bool(c >= Cn // 2) for c in range(Cn))
    out = {
        'semantic_logits': logits,
        'instance_center': center,
        'instance_offset': offset,
        'planted_centers': planted,
        'semantic_classes_is_thing': np.array(is_thing, dtype=bool),
    }
    if with_orientation:
        out['instance_orientation'] = orientation
    return out


def make_metric_inputs(
    pred_panoptic: np.ndarray,
    n_classes_with_void: int,
    seed: int = 0,
    shift_px: int = 3,
) -> Dict[str, np.ndarray]:
    """GT for the metric accumulators: target_pan = roll(pred_pan, shift_px)
    with a void band, target_sem uniform in [0, C]."""
    rng = np.random.default_rng(seed + 1000)
    target_pan = np.roll(pred_panoptic, shift=(shift_px, shift_px), axis=(-2, -1)).copy()
    target_pan[..., :shift_px, :] = 0          # void band
    target_sem = rng.integers(0, n_classes_with_void, size=pred_panoptic.shape,
                              dtype=np.int64).astype(np.uint8)
    return {'panoptic_target': target_pan.astype(np.int64),
            'semantic_target': target_sem}


def make_loss_inputs(
    batch_size: int,
    n_classes: int = 40,
    height: int = 480,
    width: int = 640,
    seed: int = 0,
    embedding_dim: int = 0,
    n_lut: int = 64,
) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed + 2000)
    B, Cn, H, W = batch_size, n_classes, height, width
    base = make_panoptic_inputs(B, Cn, H, W, seed=seed, with_orientation=True)
    out = {
        'semantic_logits': base['semantic_logits'],
        'semantic_target': rng.integers(0, Cn + 1, size=(B, H, W)).astype(np.uint8),
        'class_weights': (rng.random(Cn) + 0.5).astype(np.float32),
        # predictions: noisy versions of the targets
        'center_target': base['instance_center'][:, 0].copy(),
        'center_pred': np.clip(base['instance_center'][:, 0]
                               + 0.1 * rng.standard_normal((B, H, W)), 0, 1).astype(np.float32),
        'center_mask': rng.random((B, H, W)) < 0.7,
        'offset_target': base['instance_offset'],
        'offset_pred': (base['instance_offset']
                        + 0.01 * rng.standard_normal((B, 2, H, W))).astype(np.float32),
        'offset_mask': rng.random((B, H, W)) < 0.5,
        'orientation_target': base['instance_orientation'],
        'orientation_mask': rng.random((B, H, W)) < 0.3,
    }
    # unit-length noisy orientation prediction (normalised with sqrt only)
    op = base['instance_orientation'].astype(np.float64) + 0.3 * rng.standard_normal((B, 2, H, W))
    op = op / (np.sqrt((op * op).sum(axis=1, keepdims=True)) + 1e-7)
    out['orientation_pred'] = op.astype(np.float32)
    if embedding_dim:
        D = embedding_dim
        lut = rng.standard_normal((B, n_lut, D))
        lut = lut / np.sqrt((lut * lut).sum(axis=-1, keepdims=True))
        out['embedding_lut'] = lut.astype(np.float32)
        out['embedding_indices'] = rng.integers(0, n_lut + 1, size=(B, H, W)).astype(np.int32)
        out['embedding_pred'] = rng.standard_normal((B, D, H, W)).astype(np.float32)
    return out


def round_to_bf16(a: np.ndarray) -> np.ndarray:
    """float32 array rounded to the nearest bfloat16 (ties to even), returned as float32 —
    what a bf16 network output holds; integer arithmetic only, platform-stable"""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def make_embedding_inputs(batch_size: int, embedding_dim: int, height: int, width: int,
                          n_lut: int, seed: int = 0, block: int = 4, bf16: bool = False
                          ) -> Dict[str, np.ndarray]:
    """Dense visual-embedding loss inputs (SURVEY §8d, configs[4]): prediction [B,D,H,W],
    per-image LUT [B,L,D] of unit rows, index map [B,H,W] int32 in [0, L] (0 = no target) made
    of `block` x `block` segments with ~15 % of the pixels re-drawn individually, so that lanes
    of a wave see both shared and distinct LUT rows.  `bf16`: the prediction holds
    bf16-representable values (as float32)."""
    rng = np.random.default_rng(seed + 4000)
    B, D, H, W, L = batch_size, embedding_dim, height, width, n_lut
    lut = rng.standard_normal((B, L, D))
    lut = lut / np.sqrt((lut * lut).sum(axis=-1, keepdims=True))
    coarse = rng.integers(0, L + 1, size=(B, (H + block - 1) // block, (W + block - 1) // block))
    idx = np.kron(coarse, np.ones((block, block), np.int64))[:, :H, :W]
    redraw = rng.random((B, H, W)) < 0.15
    idx = np.where(redraw, rng.integers(0, L + 1, size=(B, H, W)), idx)
    # prediction: the target row plus noise for most pixels (cosine ~0.5), pure noise elsewhere
    tgt = np.take_along_axis(lut, np.clip(idx - 1, 0, L - 1).reshape(B, -1, 1), axis=1)
    tgt = tgt.reshape(B, H, W, D).transpose(0, 3, 1, 2)
    pred = (tgt * (rng.random((B, 1, H, W)) < 0.8)
            + rng.standard_normal((B, D, H, W)) / np.sqrt(D)) * 3.0
    pred = pred.astype(np.float32)
    if bf16:
        pred = round_to_bf16(pred)
    return {'embedding_pred': pred, 'embedding_lut': lut.astype(np.float32),
            'embedding_indices': idx.astype(np.int32)}


def make_training_case(batch_size=2, n_classes=9, height=64, width=96, seed=0,
                       embedding_dim=16, n_lut=9, scales=(2, 4)):
    """One training batch for the task helpers (numpy): main output + side outputs at `scales`
    (every s-th pixel of the main prediction, scaled by 0.9) and the targets of the matching
    resolutions under the '_down_<s>' batch keys.  -> (batch, preds); per-image LUTs have
    different row counts (not stackable, like the reference's batches)."""
    d = make_loss_inputs(batch_size, n_classes, height, width, seed=seed,
                         embedding_dim=embedding_dim, n_lut=n_lut)
    rng = np.random.default_rng(seed + 3000)
    rows = [int(r) for r in rng.integers(max(n_lut // 2, 1), n_lut + 1, size=batch_size)]
    idx = d['embedding_indices'].copy()
    for b, r in enumerate(rows):
        idx[b][idx[b] > r] = 0                           # indices beyond the image's rows: no target
    batch = {
        'semantic': d['semantic_target'], 'instance_center': d['center_target'],
        'instance_center_mask': d['center_mask'], 'instance_offset': d['offset_target'],
        'instance_foreground': d['offset_mask'], 'orientation': d['orientation_target'],
        'orientation_foreground': d['orientation_mask'],
        'dense_visual_embedding_indices': idx,
    }

    def down(a, s):
        return np.ascontiguousarray(a[..., ::s, ::s])
    for s in scales:
        batch[f'_down_{s}'] = {k: down(v, s) for k, v in batch.items() if isinstance(v, np.ndarray)}
    luts = [np.ascontiguousarray(d['embedding_lut'][b, :r]) for b, r in enumerate(rows)]
    batch['dense_visual_embedding_lut'] = luts
    for s in scales:
        batch[f'_down_{s}']['dense_visual_embedding_lut'] = luts
    main = (d['center_pred'][:, None], d['offset_pred'], d['orientation_pred'])
    f32 = np.float32
    preds = {
        'semantic_output': d['semantic_logits'],
        'semantic_side_outputs': tuple((down(d['semantic_logits'], s) * f32(0.9)) for s in scales),
        'instance_output': main,
        'instance_side_outputs': tuple(tuple(down(x, s) * f32(0.9) for x in main) for s in scales),
        'dense_visual_embedding_output': d['embedding_pred'],
        'dense_visual_embedding_side_outputs': tuple(down(d['embedding_pred'], s) * f32(0.9)
                                                     for s in scales),
    }
    return batch, preds, d['class_weights']


def make_predictions_from_targets(semantic, center, offset, orientation, n_classes, seed=0):
    """Network-like outputs that mostly agree with the ground truth (so that validation metrics
    are neither 0 nor 1): logits = 3 * onehot(label - 1) + N(0, 1) (void pixels: pure noise),
    center / offset / orientation = target + noise.  numpy only, float32 results."""
    rng = np.random.default_rng(seed + 5000)
    B, H, W = semantic.shape
    logits = rng.standard_normal((B, n_classes, H, W))
    lab = semantic.astype(np.int64) - 1
    onehot = (np.arange(n_classes)[None, :, None, None] == lab[:, None]).astype(np.float64)
    logits = logits + 3.0 * onehot
    c = np.clip(center + 0.03 * rng.standard_normal(center.shape), 0.0, 1.0)
    o = offset + 0.004 * rng.standard_normal(offset.shape)
    q = orientation.astype(np.float64) + 0.25 * rng.standard_normal(orientation.shape)
    q = q / (np.sqrt((q * q).sum(axis=1, keepdims=True)) + 1e-7)
    return (logits.astype(np.float32), c.astype(np.float32)[:, None], o.astype(np.float32),
            q.astype(np.float32))


def make_label_maps(batch_size, n_classes=41, height=480, width=640, n_instances=30, seed=0,
                    max_id=65535, mixed_fraction=0.3, max_radius=None):
    """Ground-truth style label maps for the target generators.

    semantic u8 [B,H,W] in [0, n_classes) (0 = void): blocky regions; instance int32 [B,H,W]:
    `n_instances` random ellipses with sparse ids in [1, max_id] (uint16 range, as the datasets
    store them), later ones painted over earlier ones.  A `mixed_fraction` of the instances keeps
    the underlying (mixed) semantic labels, the others get one thing class painted in.
    is_thing = class index >= n_classes // 2 (void and the lower half are stuff)."""
    rng = np.random.default_rng(seed)
    is_thing = np.arange(n_classes) >= max(1, n_classes // 2)
    thing_ids = np.where(is_thing)[0]
    sem = np.empty((batch_size, height, width), np.uint8)
    ins = np.zeros((batch_size, height, width), np.int32)
    yy, xx = np.mgrid[0:height, 0:width]
    for b in range(batch_size):
        coarse = rng.integers(0, n_classes, ((height + 31) // 32, (width + 31) // 32))
        sem[b] = np.kron(coarse, np.ones((32, 32), np.int64))[:height, :width]
        ids = rng.choice(np.arange(1, max_id + 1), size=n_instances, replace=False)
        for k, iid in enumerate(ids):
            cy, cx = rng.integers(0, height), rng.integers(0, width)
            ry = rng.integers(3, max_radius or max(4, height // 5))
            rx = rng.integers(3, max_radius or max(4, width // 5))
            m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
            ins[b][m] = iid
            if rng.random() >= mixed_fraction:
                sem[b][m] = rng.choice(thing_ids)
    return {'semantic': sem, 'instance': ins, 'semantic_classes_is_thing': is_thing}


def input_digest(*arrays: np.ndarray) -> str:
    h = hashlib.sha256()
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def make_panoptic_inputs_torch(batch_size, n_classes=40, height=480, width=640,
                               n_centers=24, seed=0, device='cuda', sigma=8.0,
                               offset_noise_px=2.0, quantize_offsets=True,
                               logits_dtype=None, chunk=8):
    """Same recipe as `make_panoptic_inputs`, generated on `device` with torch
    (for bench.py / full-size property tests: fast, not bit-reproducible
    across devices — parity there is checked against the oracle on the very
    tensors that were generated)."""
    import torch
    B, Cn, H, W = batch_size, n_classes, height, width
    g = torch.Generator(device=device).manual_seed(seed)
    gh, gw = max(H // 32, 2), max(W // 32, 2)
    coarse = torch.randn((B, Cn, gh, gw), device=device, generator=g)
    logits = 4.0 * torch.nn.functional.interpolate(coarse, size=(H, W), mode='bilinear',
                                                   align_corners=False)
    if logits_dtype is not None:
        logits = logits.to(logits_dtype)
    border = int(min(8, H // 4, W // 4))
    cy = torch.randint(border, H - border, (B, n_centers), device=device, generator=g)
    cx = torch.randint(border, W - border, (B, n_centers), device=device, generator=g)
    yy = torch.arange(H, device=device, dtype=torch.float32).view(1, 1, H, 1)
    xx = torch.arange(W, device=device, dtype=torch.float32).view(1, 1, 1, W)
    center = torch.empty((B, 1, H, W), device=device)
    offset = torch.empty((B, 2, H, W), device=device)
    for b0 in range(0, B, chunk):
        b1 = min(b0 + chunk, B)
        dy = cy[b0:b1].float().view(-1, n_centers, 1, 1) - yy
        dx = cx[b0:b1].float().view(-1, n_centers, 1, 1) - xx
        d2 = dy * dy + dx * dx
        center[b0:b1, 0] = torch.exp(-d2 / (2.0 * sigma * sigma)).amax(dim=1)
        near = d2.argmin(dim=1, keepdim=True)
        oy = torch.gather(dy.expand_as(d2), 1, near)[:, 0]
        ox = torch.gather(dx.expand_as(d2), 1, near)[:, 0]
        oy = oy + offset_noise_px * torch.randn(oy.shape, device=device, generator=g)
        ox = ox + offset_noise_px * torch.randn(ox.shape, device=device, generator=g)
        if quantize_offsets:
            oy = torch.round(oy * 2.0) / 2.0
            ox = torch.round(ox * 2.0) / 2.0
        offset[b0:b1, 0] = oy / H
        offset[b0:b1, 1] = ox / W
    is_thing = torch.arange(Cn, device=device) >= Cn // 2
    return {'semantic_logits': logits.contiguous(), 'instance_center': center,
            'instance_offset': offset, 'semantic_classes_is_thing': is_thing}
