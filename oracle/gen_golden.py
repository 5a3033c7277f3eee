"""
TEST INFRASTRUCTURE — golden-vector generator.  Runs ONLY in the build
container (needs the read-only reference mount); its outputs (small .npz data
files: inputs + the reference's outputs) are committed under tests/golden/.

    python oracle/gen_golden.py            # (re)generate everything
    python oracle/gen_golden.py norm       # only check the norm formula

Every case calls the reference's own, unmodified Python (loaded by
oracle/ref_loader.py) on seeded inputs.
"""
import json
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle.ref_loader import load_reference          # noqa: E402
from nicr_mt_scene_analysis_amd.testing import synthetic as syn   # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                      'tests', 'golden')
os.makedirs(GOLDEN, exist_ok=True)
torch.set_num_threads(8)


def save(name, **arrays):
    path = os.path.join(GOLDEN, name + '.npz')
    np.savez_compressed(path, **arrays)
    print(f'  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)')


def jdump(obj) -> np.ndarray:
    return np.frombuffer(json.dumps(obj).encode(), dtype=np.uint8)


def make_batch(ref, B, H, W, extra=None):
    batch = {
        'rgb_fullres': torch.zeros((B, 3, H, W)),
        ref.APPLIED_PREPROCESSING_KEY: [[{
            'type': 'Resize',
            'valid_region_slice_y': slice(0, H),
            'valid_region_slice_x': slice(0, W),
        }]] * B,
    }
    if extra:
        batch.update(extra)
    return batch


def meta_to_arrays(meta_list, cap=320):
    """list[dict[id -> {center_yx, area, score}]] -> dense arrays."""
    B = len(meta_list)
    n = np.zeros((B,), np.int32)
    cyx = np.zeros((B, cap, 2), np.int32)
    area = np.zeros((B, cap), np.int64)
    score = np.zeros((B, cap), np.float32)
    for b, meta in enumerate(meta_list):
        n[b] = len(meta)
        for i, m in meta.items():
            cyx[b, i - 1] = m['center_yx']
            area[b, i - 1] = m['area']
            score[b, i - 1] = m['score']
    return n, cyx, area, score


def ids_to_arrays(dicts, cap=320):
    B = len(dicts)
    n = np.zeros((B,), np.int32)
    k = np.zeros((B, cap), np.int64)
    v = np.zeros((B, cap), np.int64)
    for b, d in enumerate(dicts):
        n[b] = len(d)
        for i, (pk, pv) in enumerate(d.items()):    # insertion order matters
            k[b, i] = pk
            v[b, i] = pv
    return n, k, v


# ---------------------------------------------------------------------------
def check_norm_formula(ref):
    """torch.norm(int32 centers - f32 loc, dim=-1) == sqrtf(fmaf(dx,dx,dy*dy))."""
    g = torch.Generator().manual_seed(0)
    n, P = 64, 200000
    centers = torch.stack([torch.randint(0, 480, (n,), generator=g),
                           torch.randint(0, 640, (n,), generator=g)], 1).int()
    loc = torch.stack([torch.rand(P, generator=g) * 480,
                       torch.rand(P, generator=g) * 640], 1).float()
    d = centers.unsqueeze(1) - loc.unsqueeze(0)
    r = torch.norm(d, dim=-1).numpy()
    dy = d[..., 0].numpy()
    dx = d[..., 1].numpy()
    emu = np.sqrt((dx.astype(np.float64) ** 2
                   + (dy * dy).astype(np.float64)).astype(np.float32))
    mism = int((emu != r).sum())
    print(f'norm formula sqrt(fma(dx,dx,dy*dy)): {mism} mismatches / {r.size}')
    assert mism == 0


# ---------------------------------------------------------------------------
def run_panoptic(ref, inp, heatmap_kwargs=None, with_orientation=False):
    is_thing = tuple(bool(x) for x in inp['semantic_classes_is_thing'])
    sem_post = ref.post_semantic.SemanticPostprocessing()
    ins_post = ref.post_instance.InstancePostprocessing(**(heatmap_kwargs or {}))
    post = ref.post_panoptic.PanopticPostprocessing(
        semantic_postprocessing=sem_post, instance_postprocessing=ins_post,
        semantic_classes_is_thing=is_thing,
        semantic_class_has_orientation=is_thing)
    logits = torch.from_numpy(inp['semantic_logits'])
    center = torch.from_numpy(inp['instance_center'])
    offset = torch.from_numpy(inp['instance_offset'])
    B, _, H, W = logits.shape
    i_out = (center, offset)
    if with_orientation:
        i_out = i_out + (torch.from_numpy(inp['instance_orientation']),)
    data = ((logits, i_out), (None, None))
    batch = make_batch(ref, B, H, W)
    return post.postprocess(data, batch, is_training=False)


def panoptic_outputs(r, with_orientation=False, score_stride=1):
    n, cyx, area, score = meta_to_arrays(r['panoptic_segmentation_deeplab_instance_meta'])
    idn, idk, idv = ids_to_arrays(r['panoptic_segmentation_deeplab_ids'])
    out = dict(
        semantic_idx=r['semantic_segmentation_idx'].numpy().astype(np.uint8),
        semantic_score=r['semantic_segmentation_score'].numpy()[:, ::score_stride, ::score_stride],
        semantic_score_stride=np.int32(score_stride),
        foreground=r['panoptic_foreground_mask'].numpy(),
        instance=r['panoptic_segmentation_deeplab_instance_idx'].numpy(),
        panoptic=r['panoptic_segmentation_deeplab'].numpy(),
        panoptic_semantic=r['panoptic_segmentation_deeplab_semantic_idx'].numpy().astype(np.uint8),
        meta_n=n, meta_center_yx=cyx, meta_area=area, meta_score=score,
        ids_n=idn, ids_pan=idk, ids_ins=idv,
    )
    if with_orientation:
        ori = r['orientations_panoptic_segmentation_deeplab_instance']
        arr = np.full((len(ori), 256), np.nan, np.float32)
        for b, d in enumerate(ori):
            for k, v in d.items():
                arr[b, k] = v
        out['orientation'] = arr
    return out


def gen_panoptic(ref):
    print('panoptic pipeline (reference PanopticPostprocessing.postprocess)')
    # small: inputs stored
    inp = syn.make_panoptic_inputs(2, n_classes=8, height=96, width=128,
                                   n_centers=6, seed=3, with_orientation=True)
    r = run_panoptic(ref, inp, with_orientation=True)
    save('panoptic_small', **{f'in_{k}': v for k, v in inp.items()},
         **panoptic_outputs(r, with_orientation=True))

    # small, unrounded offsets + distance threshold + fg-masked heatmap
    inp = syn.make_panoptic_inputs(3, n_classes=5, height=64, width=96,
                                   n_centers=5, seed=4, quantize_offsets=False)
    kw = dict(heatmap_threshold=0.2, heatmap_nms_kernel_size=5,
              heatmap_apply_foreground_mask=True, top_k_instances=3,
              offset_distance_threshold=20)
    r = run_panoptic(ref, inp, heatmap_kwargs=kw)
    save('panoptic_small_kwargs', kwargs=jdump(kw),
         **{f'in_{k}': v for k, v in inp.items()}, **panoptic_outputs(r))

    # cfg1: B=4, C=40, 480x640, 24 centers, quantised offsets (digest only)
    for name, B, seed, quant in (('panoptic_cfg1_q', 4, 0, True),
                                 ('panoptic_cfg1_r', 2, 1, False)):
        inp = syn.make_panoptic_inputs(B, seed=seed, quantize_offsets=quant)
        r = run_panoptic(ref, inp)
        digest = syn.input_digest(inp['semantic_logits'], inp['instance_center'],
                                  inp['instance_offset'])
        save(name, params=jdump(dict(batch_size=B, seed=seed, quantize_offsets=quant)),
             digest=jdump(digest), **panoptic_outputs(r, score_stride=8))


def gen_edges(ref):
    """degenerate images: no centers, no foreground, all foreground with one center, a tight
    distance threshold that un-assigns most pixels."""
    print('panoptic pipeline edge cases (reference PanopticPostprocessing.postprocess)')
    inp = syn.make_panoptic_inputs(4, n_classes=6, height=48, width=64, n_centers=4, seed=13)
    lg = inp['semantic_logits']
    is_thing = inp['semantic_classes_is_thing']
    inp['instance_center'][0] = 0.0                       # image 0: no centers at all
    lg[1, np.where(is_thing)[0]] -= 100.0                 # image 1: every pixel is stuff
    lg[2, np.where(~is_thing)[0]] -= 100.0                # image 2: every pixel is a thing ...
    heat = inp['instance_center'][2]
    heat[:] = 0.0
    heat[0, 20, 30] = 0.9                                 # ... and exactly one center
    for name, kw in (('plain', None), ('thr', dict(offset_distance_threshold=6))):
        r = run_panoptic(ref, inp, heatmap_kwargs=kw)
        save(f'panoptic_edges_{name}', kwargs=jdump(kw or {}),
             **{f'in_{k}': v for k, v in inp.items()}, **panoptic_outputs(r))


def gen_centers(ref):
    print('center NMS / top-k adversarial cases (reference _get_instance_centers)')
    rng = np.random.default_rng(7)
    cases = {}
    H, W = 24, 40

    def add(name, heat, fg=None, **kw):
        cases[name] = (heat.astype(np.float32), fg, kw)

    # 1) random noise, many candidates, small top-k -> exercises kth selection
    add('noise_k4', rng.random((3, 1, H, W)), top_k_instances=4)
    add('noise_k64', rng.random((2, 1, H, W)), top_k_instances=64)
    # 2) quantised heatmap -> many ties (plateaus, equal peaks)
    q = np.round(rng.random((3, 1, H, W)) * 4) / 4
    add('quant_ties_k3', q, top_k_instances=3)
    add('quant_ties_ks1', q, top_k_instances=5, heatmap_nms_kernel_size=1)
    add('quant_ties_ks5', q, top_k_instances=5, heatmap_nms_kernel_size=5)
    # 3) peaks on/next to the border (never centers with ks=3)
    b = np.zeros((1, 1, H, W))
    b[0, 0, 0, 0] = 1.0
    b[0, 0, 0, 7] = 0.9
    b[0, 0, 1, 1] = 0.8
    b[0, 0, H - 1, W - 1] = 0.7
    b[0, 0, H - 2, W - 2] = 0.6
    b[0, 0, 12, 0] = 0.95
    b[0, 0, 12, 20] = 0.5
    add('border', b)
    add('border_ks1', b, heatmap_nms_kernel_size=1)
    # 4) fewer than k survivors / none at all / all below threshold
    add('empty', np.zeros((2, 1, H, W)))
    add('below_thr', np.full((1, 1, H, W), 0.05))
    # 5) foreground-masked heatmap
    fg = rng.random((3, H, W)) < 0.5
    add('noise_fg', rng.random((3, 1, H, W)), fg=fg, top_k_instances=6,
        heatmap_apply_foreground_mask=True)
    # 6) equal isolated peaks, more than k (all kept by the >= kth rule)
    e = np.zeros((1, 1, H, W))
    e[0, 0, 2:H - 2:3, 2:W - 2:3] = 1.0
    add('equal_peaks_k5', e, top_k_instances=5)
    # 7) NaN / inf in the heatmap
    nn = rng.random((2, 1, H, W))
    nn[0, 0, 5, 5] = np.nan
    nn[0, 0, 10, 30] = np.inf
    nn[1, 0, 3, 3] = -np.inf
    add('nonfinite', nn, top_k_instances=8)
    # 8) negative threshold, zero-valued heatmap corner (pixel-0 corner case)
    z = rng.random((1, 1, H, W)) - 0.5
    z[0, 0, 0, 0] = 0.0
    add('neg_threshold', z, heatmap_threshold=-0.25, top_k_instances=10)

    out = {}
    for name, (heat, fg, kw) in cases.items():
        post = ref.post_instance.InstancePostprocessing(**kw)
        mask, clist = post._get_instance_centers(
            torch.from_numpy(heat.copy()),
            None if fg is None else torch.from_numpy(fg))
        B = heat.shape[0]
        cap = max(max(len(c) for c in clist), 1)
        cyx = np.zeros((B, cap, 2), np.int32)
        n = np.zeros((B,), np.int32)
        for bi, c in enumerate(clist):
            n[bi] = len(c)
            cyx[bi, :len(c)] = c.numpy()
        out[f'{name}__heat'] = heat
        if fg is not None:
            out[f'{name}__fg'] = fg
        out[f'{name}__kwargs'] = jdump(kw)
        out[f'{name}__mask'] = mask.numpy()
        out[f'{name}__n'] = n
        out[f'{name}__centers'] = cyx
    out['names'] = jdump(list(cases.keys()))
    save('centers_adversarial', **out)


def gen_grouping(ref):
    print('offset grouping adversarial cases (reference _get_instance_segmentation)')
    rng = np.random.default_rng(11)
    out = {}
    names = []

    def run(name, heat, offset, fg, **kw):
        post = ref.post_instance.InstancePostprocessing(normalized_offset=False, **kw)
        seg, meta = post._get_instance_segmentation(
            torch.from_numpy(heat.copy()), torch.from_numpy(offset.copy()),
            torch.from_numpy(fg))
        n, cyx, area, score = meta_to_arrays(meta, cap=400)
        out.update({f'{name}__heat': heat, f'{name}__offset': offset,
                    f'{name}__fg': fg, f'{name}__kwargs': jdump(kw),
                    f'{name}__inst': seg.numpy(), f'{name}__meta_n': n,
                    f'{name}__meta_center_yx': cyx, f'{name}__meta_area': area,
                    f'{name}__meta_score': score})
        names.append(name)

    # exact ties: zero offsets, symmetric centers -> lowest center index wins
    H, W = 33, 49
    heat = np.zeros((1, 1, H, W), np.float32)
    for (y, x) in ((8, 8), (8, 40), (24, 8), (24, 40), (16, 24)):
        heat[0, 0, y, x] = 1.0
    off = np.zeros((1, 2, H, W), np.float32)
    fg = np.ones((1, H, W), bool)
    run('ties_zero_offset', heat, off, fg)
    # with a distance threshold (int, like the reference kwarg)
    run('ties_thr10', heat, off, fg, offset_distance_threshold=10)
    # random offsets, random fg, continuous values
    off = (rng.standard_normal((1, 2, H, W)) * 6).astype(np.float32)
    fg = rng.random((1, H, W)) < 0.6
    run('random_offsets', heat, off, fg)
    run('random_offsets_thr', heat, off, fg, offset_distance_threshold=7)
    # NaN / inf offsets
    offn = off.copy()
    offn[0, 0, 3, 3] = np.nan
    offn[0, 1, 4, 4] = np.inf
    offn[0, 0, 5, 5] = -np.inf
    run('nonfinite_offsets', heat, offn, np.ones((1, H, W), bool))
    # no centers at all / empty foreground
    run('no_centers', np.zeros((2, 1, H, W), np.float32),
        np.zeros((2, 2, H, W), np.float32), np.ones((2, H, W), bool))
    run('empty_fg', heat, off, np.zeros((1, H, W), bool))

    # > 255 centers: uint8 id wrap (instance.py:236); 300 equal isolated peaks
    H, W = 64, 96
    heat = np.zeros((1, 1, H, W), np.float32)
    ys, xs = np.meshgrid(np.arange(2, H - 2, 3), np.arange(2, W - 2, 3), indexing='ij')
    pts = np.stack([ys.ravel(), xs.ravel()], 1)[:300]
    heat[0, 0, pts[:, 0], pts[:, 1]] = 1.0
    off = (rng.standard_normal((1, 2, H, W)) * 2).astype(np.float32)
    fg = rng.random((1, H, W)) < 0.8
    run('wrap_300_centers', heat, off, fg, top_k_instances=254)

    out['names'] = jdump(names)
    save('grouping_adversarial', **out)


def gen_merge(ref):
    print('merge (reference deeplab_merge_batch + numpy twins + naive)')
    pm = ref.panoptic_merge
    rng = np.random.default_rng(13)
    H, W = 40, 56
    out = {}
    names = []

    def blobs(n_vals, B, lo=0):
        coarse = rng.integers(lo, n_vals, size=(B, H // 8, W // 8))
        return np.repeat(np.repeat(coarse, 8, axis=1), 8, axis=2)

    def run(name, sem, ins, thing, max_inst, thing_ids, void):
        pan, dicts = pm.deeplab_merge_batch(
            torch.from_numpy(sem), torch.from_numpy(ins), torch.from_numpy(thing),
            max_inst, thing_ids, void)
        n, k, v = ids_to_arrays(dicts, cap=2048)
        out.update({f'{name}__sem': sem, f'{name}__ins': ins, f'{name}__thing': thing,
                    f'{name}__params': jdump(dict(max_inst=max_inst,
                                                  thing_ids=[int(t) for t in thing_ids],
                                                  void=void)),
                    f'{name}__pan': pan.numpy(), f'{name}__ids_n': n,
                    f'{name}__ids_pan': k, f'{name}__ids_ins': v})
        names.append(name)
        return pan.numpy(), dicts

    # prediction-style: sem 1..C (no void), instance blobs cut across classes,
    # instances only inside things
    C = 6
    sem = blobs(C, 3, lo=0).astype(np.int64) + 1
    thing_ids = [4, 5, 6]
    thing = np.isin(sem, thing_ids)
    ins = (np.roll(blobs(9, 3), 3, axis=2) * thing).astype(np.uint8)
    run('pred_style', sem, ins, thing, 1 << 16, thing_ids, 0)
    # GT-style with void (0) in sem, int32 instance ids, thing mask = ins != 0,
    # ids spanning void-majority regions and tie votes
    sem = blobs(C + 1, 3).astype(np.int64)
    ins = blobs(12, 3).astype(np.int32)
    ins[:, ::2, :] = np.roll(ins, 5, axis=2)[:, ::2, :]
    thing = ins != 0
    run('gt_style_void', sem, ins, thing, 1 << 16, [3, 4], 0)
    # thing mask disagreeing with ins>0, small max_inst, non-zero void label
    thing2 = rng.random(sem.shape) < 0.5
    run('mask_mismatch', sem, ins, thing2, 256, [1, 2, 6], 0)
    run('void_label_7', sem, ins, thing, 100, [3, 4], 7)
    # ground-truth style ids beyond uint8: 300 distinct ids up to 60000, and 1500 distinct ids
    # (more than the first table size of the HIP path)
    for nm, n_ids, hi in (('wide_ids', 300, 60000), ('many_ids', 1500, 65535)):
        ids = np.sort(rng.choice(np.arange(1, hi + 1), size=n_ids, replace=False))
        cells = rng.integers(0, n_ids + 1, size=(2, H // 2, W // 2))      # 0 = no instance
        ins_w = np.where(cells > 0, ids[np.maximum(cells - 1, 0)], 0)
        ins_w = np.repeat(np.repeat(ins_w, 2, axis=1), 2, axis=2).astype(np.int32)
        sem_w = blobs(C + 1, 2).astype(np.int64)
        run(nm, sem_w, ins_w, ins_w != 0, 1 << 16, [3, 4, 5], 0)
    # everything stuff / everything void
    run('all_stuff', sem, np.zeros_like(ins), np.zeros_like(thing), 1 << 16, [], 0)
    run('all_void', np.zeros_like(sem), ins, thing, 1 << 16, [1], 0)

    # numpy twins + naive on consistent GT-style maps (instances never span
    # classes: the invariant of reference tests/test_merge.py:97-102)
    B = 2
    ins_u16 = blobs(10, B).astype(np.uint16)
    cls_of_ins = np.array([0, 3, 4, 3, 4, 4, 3, 3, 4, 3])
    stuff = blobs(3, B).astype(np.uint8)          # 0,1,2 (0 = void)
    sem_u8 = np.where(ins_u16 > 0, cls_of_ins[ins_u16], stuff).astype(np.uint8)
    naive_p, deeplab_p, torch_p = [], [], []
    for b in range(B):
        n_p, n_d = pm.naive_merge_semantic_and_instance_np(
            sem_u8[b], ins_u16[b], 1 << 16, [3, 4], 0)
        d_p, d_d = pm.deeplab_merge_semantic_and_instance_np(
            sem_u8[b], ins_u16[b], ins_u16[b] != 0, 1 << 16, [3, 4], 0)
        t_p, t_d = pm.deeplab_merge_semantic_and_instance(
            torch.from_numpy(sem_u8[b].astype(np.int64)),
            torch.from_numpy(ins_u16[b].astype(np.int32)),
            torch.from_numpy(ins_u16[b] != 0), 1 << 16, [3, 4], 0)
        assert (n_p == d_p).all() and (n_p == t_p.numpy()).all()
        assert n_d == d_d == t_d
        naive_p.append(n_p)
    n, k, v = ids_to_arrays([n_d], cap=64)
    out.update(consistent__sem=sem_u8, consistent__ins=ins_u16,
               consistent__pan=np.stack(naive_p).astype(np.int64),
               consistent__last_ids_n=n, consistent__last_ids_pan=k,
               consistent__last_ids_ins=v)
    # naive on an inconsistent map (instances split by class)
    sem_b = blobs(5, 1).astype(np.uint8)
    n_p, n_d = pm.naive_merge_semantic_and_instance_np(
        sem_b[0], ins_u16[0], 1 << 16, [3, 4], 0)
    n, k, v = ids_to_arrays([n_d], cap=128)
    out.update(naive_split__sem=sem_b, naive_split__ins=ins_u16[:1],
               naive_split__pan=n_p[None].astype(np.int64), naive_split__ids_n=n,
               naive_split__ids_pan=k, naive_split__ids_ins=v)
    out['names'] = jdump(names)
    save('merge_cases', **out)


def gen_metrics(ref):
    print('metrics (reference MeanIntersectionOverUnion / compare_and_accumulate)')
    rng = np.random.default_rng(17)
    out = {}
    # mIoU: n in (5, 41, 101), with/without void, incl. absent classes
    for n in (5, 41, 101):
        pred = rng.integers(0, n, size=(4, 50, 60))
        tgt = rng.integers(0, n, size=(4, 50, 60))
        tgt[tgt == 3] = 2                        # class 3 has no GT
        for ign in (False, True):
            m = ref.metric_miou.MeanIntersectionOverUnion(n, ignore_first_class=ign)
            m.update(torch.from_numpy(pred[:2]), torch.from_numpy(tgt[:2]))
            m.update(torch.from_numpy(pred[2:]), torch.from_numpy(tgt[2:]))
            miou, ious = m.compute(return_ious=True)
            out[f'miou_{n}_{int(ign)}__confmat'] = m.confmat.numpy()
            out[f'miou_{n}_{int(ign)}__miou'] = np.float32(miou.item())
            out[f'miou_{n}_{int(ign)}__ious'] = ious.numpy()
        out[f'miou_{n}__pred'] = pred.astype(np.uint8)
        out[f'miou_{n}__target'] = tgt.astype(np.uint8)

    # PQ on blobby random panoptic maps (class*65536 + inst), offset 256**3
    H, W = 48, 64

    def pan_map(B, n_cls, seed):
        r = np.random.default_rng(seed)
        cls = np.repeat(np.repeat(r.integers(0, n_cls, (B, H // 8, W // 8)), 8, 1), 8, 2)
        ins = np.repeat(np.repeat(r.integers(0, 4, (B, H // 4, W // 4)), 4, 1), 4, 2)
        thing = cls >= n_cls // 2
        return (cls.astype(np.int64) * 65536 + ins * thing).astype(np.int64)

    n_cls = 9
    pred = pan_map(3, n_cls, 1)
    tgt = np.roll(pred, 2, axis=2)
    tgt[:, :6] = 0
    tgt2 = pan_map(3, n_cls, 2)
    states = []
    all_matches = []
    for name, (p, t) in (('shift', (pred, tgt)), ('indep', (pred, tgt2))):
        st = [np.zeros(n_cls) for _ in range(4)]
        per_image = []
        for b in range(p.shape[0]):
            iou, tp, fn, fp, matched = ref.metric_pq.compare_and_accumulate(
                torch.from_numpy(p[b]), torch.from_numpy(t[b]),
                n_cls, 0, 65536, 256 ** 3, 0)
            for s, x in zip(st, (iou, tp, fn, fp)):
                s += x.numpy()
            per_image.append(sorted(matched))
        out[f'pq_{name}__pred'] = p
        out[f'pq_{name}__target'] = t
        out[f'pq_{name}__state'] = np.stack(st)
        out[f'pq_{name}__matches'] = jdump(per_image)
    out['pq_params'] = jdump(dict(num_categories=n_cls, ignored_label=0,
                                  max_instances_per_category=65536, offset=256 ** 3))
    save('metric_cases', **out)


def gen_losses(ref):
    print('losses (reference loss classes + task-helper masking, with autograd grads)')
    B, Cn, H, W, D = 2, 7, 24, 32, 16
    inp = syn.make_loss_inputs(B, Cn, H, W, seed=5, embedding_dim=D, n_lut=9)
    out = {f'in_{k}': v for k, v in inp.items()}
    t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in inp.items()}

    # --- CE (ce.py) : plain / weighted / label smoothing / weighted reduction
    for name, kw in (('plain', {}),
                     ('weighted', dict(weights=t['class_weights'])),
                     ('smooth', dict(weights=t['class_weights'], label_smoothing=0.25)),
                     ('smooth_nw', dict(label_smoothing=0.5)),
                     ('wred', dict(weights=t['class_weights'], weighted_reduction=True))):
        x = t['semantic_logits'].clone().requires_grad_(True)
        fn = ref.loss_ce.CrossEntropyLossSemantic(**kw)
        (loss, n), = fn([x], [t['semantic_target']])
        loss.backward()
        out[f'ce_{name}__loss'] = np.float32(loss.item())
        out[f'ce_{name}__n'] = np.int64(n)
        out[f'ce_{name}__grad'] = x.grad.numpy()

    # --- center (mse / l1), masking of task_helper/instance.py:129-139
    for kind, cls in (('mse', ref.loss_mse.MSELoss), ('l1', ref.loss_l1.L1Loss)):
        x = t['center_pred'].clone().requires_grad_(True)
        mask = t['center_mask']
        (loss, n), = cls(reduction='sum')([x * mask], [t['center_target']])
        loss.backward()
        out[f'center_{kind}__loss'] = np.float32(loss.item())
        out[f'center_{kind}__n_loss'] = np.int64(n)
        out[f'center_{kind}__n_mask'] = np.int64(mask.sum().item())
        out[f'center_{kind}__grad'] = x.grad.numpy()
    # --- offset (l1), task_helper/instance.py:154-167
    x = t['offset_pred'].clone().requires_grad_(True)
    mask = t['offset_mask']
    (loss, n), = ref.loss_l1.L1Loss(reduction='sum')(
        [x * mask.unsqueeze(1).expand_as(x)], [t['offset_target']])
    loss.backward()
    out['offset_l1__loss'] = np.float32(loss.item())
    out['offset_l1__n_loss'] = np.int64(n)
    out['offset_l1__n_mask'] = np.int64(mask.sum().item())
    out['offset_l1__grad'] = x.grad.numpy()
    # --- orientation (von Mises), task_helper/instance.py:186-216
    x = t['orientation_pred'].clone().requires_grad_(True)
    mask = t['orientation_mask'].flatten()
    p2 = x.permute(0, 2, 3, 1).reshape(-1, 2)[mask, :]
    y2 = t['orientation_target'].permute(0, 2, 3, 1).reshape(-1, 2)[mask, :]
    for kappa in (1.0, 2.5):
        x.grad = None
        (loss, n), = ref.loss_vonmises.VonMisesLossBiternion(kappa=kappa)([p2], [y2])
        loss.backward(retain_graph=True)
        out[f'vonmises_{kappa}__loss'] = np.float32(loss.item())
        out[f'vonmises_{kappa}__n'] = np.int64(n)
        out[f'vonmises_{kappa}__grad'] = x.grad.numpy().copy()
    # --- cosine embedding, task_helper/dense_visual_embedding.py:110-171
    x = t['embedding_pred'].clone().requires_grad_(True)
    idx = t['embedding_indices']
    valid = idx != 0
    pm_ = x.permute(0, 2, 3, 1)[valid]
    keep = (idx - 1)[valid].long()
    bidx = torch.where(valid)[0]
    tg = torch.cat([t['embedding_lut'][b][keep[bidx == b]] for b in range(B)], 0)
    (loss, n), = ref.loss_cos_emb.CosineEmbeddingLoss()([pm_], [tg])
    loss.backward()
    out['cos_emb__loss'] = np.float32(loss.item())
    out['cos_emb__n'] = np.int64(n)
    out['cos_emb__grad'] = x.grad.numpy()
    save('loss_cases', **out)


def gen_loss_forms(ref):
    """the forms of the loss classes no task helper calls: reduction none / sum / mean on 2-D, 3-D
    and 4-D inputs (mse.py:21-41, l1.py:21-41) and labelled cosine pairs (cos_emb.py:21-56)"""
    print('loss forms (reductions x ranks of MSE / L1, labelled cosine rows; with autograd grads)')
    rng = np.random.default_rng(77)
    out = {}
    shapes = {'r2': (7, 5), 'r3': (2, 6, 8), 'r4': (2, 3, 6, 8)}
    for rk, shp in shapes.items():
        out[f'{rk}__x'] = rng.standard_normal(shp).astype(np.float32)
        out[f'{rk}__t'] = rng.standard_normal(shp).astype(np.float32)
        out[f'{rk}__w'] = rng.standard_normal(shp).astype(np.float32)      # upstream of the 'none' form
    for kind, cls in (('mse', ref.loss_mse.MSELoss), ('l1', ref.loss_l1.L1Loss)):
        for rk in shapes:
            for red in ('none', 'sum', 'mean'):
                x = torch.from_numpy(out[f'{rk}__x']).clone().requires_grad_(True)
                t = torch.from_numpy(out[f'{rk}__t'])
                (loss, n), = cls(reduction=red)([x], [t])
                (loss * torch.from_numpy(out[f'{rk}__w'])).sum().backward() if red == 'none' else loss.backward()
                out[f'{kind}_{rk}_{red}__loss'] = loss.detach().numpy().astype(np.float32)
                out[f'{kind}_{rk}_{red}__n'] = np.int64(n)
                out[f'{kind}_{rk}_{red}__grad'] = x.grad.numpy()
    N, D = 11, 16
    out['cos__x'] = rng.standard_normal((N, D)).astype(np.float32)
    tgt = rng.standard_normal((N, D)).astype(np.float32)
    tgt[:4] = out['cos__x'][:4] + 0.3 * tgt[:4]                # some similar pairs, some not
    tgt[4:6] = -out['cos__x'][4:6] + 0.2 * tgt[4:6]            # opposite directions: the clamp is active
    out['cos__t'] = tgt
    out['cos__labels'] = np.array([1, -1, 1, -1, -1, 1, -1, 0, 1, -1, -1], dtype=np.float32)
    out['cos__w'] = rng.standard_normal((N,)).astype(np.float32)
    for lab in ('labelled', 'plain'):
        for red in ('none', 'sum', 'mean'):
            x = torch.from_numpy(out['cos__x']).clone().requires_grad_(True)
            fn = ref.loss_cos_emb.CosineEmbeddingLoss(reduction=red)
            args = (x, torch.from_numpy(out['cos__t'])) + \
                ((torch.from_numpy(out['cos__labels']),) if lab == 'labelled' else ())
            loss, n = fn._compute_loss(*args)
            (loss * torch.from_numpy(out['cos__w'])).sum().backward() if red == 'none' else loss.backward()
            out[f'cos_{lab}_{red}__loss'] = loss.detach().numpy().astype(np.float32)
            out[f'cos_{lab}_{red}__n'] = np.int64(n)
            out[f'cos_{lab}_{red}__grad'] = x.grad.numpy()
    save('loss_forms', **out)


def gen_argmax_ties(ref):
    """a1 boundary: the reference takes max(softmax(x)) (model/postprocessing/semantic.py:52-53),
    the kernels take argmax(x).  Adversarial columns: classes c1 < c2 with x[c2] = x[c1] + k ulp,
    every other class at least 1 below, |x| from 1e-3 to 16, C = 40, f32.  Stored: the columns,
    the reference's `semantic_segmentation_idx`, delta = x[c2] - x[c1].  Also the natural rate:
    the reference vs argmax(x) on 8.4 Mpx of the bench's blobby logits and on 8.4 Mpx of
    small-magnitude logits (|x| < 0.3, where adjacent floats are < 2^-25 apart)."""
    print('a1: max(softmax(x)) vs argmax(x) on adversarial near-tie columns')
    post = ref.post_semantic.SemanticPostprocessing()
    rng = np.random.default_rng(0)
    C = 40
    cols, c1s, c2s = [], [], []
    for mag in (1e-3, 1e-2, 0.1, 0.25, 0.5, 1, 2, 4, 8, 16):
        for k in (1, 2, 3, 4, 8, 16, 64):
            for sign in (1, -1):
                for _ in range(12):
                    a = np.float32(sign * mag * (1 + rng.random() * 0.5))
                    x = np.minimum((rng.standard_normal(C) * 0.5 - 3).astype(np.float32),
                                   a - np.float32(1.0)).astype(np.float32)
                    c1, c2 = sorted(rng.choice(C, 2, replace=False))
                    b = a
                    for _ in range(k):
                        b = np.nextafter(b, np.float32(np.inf))
                    x[c1], x[c2] = a, b
                    cols.append(x)
                    c1s.append(c1)
                    c2s.append(c2)
    N = len(cols)
    W = 40
    assert N % W == 0
    logits = np.stack(cols).T.reshape(1, C, N // W, W).copy()
    r = post.postprocess((torch.from_numpy(logits), None),
                         make_batch(ref, 1, N // W, W), is_training=False)
    ref_idx = r['semantic_segmentation_idx'].numpy().astype(np.uint8)
    c1s, c2s = np.array(c1s, np.uint8), np.array(c2s, np.uint8)
    delta = np.array([float(c[j]) - float(c[i]) for c, i, j in zip(cols, c1s, c2s)])
    flat = ref_idx.reshape(-1)
    assert ((flat == c1s) | (flat == c2s)).all()
    for lo, hi in ((0, 2 ** -25), (2 ** -25, 2 ** -23), (2 ** -23, 1)):
        m = (delta > lo) & (delta <= hi)
        print(f'  delta in ({lo:.3g}, {hi:.3g}]: {m.sum()} columns, reference returns the LOWER '
              f'index in {(flat[m] == c1s[m]).sum()}')

    # natural rate on continuous logits (counts only; the tensors are not stored)
    def rate(make, n_img):
        n_px = n_diff = 0
        for s in range(n_img):
            x = make(s)
            pr = post.postprocess((x, None), make_batch(ref, x.shape[0], *x.shape[-2:]),
                                  is_training=False)['semantic_segmentation_idx']
            n_diff += int((pr != x.argmax(dim=1)).sum())
            n_px += pr.numel()
        return n_px, n_diff

    def blobby(seed):
        g = torch.Generator().manual_seed(seed)
        coarse = torch.randn((4, C, 15, 20), generator=g)
        return 4.0 * torch.nn.functional.interpolate(coarse, size=(480, 640), mode='bilinear',
                                                     align_corners=False)

    def small(seed):
        g = torch.Generator().manual_seed(100 + seed)
        return (torch.rand((4, C, 480, 640), generator=g) - 0.5) * 0.6
    nb = rate(blobby, 7)
    ns = rate(small, 7)
    print(f'  natural rate: blobby logits {nb[1]} of {nb[0]} px differ; |x| < 0.3 logits '
          f'{ns[1]} of {ns[0]} px differ')
    def grid_map():
        # 4 random classes per pixel on a grid of 2^-26 (gaps of 0..11 steps: inside the 2^-25
        # band, between the bands and outside), the other classes far below
        x = (-1.0 - rng.random((1, C, 48, 64))).astype(np.float32)
        for _ in range(4):
            c = rng.integers(0, C, (48, 64))
            v = (rng.integers(0, 12, (48, 64)) * 2.0 ** -26).astype(np.float32)
            np.put_along_axis(x[0], c[None], v[None], axis=0)
        return x

    # whole maps in the regime where probabilities tie by themselves, through the reference
    # (network resolution and the full-resolution resize): every class within 2^-25 of the
    # maximum ('tiny': all of them -> index 0) and a map on a 2^-26 grid ('small')
    key_full = 'semantic_segmentation_idx_fullres'
    extra = {}
    for name, x, full in (
            ('tiny', (rng.standard_normal((1, C, 24, 40)) * 1e-9).astype(np.float32), (36, 60)),
            ('small', grid_map(), (72, 96))):
        r = post.postprocess((torch.from_numpy(x), None), make_batch(ref, x.shape[0], *full),
                             is_training=False)
        extra[f'{name}_logits'] = x
        extra[f'{name}_ref_idx'] = r['semantic_segmentation_idx'].numpy().astype(np.uint8)
        extra[f'{name}_ref_idx_fullres'] = r[key_full].numpy().astype(np.uint8)
        am = torch.from_numpy(x).argmax(dim=1).numpy()
        print(f'  {name}: reference differs from argmax(x) on '
              f'{int((extra[f"{name}_ref_idx"] != am).sum())} of {am.size} px')
    # which ATen build resolved these near-ties: its CPU softmax (Sleef expf_u10, vectorised
    # summation order) is what csrc/argmax_state.hpp restates; another build (exp_u20, no FMA,
    # a GPU softmax) may resolve columns inside the 2^-25 .. 2^-23 band differently
    import platform
    producer = {'torch': torch.__version__, 'cpu_capability': torch.backends.cpu.get_cpu_capability(),
                'machine': platform.machine(), 'threads': torch.get_num_threads()}
    save('argmax_ties', logits=logits, ref_idx=ref_idx, c1=c1s, c2=c2s, delta=delta,
         natural_blobby=np.array(nb, np.int64), natural_small=np.array(ns, np.int64),
         producer=jdump(producer), **extra)


COS_LARGE_CASES = (
    # name, B, D, H, W, L, bf16      (kernel path on the GPU, csrc/losses.hip::cos_chunk)
    ('d512_l64', 2, 512, 24, 32, 64, False),       # one LDS chunk of 131 KB
    ('d512_l64_bf16', 2, 512, 24, 32, 64, True),
    ('d768_l64', 2, 768, 24, 32, 64, False),       # 197 KB as fp32: two chunks of 384
    ('d768_l64_bf16', 2, 768, 24, 32, 64, True),
    ('d768_l40_ragged', 1, 768, 17, 23, 40, False),  # single chunk, P % 4 != 0 (scalar loads)
    ('d520_l90', 1, 520, 16, 24, 90, True),        # chunks of 264 + 256, D % chunk != 0
    ('d48_l1300', 1, 48, 16, 16, 1300, False),     # LUT too tall for LDS: generic kernel
)


def gen_cos_emb_large(ref):
    """reference CosineEmbeddingLoss with the masking / LUT gather of
    task_helper/dense_visual_embedding.py:110-171 at the dense-visual-embedding sizes
    (D = 512 / 768, L up to 64).  Inputs are regenerated from the seed by the tests
    (`make_embedding_inputs`, digest stored); the full gradient would be 5 MB per case, so the
    golden keeps it for 96 sampled pixels plus three f64 projections of the whole tensor."""
    print('cosine embedding at D >= 512 (reference loss + autograd grads)')
    out = {'names': jdump([c[0] for c in COS_LARGE_CASES])}
    for seed, (name, B, D, H, W, L, bf16) in enumerate(COS_LARGE_CASES):
        inp = syn.make_embedding_inputs(B, D, H, W, L, seed=seed, bf16=bf16)
        x = torch.from_numpy(inp['embedding_pred']).clone().requires_grad_(True)
        idx = torch.from_numpy(inp['embedding_indices'])
        lut = torch.from_numpy(inp['embedding_lut'])
        valid = idx != 0
        pm_ = x.permute(0, 2, 3, 1)[valid]
        keep = (idx - 1)[valid].long()
        bidx = torch.where(valid)[0]
        parts = [lut[b][keep[bidx == b]] for b in range(B) if int((bidx == b).sum())]
        tg = torch.cat(parts, 0)
        (loss, n), = ref.loss_cos_emb.CosineEmbeddingLoss()([pm_], [tg])
        loss.backward()
        grad = x.grad.numpy().astype(np.float64)
        rng = np.random.default_rng(900 + seed)
        pix = np.sort(rng.choice(B * H * W, size=min(96, B * H * W), replace=False))
        g_rows = x.grad.permute(0, 2, 3, 1).reshape(-1, D)[torch.from_numpy(pix)].numpy()
        proj = rng.standard_normal(grad.shape)
        out[f'{name}__params'] = jdump(dict(B=B, D=D, H=H, W=W, L=L, bf16=bf16, seed=seed))
        out[f'{name}__digest'] = jdump(syn.input_digest(inp['embedding_pred'], inp['embedding_lut'],
                                                        inp['embedding_indices']))
        out[f'{name}__loss'] = np.float32(loss.item())
        out[f'{name}__n'] = np.int64(n)
        out[f'{name}__grad_pixels'] = pix.astype(np.int64)
        out[f'{name}__grad_rows'] = g_rows
        out[f'{name}__grad_sums'] = np.array([grad.sum(), np.abs(grad).sum(), (grad * proj).sum()])
        out[f'{name}__proj_seed'] = np.int64(900 + seed)
        print(f'  {name}: loss {loss.item():.6f} n {n}')
    save('cos_emb_large', **out)


def gen_orientation(ref):
    print('instance orientation (reference _get_instance_orientation)')
    inp = syn.make_panoptic_inputs(2, n_classes=6, height=48, width=64, n_centers=5,
                                   seed=9, with_orientation=True)
    rng = np.random.default_rng(19)
    inst = np.repeat(np.repeat(rng.integers(0, 5, (2, 6, 8)), 8, 1), 8, 2).astype(np.uint8)
    mask = rng.random((2, 48, 64)) < 0.7
    post = ref.post_instance.InstancePostprocessing()
    res = {}
    for name, m in (('masked', mask), ('nomask', None)):
        r = post._get_instance_orientation(
            torch.from_numpy(inp['instance_orientation']), torch.from_numpy(inst),
            None if m is None else torch.from_numpy(m))
        arr = np.full((2, 256), np.nan, np.float32)
        for b, d in enumerate(r):
            for k, v in d.items():
                arr[b, k] = v
        res[f'{name}__angle'] = arr
    save('orientation_cases', orientation=inp['instance_orientation'], inst=inst,
         mask=mask, **res)


def gen_fullres(ref):
    """f2: crop to the valid region + resize to the dataset resolution
    (dense_base.py:15-58, semantic.py:61-80, panoptic.py:240-290)."""
    print('full-resolution step (reference _crop_to_valid_region_and_resize_prediction)')
    from oracle import oracle as orc
    post = ref.post_semantic.SemanticPostprocessing()
    rng = np.random.default_rng(31)
    Hs, Ws = 96, 128
    geoms = {                       # name -> (crop, out shape)
        'up_crop': ((slice(4, 92), slice(0, 128)), (150, 200)),
        'up_crop_xy': ((slice(0, 96), slice(7, 121)), (171, 203)),
        'down': ((slice(0, 96), slice(0, 128)), (70, 90)),
        'double': ((slice(0, 96), slice(0, 128)), (192, 256)),
        'odd': ((slice(3, 90), slice(5, 126)), (131, 197)),
    }
    out = {'geoms': jdump({k: [[c[0].start, c[0].stop, c[1].start, c[1].stop], list(sz)]
                           for k, (c, sz) in geoms.items()})}
    maps = {
        'u8': rng.integers(0, 256, (1, Hs, Ws)).astype(np.uint8),
        'bool': rng.random((1, Hs, Ws)) < 0.5,
        'i32': rng.integers(-5, 1 << 20, (1, Hs, Ws)).astype(np.int32),
        # panoptic ids; a few beyond 2^24 to pin the float32 round trip of the reference
        'i64': np.where(rng.random((1, Hs, Ws)) < 0.05,
                        rng.integers(1 << 24, 1 << 28, (1, Hs, Ws)),
                        rng.integers(0, 40 * 65536, (1, Hs, Ws))).astype(np.int64),
        'f32': rng.random((1, Hs, Ws)).astype(np.float32),
    }
    logits = (rng.standard_normal((1, 3, Hs, Ws)) * 3).astype(np.float32)
    for k, v in maps.items():
        out[f'in__{k}'] = v
    out['in__logits'] = logits
    for name, (crop, size) in geoms.items():
        for k, v in maps.items():
            r = post._crop_to_valid_region_and_resize_prediction(
                torch.from_numpy(v), valid_region_slices=crop, shape=size, mode='nearest').numpy()
            assert np.array_equal(r, orc.resize_nearest(v, size, crop)), (name, k)
            if name in ('up_crop', 'odd') or k in ('u8', 'i64'):     # keep the file small
                out[f'{name}__nearest_{k}'] = r
        r = post._crop_to_valid_region_and_resize_prediction(
            torch.from_numpy(logits), valid_region_slices=crop, shape=size, mode='bilinear').numpy()
        mine = orc.resize_bilinear(logits, size, crop)
        # bit-exact, NaN-free inputs (the committed vectors are the reference's)
        assert np.array_equal(r, mine), (name, float(np.abs(r - mine).max()))
        out[f'{name}__bilinear'] = r
    save('fullres_cases', **out)

    # ---- end to end: the reference panoptic postprocessing with a real fullres step ----
    inp = syn.make_panoptic_inputs(2, n_classes=40, height=Hs, width=Ws, n_centers=9, seed=77)
    is_thing = tuple(bool(x) for x in inp['semantic_classes_is_thing'])
    pan = ref.post_panoptic.PanopticPostprocessing(
        semantic_postprocessing=ref.post_semantic.SemanticPostprocessing(),
        instance_postprocessing=ref.post_instance.InstancePostprocessing(),
        semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing)
    crop, size = geoms['up_crop']
    batch = {
        'rgb_fullres': torch.zeros((2, 3) + size),
        ref.APPLIED_PREPROCESSING_KEY: [[{
            'type': 'Resize', 'valid_region_slice_y': crop[0], 'valid_region_slice_x': crop[1],
        }]] * 2,
    }
    data = ((torch.from_numpy(inp['semantic_logits']),
             (torch.from_numpy(inp['instance_center']), torch.from_numpy(inp['instance_offset']))),
            (None, None))
    r = pan.postprocess(data, batch, is_training=False)
    e2e = dict(input_digest=jdump(syn.input_digest(inp['semantic_logits'], inp['instance_center'],
                                                    inp['instance_offset'])),
               crop=np.array([crop[0].start, crop[0].stop, crop[1].start, crop[1].stop], np.int32),
               size=np.array(size, np.int32))
    for k, v in r.items():
        if not (k.endswith('_fullres') and torch.is_tensor(v)):
            continue
        a = v.numpy()
        if k == 'semantic_output_fullres':
            a = a[:, ::13]                                   # 4 of 40 classes
        elif k == 'semantic_softmax_scores_fullres':
            continue                                          # = softmax of the entry above
        e2e[k] = a
        print('   ', k, a.dtype, a.shape)
    # the argmax boundary (argmax of logits vs max of softmax) is documented; make sure the
    # committed vector does not sit on it
    lf = orc.resize_bilinear(inp['semantic_logits'], size, crop)
    idx, _ = orc.semantic_argmax(lf)
    assert np.array_equal(idx, e2e['semantic_segmentation_idx_fullres'])
    save('fullres_panoptic', **e2e)


def gen_scores(ref):
    """f3: the compute_scores branch of the reference's PanopticPostprocessing."""
    print('compute_scores (reference panoptic.py:171-239)')
    from oracle import oracle as orc
    inp = syn.make_panoptic_inputs(2, n_classes=8, height=96, width=128, n_centers=7, seed=5)
    is_thing = tuple(bool(x) for x in inp['semantic_classes_is_thing'])
    pan = ref.post_panoptic.PanopticPostprocessing(
        semantic_postprocessing=ref.post_semantic.SemanticPostprocessing(),
        instance_postprocessing=ref.post_instance.InstancePostprocessing(),
        semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing,
        compute_scores=True)
    logits = torch.from_numpy(inp['semantic_logits'])
    data = ((logits, (torch.from_numpy(inp['instance_center']),
                      torch.from_numpy(inp['instance_offset']))), (None, None))
    r = pan.postprocess(data, make_batch(ref, 2, 96, 128), is_training=False)
    meta = r['panoptic_segmentation_deeplab_instance_meta']
    cap = 256
    m_sem = np.full((2, cap), np.nan, np.float32)
    m_pan = np.full((2, cap), np.nan, np.float32)
    m_idx = np.full((2, cap), -1, np.int64)
    m_pid = np.full((2, cap), -1, np.int64)
    tab = np.zeros((2, 256), np.float32)
    for b, md in enumerate(meta):
        for i, m in md.items():
            tab[b, i] = m['score']
            if 'panoptic_id' in m:
                m_sem[b, i] = m['semantic_score']
                m_pan[b, i] = m['panoptic_score']
                m_idx[b, i] = m['semantic_idx']
                m_pid[b, i] = m['panoptic_id']
    idn, idk, idv = ids_to_arrays(r['panoptic_segmentation_deeplab_ids'])
    out = dict(
        in_semantic_logits=inp['semantic_logits'], in_instance_center=inp['instance_center'],
        in_instance_offset=inp['instance_offset'],
        in_semantic_classes_is_thing=inp['semantic_classes_is_thing'],
        panoptic=r['panoptic_segmentation_deeplab'].numpy(),
        panoptic_semantic=r['panoptic_segmentation_deeplab_semantic_idx'].numpy(),
        ids_n=idn, ids_pan=idk, ids_ins=idv, inst_score_by_id=tab,
        semantic_score=r['panoptic_segmentation_deeplab_semantic_score'].numpy(),
        instance_score=r['panoptic_segmentation_deeplab_instance_score'].numpy(),
        panoptic_score=r['panoptic_segmentation_deeplab_panoptic_score'].numpy(),
        meta_semantic_score=m_sem, meta_panoptic_score=m_pan, meta_semantic_idx=m_idx,
        meta_panoptic_id=m_pid)
    # generation-time check of the restatement
    ids = [dict(zip(idk[b, :idn[b]].tolist(), idv[b, :idn[b]].tolist())) for b in range(2)]
    sem, ins, pns, mean = orc.panoptic_scores(inp['semantic_logits'], out['panoptic_semantic'],
                                              out['panoptic'], ids, tab)
    np.testing.assert_allclose(sem, out['semantic_score'], rtol=1e-5, atol=1e-7)
    assert np.array_equal(ins, out['instance_score'])
    np.testing.assert_allclose(pns, out['panoptic_score'], rtol=1e-5, atol=1e-7)
    save('scores_cases', **out)


def gen_targets(ref):
    """f4: the reference's numpy target generators, sample by sample."""
    print('target generation (reference data/preprocessing/{instance,panoptic,dense_visual_embedding}.py)')
    from oracle import oracle as orc
    B, NC, H, W = 3, 9, 96, 128
    maps = syn.make_label_maps(B, NC, H, W, n_instances=14, seed=1)
    is_thing = tuple(bool(x) for x in maps['semantic_classes_is_thing'])
    out = dict(in_semantic=maps['semantic'], in_instance=maps['instance'],
               in_is_thing=maps['semantic_classes_is_thing'])
    clear = ref.prep_instance.InstanceClearStuffIDs(semantic_classes_is_thing=is_thing)
    gens = {
        's8n': ref.prep_instance.InstanceTargetGenerator(sigma=8, semantic_classes_is_thing=is_thing),
        's3u': ref.prep_instance.InstanceTargetGenerator(sigma=3, semantic_classes_is_thing=is_thing,
                                                         normalized_offset=False),
        's5nothing': ref.prep_instance.InstanceTargetGenerator(sigma=5),
    }
    pgen = ref.prep_panoptic.PanopticTargetGenerator(semantic_classes_is_thing=is_thing)
    dgen = ref.prep_dve.DenseVisualEmbeddingTargetGenerator(diff_factor=0.65)
    rng = np.random.default_rng(3)
    D, K = 16, 12
    cleared = np.empty((B, H, W), np.int32)
    res = {k: dict(center=[], offset=[], fg=[], cmask=[]) for k in gens}
    pan, pan_ids = [], []
    dve_keys = np.zeros((B, K), np.int64)
    dve_n = np.zeros((B,), np.int32)
    dve_emb = rng.standard_normal((B, K, D)).astype(np.float32)
    dve_img = rng.standard_normal((B, D)).astype(np.float32)
    dve_lut = np.zeros((B, K, D), np.float32)
    dve_idx = np.zeros((B, H, W), np.int32)
    enc = {k: [] for k in gens}
    for b in range(B):
        smp = clear({'semantic': maps['semantic'][b].copy(),
                     'instance': maps['instance'][b].astype(np.uint16)})
        cleared[b] = smp['instance']
        for k, gen in gens.items():
            sample = {'semantic': smp['semantic'].copy(), 'instance': smp['instance'].copy()}
            r = gen(sample)
            res[k]['center'].append(r['instance_center'])
            res[k]['offset'].append(np.moveaxis(r['instance_offset'], -1, 0))      # HWC -> CHW
            res[k]['fg'].append(r['instance_foreground'])
            res[k]['cmask'].append(r['instance_center_mask'])
        r = pgen({'semantic': smp['semantic'].copy(), 'instance': smp['instance'].copy()})
        pan.append(r['panoptic'].astype(np.int64))
        pan_ids.append(dict(r['panoptic_ids_to_instance_dict']))
        present = [int(v) for v in np.unique(r['panoptic']) if v != 0]
        keys = [present[i] for i in rng.permutation(len(present))[:K - 2]] + [123456789]
        dve_n[b] = len(keys)
        dve_keys[b, :len(keys)] = keys
        smp_d = {'image_embedding': dve_img[b],
                 'panoptic_embedding': {k_: dve_emb[b, i] for i, k_ in enumerate(keys)},
                 'panoptic': r['panoptic']}
        rd = dgen(smp_d)
        dve_lut[b, :len(keys)] = rd['dense_visual_embedding_lut']
        dve_idx[b] = rd['dense_visual_embedding_indices']
    out['cleared_instance'] = cleared
    for k in gens:
        out[f'{k}__center'] = np.stack(res[k]['center'])
        out[f'{k}__offset'] = np.stack(res[k]['offset'])
        out[f'{k}__foreground'] = np.stack(res[k]['fg'])
        out[f'{k}__center_mask'] = np.stack(res[k]['cmask'])
    out['panoptic'] = np.stack(pan)
    n, kk, vv = ids_to_arrays(pan_ids, cap=256)
    out.update(pan_ids_n=n, pan_ids_pan=kk, pan_ids_ins=vv)
    out.update(dve_keys=dve_keys, dve_n=dve_n, dve_emb=dve_emb, dve_img=dve_img, dve_lut=dve_lut,
               dve_indices=dve_idx)
    # the reference asserts when an instance is skipped (stuff majority) — record that it does
    smp = {'semantic': maps['semantic'][0].copy(), 'instance': maps['instance'][0].astype(np.uint16)}
    try:
        gens['s8n'](smp)
        out['uncleared_raises'] = np.int32(0)
    except AssertionError:
        out['uncleared_raises'] = np.int32(1)
    print('    uncleared instance map raises AssertionError:', int(out['uncleared_raises']))

    # generation-time check of the restatement
    stuff = np.zeros((NC,), np.uint8)
    stuff[np.where(~maps['semantic_classes_is_thing'])[0][1:]] = 1
    o = orc.instance_targets(maps['semantic'], cleared, NC, maps['semantic_classes_is_thing'], stuff, 8, True)
    assert np.array_equal(o['center'], out['s8n__center'])
    assert np.array_equal(o['offset'], out['s8n__offset'])
    assert np.array_equal(o['foreground'], out['s8n__foreground'])
    assert np.array_equal(o['center_mask'], out['s8n__center_mask'])
    o = orc.instance_targets(maps['semantic'], cleared, NC, maps['semantic_classes_is_thing'], stuff, 3, False)
    assert np.array_equal(o['center'], out['s3u__center']) and np.array_equal(o['offset'], out['s3u__offset'])
    o = orc.instance_targets(maps['semantic'], cleared, NC, None, None, 5, True)
    assert np.array_equal(o['center'], out['s5nothing__center'])
    assert np.array_equal(o['center_mask'], out['s5nothing__center_mask'])
    p2, d2 = orc.naive_merge(maps['semantic'], cleared, 1 << 16, np.where(maps['semantic_classes_is_thing'])[0], 0)
    assert np.array_equal(p2, out['panoptic'])
    assert [list(d.items()) for d in d2] == [list(d.items()) for d in pan_ids]
    assert np.array_equal(orc.dve_indices(out['panoptic'], [dve_keys[b, :dve_n[b]] for b in range(B)]), dve_idx)
    save('target_cases', **out)


def gen_instance_post(ref):
    """InstancePostprocessing.postprocess (instance.py:337-468) with every ground-truth key:
    GT foreground, debug all-foreground, a real crop + upscale, and the four orientation dicts —
    incl. GT instance ids beyond 255 (dataset maps are uint16)."""
    print('instance postprocessing with GT keys (reference InstancePostprocessing.postprocess)')
    H, W = 96, 128
    inp = syn.make_panoptic_inputs(2, n_classes=6, height=H, width=W, n_centers=6, seed=23,
                                   with_orientation=True)
    rng = np.random.default_rng(29)
    fg = np.repeat(np.repeat(rng.random((2, H // 8, W // 8)) < 0.6, 8, 1), 8, 2)
    gt_ids = rng.choice(np.arange(1, 60000), size=12, replace=False)
    gt_inst = np.repeat(np.repeat(gt_ids[rng.integers(0, 12, (2, H // 16, W // 16))], 16, 1), 16, 2)
    gt_inst = np.where(rng.random((2, H, W)) < 0.15, 0, gt_inst).astype(np.int32)
    ori_fg = rng.random((2, H, W)) < 0.7
    crop, size = (slice(4, 92), slice(0, 128)), (150, 200)
    post = ref.post_instance.InstancePostprocessing(debug=True)
    batch = {
        'rgb_fullres': torch.zeros((2, 3) + size),
        ref.APPLIED_PREPROCESSING_KEY: [[{
            'type': 'Resize', 'valid_region_slice_y': crop[0], 'valid_region_slice_x': crop[1],
        }]] * 2,
        'instance_foreground': torch.from_numpy(fg),
        'instance': torch.from_numpy(gt_inst),
        'orientation_foreground': torch.from_numpy(ori_fg),
    }
    data = ((torch.from_numpy(inp['instance_center']), torch.from_numpy(inp['instance_offset']),
             torch.from_numpy(inp['instance_orientation'])), None)
    r = post.postprocess(data, batch, is_training=False)
    out = dict(in_center=inp['instance_center'], in_offset=inp['instance_offset'],
               in_orientation=inp['instance_orientation'], in_fg=fg, in_gt_instance=gt_inst,
               in_orientation_fg=ori_fg,
               crop=np.array([4, 92, 0, 128], np.int32), size=np.array(size, np.int32))
    for k in ('instance_segmentation_gt_foreground', 'instance_segmentation_all_foreground'):
        out[k] = r[k].numpy()
        out[k + '_fullres'] = r[k + '_fullres'].numpy()
    n, cyx, area, score = meta_to_arrays(r['instance_segmentation_gt_meta'])
    out.update(meta_n=n, meta_center_yx=cyx, meta_area=area, meta_score=score)
    for k in ('orientations_gt_instance_gt_orientation_foreground',
              'orientations_instance_segmentation_gt_orientation_foreground',
              'orientations_gt_instance', 'orientations_instance_segmentation'):
        keys = np.full((2, 64), -1, np.int64)
        vals = np.full((2, 64), np.nan, np.float32)
        for b, d in enumerate(r[k]):
            for i, (kk, vv) in enumerate(sorted(d.items())):
                keys[b, i], vals[b, i] = kk, vv
        out[k + '__ids'] = keys
        out[k + '__angles'] = vals
    save('instance_post_cases', **out)


def _to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x))
    if isinstance(x, dict):
        return {k: _to_torch(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_to_torch(v) for v in x)
    return x


def _scalars(d):
    """ordered dict of 0-d tensors / numbers -> (key list, float64 values)"""
    keys = list(d.keys())
    return jdump(keys), np.array([float(d[k]) for k in keys], np.float64)


VALIDATION_CASE = dict(B=3, C=8, H=64, W=96, n_instances=8, seed=4, max_radius=24, sigma=8)


def build_validation_case(ref):
    """ground truth through the reference's own target generators (per sample, numpy), then
    network-like predictions derived from it; everything as numpy (stored in the golden)"""
    c = VALIDATION_CASE
    B, C, H, W, NC = c['B'], c['C'], c['H'], c['W'], c['C'] + 1
    maps = syn.make_label_maps(B, NC, H, W, n_instances=c['n_instances'], seed=c['seed'],
                               max_radius=c['max_radius'])
    is_thing = tuple(bool(x) for x in maps['semantic_classes_is_thing'])        # incl. void
    clear = ref.prep_instance.InstanceClearStuffIDs(semantic_classes_is_thing=is_thing)
    igen = ref.prep_instance.InstanceTargetGenerator(sigma=c['sigma'], semantic_classes_is_thing=is_thing)
    pgen = ref.prep_panoptic.PanopticTargetGenerator(semantic_classes_is_thing=is_thing)
    ogen = ref.prep_orientation.OrientationTargetGenerator(semantic_classes_estimate_orientation=is_thing)
    rng = np.random.default_rng(77)
    out = {k: [] for k in ('instance', 'instance_center', 'instance_center_mask', 'instance_offset',
                           'instance_foreground', 'panoptic', 'orientation', 'orientation_foreground')}
    pan_ids, ori_present = [], []
    for b in range(B):
        smp = clear({'semantic': maps['semantic'][b].copy(),
                     'instance': maps['instance'][b].astype(np.uint16)})
        ids = [int(i) for i in np.unique(smp['instance']) if i != 0]
        smp['orientations'] = {i: float(rng.random() * 2 * np.pi) for i in ids if rng.random() < 0.8}
        smp = ogen(igen(pgen(smp)))
        out['instance'].append(smp['instance'].astype(np.int32))
        out['instance_center'].append(smp['instance_center'])
        out['instance_center_mask'].append(smp['instance_center_mask'])
        out['instance_offset'].append(np.moveaxis(smp['instance_offset'], -1, 0))
        out['instance_foreground'].append(smp['instance_foreground'])
        out['panoptic'].append(smp['panoptic'].astype(np.int64))
        out['orientation'].append(np.moveaxis(smp['orientation'], -1, 0))
        out['orientation_foreground'].append(smp['orientation_foreground'])
        pan_ids.append({int(k): int(v) for k, v in smp['panoptic_ids_to_instance_dict'].items()})
        ori_present.append({int(k): float(v) for k, v in smp['orientations_present'].items()})
    gt = {k: np.ascontiguousarray(np.stack(v)) for k, v in out.items()}
    gt['semantic'] = maps['semantic']
    return gt, pan_ids, ori_present, is_thing


def gen_task_helpers(ref_unused):
    """a10 / a14: the reference's task helpers themselves (task_helper/{semantic,instance,
    panoptic,dense_visual_embedding}.py) — loss dicts of a training step with main + 2 side
    outputs, and two validation steps + validation_epoch_end through the reference's
    postprocessing.  Stored: every key in order and every value."""
    from oracle.ref_loader import load_reference as _load
    ref = _load(task_helpers=True)
    print('task helpers (reference training_step / validation_step / validation_epoch_end)')
    cpu = torch.device('cpu')
    out = {}

    # ---- training: loss dicts ----------------------------------------------------------------
    batch_np, preds_np, weights = syn.make_training_case()
    out['train__digest'] = jdump(syn.input_digest(
        preds_np['semantic_output'], preds_np['instance_output'][1],
        preds_np['dense_visual_embedding_output'], batch_np['semantic'],
        batch_np['dense_visual_embedding_indices']))
    batch, preds = _to_torch(batch_np), _to_torch(preds_np)
    C = preds_np['semantic_output'].shape[1]
    is_thing = (False,) * (C // 2 + 1) + (True,) * (C - C // 2)
    cases = {
        'sem_plain': ref.th_semantic.SemanticTaskHelper(n_classes=C),
        'sem_weighted_smooth': ref.th_semantic.SemanticTaskHelper(
            n_classes=C, class_weights=weights, label_smoothing=0.1),
        'sem_single_scale': ref.th_semantic.SemanticTaskHelper(
            n_classes=C, disable_multiscale_supervision=True),
        'ins_mse': ref.th_instance.InstanceTaskHelper(C + 1, is_thing),
        'ins_l1': ref.th_instance.InstanceTaskHelper(C + 1, is_thing, loss_name_instance_center='l1'),
        'dve_cos': ref.th_dve.DenseVisualEmbeddingTaskHelper(n_classes=C, loss_name='cos_emb'),
    }
    for name, helper in cases.items():
        helper.initialize(cpu)
        losses, logs = helper.training_step(batch, 0, preds)
        out[f'train__{name}__keys'], out[f'train__{name}__values'] = _scalars(losses)
        out[f'train__{name}__log_keys'] = jdump(sorted(logs.keys()))
        print(f'  {name}: {len(losses)} losses, total '
              f'{[round(float(v), 6) for k, v in losses.items() if k.endswith("total_loss")]}')
    # instance helper without an orientation head (2-tuple outputs)
    preds2 = dict(preds, instance_output=preds['instance_output'][:2],
                  instance_side_outputs=tuple(p[:2] for p in preds['instance_side_outputs']))
    helper = ref.th_instance.InstanceTaskHelper(C + 1, is_thing)
    helper.initialize(cpu)
    losses, _ = helper.training_step(batch, 0, preds2)
    out['train__ins_no_orientation__keys'], out['train__ins_no_orientation__values'] = _scalars(losses)

    # ---- validation: postprocessing -> helpers -> epoch end ----------------------------------------
    c = VALIDATION_CASE
    B, C, H, W = c['B'], c['C'], c['H'], c['W']
    gt, pan_ids, ori_present, is_thing_nc = build_validation_case(ref)
    for k, v in gt.items():
        out[f'val__gt_{k}'] = v
    n, kk, vv = ids_to_arrays(pan_ids, cap=64)
    out.update(val__pan_ids_n=n, val__pan_ids_pan=kk, val__pan_ids_ins=vv)
    out['val__orientations_present'] = jdump([{str(k): v for k, v in d.items()} for d in ori_present])
    out['val__is_thing_with_void'] = np.array(is_thing_nc)
    is_thing_c = is_thing_nc[1:]
    post = ref.post_panoptic.PanopticPostprocessing(
        semantic_postprocessing=ref.post_semantic.SemanticPostprocessing(),
        instance_postprocessing=ref.post_instance.InstancePostprocessing(),
        semantic_classes_is_thing=is_thing_c, semantic_class_has_orientation=is_thing_c)
    batch = _to_torch(gt)
    for k in ('semantic', 'instance', 'panoptic'):
        batch[f'{k}_fullres'] = batch[k]
    batch['panoptic_ids_to_instance_dict'] = pan_ids
    batch['orientations_present'] = ori_present
    batch.update(make_batch(ref, B, H, W))
    label_list = types.SimpleNamespace(colors=None, classes_is_thing=is_thing_nc, colors_array=None)
    sem = ref.th_semantic.SemanticTaskHelper(n_classes=C)
    ins = ref.th_instance.InstanceTaskHelper(C + 1, is_thing_nc)
    pan = ref.th_panoptic.PanopticTaskHelper(C + 1, is_thing_nc, label_list)
    for h in (sem, ins, pan):
        h.initialize(cpu)
    digests = []
    for step in range(2):
        logits, center, offset, ori = syn.make_predictions_from_targets(
            gt['semantic'], gt['instance_center'], gt['instance_offset'], gt['orientation'], C,
            seed=step)
        digests.append(syn.input_digest(logits, center, offset, ori))
        data = ((torch.from_numpy(logits), (torch.from_numpy(center), torch.from_numpy(offset),
                                            torch.from_numpy(ori))), ((None, None), (None, None)))
        r = post.postprocess(data, batch, is_training=False)     # eval mode: side outputs are None
        if step == 0:
            out['val__pred_panoptic_step0'] = r['panoptic_segmentation_deeplab_fullres'].numpy()
        for name, h in (('sem', sem), ('ins', ins), ('pan', pan)):
            losses, logs = h.validation_step(batch, step, r)
            out[f'val__{name}__step{step}__loss_keys'], out[f'val__{name}__step{step}__loss_values'] = \
                _scalars(losses)
            out[f'val__{name}__step{step}__log_keys'] = jdump(sorted(logs.keys()))
    out['val__pred_digests'] = jdump(digests)
    for name, h in (('sem', sem), ('ins', ins), ('pan', pan)):
        artifacts, examples, logs = h.validation_epoch_end()
        scal = {k: v for k, v in logs.items() if k.endswith('_time') is False}
        out[f'val__{name}__log_keys'], out[f'val__{name}__log_values'] = _scalars(scal)
        out[f'val__{name}__artifact_keys'] = jdump(list(artifacts.keys()))
        for k, v in artifacts.items():
            out[f'val__{name}__artifact__{k}'] = v.numpy()
        print(f'  validation {name}: ' + ', '.join(f'{k}={float(v):.5f}' for k, v in scal.items()
                                                   if 'pq' in k or 'miou' in k or 'mae' in k)[:300])
    save('task_helper_cases', **out)


def main():
    ref = load_reference()
    only = set(sys.argv[1:])

    def want(k):
        return not only or k in only
    if want('norm'):
        check_norm_formula(ref)
    if want('panoptic'):
        gen_panoptic(ref)
    if want('edges'):
        gen_edges(ref)
    if want('centers'):
        gen_centers(ref)
    if want('grouping'):
        gen_grouping(ref)
    if want('merge'):
        gen_merge(ref)
    if want('metrics'):
        gen_metrics(ref)
    if want('losses'):
        gen_losses(ref)
    if want('loss_forms'):
        gen_loss_forms(ref)
    if want('argmax_ties'):
        gen_argmax_ties(ref)
    if want('cos_emb_large'):
        gen_cos_emb_large(ref)
    if want('orientation'):
        gen_orientation(ref)
    if want('fullres'):
        gen_fullres(ref)
    if want('scores'):
        gen_scores(ref)
    if want('instance_post'):
        gen_instance_post(ref)
    if want('targets'):
        gen_targets(ref)
    if want('task_helpers'):
        gen_task_helpers(ref)


if __name__ == '__main__':
    main()
