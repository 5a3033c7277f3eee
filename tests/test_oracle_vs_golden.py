"""
CPU tier: pins the C oracle (oracle/nmsa_oracle.c) against the golden vectors
produced by the reference's own Python (oracle/gen_golden.py) and against the
reference's known-answer PQ tables (reference tests/test_metrics.py:76-446).
"""
import numpy as np
import pytest

from _golden import load, jload, meta_from_arrays, ids_from_arrays
from nicr_mt_scene_analysis_amd.testing import synthetic as syn


# ---------------------------------------------------------------------------
# full panoptic pipeline a1..a5
def _run_pipeline_oracle(oracle, logits, center, offset, is_thing, kw=None):
    kw = dict(kw or {})
    B, C, H, W = logits.shape
    idx, score = oracle.semantic_argmax(logits)
    fg = is_thing[idx]
    cyx, n, scores, _ = oracle.center_nms_topk(
        center, fg=fg,
        threshold=kw.get('heatmap_threshold', 0.1),
        ksize=kw.get('heatmap_nms_kernel_size', 3),
        topk=kw.get('top_k_instances', 64),
        apply_fg=kw.get('heatmap_apply_foreground_mask', False),
        max_centers=512)
    inst, area = oracle.group_offsets(offset, fg, cyx, n, scale_y=H, scale_x=W,
                                      dist_thr=kw.get('offset_distance_threshold'))
    thing_ids = np.where(is_thing)[0] + 1
    pan, ids = oracle.deeplab_merge(idx + 1, inst, fg, 1 << 16, thing_ids, 0)
    return dict(idx=idx, score=score, fg=fg, cyx=cyx, n=n, scores=scores,
                inst=inst, area=area, pan=pan, ids=ids)


def _check_pipeline(g, r):
    assert (r['idx'] == g['semantic_idx']).all()
    st = int(g['semantic_score_stride']) if 'semantic_score_stride' in g else 1
    np.testing.assert_allclose(r['score'][:, ::st, ::st], g['semantic_score'],
                               rtol=1e-5, atol=1e-7)
    assert (r['fg'] == g['foreground']).all()
    assert (r['n'] == g['meta_n']).all()
    for b in range(len(r['n'])):
        nb = int(r['n'][b])
        assert (r['cyx'][b, :nb] == g['meta_center_yx'][b, :nb]).all()
        assert (r['scores'][b, :nb] == g['meta_score'][b, :nb]).all()
        assert (r['area'][b, 1:min(nb, 255) + 1] == g['meta_area'][b, :min(nb, 255)]).all()
    assert (r['inst'] == g['instance']).all()
    assert (r['pan'] == g['panoptic']).all()
    assert r['ids'] == ids_from_arrays(g['ids_n'], g['ids_pan'], g['ids_ins'])
    for rd, gd in zip(r['ids'], ids_from_arrays(g['ids_n'], g['ids_pan'], g['ids_ins'])):
        assert list(rd.items()) == list(gd.items())      # insertion order too


@pytest.mark.parametrize('name', ['panoptic_small', 'panoptic_small_kwargs',
                                  'panoptic_edges_plain', 'panoptic_edges_thr'])
def test_pipeline_small(oracle, name):
    g = load(name)
    kw = jload(g['kwargs']) if 'kwargs' in g else None
    r = _run_pipeline_oracle(oracle, g['in_semantic_logits'], g['in_instance_center'],
                             g['in_instance_offset'], g['in_semantic_classes_is_thing'], kw)
    _check_pipeline(g, r)


@pytest.mark.parametrize('name', ['panoptic_cfg1_q', 'panoptic_cfg1_r'])
def test_pipeline_cfg1(oracle, name):
    g = load(name)
    p = jload(g['params'])
    inp = syn.make_panoptic_inputs(p['batch_size'], seed=p['seed'],
                                   quantize_offsets=p['quantize_offsets'])
    digest = syn.input_digest(inp['semantic_logits'], inp['instance_center'],
                              inp['instance_offset'])
    if digest != jload(g['digest']):
        pytest.skip('synthetic inputs differ bit-wise on this host (numpy/libm)')
    r = _run_pipeline_oracle(oracle, inp['semantic_logits'], inp['instance_center'],
                             inp['instance_offset'], inp['semantic_classes_is_thing'])
    _check_pipeline(g, r)


# ---------------------------------------------------------------------------
def test_centers_adversarial(oracle):
    g = load('centers_adversarial')
    for name in jload(g['names']):
        kw = jload(g[f'{name}__kwargs'])
        heat = g[f'{name}__heat']
        fg = g[f'{name}__fg'] if f'{name}__fg' in g else None
        cyx, n, _, mask = oracle.center_nms_topk(
            heat, fg=fg,
            threshold=kw.get('heatmap_threshold', 0.1),
            ksize=kw.get('heatmap_nms_kernel_size', 3),
            topk=kw.get('top_k_instances', 64),
            apply_fg=kw.get('heatmap_apply_foreground_mask', False),
            max_centers=1024)
        assert (n == g[f'{name}__n']).all(), name
        assert (mask == g[f'{name}__mask']).all(), name
        for b in range(len(n)):
            assert (cyx[b, :n[b]] == g[f'{name}__centers'][b, :n[b]]).all(), name


def test_grouping_adversarial(oracle):
    g = load('grouping_adversarial')
    for name in jload(g['names']):
        kw = jload(g[f'{name}__kwargs'])
        heat, offset, fg = g[f'{name}__heat'], g[f'{name}__offset'], g[f'{name}__fg']
        cyx, n, scores, _ = oracle.center_nms_topk(
            heat, topk=kw.get('top_k_instances', 64), max_centers=512)
        inst, area = oracle.group_offsets(offset, fg, cyx, n,
                                          dist_thr=kw.get('offset_distance_threshold'))
        assert (inst == g[f'{name}__inst']).all(), name
        assert (n == g[f'{name}__meta_n']).all(), name
        for b in range(len(n)):
            nb = int(n[b])
            assert (cyx[b, :nb] == g[f'{name}__meta_center_yx'][b, :nb]).all(), name
            assert (scores[b, :nb] == g[f'{name}__meta_score'][b, :nb]).all(), name
            # reference: bincount(uint8 ids, minlength=n+1)[i], i = 1..n
            ref_area = g[f'{name}__meta_area'][b, :nb]
            got = np.array([area[b, i] if i <= 255 else 0 for i in range(1, nb + 1)])
            assert (got == ref_area).all(), name


def test_merge_cases(oracle):
    g = load('merge_cases')
    for name in jload(g['names']):
        p = jload(g[f'{name}__params'])
        pan, ids = oracle.deeplab_merge(g[f'{name}__sem'], g[f'{name}__ins'],
                                        g[f'{name}__thing'], p['max_inst'],
                                        p['thing_ids'], p['void'])
        assert (pan == g[f'{name}__pan']).all(), name
        want = ids_from_arrays(g[f'{name}__ids_n'], g[f'{name}__ids_pan'], g[f'{name}__ids_ins'])
        for a, b in zip(ids, want):
            assert list(a.items()) == list(b.items()), name
    # numpy twins + naive on consistent GT-style maps
    sem, ins = g['consistent__sem'], g['consistent__ins']
    pan_d, ids_d = oracle.deeplab_merge(sem, ins, ins != 0, 1 << 16, [3, 4], 0)
    pan_n, ids_n = oracle.naive_merge(sem, ins, 1 << 16, [3, 4], 0)
    assert (pan_d == g['consistent__pan']).all()
    assert (pan_n == g['consistent__pan']).all()
    want = ids_from_arrays(g['consistent__last_ids_n'], g['consistent__last_ids_pan'],
                           g['consistent__last_ids_ins'])[0]
    assert list(ids_d[-1].items()) == list(want.items())
    assert list(ids_n[-1].items()) == list(want.items())
    pan_s, ids_s = oracle.naive_merge(g['naive_split__sem'], g['naive_split__ins'],
                                      1 << 16, [3, 4], 0)
    assert (pan_s == g['naive_split__pan']).all()
    want = ids_from_arrays(g['naive_split__ids_n'], g['naive_split__ids_pan'],
                           g['naive_split__ids_ins'])[0]
    assert list(ids_s[0].items()) == list(want.items())


def test_orientation_cases(oracle):
    g = load('orientation_cases')
    for name, mask in (('masked', g['mask']), ('nomask', None)):
        res = oracle.instance_orientation(g['orientation'], g['inst'], mask)
        want = g[f'{name}__angle']
        for b, d in enumerate(res):
            present = ~np.isnan(want[b])
            assert sorted(d.keys()) == list(np.where(present)[0])
            for k, v in d.items():
                assert abs(v - want[b, k]) < 1e-5


# ---------------------------------------------------------------------------
def test_miou(oracle):
    g = load('metric_cases')
    for n in (5, 41, 101):
        pred, tgt = g[f'miou_{n}__pred'], g[f'miou_{n}__target']
        cm = oracle.confmat_update(pred[:2], tgt[:2], n)
        cm = oracle.confmat_update(pred[2:], tgt[2:], n, cm)
        for ign in (0, 1):
            assert (cm == g[f'miou_{n}_{ign}__confmat']).all()
            miou, ious = oracle.miou_compute(cm, bool(ign))
            np.testing.assert_allclose(miou, g[f'miou_{n}_{ign}__miou'], rtol=1e-5)
            np.testing.assert_allclose(ious, g[f'miou_{n}_{ign}__ious'], rtol=1e-5,
                                       equal_nan=True)


def test_pq_random(oracle):
    g = load('metric_cases')
    p = jload(g['pq_params'])
    for name in ('shift', 'indep'):
        pred, tgt = g[f'pq_{name}__pred'], g[f'pq_{name}__target']
        state = None
        all_m = []
        for b in range(pred.shape[0]):
            *state, m = oracle.pq_compare_and_accumulate(
                pred[b], tgt[b], p['num_categories'], p['ignored_label'],
                p['max_instances_per_category'], p['offset'], state=state)
            all_m.append(sorted(m))
        # counts exact, IoU sums bit-exact (same fp64 op order as the reference)
        assert (np.stack(state) == g[f'pq_{name}__state']).all()
        assert all_m == [[tuple(x) for x in im] for im in jload(g[f'pq_{name}__matches'])]


# ---------------------------------------------------------------------------
def test_losses(oracle):
    g = load('loss_cases')
    logits, tgt, w = g['in_semantic_logits'], g['in_semantic_target'], g['in_class_weights']
    for name, kw in (('plain', {}), ('weighted', dict(weights=w)),
                     ('smooth', dict(weights=w, label_smoothing=0.25)),
                     ('smooth_nw', dict(label_smoothing=0.5))):
        s, n, _, grad = oracle.loss_ce(logits, tgt, want_grad=True, **kw)
        np.testing.assert_allclose(s, g[f'ce_{name}__loss'], rtol=1e-5)
        assert n == g[f'ce_{name}__n']
        np.testing.assert_allclose(grad, g[f'ce_{name}__grad'], rtol=1e-4, atol=1e-6)
    s, n, wd, _ = oracle.loss_ce(logits, tgt, weights=w)
    np.testing.assert_allclose(s / wd, g['ce_wred__loss'], rtol=1e-5)

    for kind in ('mse', 'l1'):
        s, n, grad = oracle.loss_masked_elementwise(
            g['in_center_pred'], g['in_center_target'], g['in_center_mask'], kind, True)
        np.testing.assert_allclose(s, g[f'center_{kind}__loss'], rtol=1e-5)
        assert n == g[f'center_{kind}__n_mask']
        np.testing.assert_allclose(grad, g[f'center_{kind}__grad'], rtol=1e-5, atol=1e-7)
    s, n, grad = oracle.loss_masked_elementwise(
        g['in_offset_pred'], g['in_offset_target'], g['in_offset_mask'], 'l1', True)
    np.testing.assert_allclose(s, g['offset_l1__loss'], rtol=1e-5)
    assert n == g['offset_l1__n_mask']
    np.testing.assert_allclose(grad, g['offset_l1__grad'], rtol=1e-5, atol=1e-7)

    for kappa in (1.0, 2.5):
        s, n, grad = oracle.loss_vonmises(g['in_orientation_pred'], g['in_orientation_target'],
                                          g['in_orientation_mask'], kappa, True)
        np.testing.assert_allclose(s, g[f'vonmises_{kappa}__loss'], rtol=1e-5)
        assert n == g[f'vonmises_{kappa}__n']
        np.testing.assert_allclose(grad, g[f'vonmises_{kappa}__grad'], rtol=1e-5, atol=1e-7)

    s, n, grad = oracle.loss_cosine_embedding(g['in_embedding_pred'], g['in_embedding_indices'],
                                              g['in_embedding_lut'], True)
    np.testing.assert_allclose(s, g['cos_emb__loss'], rtol=1e-5)
    assert n == g['cos_emb__n']
    np.testing.assert_allclose(grad, g['cos_emb__grad'], rtol=1e-4, atol=1e-6)


def test_loss_forms(oracle):
    """the reductions / ranks / labelled pairs no task helper uses (reference-run fixture
    loss_forms.npz: mse.py:21-41, l1.py:21-41, cos_emb.py:21-56)"""
    g = load('loss_forms')
    for kind in ('mse', 'l1'):
        for rk in ('r2', 'r3', 'r4'):
            for red in ('none', 'sum', 'mean'):
                loss, n, grad = oracle.loss_elementwise_form(g[f'{rk}__x'], g[f'{rk}__t'], kind, red,
                                                             g[f'{rk}__w'])
                key = f'{kind}_{rk}_{red}'
                np.testing.assert_allclose(loss, g[key + '__loss'], rtol=2e-6, atol=1e-7, err_msg=key)
                assert n == g[key + '__n'], key
                np.testing.assert_allclose(grad, g[key + '__grad'], rtol=1e-5, atol=1e-7, err_msg=key)
    for lab in ('labelled', 'plain'):
        for red in ('none', 'sum', 'mean'):
            loss, n, grad = oracle.loss_cosine_rows(g['cos__x'], g['cos__t'],
                                                    g['cos__labels'] if lab == 'labelled' else None,
                                                    red, g['cos__w'])
            key = f'cos_{lab}_{red}'
            np.testing.assert_allclose(loss, g[key + '__loss'], rtol=1e-5, atol=1e-6, err_msg=key)
            assert n == g[key + '__n'], key
            np.testing.assert_allclose(grad, g[key + '__grad'], rtol=1e-4, atol=1e-6, err_msg=key)


def test_argmax_tie_band_boundary(oracle):
    """a1: the reference takes max(softmax(x)) (semantic.py:52-53).  softmax is monotone, so that
    is argmax(x) with first-index ties — except where a lower-indexed class sits so close below
    the maximum that ATen's fp32 softmax gives both the same probability.  The oracle decides
    those columns with ATen's own arithmetic (Sleef expf_u10, sequential fp32 sum, IEEE division).
    tests/golden/argmax_ties.npz (reference-run): 1680 adversarial columns with the top-2 logits
    delta apart (c1 < c2, x[c2] > x[c1]) — EVERY column must come out as the reference's, the 229
    with delta in (2^-25, 2^-23] included (38 of them return the lower index) —, two whole maps
    ('tiny': every class within 2^-25, 'small': 4 classes per pixel on a 2^-26 grid) and their
    full-resolution twins (the same rule on the bilinearly interpolated logits)."""
    from _golden import aten_softmax_argmax, jload
    g = load('argmax_ties')
    # the tie rule is that of ONE ATen build — the one that ran the reference for this fixture
    # (recorded in it): its CPU softmax (Sleef expf_u10, vectorised fp32 summation).  Oracle and
    # kernels restate that arithmetic; they do not call torch, so the check holds on any box
    producer = jload(g['producer'])
    assert producer['torch'].startswith('2.10') and producer['cpu_capability'] == 'AVX512', producer
    idx, _ = oracle.semantic_argmax(g['logits'])
    idx, ref = idx.reshape(-1), g['ref_idx'].reshape(-1)
    delta, c1, c2 = g['delta'], g['c1'], g['c2']
    outside = delta > 2.0 ** -23
    assert outside.sum() > 1000 and (ref[outside] == c2[outside]).all()
    zone_a = delta <= 2.0 ** -25
    assert zone_a.sum() > 300 and (ref[zone_a] == c1[zone_a]).all()
    zone_b = ~outside & ~zone_a
    assert zone_b.sum() == 229 and (ref[zone_b] == c1[zone_b]).sum() == 38
    assert (idx == ref).all()                                 # all 1680, the in-between ones too
    assert (aten_softmax_argmax(g['logits'])[0].reshape(-1) == ref).all()     # numpy twin
    assert int(g['natural_blobby'][1]) == 0                # bench-like logits: never observed
    assert int(g['natural_small'][1]) <= 20 and int(g['natural_small'][0]) > 8_000_000
    for name in ('tiny', 'small'):
        x, ref = g[f'{name}_logits'], g[f'{name}_ref_idx']
        got, _ = oracle.semantic_argmax(x)
        assert (got == ref).all(), name
        assert (aten_softmax_argmax(x)[0] == ref).all(), name
        assert (ref != x.argmax(axis=1)).sum() > 900, name  # the rule matters on these maps
        ref_full = g[f'{name}_ref_idx_fullres']
        up = oracle.resize_bilinear(x, ref_full.shape[-2:], None)
        got_full, _ = oracle.semantic_argmax(up)
        assert (got_full == ref_full).all(), name


def test_argmax_probability_of_the_maximum_and_far_classes(oracle):
    """the restated softmax arithmetic away from the ties: classes 100 and more below the maximum
    (Sleef's cut-off at -104, the two-factor ldexp), -inf entries, ordinary columns — oracle index
    == numpy twin == plain argmax, and the twin's maximum probability equals torch's softmax"""
    import torch
    from _golden import aten_softmax_argmax
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, 19, 16, 16)) * 30).astype(np.float32)
    x[0, 3, :4] = -np.inf
    x[1, :, 5, 5] = rng.standard_normal(19).astype(np.float32) * 1e-3 - 110.0
    idx, _ = oracle.semantic_argmax(x)
    twin, pmax = aten_softmax_argmax(x)
    assert (idx == twin).all() and (idx == x.argmax(axis=1)).all()
    want = torch.softmax(torch.from_numpy(x), dim=1).max(dim=1)[0].numpy()
    np.testing.assert_allclose(pmax, want, rtol=2e-7)


def test_cosine_embedding_large_dims(oracle):
    """a9 at the dense-visual-embedding sizes (D = 512 / 768, L up to 64 and a 1300-row LUT):
    the C oracle against the reference's loss + autograd gradient"""
    from _golden import cos_emb_large_cases, check_cos_emb_large_grad
    for name, p, inp, g in cos_emb_large_cases():
        s, n, grad = oracle.loss_cosine_embedding(inp['embedding_pred'], inp['embedding_indices'],
                                                  inp['embedding_lut'], True)
        np.testing.assert_allclose(s, g[f'{name}__loss'], rtol=1e-5, err_msg=name)
        assert n == g[f'{name}__n']
        check_cos_emb_large_grad(name, p, g, grad, rtol=1e-4, atol=1e-7, sums_rtol=1e-5)


# ---------------------------------------------------------------------------
# f2: crop + resize to the dataset resolution (dense_base.py:15-58)
def _fullres_geoms(g):
    out = {}
    for name, (c, size) in jload(g['geoms']).items():
        out[name] = ((slice(c[0], c[1]), slice(c[2], c[3])), tuple(size))
    return out


def test_fullres_resize_cases(oracle):
    g = load('fullres_cases')
    n = 0
    for name, (crop, size) in _fullres_geoms(g).items():
        for k in ('u8', 'bool', 'i32', 'i64', 'f32'):
            key = f'{name}__nearest_{k}'
            if key not in g:
                continue
            got = oracle.resize_nearest(g[f'in__{k}'], size, crop)
            assert got.dtype == g[key].dtype and np.array_equal(got, g[key]), key
            n += 1
        got = oracle.resize_bilinear(g['in__logits'], size, crop)
        assert np.array_equal(got, g[f'{name}__bilinear']), name      # bit-exact
        n += 1
    assert n >= 20


def test_fullres_panoptic(oracle):
    """the reference's PanopticPostprocessing with a real crop + upscale (96x128 -> 150x200)"""
    g = load('fullres_panoptic')
    inp = syn.make_panoptic_inputs(2, n_classes=40, height=96, width=128, n_centers=9, seed=77)
    if syn.input_digest(inp['semantic_logits'], inp['instance_center'],
                        inp['instance_offset']) != jload(g['input_digest']):
        pytest.skip('synthetic inputs differ bit-wise on this host (numpy/libm)')
    c = g['crop']
    crop, size = (slice(int(c[0]), int(c[1])), slice(int(c[2]), int(c[3]))), tuple(g['size'])
    r = _run_pipeline_oracle(oracle, inp['semantic_logits'], inp['instance_center'],
                             inp['instance_offset'], inp['semantic_classes_is_thing'])
    lf = oracle.resize_bilinear(inp['semantic_logits'], size, crop)
    assert np.array_equal(lf[:, ::13], g['semantic_output_fullres'])
    idx, score = oracle.semantic_argmax(lf)
    assert np.array_equal(idx, g['semantic_segmentation_idx_fullres'])
    np.testing.assert_allclose(score, g['semantic_segmentation_score_fullres'],
                               rtol=1e-5, atol=1e-7)
    assert np.array_equal(oracle.resize_nearest(r['pan'], size, crop),
                          g['panoptic_segmentation_deeplab_fullres'])
    assert np.array_equal(oracle.resize_nearest(r['inst'].astype(np.uint8), size, crop),
                          g['panoptic_segmentation_deeplab_instance_idx_fullres'])
    pan_sem = np.where(r['pan'] > 0, r['pan'] // (1 << 16), 0)
    assert np.array_equal(oracle.resize_nearest(pan_sem.astype(np.int64), size, crop),
                          g['panoptic_segmentation_deeplab_semantic_idx_fullres'])


# ---------------------------------------------------------------------------
# f3: compute_scores (panoptic.py:171-239)
def test_scores_cases(oracle):
    g = load('scores_cases')
    ids = ids_from_arrays(g['ids_n'], g['ids_pan'], g['ids_ins'])
    sem, ins, pns, mean = oracle.panoptic_scores(g['in_semantic_logits'], g['panoptic_semantic'],
                                                 g['panoptic'], ids, g['inst_score_by_id'])
    np.testing.assert_allclose(sem, g['semantic_score'], rtol=1e-5, atol=1e-7)
    assert np.array_equal(ins, g['instance_score'])
    np.testing.assert_allclose(pns, g['panoptic_score'], rtol=1e-5, atol=1e-7)
    used = g['meta_panoptic_id'] >= 0
    np.testing.assert_allclose(mean[used], g['meta_semantic_score'][used], rtol=1e-5)
    # the pipeline feeding it is the pinned one
    r = _run_pipeline_oracle(oracle, g['in_semantic_logits'], g['in_instance_center'],
                             g['in_instance_offset'], g['in_semantic_classes_is_thing'])
    assert np.array_equal(r['pan'], g['panoptic']) and r['ids'] == ids


# ---------------------------------------------------------------------------
# f4: target generation (data/preprocessing/{instance,panoptic,dense_visual_embedding}.py)
def _stuff_lut(is_thing):
    st = np.zeros((len(is_thing),), np.uint8)
    st[np.where(~is_thing)[0][1:]] = 1
    return st


def test_target_cases(oracle):
    g = load('target_cases')
    sem, ins, is_thing = g['in_semantic'], g['cleared_instance'], g['in_is_thing']
    NC = len(is_thing)
    for name, kw in (('s8n', dict(is_thing=is_thing, is_stuff=_stuff_lut(is_thing), sigma=8, normalized=True)),
                     ('s3u', dict(is_thing=is_thing, is_stuff=_stuff_lut(is_thing), sigma=3, normalized=False)),
                     ('s5nothing', dict(sigma=5, normalized=True))):
        o = oracle.instance_targets(sem, ins, NC, **kw)
        assert np.array_equal(o['center'], g[f'{name}__center']), name          # bit-exact
        assert o['offset'].dtype == g[f'{name}__offset'].dtype
        assert np.array_equal(o['offset'], g[f'{name}__offset']), name
        assert np.array_equal(o['foreground'], g[f'{name}__foreground']), name
        assert np.array_equal(o['center_mask'], g[f'{name}__center_mask']), name
        assert all(len(s) == 0 for s in o['skipped'])
    # an uncleared map has stuff-majority instances: the reference asserts, the oracle lists them
    assert int(g['uncleared_raises']) == 1
    o = oracle.instance_targets(sem, g['in_instance'], NC, is_thing=is_thing, sigma=8)
    assert any(len(s) for s in o['skipped'])
    pan, dicts = oracle.naive_merge(sem, ins, 1 << 16, np.where(is_thing)[0], 0)
    assert np.array_equal(pan, g['panoptic'])
    want = ids_from_arrays(g['pan_ids_n'], g['pan_ids_pan'], g['pan_ids_ins'])
    assert [list(d.items()) for d in dicts] == [list(d.items()) for d in want]
    keys = [g['dve_keys'][b, :g['dve_n'][b]] for b in range(len(sem))]
    assert np.array_equal(oracle.dve_indices(g['panoptic'], keys), g['dve_indices'])


def test_task_helper_loss_dicts_from_oracle_pieces(oracle):
    """a10 on the CPU tier: the reference task helpers' loss dicts (tests/golden/task_helper_cases)
    rebuilt from the C oracle's loss functions with the helpers' conventions — per-scale
    `loss / n`, total = sum(loss) / sum(n) (task_helper/base.py:161-182), center `pred * mask` with
    n = sum(mask) (instance.py:129-139), offset n = sum(foreground) (:154-167), orientation
    n = max(sum(mask), 1) (:206-211)"""
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn
    g = load('task_helper_cases')
    batch, preds, weights = syn.make_training_case()
    scales = [('main', None), ('down_2', 2), ('down_4', 4)]

    def want(name):
        return dict(zip(jload(g[f'train__{name}__keys']), g[f'train__{name}__values']))

    def tgt(key, s):
        return batch[key] if s is None else batch[f'_down_{s}'][key]

    def pred(key, s, i=None):
        x = preds[key] if s is None else preds[key.replace('_output', '_side_outputs')][scales_idx[s]]
        return x if i is None else x[i]
    scales_idx = {2: 0, 4: 1}

    # semantic: plain and weighted + label smoothing
    for name, kw in (('sem_plain', {}), ('sem_weighted_smooth', dict(weights=weights, label_smoothing=0.1))):
        w, tot_l, tot_n = want(name), 0.0, 0
        for key, s in scales:
            l, n, _, _ = oracle.loss_ce(pred('semantic_output', s), tgt('semantic', s), **kw)
            np.testing.assert_allclose(l / n, w[f'semantic_loss_{key}'], rtol=1e-5, err_msg=f'{name} {key}')
            tot_l, tot_n = tot_l + l, tot_n + n
        np.testing.assert_allclose(tot_l / tot_n, w['semantic_total_loss'], rtol=1e-5)

    # instance: center (mse / l1), offset (l1), orientation (von Mises)
    for name, kind in (('ins_mse', 'mse'), ('ins_l1', 'l1')):
        w = want(name)
        acc = {k: [0.0, 0] for k in ('center', 'offset', 'orientation')}
        for key, s in scales:
            l, n, _ = oracle.loss_masked_elementwise(pred('instance_output', s, 0)[:, 0],
                                                     tgt('instance_center', s),
                                                     tgt('instance_center_mask', s), kind)
            np.testing.assert_allclose(l / n, w[f'instance_center_loss_{key}'], rtol=1e-5)
            acc['center'][0] += l; acc['center'][1] += n
            l, n, _ = oracle.loss_masked_elementwise(pred('instance_output', s, 1), tgt('instance_offset', s),
                                                     tgt('instance_foreground', s), 'l1')
            np.testing.assert_allclose(l / n, w[f'instance_offset_loss_{key}'], rtol=1e-5)
            acc['offset'][0] += l; acc['offset'][1] += n
            l, n, _ = oracle.loss_vonmises(pred('instance_output', s, 2), tgt('orientation', s),
                                           tgt('orientation_foreground', s), 1.0)
            n = max(n, 1)
            np.testing.assert_allclose(l / n, w[f'instance_orientation_loss_{key}'], rtol=1e-5)
            acc['orientation'][0] += l; acc['orientation'][1] += n
        for k, (l, n) in acc.items():
            np.testing.assert_allclose(l / n, w[f'instance_{k}_total_loss'], rtol=1e-5, err_msg=f'{name} {k}')


def test_validation_logs_from_oracle_pieces(oracle):
    """a14 on the CPU tier: the reference's SemanticTaskHelper / PanopticTaskHelper validation
    artifacts and logs (two steps + epoch end, tests/golden/task_helper_cases) rebuilt from the
    C oracle: argmax -> center NMS -> grouping -> merge, confusion matrices (bit-exact), PQ states
    and the PQ / SQ / RQ / mIoU values computed from them"""
    import torch
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn
    g = load('task_helper_cases')
    gt = {k[len('val__gt_'):]: g[k] for k in g.files if k.startswith('val__gt_')}
    is_thing_nc = g['val__is_thing_with_void']
    NC = len(is_thing_nc)
    C = NC - 1
    is_thing = is_thing_nc[1:]
    B, H, W = gt['semantic'].shape
    cm_sem = np.zeros((C, C), np.int64)
    cm_pan = np.zeros((NC, NC), np.int64)
    state = None
    for step in range(2):
        logits, center, offset, _ = syn.make_predictions_from_targets(
            gt['semantic'], gt['instance_center'], gt['instance_offset'], gt['orientation'], C, seed=step)
        idx, _ = oracle.semantic_argmax(logits)
        # SemanticTaskHelper.validation_step (semantic.py:124-128): void pixels masked out
        m = gt['semantic'] != 0
        cm_sem = oracle.confmat_update(idx[m], gt['semantic'][m] - 1, C, cm_sem)
        fg = is_thing[idx]
        cyx, n, _, _ = oracle.center_nms_topk(center, max_centers=256)
        inst, _ = oracle.group_offsets(offset, fg, cyx, n, scale_y=H, scale_x=W)
        pan, _ = oracle.deeplab_merge(idx + 1, inst, fg, 1 << 16, np.where(is_thing)[0] + 1, 0)
        if step == 0:
            assert (pan == g['val__pred_panoptic_step0']).all()
        # PanopticTaskHelper.validation_step (panoptic.py:104-126)
        for b in range(B):
            *state, _ = oracle.pq_compare_and_accumulate(pan[b], gt['panoptic'][b], NC, 0, 1 << 16,
                                                         256 ** 3, state=state)
        cm_pan = oracle.confmat_update(pan // 65536, gt['semantic'], NC, cm_pan)
    assert (cm_sem == g['val__sem__artifact__semantic_cm']).all()
    assert (cm_pan == g['val__pan__artifact__panoptic_deeplab_semantic_cm']).all()

    logs = dict(zip(jload(g['val__sem__log_keys']), g['val__sem__log_values']))
    miou = MeanIntersectionOverUnion(C, device='cpu')
    miou.confmat += torch.from_numpy(cm_sem)
    np.testing.assert_allclose(float(miou.compute()), logs['semantic_miou'], rtol=1e-6)

    logs = dict(zip(jload(g['val__pan__log_keys']), g['val__pan__log_values']))
    miou = MeanIntersectionOverUnion(NC, ignore_first_class=True, device='cpu')
    miou.confmat += torch.from_numpy(cm_pan)
    np.testing.assert_allclose(float(miou.compute()), logs['panoptic_deeplab_semantic_miou'], rtol=1e-6)
    pq = PanopticQuality(NC, 0, 1 << 16, 256 ** 3, [bool(t) for t in is_thing_nc], device='cpu')
    for name, vec in zip(('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class'), state):
        getattr(pq, name).add_(torch.from_numpy(vec))
    res = pq.compute(suffix='_deeplab')
    for k, v in logs.items():
        short = k[len('panoptic_'):]
        if short in res:
            np.testing.assert_allclose(float(res[short]), v, rtol=1e-12, err_msg=k)
    for k in ('sq_per_class', 'rq_per_class', 'pq_per_class'):
        np.testing.assert_allclose(res[k].numpy(), g[f'val__pan__artifact__panoptic_{k}'], rtol=1e-12)
