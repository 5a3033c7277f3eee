"""
GPU tier (pytest -m gpu): randomised parity of the HIP hot path against the C oracle on small
adversarial inputs — quantised logits / heat-maps / offsets so that EXACT ties (equal logits,
plateaus in the heat-map, equidistant centers, equal top-k values) are common, odd shapes, tiny
top-k, every heat-map kernel size, optional foreground masking and distance threshold.
The oracle itself is pinned to the reference by tests/test_oracle_vs_golden.py.
"""
import os

import numpy as np
import pytest
import torch

from _golden import ids_from_arrays

pytestmark = pytest.mark.gpu

hypothesis = pytest.importorskip('hypothesis')
from hypothesis import example, given, settings, strategies as st, HealthCheck   # noqa: E402


# NMSA_FUZZ_SCALE=k: k times the examples, drawn at random (bug hunting on the GPU box); the
# default run is derandomized so that the suite is reproducible
_SCALE = int(os.environ.get('NMSA_FUZZ_SCALE', '1'))
_DERANDOMIZE = _SCALE == 1


_EFFECTIVE = {}


def _n(examples):
    return examples * _SCALE


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@st.composite
def cases(draw, medium=False):
    B = draw(st.integers(1, 3))
    if medium:      # several 1024-px chunks per image, vector and scalar paths, many classes
        C = draw(st.integers(2, 48))
        H = draw(st.integers(40, 150))
        W = draw(st.sampled_from([64, 96, 128, 160, 61, 99, 130, 203]))
    else:
        C = draw(st.integers(2, 9))
        H = draw(st.integers(6, 33))
        W = draw(st.integers(6, 40))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    levels = draw(st.sampled_from([2, 3, 5, 17]))            # few logit levels -> ties
    heat_levels = draw(st.sampled_from([3, 6, 64]))
    off_q = draw(st.sampled_from([1.0, 0.5, 0.125]))          # offsets on a px grid -> equal distances
    ksize = draw(st.sampled_from([3, 5, 7]))
    topk = draw(st.integers(1, 70 if medium else 6))
    thr = draw(st.sampled_from([0.0, 0.1, 0.34, 0.9]))
    apply_fg = draw(st.booleans())
    dist_thr = draw(st.sampled_from([None, 0.0, 2.0, 7.5]))
    n_thing = draw(st.integers(0, C))
    dtype = draw(st.sampled_from(['float32', 'bfloat16', 'float16']))
    specials = draw(st.sampled_from([False, False, False, True]))
    # f32 only: logit levels 2^-26 apart instead of 1 apart — neighbouring levels get the SAME
    # softmax probability and the lowest index wins (a1 probability ties)
    near_ties = draw(st.sampled_from([False, False, True])) and dtype == 'float32'
    return dict(near_ties=near_ties, specials=specials, dtype=dtype, B=B, C=C, H=H, W=W, seed=seed, levels=levels, heat_levels=heat_levels, off_q=off_q,
                ksize=ksize, topk=topk, thr=thr, apply_fg=apply_fg, dist_thr=dist_thr, n_thing=n_thing)


def make_inputs(p):
    rng = np.random.default_rng(p['seed'])
    B, C, H, W = p['B'], p['C'], p['H'], p['W']
    logits = rng.integers(0, p['levels'], (B, C, H, W)).astype(np.float32)
    if p.get('near_ties'):
        logits = (logits * np.float32(2.0 ** -26)).astype(np.float32)
    heat = (rng.integers(0, p['heat_levels'], (B, 1, H, W)) / (p['heat_levels'] - 1)).astype(np.float32)
    # offsets: multiples of off_q pixels, normalised by (H, W) like the network head
    off_px = rng.integers(-8, 9, (B, 2, H, W)) * p['off_q']
    offset = np.stack([off_px[:, 0] / H, off_px[:, 1] / W], 1).astype(np.float32)
    is_thing = np.zeros((C,), bool)
    is_thing[rng.permutation(C)[:p['n_thing']]] = True
    if p.get('specials'):       # non-finite logits: softmax-then-max semantics (index 0, NaN score)
        for _ in range(int(rng.integers(1, 6))):
            logits[rng.integers(B), rng.integers(C), rng.integers(H), rng.integers(W)] = \
                rng.choice([np.nan, np.inf, -np.inf])
        if rng.random() < 0.3:  # a column of nothing but -inf
            logits[rng.integers(B), :, rng.integers(H), rng.integers(W)] = -np.inf
    return logits, heat, offset, is_thing


@settings(max_examples=_n(300), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(p=cases())
def test_fuzz_pipeline_vs_oracle(oracle, p):
    check_pipeline(oracle, p)


@settings(max_examples=_n(25), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), C=st.integers(2, 48), scale=st.sampled_from([1.0, 0.5, 0.03, 1e-4]),
       spread=st.sampled_from([0.2, 2.0, 8.0, 40.0, 120.0]), dtype=st.sampled_from(['float32', 'float32', 'bfloat16', 'float16']))
def test_fuzz_argmax_probability_ties(oracle, seed, C, scale, spread, dtype):
    """a1 where it is decided by ATen's fp32 softmax arithmetic: columns whose maximum lies in
    (-scale, scale) with 1-3 lower-indexed classes 1..6 ulps below it and the other classes up to
    `spread` below (every column its own softmax denominator; 120: beyond Sleef's cut-off) — the
    kernels (stand-alone, with score, fused) against the C oracle, two independent restatements
    of the same arithmetic, and a sample against the numpy twin"""
    from _golden import aten_softmax_argmax
    from nicr_mt_scene_analysis_amd import ops
    rng = np.random.default_rng(seed)
    H, W = 24, 64
    tdt = getattr(torch, dtype)
    top = ((rng.random((1, 1, H, W)) * 2 - 1) * scale).astype(np.float32)
    top = torch.from_numpy(top).to(tdt).float().numpy()                   # representable maxima
    x = top - (rng.random((1, C, H, W)) * spread + 0.05 * spread).astype(np.float32)
    x = torch.from_numpy(x).to(tdt).float().numpy()
    am = rng.integers(0, C, (1, 1, H, W))
    np.put_along_axis(x, am, top, axis=1)
    info = {'float32': np.float32, 'bfloat16': None, 'float16': np.float16}[dtype]
    for _ in range(3):                                                    # near-maximum classes
        c = rng.integers(0, C, (1, 1, H, W))
        k = rng.integers(1, 7, (1, 1, H, W))
        if info is not None:
            near = top.astype(info)
            for _k in range(6):
                near = np.where(k > _k, np.nextafter(near, info(-np.inf)), near)
            near = near.astype(np.float32)
        else:                                                             # bf16: step the 16-bit pattern
            bits = (top.view(np.uint32) >> 16).astype(np.int64)
            sign = np.where(top < 0, 1, -1) * np.where(top == 0, 0, 1)
            bits = np.where(top == 0, 0x8000 + k, bits + sign * k)        # below +0: the negatives
            near = ((bits.astype(np.uint32) & 0xFFFF) << 16).view(np.float32).reshape(top.shape)
        keep = (c != am) & (near < top)
        np.put_along_axis(x, c, np.where(keep, near, np.take_along_axis(x, c, axis=1)), axis=1)
    xt = torch.from_numpy(x).to(tdt)
    assert torch.equal(xt.float(), torch.from_numpy(x))
    want, _ = oracle.semantic_argmax(x)
    want = want.astype(np.uint8)
    xd = xt.cuda()
    got = [ops.semantic_argmax(xd, want_u8=True, want_i64=False, want_score=False)['idx_u8'],
           ops.semantic_argmax(xd, want_u8=True, want_i64=False, want_score=True)['idx_u8'],
           ops.panoptic_pipeline(xd, torch.zeros((1, 1, H, W), device='cuda'),
                                 torch.zeros((1, 2, H, W), device='cuda'),
                                 torch.zeros((C,), dtype=torch.bool, device='cuda'))['semantic_idx_u8']]
    for g in got:
        assert (g.cpu().numpy() == want).all()
    twin, _ = aten_softmax_argmax(x[:, :, :4])
    assert (twin == want[:, :4]).all()
    _EFFECTIVE['argmax_ties'] = _EFFECTIVE.get('argmax_ties', 0) + int((want != x.argmax(axis=1)).sum())


@settings(max_examples=_n(40), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture,
                                 HealthCheck.data_too_large])
@given(p=cases(medium=True))
def test_fuzz_pipeline_medium_shapes_vs_oracle(oracle, p):
    """same check on images of several thousand pixels: more than one 1024-px chunk per image,
    W % 32 == 0 band NMS kernels, vector / scalar paths, up to 48 classes and 70 centers"""
    check_pipeline(oracle, p, max_centers=4096)


def check_pipeline(oracle, p, max_centers=1024):
    from nicr_mt_scene_analysis_amd import ops
    logits, heat, offset, is_thing = make_inputs(p)
    B, C, H, W = logits.shape
    idx, score = oracle.semantic_argmax(logits)
    fg = is_thing[idx]
    try:
        cyx, n, scores, _ = oracle.center_nms_topk(heat, fg=fg, threshold=p['thr'], ksize=p['ksize'],
                                                   topk=p['topk'], apply_fg=p['apply_fg'],
                                                   max_centers=max_centers)
    except oracle.OracleError:
        return          # more tied centers than the table holds (the API re-runs with a larger one)
    # the logit levels are small integers: exactly representable in bf16 / f16
    r = ops.panoptic_pipeline(
        dev(logits).to(getattr(torch, p['dtype'])), dev(heat), dev(offset), dev(is_thing),
        threshold=p['thr'],
        kernel_size=p['ksize'], top_k=p['topk'], apply_foreground_mask=p['apply_fg'],
        distance_threshold=p['dist_thr'], want_score=True, want_panoptic_semantic=True,
        max_centers=max_centers)
    torch.cuda.synchronize()
    r = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in r.items()}
    assert (r['semantic_idx_u8'] == idx).all(), p
    np.testing.assert_allclose(r['semantic_score'], score, rtol=1e-5, atol=1e-7, equal_nan=True)
    assert (r['foreground'] == fg).all(), p
    assert (r['n_centers'] == n).all(), p
    for b in range(B):
        assert (r['centers_yx'][b, :n[b]] == cyx[b, :n[b]]).all(), p
        assert (r['center_scores'][b, :n[b]] == scores[b, :n[b]]).all(), p
    if n.max() > 255:
        return                      # uint8 wrap of the ids is pinned by the golden fixture only
    inst, area = oracle.group_offsets(offset, fg, cyx, n, scale_y=H, scale_x=W, dist_thr=p['dist_thr'])
    assert (r['instance'] == inst).all(), p
    pan, ids = oracle.deeplab_merge(idx + 1, inst, fg, 1 << 16, np.where(is_thing)[0] + 1, 0)
    assert (r['panoptic'] == pan).all(), p
    got = ids_from_arrays(r['n_ids'], r['ids_pan'], r['ids_ins'])
    assert [list(d.items()) for d in got] == [list(d.items()) for d in ids], p


@settings(max_examples=_n(120), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 2), H=st.integers(1, 21),
       W=st.sampled_from([32, 64, 96, 256, 288, 320, 544]), ksize=st.sampled_from([3, 3, 3, 5, 9]),
       levels=st.sampled_from([2, 3, 7, 64]), thr=st.sampled_from([-0.5, 0.0, 0.1, 0.6]),
       topk=st.integers(1, 40), n_nan=st.integers(0, 3))
def test_fuzz_center_nms_word_aligned_widths(oracle, seed, B, H, W, ksize, levels, thr, topk, n_nan):
    """W % 32 == 0 takes the band kernels (3x3: k_nms_rows3, else k_nms_strip): plateaus, peaks
    on band / image borders, NaNs, a negative threshold (the zero-padded-pool rule for pixel 0)"""
    from nicr_mt_scene_analysis_amd import ops
    rng = np.random.default_rng(seed)
    heat = (rng.integers(0, levels, (B, 1, H, W)) / max(levels - 1, 1)).astype(np.float32)
    heat[:, :, :, ::max(W // 7, 1)] = rng.choice([0.0, 0.5, 1.0])       # columns incl. band edges
    if rng.random() < 0.3:
        heat[:, 0, 0, 0] = 0.0
    for _ in range(n_nan):
        heat[rng.integers(B), 0, rng.integers(H), rng.integers(W)] = np.nan
    if topk > H * W:                    # torch.topk raises (instance.py:130): so do both sides
        with pytest.raises(Exception):
            oracle.center_nms_topk(heat, threshold=thr, ksize=ksize, topk=topk, max_centers=4096)
        with pytest.raises(Exception):
            ops.center_nms_topk(dev(heat), threshold=thr, kernel_size=ksize, top_k=topk,
                                max_centers=4096, want_mask=True)
        return
    cyx, n, scores, mask = oracle.center_nms_topk(heat, threshold=thr, ksize=ksize, topk=topk,
                                                  max_centers=4096)
    r = ops.center_nms_topk(dev(heat), threshold=thr, kernel_size=ksize, top_k=topk,
                            max_centers=4096, want_mask=True)
    assert (r['n_centers'].cpu().numpy() == n).all()
    assert (r['center_mask'].cpu().numpy() == mask).all()
    got = r['centers_yx'].cpu().numpy()
    for b in range(B):
        assert (got[b, :n[b]] == cyx[b, :n[b]]).all()


@settings(max_examples=_n(150), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 3), H=st.integers(4, 30), W=st.integers(4, 37),
       n_cat=st.integers(2, 7), n_seg=st.integers(1, 9))
@example(seed=7, B=3, H=4, W=31, n_cat=2, n_seg=7)     # IoU sums: per image first, then the states
def test_fuzz_metrics_vs_oracle(oracle, seed, B, H, W, n_cat, n_seg):
    """random blocky panoptic maps: PQ states and confusion matrix bit-exact vs the oracle"""
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality
    rng = np.random.default_rng(seed)

    def pan_map():
        out = np.zeros((B, H, W), np.int64)
        for b in range(B):
            for _ in range(n_seg):
                y0, x0 = rng.integers(0, H), rng.integers(0, W)
                y1, x1 = rng.integers(y0, H) + 1, rng.integers(x0, W) + 1
                cat = rng.integers(0, n_cat)
                out[b, y0:y1, x0:x1] = cat * 65536 + (rng.integers(0, 4) if cat else 0)
        return out
    pred, tgt = pan_map(), pan_map()
    is_thing = [False] + [bool(rng.integers(0, 2)) for _ in range(n_cat - 1)]
    pq = PanopticQuality(n_cat, 0, 1 << 16, 256 ** 3, is_thing, device='cuda')
    pq.update(dev(pred), dev(tgt))
    state = None
    for b in range(B):
        *state, _ = oracle.pq_compare_and_accumulate(pred[b], tgt[b], n_cat, 0, 1 << 16, 256 ** 3,
                                                     state=state)
    torch.cuda.synchronize()
    got = [pq.iou_per_class, pq.tp_per_class, pq.fn_per_class, pq.fp_per_class]
    for g, w in zip(got, state):
        assert np.array_equal(g.cpu().numpy(), np.asarray(w, dtype=np.float64)), (seed, B, H, W)
    # the matched (target id, prediction id) pairs the orientation MAE walks (mae.py:129-162)
    from nicr_mt_scene_analysis_amd.metric.pq import compare_and_accumulate
    *_, want_matches = oracle.pq_compare_and_accumulate(pred[0], tgt[0], n_cat, 0, 1 << 16, 256 ** 3)
    *_, got_matches = compare_and_accumulate(dev(pred[0]), dev(tgt[0]), n_cat, 0, 1 << 16, 256 ** 3, 0)
    assert got_matches == set(want_matches)
    miou = MeanIntersectionOverUnion(n_cat, device='cuda')
    miou.update(dev(pred // 65536), dev(tgt // 65536))
    cm = oracle.confmat_update(pred // 65536, tgt // 65536, n_cat)
    assert np.array_equal(miou.confmat.cpu().numpy(), cm)
    # the fused form (one pass over the prediction for both metrics)
    tsem = rng.integers(0, n_cat, (B, H, W)).astype(np.uint8)
    pq2 = PanopticQuality(n_cat, 0, 1 << 16, 256 ** 3, is_thing, device='cuda')
    miou2 = MeanIntersectionOverUnion(n_cat, device='cuda')
    pq2.update_with_miou(dev(pred), dev(tgt), miou2, dev(tsem), 65536)
    torch.cuda.synchronize()
    for g2, w in zip([pq2.iou_per_class, pq2.tp_per_class, pq2.fn_per_class, pq2.fp_per_class], state):
        assert np.array_equal(g2.cpu().numpy(), np.asarray(w, dtype=np.float64)), (seed, 'fused')
    assert np.array_equal(miou2.confmat.cpu().numpy(), oracle.confmat_update(pred // 65536, tsem, n_cat))


@settings(max_examples=_n(120), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 3), H=st.integers(3, 70),
       W=st.sampled_from([5, 16, 31, 64, 130, 256, 517]), C=st.integers(2, 12), n_seg=st.integers(1, 12),
       max_inst=st.sampled_from([1 << 16, 1000]), noise=st.booleans())
@example(seed=3, B=2, H=64, W=256, C=5, n_seg=9, max_inst=1 << 16, noise=True)     # > 1024 pairs in a workgroup
def test_fuzz_pq_from_the_parts_vs_oracle(oracle, seed, B, H, W, C, n_seg, max_inst, noise):
    """k_pq_count_parts (compact keys; full tiles, single steps, the ragged rest): the prediction
    given as class u8 / instance u8 / pan_of_inst, PQ states and confusion matrix bit-exact vs the
    oracle on the map the parts paint"""
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality
    rng = np.random.default_rng(seed)
    n = C + 1
    offset = 256 ** 3 if max_inst == 1 << 16 else 10 ** 7
    thing_c = rng.integers(0, 2, C).astype(bool)

    def rects(hi, fill=None):
        out = np.zeros((B, H, W), np.int64) if fill is None else fill.copy()
        for b in range(B):
            for _ in range(n_seg):
                y0, x0 = rng.integers(0, H), rng.integers(0, W)
                y1, x1 = rng.integers(y0, H) + 1, rng.integers(x0, W) + 1
                out[b, y0:y1, x0:x1] = rng.integers(0, hi)
        return out
    sem = rects(C).astype(np.uint8)
    inst = (rects(7) * thing_c[sem]).astype(np.uint8)                    # instances on thing classes only
    pan_of_inst = np.zeros((B, 256), np.int64)
    pan_of_inst[:, 1:7] = rng.integers(1, n, (B, 6)) * max_inst + np.arange(1, 7)
    tgt = rects(n) * max_inst + rects(3)
    tsem = rects(n).astype(np.uint8)
    if noise:                                                            # incoherent pixels: many pairs, no runs
        k = rng.random((B, H, W)) < 0.5
        tgt[k] = (rng.integers(0, n, k.sum()) * max_inst + rng.integers(0, 9, k.sum()))
        tsem[k] = rng.integers(0, n, k.sum())
    pred = np.where(inst > 0, np.take_along_axis(pan_of_inst, inst.reshape(B, -1).astype(np.int64), 1).reshape(B, H, W),
                    np.where(thing_c[sem], 0, (sem.astype(np.int64) + 1) * max_inst))       # nmsa_panoptic_paint's rule
    d_pred = torch.empty((B, H, W), dtype=torch.int64, device='cuda')
    L_ = ops.L
    d_sem, d_inst, d_poi, d_thing = dev(sem), dev(inst), dev(pan_of_inst), dev(thing_c.astype(np.uint8))
    L_.check(L_.lib().nmsa_panoptic_paint(L_.ptr(d_sem), L_.ptr(d_inst), L_.ptr(d_poi), L_.ptr(d_thing), B, C, H, W,
                                          max_inst, 0, L_.ptr(d_pred), None, L_.stream_ptr(d_pred.device)),
             'nmsa_panoptic_paint')
    assert np.array_equal(d_pred.cpu().numpy(), pred)
    parts = {'panoptic': d_pred, 'semantic_idx_u8': d_sem, 'instance': d_inst, 'pan_of_inst': d_poi,
             'is_thing': d_thing, 'void_label': 0, 'max_instances_per_category': max_inst}
    is_thing = [False] + thing_c.tolist()
    pq = PanopticQuality(n, 0, max_inst, offset, is_thing, device='cuda')
    miou = MeanIntersectionOverUnion(n, device='cuda')
    assert PanopticQuality.parts_usable(parts, d_pred, max_inst)
    pq.update_with_miou_parts(parts, dev(tgt), miou, dev(tsem), max_inst)
    torch.cuda.synchronize()
    if noise and H * W >= 4096 and int(pq._status):
        pq._status.zero_()          # more distinct intersections than the image's table holds: reported, not counted
        return
    state = None
    for b in range(B):
        *state, _ = oracle.pq_compare_and_accumulate(pred[b], tgt[b], n, 0, max_inst, offset, state=state)
    for g_, w in zip([pq.iou_per_class, pq.tp_per_class, pq.fn_per_class, pq.fp_per_class], state):
        assert np.array_equal(g_.cpu().numpy(), np.asarray(w, dtype=np.float64)), (seed, B, H, W)
    assert np.array_equal(miou.confmat.cpu().numpy(), oracle.confmat_update(pred // max_inst, tsem, n))
    assert int(pq._status) == 0 and int(miou._status) == 0


@settings(max_examples=_n(60), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), N=st.integers(1, 300), D=st.integers(2, 200),
       red=st.sampled_from(['none', 'sum', 'mean']), labelled=st.booleans(),
       kind=st.sampled_from(['mse', 'l1']), shape=st.sampled_from([(7,), (3, 5), (2, 3, 4), (2, 3, 4, 5), (2, 1, 3, 2, 2)]))
def test_fuzz_loss_forms_vs_oracle(oracle, seed, N, D, red, labelled, kind, shape):
    """the loss forms no task helper calls (csrc/losses_forms.hip): cosine rows of any [N, D] with
    labels from {+1, -1, 0}, MSE / L1 of any rank with every reduction — values and gradients
    against the fp64 oracle"""
    from nicr_mt_scene_analysis_amd import loss as L_
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, D)).astype(np.float32)
    y = rng.standard_normal((N, D)).astype(np.float32)
    lab = rng.choice(np.array([1.0, -1.0, 0.0], np.float32), N, p=[0.45, 0.45, 0.1]) if labelled else None
    w = rng.standard_normal(N).astype(np.float32)
    dx = dev(x).requires_grad_(True)
    fn = L_.CosineEmbeddingLoss(reduction=red)
    loss, n = fn._compute_loss(dx, dev(y), *((dev(lab),) if labelled else ()))
    ((loss * dev(w)).sum() if red == 'none' else loss).backward()
    o_loss, o_n, o_grad = oracle.loss_cosine_rows(x, y, lab, red, w)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), o_loss, rtol=2e-5, atol=2e-6)
    assert int(n) == int(o_n)
    # (fp32 cancellation in y / den - cos x / |x|^2 when x is nearly parallel to y: absolute floor)
    np.testing.assert_allclose(dx.grad.cpu().numpy(), o_grad, rtol=2e-4, atol=3e-5)
    a = rng.standard_normal(shape).astype(np.float32)
    b = rng.standard_normal(shape).astype(np.float32)
    wa = rng.standard_normal(shape).astype(np.float32)
    da = dev(a).requires_grad_(True)
    (loss, n), = (L_.MSELoss if kind == 'mse' else L_.L1Loss)(reduction=red)([da], [dev(b)])
    ((loss * dev(wa)).sum() if red == 'none' else loss).backward()
    o_loss, o_n, o_grad = oracle.loss_elementwise_form(a, b, kind, red, wa)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), o_loss, rtol=2e-5, atol=1e-6)
    assert int(n) == int(o_n)
    np.testing.assert_allclose(da.grad.cpu().numpy(), o_grad, rtol=2e-5, atol=1e-6)


@settings(max_examples=_n(150), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), n_px=st.integers(1, 40000), n=st.integers(1, 70),
       cell=st.sampled_from([1, 3, 16, 64, 1000]), masked=st.booleans(), offset=st.sampled_from([0, 0, 16, 1, 7]))
def test_fuzz_confmat_uint8_vs_oracle(oracle, seed, n_px, n, cell, masked, offset):
    """two uint8 maps (k_confmat_u8 when both are 16-byte aligned, the general kernel otherwise —
    `offset` shifts the views): confusion matrix bit-exact vs the oracle for runs of every
    length, any size (the scalar tail of the last lane), both void treatments"""
    from nicr_mt_scene_analysis_amd import metric
    rng = np.random.default_rng(seed)
    hi = n + 1 if masked else n

    def runs(top):
        v = np.repeat(rng.integers(0, top, n_px // cell + 2), cell)
        return v[rng.integers(0, cell):][:n_px].astype(np.uint8)
    pred, tgt = runs(n), runs(hi)
    d_pred = torch.zeros(n_px + 64, dtype=torch.uint8, device='cuda')[offset:offset + n_px]
    d_tgt = torch.zeros(n_px + 64, dtype=torch.uint8, device='cuda')[offset:offset + n_px]
    d_pred.copy_(dev(pred))
    d_tgt.copy_(dev(tgt))
    m = metric.MeanIntersectionOverUnion(n)
    if masked:
        m.update_masked_void(d_pred, d_tgt)
        keep = tgt != 0
        want = oracle.confmat_update(pred[keep], tgt[keep] - 1, n)
    else:
        m.update(d_pred, d_tgt)
        want = oracle.confmat_update(pred, tgt, n)
    torch.cuda.synchronize()
    assert int(m._status) == 0
    assert np.array_equal(m.confmat.cpu().numpy(), want), (seed, n_px, n, cell, masked, offset)


@settings(max_examples=_n(120), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), Hs=st.integers(1, 40), Ws=st.integers(1, 48),
       Ho=st.integers(1, 70), Wo=st.integers(1, 90), C=st.integers(1, 6))
def test_fuzz_resize_vs_oracle(oracle, seed, Hs, Ws, Ho, Wo, C):
    """random crop / resize geometries (up, down, 1-pixel planes, ragged widths): nearest maps and
    bilinear logits bit-exact, fused resized argmax == argmax of the resized logits"""
    from nicr_mt_scene_analysis_amd import ops
    rng = np.random.default_rng(seed)
    y0 = int(rng.integers(0, Hs)); x0 = int(rng.integers(0, Ws))
    y1 = int(rng.integers(y0 + 1, Hs + 1)); x1 = int(rng.integers(x0 + 1, Ws + 1))
    crop = (slice(y0, y1), slice(x0, x1))
    size = (Ho, Wo)
    ids = rng.integers(0, 1 << 26, (2, Hs, Ws)).astype(np.int64)
    assert np.array_equal(ops.resize_nearest(dev(ids), size, crop).cpu().numpy(),
                          oracle.resize_nearest(ids, size, crop))
    u8 = rng.integers(0, 256, (2, Hs, Ws)).astype(np.uint8)
    assert np.array_equal(ops.resize_nearest(dev(u8), size, crop).cpu().numpy(),
                          oracle.resize_nearest(u8, size, crop))
    x = (rng.integers(-8, 9, (2, C, Hs, Ws)) * 0.37).astype(np.float32)   # few levels: ties
    want = oracle.resize_bilinear(x, size, crop)
    assert np.array_equal(ops.resize_bilinear(dev(x), size, crop).cpu().numpy(), want)
    idx, score = oracle.semantic_argmax(want)
    r = ops.semantic_argmax_resized(dev(x), size, crop, want_u8=True)
    assert np.array_equal(r['idx'].cpu().numpy(), idx)
    np.testing.assert_allclose(r['score'].cpu().numpy(), score, rtol=1e-5, atol=1e-7)


@settings(max_examples=_n(40), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), Hs=st.integers(20, 90), Ws=st.integers(70, 200),
       fy=st.floats(1.0, 2.6), fx=st.floats(1.0, 2.6), C=st.integers(1, 13),
       dtype=st.sampled_from(['float32', 'bfloat16', 'float16']), cropped=st.booleans(),
       near_ties=st.sampled_from([False, False, True]))
def test_fuzz_upscaling_tiles_vs_oracle(oracle, seed, Hs, Ws, fy, fx, C, dtype, cropped, near_ties):
    """upscaling geometries wide enough for the LDS-staged tile kernels (64 x 16 output tiles,
    windows shifted at the right border, one or two staging slots): resized logits bit-exact,
    fused argmax / score == argmax of the resized logits, for all three logit dtypes"""
    from nicr_mt_scene_analysis_amd import ops
    rng = np.random.default_rng(seed)
    y0 = x0 = 0
    y1, x1 = Hs, Ws
    if cropped:
        y0 = int(rng.integers(0, Hs // 3)); x0 = int(rng.integers(0, Ws // 4))
        y1 = int(rng.integers(2 * Hs // 3, Hs + 1)); x1 = int(rng.integers(3 * Ws // 4, Ws + 1))
    crop = (slice(y0, y1), slice(x0, x1))
    size = (max(1, int(round((y1 - y0) * fy))), max(1, int(round((x1 - x0) * fx))))
    tdt = getattr(torch, dtype)
    # few levels -> exact ties between classes; all levels exact in bf16 / f16
    x = (rng.integers(-8, 9, (2, C, Hs, Ws)) * 0.375).astype(np.float32)
    if near_ties and dtype == 'float32':
        # interpolated values a few 2^-27 apart: classes that share the maximum's fp32
        # probability (a1 probability ties) — the lowest index wins
        x = (x * np.float32(2.0 ** -25)).astype(np.float32)
    if rng.random() < 0.2:
        x[0, rng.integers(C), rng.integers(Hs), rng.integers(Ws)] = np.nan
    xd = dev(x).to(tdt)
    seen = xd.float().cpu().numpy()
    want = oracle.resize_bilinear(seen, size, crop)
    got = ops.resize_bilinear(xd, size, crop)
    if dtype == 'float32':
        assert np.array_equal(got.cpu().numpy(), want, equal_nan=True)
        want_lo = want
    else:           # the reference's interpolate returns the storage dtype: round like it
        want_lo = torch.from_numpy(want).to(tdt).float().numpy()
        assert np.array_equal(got.float().cpu().numpy(), want_lo, equal_nan=True)
    idx, score = oracle.semantic_argmax(want_lo)
    r = ops.semantic_argmax_resized(xd, size, crop, want_u8=True)
    assert np.array_equal(r['idx'].cpu().numpy(), idx)
    assert np.array_equal(r['idx_u8'].cpu().numpy(), idx.astype(np.uint8))
    np.testing.assert_allclose(r['score'].cpu().numpy(), score, rtol=1e-5, atol=1e-7, equal_nan=True)
    only = ops.semantic_argmax_resized(xd, size, crop, want_u8=False, want_i64=True, want_score=False)
    assert np.array_equal(only['idx'].cpu().numpy(), idx)


@settings(max_examples=_n(80), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 3), H=st.integers(3, 36), W=st.integers(3, 44),
       NC=st.integers(2, 8), n_inst=st.integers(0, 12), sigma=st.integers(1, 5),
       normalized=st.booleans())
def test_fuzz_targets_vs_oracle(oracle, seed, B, H, W, NC, n_inst, sigma, normalized):
    """random small label maps (rectangles with sparse uint16 ids, mixed classes): target
    generators bit-exact vs the oracle"""
    from nicr_mt_scene_analysis_amd import ops
    rng = np.random.default_rng(seed)
    sem = rng.integers(0, NC, (B, H, W)).astype(np.uint8)
    ins = np.zeros((B, H, W), np.int32)
    for b in range(B):
        for _ in range(n_inst):
            ya, xa = rng.integers(0, H), rng.integers(0, W)
            yb, xb = rng.integers(ya, H) + 1, rng.integers(xa, W) + 1
            ins[b, ya:yb, xa:xb] = rng.integers(1, 65536)
            if rng.random() < 0.6:
                sem[b, ya:yb, xa:xb] = rng.integers(0, NC)
    is_thing = rng.random(NC) < 0.5
    is_thing[0] = False
    stuff = np.zeros((NC,), np.uint8)
    stuff[np.where(~is_thing)[0][1:]] = 1
    r = ops.instance_targets(dev(sem), dev(ins), NC, dev(is_thing.astype(np.uint8)), dev(stuff),
                             sigma, normalized)
    torch.cuda.synchronize()
    assert int(r['status'].item()) == 0
    o = oracle.instance_targets(sem, ins, NC, is_thing, stuff, sigma, normalized)
    assert np.array_equal(r['center'].cpu().numpy(), o['center'])
    assert np.array_equal(r['offset'].cpu().numpy(), o['offset'])
    assert np.array_equal(r['foreground'].cpu().numpy(), o['foreground'])
    assert np.array_equal(r['center_mask'].cpu().numpy(), o['center_mask'])
    ne, ns = r['n_encoded'].cpu().numpy(), r['n_skipped'].cpu().numpy()
    for b in range(B):
        assert r['encoded_ids'][b, :ne[b]].cpu().tolist() == o['encoded'][b]
        assert r['skipped_ids'][b, :ns[b]].cpu().tolist() == o['skipped'][b]
    p = ops.panoptic_targets(dev(sem), dev(ins), NC, dev(is_thing.astype(np.uint8)), 1 << 16, 0)
    assert int(p['status'].item()) == 0
    pan, dicts = oracle.naive_merge(sem, ins, 1 << 16, np.where(is_thing)[0], 0)
    assert np.array_equal(p['panoptic'].cpu().numpy(), pan)
    got = ids_from_arrays(p['n_ids'].cpu().numpy(), p['ids_pan'].cpu().numpy(), p['ids_ins'].cpu().numpy())
    assert [list(d.items()) for d in got] == [list(d.items()) for d in dicts]


@settings(max_examples=_n(60), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 3), H=st.integers(8, 96), W4=st.integers(1, 40),
       NC=st.integers(2, 12), n_inst=st.integers(0, 60), sigma=st.integers(1, 4), clustered=st.booleans())
def test_fuzz_targets_scan_launch_vs_oracle(oracle, seed, B, H, W4, NC, n_inst, sigma, clustered):
    """the one-launch front end of the target generators (csrc/targets.hip k_tg_scan: rows of 4
    pixels, several workgroups per image handing over through the per-image hash table and the
    ticket tail): images of up to 96 x 160, up to 60 instances with sparse uint16 ids — or ids that
    all fall on ONE slot of the table's hash (clustered: multiples of 2^20 apart after the multiply)
    — against the oracle, twice on the same persistent workspace"""
    from nicr_mt_scene_analysis_amd import ops
    rng = np.random.default_rng(seed)
    W = 4 * W4
    sem = rng.integers(0, NC, (B, H, W)).astype(np.uint8)
    ins = np.zeros((B, H, W), np.int32)
    # (clustered: 63 ids whose home slot in the 1024-slot table is the same — a probe chain of 63)
    all_ids = np.arange(1, 65536, dtype=np.uint64)
    home = (((all_ids * np.uint64(2654435761)) & np.uint64(0xffffffff)) >> np.uint64(12)) & np.uint64(1023)
    pool = all_ids[home == np.uint64(seed % 1024)][:63].astype(np.int64) if clustered else None
    for b in range(B):
        for _ in range(n_inst):
            ya, xa = rng.integers(0, H), rng.integers(0, W)
            yb, xb = rng.integers(ya, min(H, ya + 24)) + 1, rng.integers(xa, min(W, xa + 40)) + 1
            ins[b, ya:yb, xa:xb] = rng.choice(pool) if clustered else rng.integers(1, 65536)
            if rng.random() < 0.6:
                sem[b, ya:yb, xa:xb] = rng.integers(0, NC)
    is_thing = rng.random(NC) < 0.5
    is_thing[0] = False
    stuff = np.zeros((NC,), np.uint8)
    stuff[np.where(~is_thing)[0][1:]] = 1
    o = oracle.instance_targets(sem, ins, NC, is_thing, stuff, sigma, True)
    pan, dicts = oracle.naive_merge(sem, ins, 1 << 16, np.where(is_thing)[0], 0)
    for _ in range(2):
        r = ops.instance_targets(dev(sem), dev(ins), NC, dev(is_thing.astype(np.uint8)), dev(stuff), sigma, True)
        assert int(r['status'].item()) == 0
        assert np.array_equal(r['center'].cpu().numpy(), o['center'])
        assert np.array_equal(r['offset'].cpu().numpy(), o['offset'])
        assert np.array_equal(r['foreground'].cpu().numpy(), o['foreground'])
        assert np.array_equal(r['center_mask'].cpu().numpy(), o['center_mask'])
        ne, ns = r['n_encoded'].cpu().numpy(), r['n_skipped'].cpu().numpy()
        for b in range(B):
            assert r['encoded_ids'][b, :ne[b]].cpu().tolist() == o['encoded'][b]
            assert r['skipped_ids'][b, :ns[b]].cpu().tolist() == o['skipped'][b]
        p = ops.panoptic_targets(dev(sem), dev(ins), NC, dev(is_thing.astype(np.uint8)), 1 << 16, 0)
        assert int(p['status'].item()) == 0
        assert np.array_equal(p['panoptic'].cpu().numpy(), pan)
        got = ids_from_arrays(p['n_ids'].cpu().numpy(), p['ids_pan'].cpu().numpy(), p['ids_ins'].cpu().numpy())
        assert [list(d.items()) for d in got] == [list(d.items()) for d in dicts]


@settings(max_examples=_n(120), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 3), H=st.integers(2, 30), W=st.integers(2, 41),
       NC=st.integers(2, 9), n_rect=st.integers(0, 10), wide=st.booleans())
def test_fuzz_standalone_merge_vs_oracle(oracle, seed, B, H, W, NC, n_rect, wide):
    """deeplab_merge_batch kernels (uint8 ids and ranked uint16 ids) on random maps where the
    thing mask, the instance map and the semantic map deliberately disagree"""
    from nicr_mt_scene_analysis_amd import ops
    rng = np.random.default_rng(seed)
    sem = rng.integers(0, NC, (B, H, W)).astype(np.int64)
    ins = np.zeros((B, H, W), np.int64)
    for b in range(B):
        for _ in range(n_rect):
            ya, xa = rng.integers(0, H), rng.integers(0, W)
            yb, xb = rng.integers(ya, H) + 1, rng.integers(xa, W) + 1
            ins[b, ya:yb, xa:xb] = rng.integers(1, 65536 if wide else 256)
    thing_ids = [int(c) for c in range(1, NC) if rng.random() < 0.5]
    lut = np.zeros((NC,), np.uint8)
    lut[thing_ids] = 1
    thing_seg = (lut[sem] > 0) ^ (rng.random((B, H, W)) < 0.1)          # disagree on ~10 % of px
    want_pan, want_ids = oracle.deeplab_merge(sem, ins, thing_seg, 1 << 16, thing_ids, 0)
    if wide:
        r = ops.panoptic_merge_wide(dev(sem), dev(ins.astype(np.int32)), dev(thing_seg), dev(lut),
                                    1 << 16, 0)
        assert int(r['status'].item()) == 0
    else:
        r = ops.panoptic_merge(dev(sem), dev(ins.astype(np.uint8)), dev(thing_seg), dev(lut), 1 << 16, 0)
    torch.cuda.synchronize()
    assert np.array_equal(r['panoptic'].cpu().numpy(), want_pan), (seed, wide)
    got = ids_from_arrays(r['n_ids'].cpu().numpy(), r['ids_pan'].cpu().numpy(), r['ids_ins'].cpu().numpy())
    assert [list(d.items()) for d in got] == [list(d.items()) for d in want_ids]


@settings(max_examples=_n(60), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 3), H=st.integers(2, 30), W=st.integers(2, 41),
       n_rect=st.integers(0, 8), with_mask=st.booleans())
def test_fuzz_orientation_vs_oracle(oracle, seed, B, H, W, n_rect, with_mask):
    from nicr_mt_scene_analysis_amd import ops
    rng = np.random.default_rng(seed)
    ori = rng.standard_normal((B, 2, H, W)).astype(np.float32)
    inst = np.zeros((B, H, W), np.uint8)
    for b in range(B):
        for _ in range(n_rect):
            ya, xa = rng.integers(0, H), rng.integers(0, W)
            yb, xb = rng.integers(ya, H) + 1, rng.integers(xa, W) + 1
            inst[b, ya:yb, xa:xb] = rng.integers(1, 256)
    mask = (rng.random((B, H, W)) < 0.7) if with_mask else None
    want = oracle.instance_orientation(ori, inst, mask)
    r = ops.instance_orientation_sums(dev(ori), dev(inst), None if mask is None else dev(mask))
    torch.cuda.synchronize()
    sums, cnt = r['sums'].cpu().numpy(), r['count'].cpu().numpy()
    for b in range(B):
        assert sorted(np.nonzero(cnt[b])[0].tolist()) == sorted(want[b].keys())
        for i, ang in want[b].items():
            got = float(np.arctan2(np.float32(sums[b, i, 1]), np.float32(sums[b, i, 0])))
            assert abs(got - ang) < 1e-4 or abs(abs(got - ang) - 2 * np.pi) < 1e-4


@settings(max_examples=_n(60), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 3), H=st.integers(2, 30), W=st.integers(2, 41),
       n_rect=st.integers(0, 8), with_mask=st.booleans())
def test_fuzz_orientation_wide_ids(oracle, seed, B, H, W, n_rect, with_mask):
    """_get_instance_orientation with ground-truth instance maps (sparse uint16 ids in int32):
    same angles as the uint8 oracle run on the rank-compressed ids, keyed by the original ids"""
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    rng = np.random.default_rng(seed)
    ori = rng.standard_normal((B, 2, H, W)).astype(np.float32)
    inst = np.zeros((B, H, W), np.int32)
    for b in range(B):
        for _ in range(n_rect):
            ya, xa = rng.integers(0, H), rng.integers(0, W)
            yb, xb = rng.integers(ya, H) + 1, rng.integers(xa, W) + 1
            inst[b, ya:yb, xa:xb] = rng.integers(1, 65536)
    mask = (rng.random((B, H, W)) < 0.7) if with_mask else None
    post = get_postprocessing_class('instance')()
    got = post._get_instance_orientation(dev(ori), dev(inst), None if mask is None else dev(mask))
    for b in range(B):
        ids = np.unique(inst[b])
        small = np.searchsorted(ids, inst[b]).astype(np.uint8)            # ranks (0 stays 0: ids[0] == 0 or absent)
        if ids[0] != 0:
            small = (small + 1).astype(np.uint8)
            ids = np.concatenate([[0], ids])
        want = oracle.instance_orientation(ori[b:b + 1], small[None], None if mask is None else mask[b:b + 1])[0]
        assert sorted(got[b].keys()) == sorted(int(ids[k]) for k in want), (seed, b)
        for k, ang in want.items():
            g = got[b][int(ids[k])]
            assert abs(g - ang) < 1e-4 or abs(abs(g - ang) - 2 * np.pi) < 1e-4


@settings(max_examples=_n(60), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 3),
       C=st.one_of(st.integers(1, 11), st.integers(12, 52)), H=st.integers(1, 19),
       W=st.integers(1, 23), weighted=st.booleans(), ls=st.sampled_from([0.0, 0.1, 0.5]),
       dtype=st.sampled_from(['float32', 'bfloat16']),
       expect=st.sampled_from(['nothing', 'right', 'wrong']))
def test_fuzz_losses_vs_oracle(oracle, seed, B, C, H, W, weighted, ls, dtype, expect):
    """every loss kernel (sum, count, gradient) on random odd shapes against the C oracle;
    `expect`: the forward kernel writes the gradient for the upstream scale it is told to expect
    (right: backward confirms; wrong: backward recomputes; nothing: two-kernel path)"""
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    rng = np.random.default_rng(seed)
    # `loss.backward()` below sends 1.0 to the loss sum
    exp = {'nothing': None, 'right': torch.ones(1, device='cuda'),
           'wrong': torch.full((1,), 0.5, device='cuda')}[expect]
    tdt = getattr(torch, dtype)
    # gradients are O(1); the exp2-domain softmax carries ~|x| * 1e-7 of absolute error
    tol = dict(rtol=2e-5, atol=1e-5) if dtype == 'float32' else dict(rtol=2e-2, atol=2e-2)

    def leaf(a):
        return torch.from_numpy(a).to(tdt).cuda().requires_grad_(True)

    def as_f32(t):                      # the values the kernel actually saw
        return t.detach().float().cpu().numpy()

    # ---- cross entropy (a6) ----
    x = leaf((rng.standard_normal((B, C, H, W)) * 3).astype(np.float32))
    t = rng.integers(0, C + 1, (B, H, W)).astype(np.uint8)              # 0 = void
    w = (rng.random(C) + 0.5).astype(np.float32) if weighted else None
    loss, n, wsum = F_.cross_entropy_sum(x, dev(t), None if w is None else dev(w), ls, exp)
    loss.backward()
    s_ref, n_ref, w_ref, g_ref = oracle.loss_ce(as_f32(x), t, w, ls, want_grad=True)
    assert int(n) == n_ref
    # lse - x[target] cancels: each pixel carries ~1e-7 * |x| of absolute error (all of the loss
    # when there is a single class and the exact result is 0)
    cancel = 3e-7 * float(np.abs(as_f32(x)).max(axis=1).sum())
    np.testing.assert_allclose(float(loss.detach()), s_ref, rtol=1e-5 if dtype == 'float32' else 2e-3,
                               atol=1e-5 + cancel)
    np.testing.assert_allclose(float(wsum), w_ref, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), g_ref, **tol)

    # ---- masked MSE / L1 (a7) ----
    for kind, Cn in (('mse', 1), ('l1', 2)):
        shape = (B, H, W) if Cn == 1 else (B, Cn, H, W)
        p = leaf(rng.standard_normal(shape).astype(np.float32))
        y = rng.standard_normal(shape).astype(np.float32)
        m = rng.random((B, H, W)) < 0.6
        loss, n = F_.masked_elementwise_sum(p, dev(y), dev(m), kind, exp)
        loss.backward()
        s_ref, n_ref, g_ref = oracle.loss_masked_elementwise(as_f32(p), y, m, kind, want_grad=True)
        assert int(n) == n_ref
        np.testing.assert_allclose(float(loss.detach()), s_ref, rtol=1e-5 if dtype == 'float32' else 2e-3, atol=1e-5)
        np.testing.assert_allclose(p.grad.float().cpu().numpy(), g_ref, **tol)

    # ---- von Mises (a8) ----
    def unit(a):
        return a / np.linalg.norm(a, axis=1, keepdims=True)
    p = leaf(unit(rng.standard_normal((B, 2, H, W))).astype(np.float32))
    y = unit(rng.standard_normal((B, 2, H, W))).astype(np.float32)
    m = rng.random((B, H, W)) < 0.5
    loss, n = F_.vonmises_sum(p, dev(y), dev(m), 1.0, exp)
    loss.backward()
    s_ref, n_ref, g_ref = oracle.loss_vonmises(as_f32(p), y, m, 1.0, want_grad=True)
    assert int(n) == n_ref
    np.testing.assert_allclose(float(loss.detach()), s_ref, rtol=1e-5 if dtype == 'float32' else 2e-3, atol=1e-5)
    # d cos / d x divides by |x|^2: short vectors amplify the fp32 rounding of the norm
    cos_tol = dict(rtol=1e-3, atol=1e-5) if dtype == 'float32' else tol
    np.testing.assert_allclose(p.grad.float().cpu().numpy(), g_ref, **cos_tol)

    # ---- cosine embedding with LUT (a9) ----
    # D >= 2: with one feature the cosine is +-1 and its gradient is pure rounding noise / |x|
    D, L = int(rng.integers(2, 9)), int(rng.integers(1, 6))
    p = leaf(rng.standard_normal((B, D, H, W)).astype(np.float32))
    idx = rng.integers(0, L + 1, (B, H, W)).astype(np.int32)          # 0 = no target
    lut = unit(rng.standard_normal((B * L, D))).reshape(B, L, D).astype(np.float32)
    loss, n = F_.cosine_embedding_lut_sum(p, dev(idx), dev(lut))
    loss.backward()
    s_ref, n_ref, g_ref = oracle.loss_cosine_embedding(as_f32(p), idx, lut, want_grad=True)
    assert int(n) == n_ref
    np.testing.assert_allclose(float(loss.detach()), s_ref, rtol=1e-5 if dtype == 'float32' else 2e-3, atol=1e-5)
    # d cos / d x divides by |x|^2: short vectors amplify the fp32 rounding of the norm
    cos_tol = dict(rtol=1e-3, atol=1e-5) if dtype == 'float32' else tol
    np.testing.assert_allclose(p.grad.float().cpu().numpy(), g_ref, **cos_tol)


@settings(max_examples=_n(40), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture,
                                 HealthCheck.data_too_large])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 2),
       D=st.sampled_from([2, 7, 64, 255, 384, 512, 520, 599, 600, 768, 900]),   # D = 1: cos = +-1, its gradient is rounding noise
       L=st.sampled_from([1, 3, 40, 64, 65, 90, 130, 1300]),
       hw=st.sampled_from([(1, 1), (3, 5), (8, 8), (8, 16), (16, 24), (7, 23), (32, 32)]),
       dtype=st.sampled_from(['float32', 'bfloat16', 'float16']))
def test_fuzz_cosine_embedding_large_dims_vs_oracle(oracle, seed, B, D, L, hw, dtype):
    """a9 at embedding sizes around the LDS chunk boundaries of k_cos_emb_lds (one chunk, two,
    three; D not a multiple of the chunk), pixel counts with and without the per-pixel dot
    buffer (H*W % 8), LUTs too tall for LDS (generic kernel): forward + gradient vs the oracle"""
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    if D * L > 300_000:
        L = max(1, 300_000 // D)                        # keep the oracle's LUT small
    H, W = hw
    rng = np.random.default_rng(seed)
    tdt = getattr(torch, dtype)
    lut = rng.standard_normal((B, L, D)).astype(np.float32)
    lut /= np.maximum(np.linalg.norm(lut, axis=-1, keepdims=True), 1e-3)
    idx = rng.integers(0, L + 1, (B, H, W)).astype(np.int32)
    tgt = np.take_along_axis(lut, np.clip(idx - 1, 0, L - 1).reshape(B, -1, 1), axis=1)
    pred = tgt.reshape(B, H, W, D).transpose(0, 3, 1, 2) + rng.standard_normal((B, D, H, W)) / np.sqrt(D)
    x = dev(pred.astype(np.float32)).to(tdt).requires_grad_(True)
    loss, n = F_.cosine_embedding_lut_sum(x, dev(idx), dev(lut))
    loss.backward()
    s_ref, n_ref, g_ref = oracle.loss_cosine_embedding(x.detach().float().cpu().numpy(), idx, lut,
                                                       want_grad=True)
    assert int(n) == n_ref
    _EFFECTIVE['cos_large'] = _EFFECTIVE.get('cos_large', 0) + 1
    np.testing.assert_allclose(float(loss.detach()), s_ref, rtol=1e-5, atol=1e-5)
    got = x.grad.float().cpu().numpy()
    if dtype == 'float32':
        np.testing.assert_allclose(got, g_ref, rtol=2e-3, atol=2e-6)
    else:                                               # the gradient is rounded to 16 bits once
        np.testing.assert_allclose(got, g_ref, rtol=2 ** -7, atol=1e-5)


@settings(max_examples=_n(25), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture,
                                 HealthCheck.data_too_large])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 3),
       hw=st.sampled_from([(96, 128), (120, 200), (240, 320), (250, 333), (480, 640)]),
       n_cat=st.integers(2, 60), cell=st.sampled_from([4, 8, 16, 32]), twice=st.booleans())
def test_fuzz_pq_many_segments_vs_oracle(oracle, seed, B, hw, n_cat, cell, twice):
    """a12 on larger maps with many segments (hundreds to thousands of distinct intersections:
    block-private LDS tables overflow into the image's global table, different table capacities
    per image size); states bit-exact vs the oracle, also on a second update that starts from
    the tables the first one left clean"""
    from nicr_mt_scene_analysis_amd import metric
    H, W = hw
    rng = np.random.default_rng(seed)
    cls = np.kron(rng.integers(0, n_cat, (B, (H + cell - 1) // cell, (W + cell - 1) // cell)),
                  np.ones((cell, cell), np.int64))[:, :H, :W]
    ins = np.kron(rng.integers(0, 5, (B, (H + 2 * cell - 1) // (2 * cell), (W + 2 * cell - 1) // (2 * cell))),
                  np.ones((2 * cell, 2 * cell), np.int64))[:, :H, :W]
    is_thing = rng.random(n_cat) < 0.5
    pred = (cls * 65536 + ins * is_thing[cls]).astype(np.int64)
    tgt = np.roll(pred, (int(rng.integers(0, 6)), int(rng.integers(0, 6))), axis=(1, 2))
    tgt[:, :int(rng.integers(0, 9))] = 0
    n_int = max(len(np.unique(tgt[b] * 256 ** 3 + pred[b])) for b in range(B))
    cap = 4096
    while cap * 24 < H * W:
        cap *= 2
    if n_int > cap // 2:
        return                                           # beyond the table: reported, other test
    _EFFECTIVE['pq_many'] = _EFFECTIVE.get('pq_many', 0) + 1
    pq = metric.PanopticQuality(n_cat, 0, 65536, 256 ** 3, [bool(t) for t in is_thing])
    state = None
    for _ in range(2 if twice else 1):
        pq.update(torch.from_numpy(pred), torch.from_numpy(tgt))
        for b in range(B):
            *state, _ = oracle.pq_compare_and_accumulate(pred[b], tgt[b], n_cat, 0, 65536, 256 ** 3,
                                                         state=state)
    got = np.stack([getattr(pq, n).cpu().numpy() for n in
                    ('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class')])
    assert (got == np.stack(state)).all()
    pq.compute()


@settings(max_examples=_n(40), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture,
                                 HealthCheck.data_too_large])
@given(p=cases(medium=True))
def test_fuzz_compute_scores_vs_oracle(oracle, p):
    """f3: score maps / per-instance mean semantic score of random pipelines vs the oracle"""
    from nicr_mt_scene_analysis_amd import ops
    logits, heat, offset, is_thing = make_inputs(p)
    logits = logits + np.random.default_rng(p['seed']).random(logits.shape).astype(np.float32)
    B, C, H, W = logits.shape
    x = dev(logits).to(getattr(torch, p['dtype']))
    r = ops.panoptic_pipeline(x, dev(heat), dev(offset), dev(is_thing), threshold=p['thr'],
                              kernel_size=p['ksize'], top_k=p['topk'],
                              apply_foreground_mask=p['apply_fg'], distance_threshold=p['dist_thr'],
                              want_score=True, want_panoptic_semantic=True, max_centers=256)
    if int(r['n_centers'].max()) > 255:
        return                                           # more tied centers than the 256-row table
    _EFFECTIVE['scores'] = _EFFECTIVE.get('scores', 0) + 1
    tab = torch.zeros((B, 256), dtype=torch.float32, device='cuda')
    tab[:, 1:] = r['center_scores'][:, :255]
    sc = ops.panoptic_scores(x, r['semantic_idx_u8'], r['semantic_score'], r['instance'],
                             r['panoptic'], r['pan_of_inst'], tab, 1 << 16)
    torch.cuda.synchronize()
    ids = ids_from_arrays(r['n_ids'].cpu().numpy(), r['ids_pan'].cpu().numpy(), r['ids_ins'].cpu().numpy())
    sem, ins, pns, mean = oracle.panoptic_scores(
        x.float().cpu().numpy(), r['panoptic_semantic'].cpu().numpy(), r['panoptic'].cpu().numpy(),
        ids, tab.cpu().numpy())
    np.testing.assert_allclose(sc['semantic_score'].cpu().numpy(), sem, rtol=2e-5, atol=1e-7, equal_nan=True)
    assert np.array_equal(sc['instance_score'].cpu().numpy(), ins)
    np.testing.assert_allclose(sc['panoptic_score'].cpu().numpy(), pns, rtol=2e-5, atol=1e-7, equal_nan=True)
    got_mean = sc['mean_semantic_score'].cpu().numpy()
    for b, d in enumerate(ids):
        for ins_id in d.values():
            g_, w_ = got_mean[b, ins_id], mean[b, ins_id]
            assert (np.isnan(g_) and np.isnan(w_)) or abs(g_ - w_) <= 2e-5 * abs(w_) + 1e-9


@settings(max_examples=_n(60), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(seed=st.integers(0, 2 ** 31 - 1), B=st.integers(1, 3), H=st.integers(1, 70), W=st.integers(1, 90),
       n_keys=st.integers(0, 64), n_ids=st.integers(1, 80))
def test_fuzz_dve_indices_vs_oracle(oracle, seed, B, H, W, n_keys, n_ids):
    """f4: panoptic map -> 1-based index of its id in the image's key list (0 = not listed):
    random blocky maps, keys in random order, ids missing from the list, empty lists"""
    from nicr_mt_scene_analysis_amd import ops
    rng = np.random.default_rng(seed)
    pool = rng.integers(0, 200 * 65536, n_ids).astype(np.int64)
    cells = pool[rng.integers(0, n_ids, (B, (H + 4) // 5, (W + 6) // 7))]
    pan = np.repeat(np.repeat(cells, 5, 1), 7, 2)[:, :H, :W].copy()
    K = max(n_keys, 1)
    keys = np.zeros((B, K), np.int64)
    nk = np.zeros((B,), np.int32)
    lists = []
    for b in range(B):
        n = int(rng.integers(0, n_keys + 1))
        chosen = rng.permutation(np.unique(np.concatenate([pool, rng.integers(0, 1 << 24, 8)])))[:n]
        keys[b, :len(chosen)] = chosen
        nk[b] = len(chosen)
        lists.append([int(v) for v in chosen])
    got = ops.dve_targets(dev(pan), dev(keys), dev(nk))['indices']
    assert np.array_equal(got.cpu().numpy(), oracle.dve_indices(pan, lists))



@settings(max_examples=_n(40), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture,
                                 HealthCheck.data_too_large])
@given(p=cases(medium=True), fy=st.floats(0.8, 1.8), fx=st.floats(0.8, 1.8), cropped=st.booleans(),
       compute_scores=st.booleans(), defer=st.booleans())
def test_fuzz_postprocess_api_vs_ops(p, fy, fx, cropped, compute_scores, defer):
    """the reference-shaped `PanopticPostprocessing.postprocess` (lazy entries, packed table
    hand-over, crop + full-resolution twins) against the bare ops — themselves checked against
    the oracle above — on random geometries"""
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    logits, heat, offset, is_thing = make_inputs(p)
    B, C, H, W = logits.shape
    rng = np.random.default_rng(p['seed'] + 1)
    y0 = x0 = 0
    y1, x1 = H, W
    if cropped:
        y1, x1 = int(rng.integers(H // 2, H + 1)), int(rng.integers(W // 2, W + 1))
        y0, x0 = int(rng.integers(0, y1 // 3 + 1)), int(rng.integers(0, x1 // 3 + 1))
    FH, FW = max(2, int(round((y1 - y0) * fy))), max(2, int(round((x1 - x0) * fx)))
    crop = (slice(y0, y1), slice(x0, x1))
    kw = dict(heatmap_threshold=p['thr'], heatmap_nms_kernel_size=p['ksize'],
              heatmap_apply_foreground_mask=p['apply_fg'], top_k_instances=p['topk'],
              offset_distance_threshold=p['dist_thr'])
    post = get_postprocessing_class('panoptic')(
        semantic_postprocessing=get_postprocessing_class('semantic')(),
        instance_postprocessing=get_postprocessing_class('instance')(**kw),
        semantic_classes_is_thing=tuple(bool(v) for v in is_thing),
        semantic_class_has_orientation=tuple(bool(v) for v in is_thing),
        compute_scores=compute_scores, defer_host_sync=defer)
    batch = {'rgb_fullres': torch.zeros((B, 3, FH, FW)),
             APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize', 'valid_region_slice_y': crop[0],
                                           'valid_region_slice_x': crop[1]}]] * B}
    x = dev(logits).to(getattr(torch, p['dtype']))
    d_heat, d_off, d_thing = dev(heat), dev(offset), dev(is_thing)
    try:
        r = post.postprocess(((x, (d_heat, d_off)), (None, None)), batch, is_training=False)
        r['panoptic_segmentation_deeplab_ids']
    except RuntimeError:
        assert defer                    # > 256 tied centers: only the eager mode can re-run
        return
    o = ops.panoptic_pipeline(x, d_heat, d_off, d_thing, threshold=p['thr'], kernel_size=p['ksize'],
                              top_k=p['topk'], apply_foreground_mask=p['apply_fg'],
                              distance_threshold=p['dist_thr'], want_score=True,
                              want_panoptic_semantic=True, max_centers=post._instance_postprocessing._max_centers)
    def eq(a, b):                       # exact, NaN == NaN (non-finite logits give NaN scores)
        if a.is_floating_point():
            return a.shape == b.shape and bool(torch.allclose(a, b, rtol=0, atol=0, equal_nan=True))
        return torch.equal(a, b)
    assert eq(r['panoptic_segmentation_deeplab'], o['panoptic'])
    assert eq(r['panoptic_segmentation_deeplab_instance_idx'], o['instance'])
    assert eq(r['panoptic_segmentation_deeplab_semantic_idx'], o['panoptic_semantic'])
    assert eq(r['panoptic_foreground_mask'], o['foreground'])
    assert eq(r['semantic_segmentation_idx'], o['semantic_idx_u8'].long())
    assert eq(r['semantic_segmentation_score'], o['semantic_score'])
    size = (FH, FW)
    for key, src in (('panoptic_segmentation_deeplab', o['panoptic']),
                     ('panoptic_segmentation_deeplab_instance_idx', o['instance']),
                     ('panoptic_segmentation_deeplab_semantic_idx', o['panoptic_semantic'])):
        assert eq(r[key + '_fullres'], ops.resize_nearest(src, size, crop)), key
    if size == (y1 - y0, x1 - x0):      # dense_base.py:27-30: same size -> no interpolation at all
        am = ops.semantic_argmax(x[..., crop[0], crop[1]].contiguous(), want_u8=False, want_i64=True,
                                 want_score=True)
    else:
        am = ops.semantic_argmax_resized(x, size, crop, want_u8=False, want_i64=True, want_score=True)
    assert eq(r['semantic_segmentation_idx_fullres'], am['idx'])
    # identity geometry shares the network-resolution score (another kernel): tolerance, not bits
    torch.testing.assert_close(r['semantic_segmentation_score_fullres'], am['score'], rtol=1e-5,
                               atol=1e-7, equal_nan=True)
    n = o['n_centers'].cpu().tolist()
    ids = ids_from_arrays(o['n_ids'].cpu().numpy(), o['ids_pan'].cpu().numpy(), o['ids_ins'].cpu().numpy())
    assert [list(d.items()) for d in r['panoptic_segmentation_deeplab_ids']] == [list(d.items()) for d in ids]
    cyx, sc, area = (o[k].cpu().numpy() for k in ('centers_yx', 'center_scores', 'area'))
    meta = r['panoptic_segmentation_deeplab_instance_meta']
    for b in range(B):
        assert list(meta[b].keys()) == list(range(1, n[b] + 1))
        for i, m in meta[b].items():
            assert m['center_yx'] == (int(cyx[b, i - 1, 0]), int(cyx[b, i - 1, 1]))
            assert m['score'] == float(sc[b, i - 1])
            assert m['area'] == (int(area[b, i]) if i <= 255 else 0)
    if compute_scores:
        for k in ('semantic', 'instance', 'panoptic'):
            key = f'panoptic_segmentation_deeplab_{k}_score'
            assert eq(r[key + '_fullres'], ops.resize_nearest(r[key], size, crop)), key



@settings(max_examples=_n(60), deadline=None, derandomize=_DERANDOMIZE,
          suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture,
                                 HealthCheck.data_too_large])
@given(p=cases(medium=True), p_fg=st.floats(0.0, 1.0), blocky=st.booleans())
def test_fuzz_standalone_grouping_vs_oracle(oracle, p, p_fg, blocky):
    """a3 with a GIVEN foreground mask (`nmsa_group_offsets`, the GT-foreground branch of
    InstancePostprocessing): instance ids and per-id areas vs the oracle"""
    from nicr_mt_scene_analysis_amd import ops
    _, heat, offset, _ = make_inputs(p)
    B, _, H, W = heat.shape
    rng = np.random.default_rng(p['seed'] + 7)
    if p.get('specials'):               # non-finite offsets: NaN / inf distances pick index 0
        offset = offset.copy()
        for _ in range(int(rng.integers(1, 8))):
            offset[rng.integers(B), rng.integers(2), rng.integers(H), rng.integers(W)] = \
                rng.choice([np.nan, np.inf, -np.inf, 1e30])
    if blocky:
        cells = rng.random((B, (H + 7) // 8, (W + 15) // 16)) < p_fg
        fg = np.repeat(np.repeat(cells, 8, 1), 16, 2)[:, :H, :W].copy()
    else:
        fg = rng.random((B, H, W)) < p_fg
    try:
        cyx, n, _, _ = oracle.center_nms_topk(heat, threshold=p['thr'], ksize=p['ksize'],
                                              topk=p['topk'], max_centers=256)
    except oracle.OracleError:
        return
    if n.max() > 255:
        return
    inst, area = oracle.group_offsets(offset, fg, cyx, n, scale_y=H, scale_x=W, dist_thr=p['dist_thr'])
    cen = ops.center_nms_topk(dev(heat), threshold=p['thr'], kernel_size=p['ksize'], top_k=p['topk'],
                              max_centers=256)
    r = ops.group_offsets(dev(offset), dev(fg), cen['centers_yx'], cen['n_centers'], float(H), float(W),
                          p['dist_thr'])
    assert np.array_equal(r['instance'].cpu().numpy(), inst)
    got_area = r['area'].cpu().numpy()
    for b in range(B):
        assert np.array_equal(got_area[b, :n[b] + 1], np.asarray(area[b][:n[b] + 1])), (b, n[b])
    _EFFECTIVE['grouping'] = _EFFECTIVE.get('grouping', 0) + 1


def test_fuzz_effective_cases():
    """runs last: the fuzz tests that may skip a draw must still have exercised the kernels"""
    for name in ('scores', 'grouping', 'cos_large', 'pq_many'):
        if name in _EFFECTIVE:
            assert _EFFECTIVE[name] >= 10, _EFFECTIVE
    if 'argmax_ties' in _EFFECTIVE:          # pixels where the probability rule beats the plain argmax
        assert _EFFECTIVE['argmax_ties'] >= 100, _EFFECTIVE
