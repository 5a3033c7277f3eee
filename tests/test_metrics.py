"""
Metric accumulators.

* The six 6x6 PQ known-answer tables and the 0.63177083 / 0.84236111 case restate
  reference tests/test_metrics.py:76-446 (values are the reference's own).  They
  run against the C oracle on the CPU tier and against the HIP path (through
  the PanopticQuality class, like the reference's tests) on the GPU tier.
* Random-map goldens come from the reference's compare_and_accumulate /
  MeanIntersectionOverUnion (oracle/gen_golden.py).
* MAAE restates reference tests/test_metrics.py:650-688.
"""
import math

import numpy as np
import pytest
import torch

from _golden import load, jload

T = torch.tensor


# ---- the reference's 6x6 tables ------------------------------------------------------
INST_A = [[1, 1, 1, 1, 1, 1],
          [1, 2, 2, 2, 2, 1],
          [1, 2, 2, 2, 2, 1],
          [1, 2, 2, 2, 2, 1],
          [1, 2, 2, 1, 1, 1],
          [1, 2, 1, 1, 1, 1]]
CAT_WRONG = [[0, 0, 0, 0, 0, 0],
             [0, 1, 0, 0, 1, 0],
             [0, 1, 1, 1, 1, 0],
             [0, 1, 1, 1, 1, 0],
             [0, 0, 0, 0, 0, 0],
             [0, 0, 0, 0, 0, 0]]
GT_IOU = [[1, 1, 1, 1, 1, 1],
          [1, 1, 1, 1, 1, 1],
          [1, 1, 2, 2, 2, 1],
          [1, 2, 2, 2, 2, 1],
          [1, 1, 1, 1, 1, 1],
          [1, 1, 1, 1, 1, 1]]
GOOD_DET = [[1, 1, 1, 1, 1, 1],
            [1, 1, 1, 1, 1, 1],
            [1, 2, 2, 2, 2, 1],
            [1, 2, 2, 2, 1, 1],
            [1, 1, 1, 1, 1, 1],
            [1, 1, 1, 1, 1, 1]]
BAD_DET = [[1, 1, 1, 1, 1, 1],
           [1, 1, 1, 1, 1, 1],
           [1, 1, 1, 2, 2, 1],
           [1, 1, 1, 2, 2, 1],
           [1, 1, 1, 2, 2, 1],
           [1, 1, 1, 1, 1, 1]]
CAT_STRIPES = [[1, 1, 1, 1, 1, 1],
               [1, 1, 1, 1, 1, 1],
               [1, 2, 2, 1, 2, 2],
               [1, 2, 2, 1, 2, 2],
               [1, 1, 1, 1, 1, 1],
               [1, 1, 1, 1, 1, 1]]
INST_RIGHT = [[0, 0, 0, 0, 0, 0],
              [0, 0, 0, 0, 0, 0],
              [0, 0, 0, 0, 1, 1],
              [0, 0, 0, 0, 1, 1],
              [0, 0, 0, 0, 0, 0],
              [0, 0, 0, 0, 0, 0]]
INST_LEFT = [[0, 0, 0, 0, 0, 0],
             [0, 0, 0, 0, 0, 0],
             [0, 1, 1, 0, 0, 0],
             [0, 1, 1, 0, 0, 0],
             [0, 0, 0, 0, 0, 0],
             [0, 0, 0, 0, 0, 0]]


def A(x):
    return np.asarray(x, dtype=np.int64)[None]


KNOWN = {
    # name: (params, [(pred, target), ...], expected state rows iou/tp/fn/fp)
    'perfect_match': (dict(num_categories=1, ignored_label=2, max_instances_per_category=16,
                           offset=16, is_thing=[True]),
                      [(A(INST_A), A(INST_A))],
                      [[2.0], [2], [0], [0]]),
    'totally_wrong': (dict(num_categories=2, ignored_label=2, max_instances_per_category=1,
                           offset=16, is_thing=[True, True]),
                      [(1 - A(CAT_WRONG), A(CAT_WRONG))],
                      [[0.0, 0.0], [0, 0], [1, 1], [1, 1]]),
    'matches_by_iou_good': (dict(num_categories=1, ignored_label=2,
                                 max_instances_per_category=16, offset=16, is_thing=[True]),
                            [(A(GOOD_DET), A(GT_IOU))],
                            [[28 / 30 + 6 / 8], [2], [0], [0]]),
    'matches_by_iou_bad': (dict(num_categories=1, ignored_label=2,
                                max_instances_per_category=16, offset=16, is_thing=[True]),
                           [(A(BAD_DET), A(GT_IOU))],
                           [[27 / 32], [1], [1], [1]]),
    'wrong_instances': (dict(num_categories=3, ignored_label=0, max_instances_per_category=10,
                             offset=100, is_thing=[True, True, True]),
                        [(A(CAT_STRIPES) * 10 + A(INST_RIGHT), A(CAT_STRIPES) * 10)],
                        [[0.0, 1.0, 0.0], [0, 1, 0], [0, 0, 1], [0, 0, 2]]),
    'instance_order_is_arbitrary': (dict(num_categories=3, ignored_label=0,
                                         max_instances_per_category=10, offset=100,
                                         is_thing=[True, True, True]),
                                    [(A(CAT_STRIPES) * 10 + A(INST_RIGHT),
                                      A(CAT_STRIPES) * 10 + A(INST_LEFT))],
                                    [[0.0, 1.0, 2.0], [0, 1, 2], [0, 0, 0], [0, 0, 0]]),
}


@pytest.mark.parametrize('name', list(KNOWN.keys()))
def test_pq_known_answers_oracle(oracle, name):
    params, updates, want = KNOWN[name]
    state = None
    for pred, tgt in updates:
        for b in range(pred.shape[0]):
            *state, _ = oracle.pq_compare_and_accumulate(
                pred[b], tgt[b], params['num_categories'], params['ignored_label'],
                params['max_instances_per_category'], params['offset'], state=state)
    np.testing.assert_array_almost_equal(state[0], want[0])
    for s, w in zip(state[1:], want[1:]):
        np.testing.assert_array_equal(s, w)


# ---- GPU tier ------------------------------------------------------------------------
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize('name', list(KNOWN.keys()))
def test_pq_known_answers_hip(name):
    from nicr_mt_scene_analysis_amd import metric
    params, updates, want = KNOWN[name]
    pq = metric.PanopticQuality(**params)
    for pred, tgt in updates:
        pq.update(torch.from_numpy(pred), torch.from_numpy(tgt))
    np.testing.assert_array_almost_equal(pq.iou_per_class.cpu().numpy(), want[0])
    np.testing.assert_array_equal(pq.tp_per_class.cpu().numpy(), want[1])
    np.testing.assert_array_equal(pq.fn_per_class.cpu().numpy(), want[2])
    np.testing.assert_array_equal(pq.fp_per_class.cpu().numpy(), want[3])


@gpu
def test_pq_compute_values_hip():
    """compute(): numbers of reference tests/test_metrics.py:76-446"""
    from nicr_mt_scene_analysis_amd import metric
    p, _, _ = KNOWN['matches_by_iou_good']
    pq = metric.PanopticQuality(**p)
    pq.update(torch.from_numpy(A(GOOD_DET)), torch.from_numpy(A(GT_IOU)))
    r = pq.compute()
    assert r['pq_per_class'].cpu().numpy() == pytest.approx([(28 / 30 + 6 / 8) / 2], abs=0)
    assert float(r['all_pq']) == (28 / 30 + 6 / 8) / 2
    assert float(r['all_rq']) == 1.0 and int(r['all_num_categories']) == 1
    pq.reset()
    pq.update(torch.from_numpy(A(BAD_DET)), torch.from_numpy(A(GT_IOU)))
    r = pq.compute()
    assert float(r['all_pq']) == 27 / 32 / 2 and float(r['all_rq']) == 0.5
    assert float(r['all_sq']) == 27 / 32

    p, updates, _ = KNOWN['wrong_instances']
    pq = metric.PanopticQuality(**p)
    pq.update(*[torch.from_numpy(x) for x in updates[0]])
    r = pq.compute()
    np.testing.assert_array_equal(r['pq_per_class'].cpu().numpy(), [0.0, 1.0, 0.0])
    assert float(r['all_pq']) == 0.5 and int(r['all_num_categories']) == 2

    # multiple batches (batch size 2, two updates; note the swapped argument order of the
    # reference test :406 / :423)
    pq = metric.PanopticQuality(num_categories=1, ignored_label=2, max_instances_per_category=16,
                                offset=16, is_thing=[True])
    gt2 = torch.from_numpy(np.concatenate([A(GT_IOU), A(GT_IOU)]))
    good2 = torch.from_numpy(np.concatenate([A(GOOD_DET), A(GOOD_DET)]))
    bad2 = torch.from_numpy(np.concatenate([A(BAD_DET), A(BAD_DET)]))
    pq.update(gt2, good2)
    pq.update(gt2, bad2)
    r = pq.compute()
    np.testing.assert_array_equal(r['pq_per_class'].cpu().numpy(),
                                  [((28 / 30 + 6 / 8) + (27 / 32)) / 2 / 2])
    np.testing.assert_array_equal(r['rq_per_class'].cpu().numpy(), [3 / 4])
    np.testing.assert_array_equal(r['sq_per_class'].cpu().numpy(),
                                  [((28 / 30 + 6 / 8) + (27 / 32)) / 3])
    np.testing.assert_almost_equal(float(r['all_pq']), 0.63177083)
    np.testing.assert_almost_equal(float(r['all_sq']), 0.84236111)
    assert float(r['all_rq']) == 0.75

    # nothing valid -> zeros (pq.py:352-359)
    pq = metric.PanopticQuality(num_categories=2, ignored_label=0, max_instances_per_category=16,
                                offset=256, is_thing=[False, True])
    r = pq.compute()
    assert int(r['all_pq']) == 0 and int(r['things_num_categories']) == 0


@gpu
def test_pq_random_vs_golden_bit_exact():
    from nicr_mt_scene_analysis_amd import metric
    g = load('metric_cases')
    p = jload(g['pq_params'])
    for name in ('shift', 'indep'):
        pq = metric.PanopticQualityWithOrientationMAE(is_thing=[c >= p['num_categories'] // 2
                                                                for c in range(p['num_categories'])],
                                                      **p)
        pred, tgt = g[f'pq_{name}__pred'], g[f'pq_{name}__target']
        B = pred.shape[0]
        res = pq._device_update(torch.from_numpy(pred), torch.from_numpy(tgt), want_matches=True)
        state = np.stack([getattr(pq, n).cpu().numpy() for n in
                          ('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class')])
        assert (state == g[f'pq_{name}__state']).all()          # fp64 sums bit-identical
        matches, n = res[0].cpu().numpy(), res[1].cpu().numpy()
        want = jload(g[f'pq_{name}__matches'])
        for b in range(B):
            assert sorted(map(tuple, matches[b, :n[b]].tolist())) == [tuple(x) for x in want[b]]
        pq._check_status()


@gpu
def test_pq_full_size_vs_oracle(oracle):
    """cfg4 shapes: 640x480 panoptic maps with ~50 segments, 41 categories."""
    from nicr_mt_scene_analysis_amd import metric
    rng = np.random.default_rng(5)
    H, W, ncat = 480, 640, 41
    B = 3
    cls = np.repeat(np.repeat(rng.integers(0, ncat, (B, H // 32, W // 32)), 32, 1), 32, 2)
    ins = np.repeat(np.repeat(rng.integers(0, 3, (B, H // 16, W // 16)), 16, 1), 16, 2)
    pred = (cls * 65536 + ins * (cls >= 20)).astype(np.int64)
    tgt = np.roll(pred, (5, 7), axis=(1, 2))
    tgt[:, :11] = 0
    pq = metric.PanopticQuality(ncat, 0, 65536, 256 ** 3, [c >= 20 for c in range(ncat)])
    pq.update(torch.from_numpy(pred), torch.from_numpy(tgt))
    state = None
    for b in range(B):
        *state, _ = oracle.pq_compare_and_accumulate(pred[b], tgt[b], ncat, 0, 65536, 256 ** 3,
                                                     state=state)
    got = np.stack([getattr(pq, n).cpu().numpy() for n in
                    ('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class')])
    assert (got == np.stack(state)).all()
    pq.compute()


@gpu
def test_pq_many_intersections_cfg5_shape(oracle):
    """configs[4] shape: 1024x768 maps with 151 categories made of 16-px blocks -> ~8500 distinct
    (target, pred) intersections per image (the round-1 tables held 4096).  States bit-exact
    vs the oracle, twice (the second update runs on the tables the first one left clean)."""
    from nicr_mt_scene_analysis_amd import metric
    rng = np.random.default_rng(11)
    H, W, ncat, B = 768, 1024, 151, 2
    cls = np.repeat(np.repeat(rng.integers(0, ncat, (B, H // 16, W // 16)), 16, 1), 16, 2)
    ins = np.repeat(np.repeat(rng.integers(0, 3, (B, H // 32, W // 32)), 32, 1), 32, 2)
    pred = (cls * 65536 + ins * (cls >= 75)).astype(np.int64)
    tgt = np.roll(pred, (5, 7), axis=(1, 2))
    tgt[:, :11] = 0
    n_int = len(np.unique(tgt[0] * 256 ** 3 + pred[0]))
    assert 4096 < n_int <= 16384, n_int
    pq = metric.PanopticQuality(ncat, 0, 65536, 256 ** 3, [c >= 75 for c in range(ncat)])
    state = None
    for rep in range(2):
        pq.update(torch.from_numpy(pred), torch.from_numpy(tgt))
        for b in range(B):
            *state, _ = oracle.pq_compare_and_accumulate(pred[b], tgt[b], ncat, 0, 65536, 256 ** 3,
                                                         state=state)
        got = np.stack([getattr(pq, n).cpu().numpy() for n in
                        ('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class')])
        assert (got == np.stack(state)).all(), rep
    pq.compute()


@gpu
def test_pq_table_overflow_is_reported_and_tables_recover():
    """per-pixel noise: more distinct intersections than the image's table holds -> ValueError
    at compute(); the next update on the same workspace is correct again"""
    from nicr_mt_scene_analysis_amd import metric
    rng = np.random.default_rng(3)
    H, W, ncat = 96, 128, 5
    pq = metric.PanopticQuality(ncat, 0, 65536, 256 ** 3, [False, False, True, True, True])
    noise = (rng.integers(2, 5, (1, H, W)) * 65536 + rng.integers(0, 2000, (1, H, W))).astype(np.int64)
    pq.update(torch.from_numpy(noise), torch.from_numpy(np.roll(noise, 1, 2)))
    with pytest.raises(ValueError, match='more distinct segments'):
        pq.compute()
    pq.reset()
    # the overflowed update left the workspace clean: a small update on it is exact again
    small = (rng.integers(0, 5, (1, H // 8, W // 8)).repeat(8, 1).repeat(8, 2) * 65536).astype(np.int64)
    pq.update(torch.from_numpy(small), torch.from_numpy(small))
    r = pq.compute()
    assert float(r['all_pq']) == 1.0


@gpu
def test_pq_error_reporting():
    from nicr_mt_scene_analysis_amd import metric
    pq = metric.PanopticQuality(2, 0, 16, 256, [False, True])
    pq.update(T([[[5 * 16 + 1, 17]]]), T([[[5 * 16 + 1, 17]]]))     # category 5 >= 2
    with pytest.raises(ValueError):
        pq.compute()


@gpu
def test_miou_vs_golden():
    from nicr_mt_scene_analysis_amd import metric
    g = load('metric_cases')
    for n in (5, 41, 101):
        pred, tgt = g[f'miou_{n}__pred'], g[f'miou_{n}__target']
        for ign in (0, 1):
            m = metric.MeanIntersectionOverUnion(n, ignore_first_class=bool(ign))
            m.update(torch.from_numpy(pred[:2]).cuda(), torch.from_numpy(tgt[:2]).cuda())
            m.update(torch.from_numpy(pred[2:]).long(), torch.from_numpy(tgt[2:]))   # CPU in, mixed dtypes
            assert (m.confmat.cpu().numpy() == g[f'miou_{n}_{ign}__confmat']).all()
            miou, ious = m.compute(return_ious=True)
            np.testing.assert_allclose(float(miou), g[f'miou_{n}_{ign}__miou'], rtol=1e-5)
            np.testing.assert_allclose(ious.cpu().numpy(), g[f'miou_{n}_{ign}__ious'], rtol=1e-5,
                                       equal_nan=True)
            m.reset()
            assert int(m.confmat.sum()) == 0


@gpu
@pytest.mark.parametrize('n_classes_without_void', (5, 40, 100))
def test_own_miou_void_handling(n_classes_without_void):
    """restates reference tests/test_miou.py:92-164 (without torchmetrics: the
    two void treatments must agree, and equal a direct confusion-matrix IoU)"""
    from nicr_mt_scene_analysis_amd import metric
    n = n_classes_without_void
    g = torch.Generator().manual_seed(n)
    m = metric.MeanIntersectionOverUnion(n_classes=n)
    m_void = metric.MeanIntersectionOverUnion(n_classes=n + 1, ignore_first_class=True)
    m_fused = metric.MeanIntersectionOverUnion(n_classes=n)
    cm = np.zeros((n, n), np.int64)
    for _ in range(4):
        preds = torch.rand((16, n, 100, 100), generator=g).argmax(dim=1)
        target = (torch.rand((16, 100, 100), generator=g) * (n + 1)).long()
        m_void.update(preds + 1, target)
        m_fused.update_masked_void(preds, target)
        mask = target != 0
        p, t = preds[mask], target[mask] - 1
        m.update(p, t)
        np.add.at(cm, (t.numpy(), p.numpy()), 1)
    miou, ious = m.compute(return_ious=True)
    miou_v, ious_v = m_void.compute(return_ious=True)
    assert (m.confmat.cpu().numpy() == cm).all()
    assert (m_fused.confmat.cpu().numpy() == cm).all()
    tp = np.diag(cm).astype(np.float64)
    ref = (tp / (cm.sum(0) + cm.sum(1) - tp)).mean()
    assert torch.allclose(miou, miou_v)
    np.testing.assert_allclose(float(miou), ref, rtol=1e-5)
    assert torch.isnan(ious_v[0])
    assert (ious_v[1:] == ious).all()


@gpu
@pytest.mark.parametrize('mode', ['plain', 'masked_void'])
@pytest.mark.parametrize('kind', ['coherent', 'noise', 'flat', 'ragged', 'bad'])
def test_confmat_uint8_maps_vs_oracle(oracle, kind, mode):
    """k_confmat_u8 (both maps uint8: 16 px per lane, runs merged in registers): equal to the
    oracle and to the general kernel on int64 copies of the maps — coherent label maps, incoherent
    pixels, a single bin (the wave-uniform shortcut), a size whose end falls inside a lane's 16
    pixels, and a class beyond the matrix (status)"""
    from nicr_mt_scene_analysis_amd import metric
    n = 41
    rng = np.random.default_rng(len(kind) * 7 + len(mode))
    shape = (3, 97, 131) if kind == 'ragged' else (4, 96, 256)
    if kind in ('coherent', 'ragged', 'bad'):
        cell = rng.integers(0, n, (shape[0], (shape[1] + 15) // 16, (shape[2] + 31) // 32))
        pred = np.repeat(np.repeat(cell, 16, 1), 32, 2)[:, :shape[1], :shape[2]]
        cell = rng.integers(0, n, (shape[0], (shape[1] + 23) // 24, (shape[2] + 23) // 24))
        tgt = np.repeat(np.repeat(cell, 24, 1), 24, 2)[:, :shape[1], :shape[2]]
    elif kind == 'noise':
        pred, tgt = rng.integers(0, n, shape), rng.integers(0, n, shape)
    else:
        pred, tgt = np.full(shape, 7), np.full(shape, 9)
    pred, tgt = np.ascontiguousarray(pred).astype(np.uint8), np.ascontiguousarray(tgt).astype(np.uint8)
    if kind == 'bad':
        pred[1, 5, 77] = 250
        tgt[2, 90, 3] = 99
    a = metric.MeanIntersectionOverUnion(n)
    b = metric.MeanIntersectionOverUnion(n)
    upd = (lambda m, p, t: m.update_masked_void(p, t)) if mode == 'masked_void' else (lambda m, p, t: m.update(p, t))
    for _ in range(2):
        upd(a, T(pred), T(tgt))
        upd(b, T(pred.astype(np.int64)), T(tgt.astype(np.int64)))
    torch.cuda.synchronize()
    assert torch.equal(a.confmat, b.confmat) and int(a._status) == int(b._status)
    assert (int(a._status) != 0) == (kind == 'bad')
    if kind != 'bad':
        if mode == 'masked_void':
            keep = tgt != 0
            want = oracle.confmat_update(pred[keep], tgt[keep] - 1, n)
        else:
            want = oracle.confmat_update(pred, tgt, n)
        assert np.array_equal(a.confmat.cpu().numpy(), 2 * want)
        assert int(a.confmat.sum()) > 0
    a._status.zero_()
    b._status.zero_()


@gpu
def test_miou_out_of_range_raises():
    from nicr_mt_scene_analysis_amd import metric
    m = metric.MeanIntersectionOverUnion(4)
    m.update(T([0, 1, 7]), T([0, 1, 3]))     # bin 3*4+7 = 19 >= 16 (reshape fails in the reference)
    with pytest.raises(ValueError):
        m.compute()


@gpu
def test_miou_from_panoptic():
    from nicr_mt_scene_analysis_amd import metric
    rng = np.random.default_rng(0)
    sem = rng.integers(0, 41, (2, 60, 80))
    pan = torch.from_numpy(sem * 65536 + rng.integers(0, 5, sem.shape))
    tgt = torch.from_numpy(rng.integers(0, 41, sem.shape).astype(np.uint8))
    a = metric.MeanIntersectionOverUnion(41, ignore_first_class=True)
    b = metric.MeanIntersectionOverUnion(41, ignore_first_class=True)
    a.update_from_panoptic(pan, tgt, 65536)
    b.update(pan // 65536, tgt)
    assert (a.confmat == b.confmat).all()


@pytest.mark.parametrize('mode', ('min', 'max'))
def test_mean_absolute_angular_error(mode):
    """restates reference tests/test_metrics.py:650-688 (pure host arithmetic)"""
    from nicr_mt_scene_analysis_amd.metric.mae import MeanAbsoluteAngularError
    mae = MeanAbsoluteAngularError(device='cpu')
    target_deg = 180.0 if mode == 'min' else 0.0
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        angles = torch.rand(50, generator=g) * 8 * math.pi - 4 * math.pi
        tgt, pred = {}, {}
        for k, a in enumerate(angles.tolist()):
            tgt[k] = a
            if mode == 'min':
                a = a - math.pi if torch.rand(1, generator=g).item() < 0.5 else a + math.pi
            pred[k] = a
        mae.update([tgt], [pred])
    rad, deg = mae.compute()
    np.testing.assert_almost_equal(float(torch.rad2deg(rad)), float(deg), decimal=5)
    np.testing.assert_almost_equal(float(deg), target_deg, decimal=4)


@gpu
def test_pq_with_orientation_mae():
    from nicr_mt_scene_analysis_amd import metric
    params, updates, _ = KNOWN['instance_order_is_arbitrary']
    pq = metric.PanopticQualityWithOrientationMAE(**params)
    pred, tgt = [torch.from_numpy(x) for x in updates[0]]
    # the left block is pred id 20 / target id 21 (the ids are swapped between the two maps):
    # pred id 20 <-> instance 1 with angle 0.5; target id 21 <-> instance 7 with angle 1.0
    pq.update(pred, [{1: 0.5}], [{20: 1}], tgt, [{7: 1.0}], [{21: 7}])
    r = pq.compute(suffix='_deeplab')
    np.testing.assert_allclose(float(r['mae_deeplab_rad']), 0.5, rtol=1e-6)
    assert int(pq.n_elements) == 1
    assert float(r['all_deeplab_pq']) == 1.0


@gpu
def test_bench_accumulators_reduce_through_metric_sync():
    """bench.py's MetricAccumulators under a (one-rank, gloo) process group: both schedules —
    `Metric.sync()` after every step into replicated totals, and local accumulation with ONE
    sync in finalize() — give the totals of direct accumulation without a group."""
    import socket
    import torch.distributed as dist
    from nicr_mt_scene_analysis_amd import ops
    from tools.bench_support import MetricAccumulators
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn
    inp = syn.make_panoptic_inputs_torch(2, 8, 96, 128, n_centers=6, seed=3, device='cuda')
    a = MetricAccumulators(9, torch.device('cuda'), inp, side_stream=False)
    assert not a.sync_every_step                        # no group yet: straight into the totals
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        b = MetricAccumulators(9, torch.device('cuda'), inp, side_stream=True, sync_every_step=True)
        c = MetricAccumulators(9, torch.device('cuda'), inp, side_stream=True, sync_every_step=False)
        assert b.sync_every_step and not c.sync_every_step
        assert b.payload_bytes == (9 * 9 + 4 * 9) * 8
        pan = ops.panoptic_pipeline(inp['semantic_logits'], inp['instance_center'],
                                    inp['instance_offset'], inp['semantic_classes_is_thing'])['panoptic']
        for _ in range(3):
            a.update_and_reduce(pan)
            b.update_and_reduce(pan, dist)
            c.update_and_reduce(pan, dist)
        assert not c.miou._is_synced
        c.finalize(dist)
        c.finalize(dist)                                # idempotent
        assert c.miou._is_synced and c.pq._is_synced    # the states ARE the summed totals now
        b.wait()
        c.wait()
        torch.cuda.synchronize()
        assert a.total_confmat.sum() > 0
        for other in (b, c):
            assert torch.equal(a.total_confmat, other.total_confmat)
            assert torch.equal(a.total_pq, other.total_pq)
    finally:
        dist.destroy_process_group()


@gpu
@pytest.mark.parametrize('max_inst,offset', [(1 << 16, 256 ** 3), (1000, 10 ** 7)])
@pytest.mark.parametrize('shape', [(3, 48, 64), (2, 37, 41), (1, 480, 640)])
def test_fused_pq_confmat_equals_separate_updates(shape, max_inst, offset):
    """nmsa_pq_update_with_confmat (one pass over the prediction) == nmsa_pq_update +
    nmsa_confmat_update, bit for bit, incl. odd sizes (scalar path) and repeated updates; with the
    reference's power-of-two divisor / offset (shifts, branch-free range tests) and with decimal
    ones (the generic instantiation: 64-bit division and multiply)"""
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality
    B, H, W = shape
    n = 9
    g = torch.Generator(device='cuda').manual_seed(B * 1000 + W)
    is_thing = [False, False, True, True, False, True, True, False, True]

    def blocky(hi):
        c = torch.randint(0, hi, (B, (H + 7) // 8, (W + 7) // 8), device='cuda', generator=g)
        return c.repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :H, :W].contiguous()
    pred = blocky(n) * max_inst + blocky(3)
    tgt = blocky(n) * max_inst + blocky(3)
    tsem = blocky(n).to(torch.uint8)
    pq_a, pq_b = (PanopticQuality(n, 0, max_inst, offset, is_thing, device='cuda') for _ in range(2))
    mi_a, mi_b = (MeanIntersectionOverUnion(n, device='cuda') for _ in range(2))
    for _ in range(2):
        pq_a.update(pred, tgt)
        mi_a.update_from_panoptic(pred, tsem, max_inst)
        pq_b.update_with_miou(pred, tgt, mi_b, tsem, max_inst)
    torch.cuda.synchronize()
    assert mi_a.confmat.sum() == 2 * B * H * W
    assert torch.equal(mi_a.confmat, mi_b.confmat)
    for name in ('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class'):
        assert torch.equal(getattr(pq_a, name), getattr(pq_b, name)), name
    # a label outside the matrix is reported through the mIoU status word
    bad = tsem.clone()
    bad[0, 0, 0] = 200
    pq_b.update_with_miou(pred, tgt, mi_b, bad, max_inst)
    with pytest.raises(ValueError):
        mi_b.compute()
    # so are a negative prediction and a class beyond the matrix (bincount's errors in the reference)
    for wrong in (-5, (n + 3) * max_inst):
        mi_c = MeanIntersectionOverUnion(n, device='cuda')
        pq_c = PanopticQuality(n, 0, max_inst, offset, is_thing, device='cuda')
        p2 = pred.clone()
        p2[-1, -1, -1] = wrong
        pq_c.update_with_miou(p2, tgt, mi_c, tsem, max_inst)
        with pytest.raises(ValueError):
            mi_c.compute()


@gpu
@pytest.mark.parametrize('max_inst,offset', [(1 << 16, 256 ** 3), (1000, 10 ** 7)])
@pytest.mark.parametrize('shape', [(3, 48, 64), (2, 37, 41), (2, 480, 640)])
def test_pq_confmat_from_the_parts_of_the_prediction(shape, max_inst, offset):
    """nmsa_pq_update_with_confmat_parts: the predicted panoptic id formed in registers from the
    parts the merge paints the map from (class u8, instance u8, pan_of_inst; nmsa_panoptic_paint's
    rule) == nmsa_pq_update_with_confmat on the painted map, bit for bit (both states, repeated
    updates, odd sizes on the scalar path), and the fall back when the parts are not the map's"""
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality
    B, H, W = shape
    C = 8
    n = C + 1
    g = torch.Generator(device='cuda').manual_seed(B * 77 + W)
    is_thing_c = torch.tensor([False, True, True, False, True, True, False, True], device='cuda')

    def blocky(hi, cell=8):
        c = torch.randint(0, hi, (B, (H + cell - 1) // cell, (W + cell - 1) // cell), device='cuda', generator=g)
        return c.repeat_interleave(cell, 1).repeat_interleave(cell, 2)[:, :H, :W].contiguous()
    sem = blocky(C).to(torch.uint8)
    inst = (blocky(6, 16) * is_thing_c[sem.long()]).to(torch.uint8)       # instances on thing classes only
    pan_of_inst = torch.zeros((B, 256), dtype=torch.int64, device='cuda')
    for b in range(B):
        for i in range(1, 6):
            pan_of_inst[b, i] = int(torch.randint(1, n, (1,), generator=g, device='cuda')) * max_inst + i
    thing_u8 = is_thing_c.to(torch.uint8)
    pred = torch.empty((B, H, W), dtype=torch.int64, device='cuda')
    L_ = ops.L
    L_.check(L_.lib().nmsa_panoptic_paint(L_.ptr(sem), L_.ptr(inst), L_.ptr(pan_of_inst), L_.ptr(thing_u8), B, C, H, W,
                                          max_inst, 0, L_.ptr(pred), None, L_.stream_ptr(pred.device)),
             'nmsa_panoptic_paint')
    tgt = blocky(n) * max_inst + blocky(3)
    tsem = blocky(n).to(torch.uint8)
    parts = {'panoptic': pred, 'semantic_idx_u8': sem, 'instance': inst, 'pan_of_inst': pan_of_inst,
             'is_thing': thing_u8, 'void_label': 0, 'max_instances_per_category': max_inst}
    is_thing = [False] + is_thing_c.tolist()
    pq_a, pq_b = (PanopticQuality(n, 0, max_inst, offset, is_thing, device='cuda') for _ in range(2))
    mi_a, mi_b = (MeanIntersectionOverUnion(n, device='cuda') for _ in range(2))
    assert PanopticQuality.parts_usable(parts, pred, max_inst)
    for _ in range(2):
        pq_a.update_with_miou(pred, tgt, mi_a, tsem, max_inst)
        pq_b.update_with_miou_parts(parts, tgt, mi_b, tsem, max_inst)
    torch.cuda.synchronize()
    assert mi_a.confmat.sum() == 2 * B * H * W and torch.equal(mi_a.confmat, mi_b.confmat)
    for name in ('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class'):
        assert torch.equal(getattr(pq_a, name), getattr(pq_b, name)), name
    assert float(pq_a.tp_per_class.sum()) + float(pq_a.fp_per_class.sum()) > 0
    # parts of ANOTHER map (a clone is not the painted tensor itself) are not used
    assert PanopticQuality.parts_usable(parts, pred[..., slice(0, H), slice(0, W)], max_inst)    # a full view of it
    assert not PanopticQuality.parts_usable(parts, pred[:, 1:], max_inst)
    assert not PanopticQuality.parts_usable(dict(parts, panoptic=pred.clone()), pred, max_inst)
    assert not PanopticQuality.parts_usable(parts, pred, max_inst + 1)


@gpu
@pytest.mark.parametrize('kind', ['noise', 'wide_ids', 'negative_ids', 'bad_classes', 'ragged'])
def test_pq_parts_hard_inputs(kind):
    """the compact-key count of the parts path (k_pq_count_parts) against the map path on what its
    fast path does not hold: more distinct (target, prediction) pairs in a workgroup than its LDS
    table has slots, target ids beyond 32 bits / negative ones (decoded pixel by pixel), classes the
    confusion matrix rejects, and sizes whose rows end inside a tile — states, confusion matrix AND
    both status words equal"""
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality
    B, H, W = (2, 63, 130) if kind == 'ragged' else (2, 96, 512)
    C, max_inst, offset = 8, 1 << 16, 256 ** 3
    n = C + 1
    g = torch.Generator(device='cuda').manual_seed(len(kind) * 1009)
    is_thing_c = torch.tensor([False, True, True, False, True, True, False, True], device='cuda')

    def blocky(hi, cell=8):
        c = torch.randint(0, hi, (B, (H + cell - 1) // cell, (W + cell - 1) // cell), device='cuda', generator=g)
        return c.repeat_interleave(cell, 1).repeat_interleave(cell, 2)[:, :H, :W].contiguous()
    sem = blocky(C).to(torch.uint8)
    inst = (blocky(6, 16) * is_thing_c[sem.long()]).to(torch.uint8)
    pan_of_inst = torch.zeros((B, 256), dtype=torch.int64, device='cuda')
    for b in range(B):
        for i in range(1, 6):
            pan_of_inst[b, i] = int(torch.randint(1, n, (1,), generator=g, device='cuda')) * max_inst + i
    if kind == 'bad_classes':
        pan_of_inst[0, 2] = (n + 3) * max_inst + 2          # a class beyond the matrix
        pan_of_inst[1, 3] = -7                              # bincount's negative
    thing_u8 = is_thing_c.to(torch.uint8)
    pred = torch.empty((B, H, W), dtype=torch.int64, device='cuda')
    L_ = ops.L
    L_.check(L_.lib().nmsa_panoptic_paint(L_.ptr(sem), L_.ptr(inst), L_.ptr(pan_of_inst), L_.ptr(thing_u8), B, C, H, W,
                                          max_inst, 0, L_.ptr(pred), None, L_.stream_ptr(pred.device)),
             'nmsa_panoptic_paint')
    tgt = blocky(n) * max_inst + blocky(3)
    tsem = blocky(n).to(torch.uint8)
    if kind == 'noise':           # 108 target ids pixel by pixel: ~1500 distinct triples per workgroup, 1024 slots
        tgt = torch.randint(0, n, (B, H, W), device='cuda', generator=g) * max_inst + \
            torch.randint(0, 12, (B, H, W), device='cuda', generator=g)
    elif kind == 'wide_ids':
        tgt[:, 10:30] += 1 << 33
        tgt[:, 50:52, ::3] = (1 << 40) + 5
    elif kind == 'negative_ids':
        tgt[0, 5:9, 100:300] = -3
        tgt[1, -1, -1] = -(1 << 35)
    elif kind == 'bad_classes':
        tsem[1, 40:44] = 200
    parts = {'panoptic': pred, 'semantic_idx_u8': sem, 'instance': inst, 'pan_of_inst': pan_of_inst,
             'is_thing': thing_u8, 'void_label': 0, 'max_instances_per_category': max_inst}
    is_thing = [False] + is_thing_c.tolist()
    pq_a, pq_b = (PanopticQuality(n, 0, max_inst, offset, is_thing, device='cuda') for _ in range(2))
    mi_a, mi_b = (MeanIntersectionOverUnion(n, device='cuda') for _ in range(2))
    assert PanopticQuality.parts_usable(parts, pred, max_inst)
    for _ in range(2):
        pq_a.update_with_miou(pred, tgt, mi_a, tsem, max_inst)
        pq_b.update_with_miou_parts(parts, tgt, mi_b, tsem, max_inst)
    torch.cuda.synchronize()
    assert int(pq_a._status) == int(pq_b._status) and int(mi_a._status) == int(mi_b._status)
    assert (int(pq_a._status) == 0) == (kind in ('noise', 'ragged'))      # (wide ids: categories out of range)
    assert (int(mi_a._status) != 0) == (kind == 'bad_classes')
    assert torch.equal(mi_a.confmat, mi_b.confmat) and int(mi_a.confmat.sum()) > 0
    for name in ('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class'):
        assert torch.equal(getattr(pq_a, name), getattr(pq_b, name)), name


@gpu
def test_pq_from_the_parts_with_more_classes_than_the_fused_matrix(oracle):
    """100 classes: the confusion matrix cannot ride in the PQ count (64 at most), so
    `update_with_miou_parts` runs the matrix over the map and the PQ count from the parts alone
    (k_pq_count_parts without its histogram) — states and matrix equal to the map path and to
    the oracle"""
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality
    B, H, W, C, max_inst, offset = 2, 96, 512, 100, 1 << 16, 256 ** 3

    def D(a):                                   # numpy -> device (the raw C-ABI takes device pointers only)
        return torch.as_tensor(np.ascontiguousarray(a)).cuda()
    n = C + 1
    rng = np.random.default_rng(5)
    thing_c = rng.integers(0, 2, C).astype(bool)

    def blocky(hi, cell):
        c = rng.integers(0, hi, (B, (H + cell - 1) // cell, (W + cell - 1) // cell))
        return np.repeat(np.repeat(c, cell, 1), cell, 2)[:, :H, :W]
    sem = blocky(C, 8).astype(np.uint8)
    inst = (blocky(9, 16) * thing_c[sem]).astype(np.uint8)
    pan_of_inst = np.zeros((B, 256), np.int64)
    pan_of_inst[:, 1:9] = rng.integers(1, n, (B, 8)) * max_inst + np.arange(1, 9)
    tgt = blocky(n, 12) * max_inst + blocky(3, 12)
    tsem = blocky(n, 12).astype(np.uint8)
    d_sem, d_inst, d_poi, d_thing = D(sem), D(inst), D(pan_of_inst), D(thing_c.astype(np.uint8))
    d_tgt, d_tsem = D(tgt), D(tsem)
    d_pred = torch.empty((B, H, W), dtype=torch.int64, device='cuda')
    L_ = ops.L
    L_.check(L_.lib().nmsa_panoptic_paint(L_.ptr(d_sem), L_.ptr(d_inst), L_.ptr(d_poi), L_.ptr(d_thing), B, C, H, W,
                                          max_inst, 0, L_.ptr(d_pred), None, L_.stream_ptr(d_pred.device)),
             'nmsa_panoptic_paint')
    parts = {'panoptic': d_pred, 'semantic_idx_u8': d_sem, 'instance': d_inst, 'pan_of_inst': d_poi,
             'is_thing': d_thing, 'void_label': 0, 'max_instances_per_category': max_inst}
    is_thing = [False] + thing_c.tolist()
    pq_a, pq_b = (PanopticQuality(n, 0, max_inst, offset, is_thing, device='cuda') for _ in range(2))
    mi_a, mi_b = (MeanIntersectionOverUnion(n, device='cuda') for _ in range(2))
    assert not pq_b._can_fuse(d_pred, mi_b, d_tsem)
    for _ in range(2):
        pq_a.update_with_miou(d_pred, d_tgt, mi_a, d_tsem, max_inst)
        pq_b.update_with_miou_parts(parts, d_tgt, mi_b, d_tsem, max_inst)
    torch.cuda.synchronize()
    assert int(pq_a._status) == 0 and int(pq_b._status) == 0 and int(mi_b._status) == 0
    assert torch.equal(mi_a.confmat, mi_b.confmat)
    pred = d_pred.cpu().numpy()
    state = None
    for _ in range(2):
        for b in range(B):
            *state, _ = oracle.pq_compare_and_accumulate(pred[b], tgt[b], n, 0, max_inst, offset, state=state)
    for name, w in zip(('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class'), state):
        assert torch.equal(getattr(pq_a, name), getattr(pq_b, name)), name
        assert np.array_equal(getattr(pq_b, name).cpu().numpy(), np.asarray(w, dtype=np.float64)), name


@gpu
def test_compare_and_accumulate_function(oracle):
    """module-level compare_and_accumulate (reference pq.py:60-179 signature) on the HIP path"""
    from nicr_mt_scene_analysis_amd.metric.pq import compare_and_accumulate
    rng = np.random.default_rng(5)
    blocks = rng.integers(0, 6, (2, 6, 8))
    pred = np.repeat(np.repeat(blocks[0] * 65536 + rng.integers(0, 3, (6, 8)), 8, 0), 8, 1).astype(np.int64)
    tgt = np.repeat(np.repeat(blocks[1] * 65536 + rng.integers(0, 3, (6, 8)), 8, 0), 8, 1).astype(np.int64)
    iou, tp, fn, fp, matched = compare_and_accumulate(T(pred), T(tgt), 6, 0, 1 << 16, 256 ** 3, 0)
    w_iou, w_tp, w_fn, w_fp, w_m = oracle.pq_compare_and_accumulate(pred, tgt, 6, 0, 1 << 16, 256 ** 3)
    assert np.array_equal(iou.cpu().numpy(), w_iou) and np.array_equal(tp.cpu().numpy(), w_tp)
    assert np.array_equal(fn.cpu().numpy(), w_fn) and np.array_equal(fp.cpu().numpy(), w_fp)
    assert matched == set(w_m)
