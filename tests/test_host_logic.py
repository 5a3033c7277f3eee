"""
CPU tier: host-side logic of the reference-shaped API (no HIP calls): lazy result dict,
factories and their error behaviour, metric `compute()` arithmetic on hand-set states
(against the reference's golden numbers), MAAE arithmetic, loss accumulation, batch helpers.
"""
import math

import numpy as np
import pytest
import torch

from _golden import load


def test_lazy_dict_semantics():
    from nicr_mt_scene_analysis_amd.model.postprocessing._lazy import LazyDict
    calls = []
    d = LazyDict(a=1)
    d.set_lazy('b', lambda: calls.append('b') or 2)
    d.set_lazy('c', lambda: calls.append('c') or d['b'] + 1)
    assert list(d.keys()) == ['a', 'b', 'c'] and 'b' in d and len(d) == 3
    assert calls == [] and d.is_pending('b')
    assert d['c'] == 3 and calls == ['c', 'b']          # forced on demand, dependency first
    assert d['b'] == 2 and calls == ['c', 'b']          # computed once
    e = LazyDict(x=0)
    e.set_lazy('y', lambda: calls.append('y') or 5)
    d.merge(e)
    assert d.is_pending('y') and calls == ['c', 'b']
    assert dict(d.items())['y'] == 5 and calls == ['c', 'b', 'y']
    d['y'] = 7
    assert d.get('y') == 7 and d.get('nope', 'dflt') == 'dflt'
    d.set_lazy('z', lambda: 9)
    assert d.pop('z') == 9 and 'z' not in d
    assert {**d}['c'] == 3                               # plain-dict unpacking works


# keys of the inference output dict the reference checks for a panoptic + orientation model
# (reference tests/test_decoders+postprocessing.py:208-258)
REFERENCE_OUTPUT_KEYS = (
    'semantic_output', 'semantic_side_outputs', 'semantic_softmax_scores',
    'semantic_segmentation_score', 'semantic_segmentation_idx', 'semantic_output_fullres',
    'semantic_softmax_scores_fullres', 'semantic_segmentation_score_fullres',
    'semantic_segmentation_idx_fullres', 'instance_output', 'instance_side_outputs',
    'instance_centers', 'instance_offsets', 'instance_segmentation_gt_foreground',
    'instance_segmentation_gt_meta', 'instance_segmentation_gt_foreground_fullres',
    'panoptic_foreground_mask', 'panoptic_segmentation_deeplab',
    'panoptic_segmentation_deeplab_fullres', 'panoptic_segmentation_deeplab_ids',
    'panoptic_segmentation_deeplab_semantic_idx',
    'panoptic_segmentation_deeplab_semantic_idx_fullres',
    'panoptic_segmentation_deeplab_semantic_score',
    'panoptic_segmentation_deeplab_semantic_score_fullres',
    'panoptic_segmentation_deeplab_instance_idx',
    'panoptic_segmentation_deeplab_instance_idx_fullres',
    'panoptic_segmentation_deeplab_instance_meta',
    'panoptic_segmentation_deeplab_instance_score',
    'panoptic_segmentation_deeplab_instance_score_fullres',
    'panoptic_segmentation_deeplab_panoptic_score',
    'panoptic_segmentation_deeplab_panoptic_score_fullres',
    'orientations_panoptic_segmentation_deeplab_instance',
    'orientations_gt_instance_gt_orientation_foreground',
    'orientations_instance_segmentation_gt_orientation_foreground')


def _all_pending_result():
    """a LazyDict with EVERY reference key pending (set_lazy / set_derived alternating)"""
    from nicr_mt_scene_analysis_amd.model.postprocessing._lazy import LazyDict
    r = LazyDict()
    for i, k in enumerate(REFERENCE_OUTPUT_KEYS):
        if i % 2:
            r.set_derived(k, (lambda kk: (lambda d: ('value', kk)))(k))
        else:
            r.set_lazy(k, (lambda kk: (lambda: ('value', kk)))(k))
    return r


def _assert_plain_and_complete(d, kind=dict):
    assert type(d) is kind, type(d)
    assert list(d.keys()) == list(REFERENCE_OUTPUT_KEYS)
    for k in REFERENCE_OUTPUT_KEYS:
        assert d[k] == ('value', k), (k, d[k])


def test_lazy_dict_is_a_drop_in_dict_under_merge_idioms():
    """the reference merges the per-task result dicts with {**a, **b}
    (model/postprocessing/panoptic.py:75,94); no idiom may see a placeholder"""
    import copy
    import pickle
    from nicr_mt_scene_analysis_amd.model.postprocessing._lazy import LazyDict

    _assert_plain_and_complete({**_all_pending_result()})
    _assert_plain_and_complete(dict(_all_pending_result()))
    x = {}
    x.update(_all_pending_result())
    _assert_plain_and_complete(x)
    _assert_plain_and_complete({**{}, **_all_pending_result()})
    _assert_plain_and_complete(dict(_all_pending_result(), **{}))
    _assert_plain_and_complete({} | _all_pending_result(), LazyDict)
    _assert_plain_and_complete(_all_pending_result() | {}, LazyDict)
    _assert_plain_and_complete((lambda **kw: kw)(**_all_pending_result()))
    _assert_plain_and_complete(pickle.loads(pickle.dumps(_all_pending_result())))
    _assert_plain_and_complete(copy.deepcopy(_all_pending_result()))
    _assert_plain_and_complete({k: v for k, v in _all_pending_result().items()})
    assert list(_all_pending_result().values()) == [('value', k) for k in REFERENCE_OUTPUT_KEYS]
    assert sorted(_all_pending_result()) == sorted(REFERENCE_OUTPUT_KEYS)       # iteration
    r = _all_pending_result()
    assert r == {k: ('value', k) for k in REFERENCE_OUTPUT_KEYS}
    assert not (_all_pending_result() != {k: ('value', k) for k in REFERENCE_OUTPUT_KEYS})
    assert 'None' not in repr(_all_pending_result())

    # shallow copies stay lazy and independent
    r = _all_pending_result()
    for c in (copy.copy(r), r.copy()):
        assert isinstance(c, LazyDict) and c.is_pending('semantic_softmax_scores')
        _assert_plain_and_complete(dict(c))
    assert r.is_pending('semantic_softmax_scores')

    # writes win over pending thunks
    k0, k1, k2, k3 = REFERENCE_OUTPUT_KEYS[:4]
    r = _all_pending_result()
    r.update({k0: 99})
    r.update([(k1, 98)], **{k2: 97})
    r |= {k3: 96}
    assert (r[k0], r[k1], r[k2], r[k3]) == (99, 98, 97, 96)
    assert dict(r)[k0] == 99 and not r.is_pending(k0)
    r = _all_pending_result()
    assert r.setdefault(k0, 'dflt') == ('value', k0) and r.setdefault('new', 5) == 5
    del r[k1]
    assert k1 not in r and not r.is_pending(k1) and k1 not in dict(r)
    r.set_lazy('tail', lambda: 'T')
    assert r.popitem() == ('tail', 'T')
    r.clear()
    assert len(r) == 0 and dict(r) == {}
    with pytest.raises(TypeError):
        hash(_all_pending_result())


def test_factory_and_constructor_errors():
    from nicr_mt_scene_analysis_amd.model.postprocessing import (
        InstancePostprocessing, get_postprocessing_class)
    with pytest.raises(ValueError, match='Unknown postprocessing'):
        get_postprocessing_class('foo')
    with pytest.raises(NotImplementedError):
        get_postprocessing_class('normal')
    cls = get_postprocessing_class('instance', top_k_instances=12)
    assert issubclass(cls, InstancePostprocessing) and cls()._top_k_instances == 12
    assert get_postprocessing_class('instance') is InstancePostprocessing
    with pytest.raises(AssertionError):
        InstancePostprocessing(heatmap_nms_kernel_size=2)
    with pytest.raises(AssertionError):
        InstancePostprocessing(top_k_instances=255)
    p = get_postprocessing_class('panoptic')(
        semantic_postprocessing=get_postprocessing_class('semantic')(),
        instance_postprocessing=cls(), semantic_classes_is_thing=(False, True, True),
        semantic_class_has_orientation=(False, False, True))
    assert p.max_instances_per_category == 65536
    assert list(p._thing_ids_panoptic) == [2, 3] and list(p._orientation_ids) == [3]


def test_training_postprocess_is_passthrough_on_cpu():
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    post = get_postprocessing_class('semantic')()
    x = torch.zeros((1, 3, 4, 4))
    assert post.postprocess((x, (None,)), {}, is_training=True) == {
        'semantic_output': x, 'semantic_side_outputs': (None,)}


def test_miou_compute_vs_reference_numbers():
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion
    g = load('metric_cases')
    for n in (5, 41, 101):
        for ign in (0, 1):
            m = MeanIntersectionOverUnion(n, ignore_first_class=bool(ign), device='cpu')
            m.confmat += torch.from_numpy(g[f'miou_{n}_{ign}__confmat'])
            miou, ious = m.compute(return_ious=True)
            np.testing.assert_allclose(float(miou), g[f'miou_{n}_{ign}__miou'], rtol=1e-6)
            np.testing.assert_allclose(ious.numpy(), g[f'miou_{n}_{ign}__ious'], rtol=1e-6,
                                       equal_nan=True)
            assert float(m.compute()) == float(miou)
            m.reset()
            assert int(m.confmat.sum()) == 0


def test_miou_update_needs_gpu():
    from nicr_mt_scene_analysis_amd import _lib as L
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion
    m = MeanIntersectionOverUnion(3, device='cpu')
    with pytest.raises(L.NmsaError):
        m.update(torch.zeros(4, dtype=torch.long), torch.zeros(4, dtype=torch.long))


def _pq_with_state(iou, tp, fn, fp, **kw):
    from nicr_mt_scene_analysis_amd.metric import PanopticQuality
    pq = PanopticQuality(device='cpu', **kw)
    for name, v in zip(('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class'),
                       (iou, tp, fn, fp)):
        getattr(pq, name).add_(torch.tensor(v, dtype=torch.float64))
    return pq


def test_pq_compute_reference_numbers():
    """states of reference tests/test_metrics.py:165-446 -> the numbers asserted there"""
    kw1 = dict(num_categories=1, ignored_label=2, max_instances_per_category=16, offset=16,
               is_thing=[True])
    r = _pq_with_state([28 / 30 + 6 / 8], [2], [0], [0], **kw1).compute()
    assert float(r['all_pq']) == (28 / 30 + 6 / 8) / 2 and float(r['all_rq']) == 1.0
    assert float(r['things_pq']) == float(r['all_pq']) and int(r['stuff_num_categories']) == 0
    r = _pq_with_state([27 / 32], [1], [1], [1], **kw1).compute()
    assert float(r['all_pq']) == 27 / 32 / 2 and float(r['all_sq']) == 27 / 32
    r = _pq_with_state([2 * (28 / 30 + 6 / 8) + 2 * (27 / 32)], [6], [2], [2], **kw1).compute()
    np.testing.assert_almost_equal(float(r['all_pq']), 0.63177083)
    np.testing.assert_almost_equal(float(r['all_sq']), 0.84236111)
    assert float(r['all_rq']) == 0.75

    kw3 = dict(num_categories=3, ignored_label=0, max_instances_per_category=10, offset=100,
               is_thing=[True, True, True])
    r = _pq_with_state([0.0, 1.0, 0.0], [0, 1, 0], [0, 0, 1], [0, 0, 2], **kw3).compute(suffix='_x')
    np.testing.assert_array_equal(r['pq_per_class'].numpy(), [0.0, 1.0, 0.0])
    assert float(r['all_x_pq']) == 0.5 and int(r['all_x_num_categories']) == 2
    assert set(k for k in r if k.endswith('_pq')) == {
        'all_x_pq', 'things_x_pq', 'stuff_x_pq', 'all_with_gt_x_pq', 'things_with_gt_x_pq',
        'stuff_with_gt_x_pq'}
    # FP-only class counts for `all` but not for `all_with_gt` (pq.py:317-338)
    r = _pq_with_state([0.0, 0.9, 0.0], [0, 1, 0], [0, 0, 0], [0, 0, 3], **kw3).compute()
    assert int(r['all_num_categories']) == 2 and int(r['all_with_gt_num_categories']) == 1
    # nothing valid -> zeros
    r = _pq_with_state([0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 0, 0], **kw3).compute()
    assert int(r['all_pq']) == 0 and int(r['all_num_categories']) == 0


def test_abs_angle_error_and_mae_arithmetic():
    from nicr_mt_scene_analysis_amd.metric.mae import (MeanAbsoluteAngularError,
                                                       abs_angle_error_rad)
    t = torch.tensor
    assert float(abs_angle_error_rad(t(0.1), t(2 * math.pi - 0.1))) == pytest.approx(0.2, abs=1e-6)
    assert float(abs_angle_error_rad(t(-3 * math.pi), t(0.0))) == pytest.approx(math.pi, abs=1e-5)
    assert float(abs_angle_error_rad(t(7.0), t(7.0 + 2 * math.pi))) == pytest.approx(0.0, abs=1e-5)
    m = MeanAbsoluteAngularError(device='cpu')
    m.update([{1: 0.5, 2: 1.0}], [{1: 0.25, 2: 1.5, 3: 9.0}])
    rad, deg = m.compute()
    assert float(rad) == pytest.approx(0.375, abs=1e-6)
    assert float(deg) == pytest.approx(math.degrees(0.375), abs=1e-4)
    assert int(m.n_elements) == 2


def test_pq_mae_matching_rules():
    """update_mae (reference metric/mae.py:129-162): id 0, unknown ids and missing
    orientations are skipped"""
    from nicr_mt_scene_analysis_amd.metric import PanopticQualityWithOrientationMAE
    pq = PanopticQualityWithOrientationMAE(num_categories=2, ignored_label=0,
                                           max_instances_per_category=16, offset=256,
                                           is_thing=[False, True], device='cpu')
    pq.update_mae(orientation_preds={1: 0.5, 2: 2.0}, panoptic_preds_id_dicts={17: 1, 18: 2},
                  orientation_target={7: 1.0}, panoptic_target_id_dicts={17: 7, 19: 8},
                  matching=[(0, 0), (17, 17), (19, 18), (20, 17), (17, 99)])
    assert int(pq.n_elements) == 1
    assert float(pq.sum_angular_error) == pytest.approx(0.5, abs=1e-6)


def test_metric_state_packing_and_reset():
    from nicr_mt_scene_analysis_amd.metric import PanopticQualityWithOrientationMAE
    pq = PanopticQualityWithOrientationMAE(num_categories=3, ignored_label=0,
                                           max_instances_per_category=16, offset=256,
                                           is_thing=[False, True, True], device='cpu')
    pq.tp_per_class += 2
    pq.n_elements += 3
    flat = pq._pack()
    assert set(flat) == {torch.float64, torch.int64}
    assert flat[torch.float64].numel() == 4 * 3 + 1 and flat[torch.int64].numel() == 1
    pq.tp_per_class += 1                                  # views stay attached to the flat buffer
    assert float(flat[torch.float64].sum()) == 9.0 and int(pq.n_elements) == 3
    pq.reset()
    assert float(flat[torch.float64].sum()) == 0.0 and int(pq.n_elements) == 0
    assert pq.state_names() == ['iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class',
                                'sum_angular_error', 'n_elements']


def test_loss_constructor_errors_and_no_cpu_path():
    """constructor / shape errors of the reference (loss/vonmises.py:32-40, asserts of the
    ctors) and: host tensors RAISE — the loss classes have no CPU path"""
    from nicr_mt_scene_analysis_amd._lib import NmsaError
    from nicr_mt_scene_analysis_amd.loss import (CosineEmbeddingLoss, CrossEntropyLossSemantic,
                                                 L1Loss, MSELoss, VonMisesLossBiternion)
    x, y = torch.rand((2, 2, 5, 7)), torch.rand((2, 2, 5, 7))
    with pytest.raises(AssertionError):
        MSELoss('avg')
    with pytest.raises(ValueError, match=r'shape \(n, 2\)'):
        VonMisesLossBiternion()([x], [y])
    with pytest.raises(AssertionError):
        VonMisesLossBiternion(reduction='mean')
    rows, tg = torch.rand((9, 2)), torch.rand((9, 2))
    for reduction in ('sum', 'mean', 'none'):
        for cls in (MSELoss, L1Loss):
            with pytest.raises(NmsaError, match='no CPU fallback'):
                cls(reduction)([x], [y])
        with pytest.raises(NmsaError, match='no CPU fallback'):
            CosineEmbeddingLoss(reduction)([rows], [tg])
    with pytest.raises(NmsaError, match='no CPU fallback'):
        VonMisesLossBiternion()([rows], [tg])
    with pytest.raises(NmsaError, match='no CPU fallback'):
        CosineEmbeddingLoss()._compute_loss(rows, tg, target_similarity=-torch.ones(9))
    with pytest.raises(NmsaError, match='no CPU fallback'):
        CrossEntropyLossSemantic()([torch.rand((1, 3, 4, 4))], [torch.zeros((1, 4, 4), dtype=torch.uint8)])


def test_task_helper_base_logic():
    from nicr_mt_scene_analysis_amd.task_helper import SemanticTaskHelper, get_total_loss_key
    from nicr_mt_scene_analysis_amd.task_helper.base import append_profile_to_logs
    h = SemanticTaskHelper(4)
    assert h.mark_as_total('semantic') == get_total_loss_key('semantic') == 'semantic_total_loss'
    out = h.accumulate_losses([torch.tensor(2.0), torch.tensor(4.0)], [3, 9])
    assert float(out) == 0.5
    out = h.accumulate_losses([torch.tensor(2.0)], [torch.tensor(4)])
    assert float(out) == 0.5
    main, side = torch.zeros((1, 4, 8, 16)), (torch.zeros((1, 4, 4, 8)), None, torch.zeros((1, 4, 1, 2)))
    t, keys, scales = h.collect_predictions_for_loss({'o': main, 's': side}, 'o', 's')
    assert keys == ['main', 'down_2', 'down_8'] and scales == [2, 8] and len(t) == 3
    batch = {'semantic': 'm', '_down_2': {'semantic': 'd2'}}
    assert h.collect_targets_for_loss(batch, 'semantic', [2, 8]) == ['m', 'd2']

    @append_profile_to_logs('t')
    def f():
        return {}, {}
    assert 't' in f()[1]


def test_batch_helpers():
    from nicr_mt_scene_analysis_amd.data.preprocessing import (
        APPLIED_PREPROCESSING_KEY, get_fullres, get_fullres_key, get_fullres_shape,
        get_valid_region_slices, get_valid_region_slices_and_fullres_shape)
    assert get_fullres_key('semantic') == 'semantic_fullres'
    b = {'depth_fullres': torch.zeros((2, 30, 40)),
         APPLIED_PREPROCESSING_KEY: [[{'type': 'Other'}, {'type': 'Resize',
                                                          'valid_region_slice_y': slice(1, 5),
                                                          'valid_region_slice_x': slice(0, 7)}]]}
    assert get_fullres(b, 'semantic') is None
    assert get_fullres_shape(b, 'semantic') == (30, 40)
    assert get_valid_region_slices(b) == (slice(1, 5), slice(0, 7))
    assert get_valid_region_slices_and_fullres_shape(b, 'x') == ((slice(1, 5), slice(0, 7)), (30, 40))
    with pytest.raises(ValueError):
        get_fullres_shape({}, 'semantic')
    with pytest.raises(ValueError):
        get_valid_region_slices({})


def test_crop_and_resize_host_side():
    """dense_base.py:15-58: no resize -> the cropped view itself; a real resize is HIP work
    (tests/test_fullres.py) and a CPU tensor must fail loudly, not fall back"""
    from nicr_mt_scene_analysis_amd._lib import NmsaError
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    post = get_postprocessing_class('semantic')()
    ids = torch.arange(2 * 6 * 8).reshape(2, 6, 8) * 65536
    same = post._crop_to_valid_region_and_resize_prediction(ids, (slice(0, 6), slice(0, 8)), (6, 8))
    assert same.data_ptr() == ids.data_ptr()
    crop = post._crop_to_valid_region_and_resize_prediction(ids, (slice(1, 5), slice(2, 8)), (4, 6))
    assert torch.equal(crop, ids[:, 1:5, 2:8])
    with pytest.raises(NmsaError):
        post._crop_to_valid_region_and_resize_prediction(ids, (slice(1, 5), slice(0, 8)), (8, 16))
    with pytest.raises(NotImplementedError):
        post._crop_to_valid_region_and_resize_prediction(ids, (slice(1, 5), slice(0, 8)), (8, 16),
                                                         mode='bicubic')


def test_bench_self_launch_command_and_host_cores(monkeypatch, capsys):
    """`python bench.py --gpus N` without a launcher: N fresh ranks as CHILD processes (no exec) with
    the environment a launcher gives them, the user's arguments forwarded, rendezvous on 127.0.0.1;
    a rank that fails is named with the tail of its stderr and its code is returned"""
    import importlib
    import sys
    bench = importlib.import_module('bench')
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, stderr=None):
            self.cmd, self.env, self.rank = cmd, env, int(env['RANK'])
            self.code = 7 if self.rank == 2 else 0
            if self.rank == 2:
                stderr.write('Traceback ...\nRuntimeError: boom on rank 2\n')
                stderr.flush()
            self.killed = False
            started.append(self)

        def poll(self):
            return self.code if self.rank in (0, 2) else None       # ranks 1 and 3 hang in a collective

        def terminate(self):
            self.killed = True

        def wait(self, timeout=None):
            return -15

        def kill(self):
            self.killed = True
    monkeypatch.setattr(bench.subprocess, 'Popen', FakeProc)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '9', '--warmup', '2'])
    monkeypatch.delenv('RANK', raising=False)
    args = bench.parse_args()
    assert bench.launch_ranks(args) == 7                        # the failed rank's return code
    assert [p.rank for p in started] == [0, 1, 2, 3]
    for p in started:
        assert p.cmd[0] == sys.executable and p.cmd[1] == bench.os.path.abspath(bench.__file__)
        assert p.cmd[2:] == ['--gpus', '4', '--steps', '9', '--warmup', '2']
        assert p.env['WORLD_SIZE'] == '4' and p.env['LOCAL_RANK'] == p.env['RANK']
        assert p.env['MASTER_ADDR'] == '127.0.0.1' and 0 < int(p.env['MASTER_PORT']) < 65536
        assert p.env.get('HSA_ENABLE_IPC_MODE_LEGACY') is not None and int(p.env['OMP_NUM_THREADS']) >= 1
    assert started[1].killed and started[3].killed and not started[0].killed
    err = capsys.readouterr().err
    assert 'rank 2 of 4 exited with code 7' in err and 'boom on rank 2' in err
    cores = bench.host_cores()
    assert 1 <= cores['usable'] <= cores['affinity'] <= max(cores['nproc'], cores['affinity'])
    assert cores['cgroup_quota'] is None or cores['usable'] <= max(1, round(cores['cgroup_quota']))


def test_metric_copies_keep_their_own_compute():
    """the rank-sync wrapper of `compute()` lives on the class: deepcopy / pickle of a metric give
    independent metrics (a wrapper bound to the instance would keep computing the original)"""
    import copy
    import pickle
    import torch
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion
    from nicr_mt_scene_analysis_amd.metric.mae import PanopticQualityWithOrientationMAE
    m = MeanIntersectionOverUnion(3, device='cpu')
    m.confmat += torch.eye(3, dtype=torch.int64)
    clone = copy.deepcopy(m)
    clone.confmat += 5
    assert float(m.compute()) == 1.0 and float(clone.compute()) < 0.5
    assert float(pickle.loads(pickle.dumps(m)).compute()) == 1.0
    q = PanopticQualityWithOrientationMAE(3, 0, 65536, 256 ** 3, [False, True, True], device='cpu')
    r = q.compute(suffix='_x')                       # subclass compute -> super().compute(): one context
    assert 'mae_x_rad' in r and 'all_x_pq' in r and q._compute_depth == 0 and not q._is_synced
