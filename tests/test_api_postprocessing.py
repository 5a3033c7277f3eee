"""
GPU tier: the reference-shaped Python API (postprocess(...) dicts,
deeplab_merge_batch) on top of the HIP kernels.

* restates reference tests/test_instance_postprocessing.py:90-150 (random
  two-rectangle scenes at 480x640: planted centers are found, every predicted
  instance maps to exactly one GT instance),
* checks the full PanopticPostprocessing dict against the golden vectors made
  by the reference's own postprocess(), incl. the key list of reference
  tests/test_decoders+postprocessing.py:208-258.
"""
import numpy as np
import pytest
import torch

from _golden import load, jload, ids_from_arrays, meta_from_arrays

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_batch(B, H, W, extra=None):
    from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY
    batch = {
        'rgb_fullres': torch.zeros((B, 3, H, W)),
        APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize',
                                      'valid_region_slice_y': slice(0, H),
                                      'valid_region_slice_x': slice(0, W)}]] * B,
    }
    batch.update(extra or {})
    return batch


# ---------------------------------------------------------------------------
def random_rectangles(batch, h, w, size_min, size_max, rng):
    inst = np.zeros((batch, h, w), np.uint8)
    heat = np.zeros((batch, 1, h, w), np.float32)
    off = np.zeros((batch, 2, h, w), np.float32)
    fg = np.zeros((batch, h, w), bool)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32),
                         indexing='ij')
    centers = []
    for b in range(batch):
        cs = []
        for k in range(2):
            cx, cy = int(rng.integers(0, w)), int(rng.integers(0, h))
            s = int(rng.integers(size_min, size_max))
            x0, x1 = max(cx - s, 0), min(cx + s, w)
            y0, y1 = max(cy - s, 0), min(cy + s, h)
            px, py = int(x1 - (x1 - x0) / 2), int(y1 - (y1 - y0) / 2)
            heat[b, 0, py, px] = 1
            cs.append((py, px))
            inst[b, y0:y1, x0:x1] = k + 1
            off[b, 0, y0:y1, x0:x1] = py - yy[y0:y1, x0:x1]
            off[b, 1, y0:y1, x0:x1] = px - xx[y0:y1, x0:x1]
            fg[b, y0:y1, x0:x1] = True
        centers.append(cs)
    return inst, heat, off, fg, centers


@pytest.mark.parametrize('batch_size', [1, 8])
@pytest.mark.parametrize('num_inst_size', [(5, 15), (20, 40), (30, 80)])
@pytest.mark.parametrize('seed', [0, 1, 2])
def test_instance_postprocessing_rectangles(batch_size, num_inst_size, seed):
    from nicr_mt_scene_analysis_amd.model.postprocessing import InstancePostprocessing
    h, w = 480, 640
    rng = np.random.default_rng(100 * seed + batch_size + num_inst_size[0])
    inst, heat, off, fg, centers = random_rectangles(batch_size, h, w, *num_inst_size, rng)
    # the reference test keeps centers away from the 1-px border implicitly by
    # construction only most of the time; drop scenes with a border center
    for cs in centers:
        for (y, x) in cs:
            if y in (0, h - 1) or x in (0, w - 1):
                pytest.skip('planted center on the image border')
        (ya, xa), (yb, xb) = cs
        if abs(ya - yb) <= 1 and abs(xa - xb) <= 1:
            pytest.skip('two equal peaks inside one NMS window (ambiguous by design)')

    post = InstancePostprocessing(normalized_offset=False)
    _, found = post._get_instance_centers(dev(heat))
    for f, c in zip(found, centers):
        got = sorted(map(tuple, f.cpu().numpy().tolist()))
        assert got == sorted(set(c))

    batch = make_batch(batch_size, h, w, {'instance_foreground': dev(fg),
                                          'instance_fullres': dev(fg)})
    r = post.postprocess(((dev(heat), dev(off)), None), batch, is_training=False)
    pred = r['instance_segmentation_gt_foreground'].cpu().numpy()
    assert r['instance_segmentation_gt_foreground_fullres'].shape == (batch_size, h, w)
    for b in range(batch_size):
        for i in np.unique(pred[b]):
            if i == 0:
                continue
            # the later rectangle overwrites the earlier one where they overlap, so a
            # predicted instance must cover exactly one GT id
            assert len(np.unique(inst[b][pred[b] == i])) == 1
        assert ((pred[b] > 0) == fg[b]).all()


# ---------------------------------------------------------------------------
EXPECTED_KEYS_SEMANTIC = [
    'semantic_output', 'semantic_side_outputs', 'semantic_softmax_scores',
    'semantic_segmentation_score', 'semantic_segmentation_idx',
    'semantic_output_fullres', 'semantic_softmax_scores_fullres',
    'semantic_segmentation_score_fullres', 'semantic_segmentation_idx_fullres']
EXPECTED_KEYS_INSTANCE = ['instance_output', 'instance_side_outputs', 'instance_centers',
                          'instance_offsets']
EXPECTED_KEYS_PANOPTIC = [
    'panoptic_foreground_mask', 'panoptic_segmentation_deeplab',
    'panoptic_segmentation_deeplab_fullres', 'panoptic_segmentation_deeplab_ids',
    'panoptic_segmentation_deeplab_semantic_idx',
    'panoptic_segmentation_deeplab_semantic_idx_fullres',
    'panoptic_segmentation_deeplab_instance_idx',
    'panoptic_segmentation_deeplab_instance_idx_fullres',
    'panoptic_segmentation_deeplab_instance_meta']
EXPECTED_KEYS_SCORES = [
    'panoptic_segmentation_deeplab_semantic_score',
    'panoptic_segmentation_deeplab_semantic_score_fullres',
    'panoptic_segmentation_deeplab_instance_score',
    'panoptic_segmentation_deeplab_instance_score_fullres',
    'panoptic_segmentation_deeplab_panoptic_score',
    'panoptic_segmentation_deeplab_panoptic_score_fullres']


def build_panoptic(is_thing, kw=None, compute_scores=False):
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    sem = get_postprocessing_class('semantic')()
    ins = get_postprocessing_class('instance')(**(kw or {}))
    return get_postprocessing_class('panoptic')(
        semantic_postprocessing=sem, instance_postprocessing=ins,
        semantic_classes_is_thing=tuple(bool(x) for x in is_thing),
        semantic_class_has_orientation=tuple(bool(x) for x in is_thing),
        compute_scores=compute_scores)


@pytest.mark.parametrize('name', ['panoptic_small', 'panoptic_small_kwargs',
                                  'panoptic_edges_plain', 'panoptic_edges_thr'])
def test_panoptic_postprocess_vs_golden(name):
    g = load(name)
    kw = jload(g['kwargs']) if 'kwargs' in g else None
    with_ori = 'in_instance_orientation' in g
    post = build_panoptic(g['in_semantic_classes_is_thing'], kw, compute_scores=True)
    logits, center, offset = (dev(g['in_semantic_logits']), dev(g['in_instance_center']),
                              dev(g['in_instance_offset']))
    B, _, H, W = logits.shape
    i_out = (center, offset) + ((dev(g['in_instance_orientation']),) if with_ori else ())
    r = post.postprocess(((logits, i_out), (None, None)), make_batch(B, H, W), is_training=False)

    keys = EXPECTED_KEYS_SEMANTIC + EXPECTED_KEYS_INSTANCE + EXPECTED_KEYS_PANOPTIC + \
        EXPECTED_KEYS_SCORES
    if with_ori:
        keys += ['orientations_panoptic_segmentation_deeplab_instance']
    for k in keys:
        assert k in list(r.keys()), k

    assert r['semantic_segmentation_idx'].dtype == torch.int64
    assert (r['semantic_segmentation_idx'].cpu().numpy() == g['semantic_idx']).all()
    assert (r['semantic_segmentation_idx_fullres'].cpu().numpy() == g['semantic_idx']).all()
    np.testing.assert_allclose(r['semantic_segmentation_score'].cpu().numpy(),
                               g['semantic_score'], rtol=1e-5, atol=1e-7)
    probs = r['semantic_softmax_scores']
    ref_probs = torch.softmax(torch.from_numpy(g['in_semantic_logits']), dim=1).numpy()
    np.testing.assert_allclose(probs.cpu().numpy(), ref_probs, rtol=1e-5, atol=1e-7)
    assert r['panoptic_foreground_mask'].dtype == torch.bool
    assert (r['panoptic_foreground_mask'].cpu().numpy() == g['foreground']).all()
    pan = r['panoptic_segmentation_deeplab']
    assert pan.dtype == torch.int64
    assert (pan.cpu().numpy() == g['panoptic']).all()
    assert (r['panoptic_segmentation_deeplab_fullres'].cpu().numpy() == g['panoptic']).all()
    assert (r['panoptic_segmentation_deeplab_semantic_idx'].cpu().numpy()
            == g['panoptic_semantic']).all()
    ins = r['panoptic_segmentation_deeplab_instance_idx']
    assert ins.dtype == torch.uint8
    assert (ins.cpu().numpy() == g['instance']).all()
    want_ids = ids_from_arrays(g['ids_n'], g['ids_pan'], g['ids_ins'])
    for a, b in zip(r['panoptic_segmentation_deeplab_ids'], want_ids):
        assert list(a.items()) == list(b.items())
    want_meta = meta_from_arrays(g['meta_n'], g['meta_center_yx'], g['meta_area'],
                                 g['meta_score'])
    for a, b in zip(r['panoptic_segmentation_deeplab_instance_meta'], want_meta):
        assert a.keys() == b.keys()
        for i in a:
            assert a[i]['center_yx'] == b[i]['center_yx']
            assert a[i]['area'] == b[i]['area']
            assert a[i]['score'] == b[i]['score']
    if with_ori:
        ori = r['orientations_panoptic_segmentation_deeplab_instance']
        for b, d in enumerate(ori):
            want = g['orientation'][b]
            assert sorted(d.keys()) == list(np.where(~np.isnan(want))[0])
            for k, v in d.items():
                assert abs(v - want[k]) < 1e-5
    # score maps: consistent with the definition (panoptic.py:171-239)
    sem_score = r['panoptic_segmentation_deeplab_semantic_score'].cpu().numpy()
    pan_sem = g['panoptic_semantic'].astype(np.int64)
    take = np.take_along_axis(ref_probs, np.clip(pan_sem - 1, 0, None)[:, None], axis=1)[:, 0]
    take[pan_sem == 0] = 0
    np.testing.assert_allclose(sem_score, take, rtol=1e-5, atol=1e-7)
    ins_score = r['panoptic_segmentation_deeplab_instance_score'].cpu().numpy()
    pan_score = r['panoptic_segmentation_deeplab_panoptic_score'].cpu().numpy()
    for b in range(B):
        for pan_id, ins_id in want_ids[b].items():
            m = g['panoptic'][b] == pan_id
            sc = want_meta[b][ins_id]['score']
            assert np.allclose(ins_score[b][m], sc)
            assert np.allclose(pan_score[b][m], sc * take[b][m].mean(), rtol=1e-5)


def _same(a, b):
    """deep equality of result entries (tensors, tuples / lists / dicts of them, scalars)"""
    if torch.is_tensor(a) or torch.is_tensor(b):
        return torch.is_tensor(a) and torch.is_tensor(b) and a.dtype == b.dtype and \
            a.shape == b.shape and torch.equal(a.cpu(), b.cpu())
    if isinstance(a, dict):
        return isinstance(b, dict) and list(a.keys()) == list(b.keys()) and \
            all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return isinstance(b, (list, tuple)) and len(a) == len(b) and \
            all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, float) and isinstance(b, float) and a != a and b != b:
        return True
    return a == b


@pytest.mark.parametrize('defer', [False, True])
def test_postprocess_result_is_a_plain_dict_under_merge_idioms(defer):
    """{**r} / dict(r) / x.update(r) / pickle of the real result: every key of
    reference tests/test_decoders+postprocessing.py:208-258 has its value (never the
    placeholder of a pending entry) and equals what r[k] gives"""
    import copy
    import pickle
    g = load('panoptic_small')
    with_ori = 'in_instance_orientation' in g
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    post = get_postprocessing_class('panoptic')(
        semantic_postprocessing=get_postprocessing_class('semantic')(),
        instance_postprocessing=get_postprocessing_class('instance')(),
        semantic_classes_is_thing=tuple(bool(x) for x in g['in_semantic_classes_is_thing']),
        semantic_class_has_orientation=tuple(bool(x) for x in g['in_semantic_classes_is_thing']),
        compute_scores=True, **({'defer_host_sync': True} if defer else {}))
    logits, center, offset = (dev(g['in_semantic_logits']), dev(g['in_instance_center']),
                              dev(g['in_instance_offset']))
    B, _, H, W = logits.shape
    i_out = (center, offset) + ((dev(g['in_instance_orientation']),) if with_ori else ())

    def run():
        return post.postprocess(((logits, i_out), (None, None)), make_batch(B, H, W),
                                is_training=False)

    keys = EXPECTED_KEYS_SEMANTIC + EXPECTED_KEYS_INSTANCE + EXPECTED_KEYS_PANOPTIC + \
        EXPECTED_KEYS_SCORES
    truth = run()
    x = {}
    x.update(run())
    merged = [{**run()}, dict(run()), x, {**{'other_task': 1}, **run()},
              pickle.loads(pickle.dumps(run())), copy.deepcopy(run()), dict(copy.copy(run()))]
    for m in merged:
        assert type(m) is dict
        for k in keys:
            want, got = truth[k], m[k]
            assert k in m and (got is None) == (want is None), k    # None only where it IS the value
            if want is None:
                continue
            assert _same(got, want), k
    assert (merged[0]['panoptic_segmentation_deeplab'].cpu().numpy() == g['panoptic']).all()
    assert (merged[1]['semantic_segmentation_idx'].cpu().numpy() == g['semantic_idx']).all()


def test_training_mode_passthrough():
    post = build_panoptic([False, True, True])
    s = torch.zeros((1, 3, 8, 8))
    c, o = torch.zeros((1, 1, 8, 8)), torch.zeros((1, 2, 8, 8))
    r = post.postprocess(((s, (c, o)), (None, None)), {}, is_training=True)
    assert set(r.keys()) == {'semantic_output', 'semantic_side_outputs', 'instance_output',
                             'instance_side_outputs'}


def test_factory_errors():
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    with pytest.raises(ValueError):
        get_postprocessing_class('does-not-exist')
    with pytest.raises(AssertionError):
        get_postprocessing_class('instance')(heatmap_nms_kernel_size=4)
    with pytest.raises(AssertionError):
        get_postprocessing_class('instance')(top_k_instances=255)


@pytest.mark.parametrize('on_cpu', [False, True])
def test_deeplab_merge_batch_vs_golden(on_cpu):
    from nicr_mt_scene_analysis_amd.utils.panoptic_merge import deeplab_merge_batch
    g = load('merge_cases')
    for name in jload(g['names']):
        p = jload(g[f'{name}__params'])
        conv = (lambda a: torch.from_numpy(np.ascontiguousarray(a))) if on_cpu else dev
        sem, ins, thing = conv(g[f'{name}__sem']), conv(g[f'{name}__ins']), conv(g[f'{name}__thing'])
        pan, ids = deeplab_merge_batch(sem, ins, thing, p['max_inst'], p['thing_ids'], p['void'])
        assert pan.dtype == torch.int64 and pan.device == sem.device
        assert (pan.cpu().numpy() == g[f'{name}__pan']).all(), name
        want = ids_from_arrays(g[f'{name}__ids_n'], g[f'{name}__ids_pan'], g[f'{name}__ids_ins'])
        for a, b in zip(ids, want):
            assert list(a.items()) == list(b.items()), name


def test_many_centers_regrow_table():
    """> 256 kept centers: the device table is re-grown and ids wrap like uint8."""
    from nicr_mt_scene_analysis_amd.model.postprocessing import InstancePostprocessing
    g = load('grouping_adversarial')
    name = 'wrap_300_centers'
    post = InstancePostprocessing(normalized_offset=False, top_k_instances=254)
    seg, meta = post._get_instance_segmentation(dev(g[f'{name}__heat']), dev(g[f'{name}__offset']),
                                                dev(g[f'{name}__fg']))
    assert (seg.cpu().numpy() == g[f'{name}__inst']).all()
    want = meta_from_arrays(g[f'{name}__meta_n'], g[f'{name}__meta_center_yx'],
                            g[f'{name}__meta_area'], g[f'{name}__meta_score'])
    assert len(meta[0]) == 300
    for i in want[0]:
        assert meta[0][i]['center_yx'] == want[0][i]['center_yx']
        assert meta[0][i]['area'] == want[0][i]['area']



def test_panoptic_postprocess_more_instances_than_fetched_columns(oracle):
    """the host tables travel cut to `_host_columns` columns; an image with more instances
    makes the postprocessing fetch again with wider tables (dicts complete, in order)."""
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn
    inp = syn.make_panoptic_inputs(2, 12, 96, 128, n_centers=48, seed=5, sigma=3.0)
    is_thing = inp['semantic_classes_is_thing']
    post = build_panoptic(is_thing)
    assert post._host_columns == 32
    i_out = (dev(inp['instance_center']), dev(inp['instance_offset']))
    r = post.postprocess(((dev(inp['semantic_logits']), i_out), (None, None)),
                         make_batch(2, 96, 128), is_training=False)
    H, W = 96, 128
    idx, _ = oracle.semantic_argmax(inp['semantic_logits'])
    fg = np.asarray(is_thing, dtype=bool)[idx]
    cyx, n, _, _ = oracle.center_nms_topk(inp['instance_center'], max_centers=256)
    inst, _ = oracle.group_offsets(inp['instance_offset'], fg, cyx, n, scale_y=H, scale_x=W)
    pan, ids = oracle.deeplab_merge(idx + 1, inst, fg, 1 << 16, np.where(is_thing)[0] + 1, 0)
    assert max(n) > 32
    assert post._host_columns >= max(n)
    assert (r['panoptic_segmentation_deeplab'].cpu().numpy() == pan).all()
    for b, d in enumerate(r['panoptic_segmentation_deeplab_ids']):
        assert list(d.items()) == list(ids[b].items())
    for b, m in enumerate(r['panoptic_segmentation_deeplab_instance_meta']):
        assert len(m) == n[b]
        for i, e in m.items():
            assert e['center_yx'] == tuple(int(v) for v in cyx[b][i - 1])
            assert e['area'] == int((inst[b] == i).sum())


@pytest.mark.parametrize('name', ['panoptic_small', 'panoptic_edges_thr'])
def test_deferred_host_sync_equals_eager(name):
    """`defer_host_sync=True`: postprocess never waits for the GPU (asynchronous table copy, host
    objects built on first read) — same entries as the eager mode"""
    g = load(name)
    kw = jload(g['kwargs']) if 'kwargs' in g else None
    with_ori = 'in_instance_orientation' in g
    logits, center, offset = (dev(g['in_semantic_logits']), dev(g['in_instance_center']),
                              dev(g['in_instance_offset']))
    B, _, H, W = logits.shape
    i_out = (center, offset) + ((dev(g['in_instance_orientation']),) if with_ori else ())
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    results = []
    for defer in (False, True):
        post = get_postprocessing_class('panoptic')(
            semantic_postprocessing=get_postprocessing_class('semantic')(),
            instance_postprocessing=get_postprocessing_class('instance')(**(kw or {})),
            semantic_classes_is_thing=tuple(bool(x) for x in g['in_semantic_classes_is_thing']),
            semantic_class_has_orientation=tuple(bool(x) for x in g['in_semantic_classes_is_thing']),
            defer_host_sync=defer)
        results.append(post.postprocess(((logits, i_out), (None, None)), make_batch(B, H, W),
                                        is_training=False))
    eager, deferred = results
    assert list(eager.keys()) == list(deferred.keys())
    assert torch.equal(eager['panoptic_segmentation_deeplab'], deferred['panoptic_segmentation_deeplab'])
    assert [list(d.items()) for d in eager['panoptic_segmentation_deeplab_ids']] == \
        [list(d.items()) for d in deferred['panoptic_segmentation_deeplab_ids']]
    assert eager['panoptic_segmentation_deeplab_instance_meta'] == \
        deferred['panoptic_segmentation_deeplab_instance_meta'] or with_ori     # NaN != NaN below
    if with_ori:
        k = 'orientations_panoptic_segmentation_deeplab_instance'
        assert eager[k] == deferred[k]
        for a, b in zip(eager['panoptic_segmentation_deeplab_instance_meta'],
                        deferred['panoptic_segmentation_deeplab_instance_meta']):
            assert a.keys() == b.keys()
            for i in a:
                assert {kk: v for kk, v in a[i].items() if v == v} == \
                    {kk: v for kk, v in b[i].items() if v == v}


def test_deferred_host_sync_reports_center_table_overflow():
    """more tied centers than the table holds: the eager mode re-runs with a larger table, the
    deferred mode cannot — it raises when the host tables are first read and enlarges the table
    for the following calls"""
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    g = load('grouping_adversarial')
    heat = dev(g['wrap_300_centers__heat'])
    B, _, H, W = heat.shape
    logits = torch.zeros((B, 3, H, W), device='cuda')
    logits[:, 1] = 1.0
    offset = torch.zeros((B, 2, H, W), device='cuda')
    post = get_postprocessing_class('panoptic')(
        semantic_postprocessing=get_postprocessing_class('semantic')(),
        instance_postprocessing=get_postprocessing_class('instance')(top_k_instances=254),
        semantic_classes_is_thing=(False, True, True),
        semantic_class_has_orientation=(False, True, True), defer_host_sync=True)
    r = post.postprocess(((logits, (heat, offset)), (None, None)), make_batch(B, H, W), is_training=False)
    with pytest.raises(RuntimeError):
        r['panoptic_segmentation_deeplab_ids']
    assert post._instance_postprocessing._max_centers >= 300
    r = post.postprocess(((logits, (heat, offset)), (None, None)), make_batch(B, H, W), is_training=False)
    assert len(r['panoptic_segmentation_deeplab_instance_meta'][0]) == 300


def test_compute_scores_vs_golden():
    """f3: score maps + meta of the reference's compute_scores branch (panoptic.py:171-239),
    produced by `nmsa_panoptic_scores`, against the reference's own output."""
    g = load('scores_cases')
    post = build_panoptic(g['in_semantic_classes_is_thing'], None, compute_scores=True)
    logits, center, offset = (dev(g['in_semantic_logits']), dev(g['in_instance_center']),
                              dev(g['in_instance_offset']))
    B, _, H, W = logits.shape
    r = post.postprocess(((logits, (center, offset)), (None, None)), make_batch(B, H, W),
                         is_training=False)
    assert (r['panoptic_segmentation_deeplab'].cpu().numpy() == g['panoptic']).all()
    for key, name in (('semantic_score', 'panoptic_segmentation_deeplab_semantic_score'),
                      ('instance_score', 'panoptic_segmentation_deeplab_instance_score'),
                      ('panoptic_score', 'panoptic_segmentation_deeplab_panoptic_score')):
        got = r[name]
        assert got.dtype == torch.float32 and got.is_cuda
        np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=1e-5, atol=1e-7, err_msg=key)
        np.testing.assert_allclose(r[name + '_fullres'].cpu().numpy(), g[key], rtol=1e-5, atol=1e-7)
    assert (r['panoptic_segmentation_deeplab_instance_score'].cpu().numpy() == g['instance_score']).all()
    meta = r['panoptic_segmentation_deeplab_instance_meta']
    for b in range(B):
        used = np.where(g['meta_panoptic_id'][b] >= 0)[0]
        assert sorted(i for i, m in meta[b].items() if 'panoptic_id' in m) == list(used)
        for i in used:
            m = meta[b][int(i)]
            assert m['panoptic_id'] == g['meta_panoptic_id'][b, i]
            assert m['semantic_idx'] == g['meta_semantic_idx'][b, i]
            assert abs(m['semantic_score'] - g['meta_semantic_score'][b, i]) <= 1e-5 * abs(g['meta_semantic_score'][b, i])
            assert abs(m['panoptic_score'] - g['meta_panoptic_score'][b, i]) <= 1e-5 * abs(g['meta_panoptic_score'][b, i])


def test_panoptic_scores_vs_oracle(oracle):
    """nmsa_panoptic_scores through ops on a larger random case incl. bf16 logits"""
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn
    inp = syn.make_panoptic_inputs(3, n_classes=19, height=120, width=160, n_centers=12, seed=11)
    for dt in (torch.float32, torch.bfloat16):
        x = torch.from_numpy(inp['semantic_logits']).to(dt)
        p = ops.panoptic_pipeline(x.cuda(), dev(inp['instance_center']), dev(inp['instance_offset']),
                                  dev(inp['semantic_classes_is_thing']), want_score=True,
                                  want_panoptic_semantic=True)
        tab = torch.zeros((3, 256), dtype=torch.float32, device='cuda')
        tab[:, 1:] = p['center_scores'][:, :255]
        sc = ops.panoptic_scores(x.cuda(), p['semantic_idx_u8'], p['semantic_score'], p['instance'],
                                 p['panoptic'], p['pan_of_inst'], tab, 1 << 16)
        torch.cuda.synchronize()
        ids = ids_from_arrays(p['n_ids'].cpu().numpy(), p['ids_pan'].cpu().numpy(),
                              p['ids_ins'].cpu().numpy())
        sem, ins, pns, mean = oracle.panoptic_scores(
            x.float().numpy(), p['panoptic_semantic'].cpu().numpy(), p['panoptic'].cpu().numpy(),
            ids, tab.cpu().numpy())
        np.testing.assert_allclose(sc['semantic_score'].cpu().numpy(), sem, rtol=1e-5, atol=1e-7)
        assert np.array_equal(sc['instance_score'].cpu().numpy(), ins)
        np.testing.assert_allclose(sc['panoptic_score'].cpu().numpy(), pns, rtol=1e-5, atol=1e-7)
        got_mean = sc['mean_semantic_score'].cpu().numpy()
        for b, d in enumerate(ids):
            for ins_id in d.values():
                assert abs(got_mean[b, ins_id] - mean[b, ins_id]) <= 1e-5 * abs(mean[b, ins_id])


def test_instance_postprocess_gt_keys_vs_golden():
    """InstancePostprocessing.postprocess with every GT key of the reference (GT foreground, debug
    all-foreground, real crop + upscale, four orientation dicts incl. uint16-range GT instance ids)
    against the reference's own output (oracle/gen_golden.py::gen_instance_post)."""
    from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    g = load('instance_post_cases')
    c = [int(v) for v in g['crop']]
    size = tuple(int(v) for v in g['size'])
    post = get_postprocessing_class('instance')(debug=True)
    batch = {
        'rgb_fullres': torch.zeros((2, 3) + size),
        APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize', 'valid_region_slice_y': slice(c[0], c[1]),
                                      'valid_region_slice_x': slice(c[2], c[3])}]] * 2,
        'instance_foreground': dev(g['in_fg']),
        'instance': dev(g['in_gt_instance']),
        'orientation_foreground': dev(g['in_orientation_fg']),
    }
    data = ((dev(g['in_center']), dev(g['in_offset']), dev(g['in_orientation'])), None)
    r = post.postprocess(data, batch, is_training=False)
    for k in ('instance_segmentation_gt_foreground', 'instance_segmentation_all_foreground'):
        assert r[k].dtype == torch.uint8
        assert np.array_equal(r[k].cpu().numpy(), g[k]), k
        assert np.array_equal(r[k + '_fullres'].cpu().numpy(), g[k + '_fullres']), k
    want_meta = meta_from_arrays(g['meta_n'], g['meta_center_yx'], g['meta_area'], g['meta_score'])
    for a, b in zip(r['instance_segmentation_gt_meta'], want_meta):
        assert a.keys() == b.keys()
        for i in a:
            assert a[i]['center_yx'] == b[i]['center_yx'] and a[i]['area'] == b[i]['area']
            assert a[i]['score'] == b[i]['score']
    for k in ('orientations_gt_instance_gt_orientation_foreground',
              'orientations_instance_segmentation_gt_orientation_foreground',
              'orientations_gt_instance', 'orientations_instance_segmentation'):
        for b in range(2):
            ids = [int(v) for v in g[k + '__ids'][b] if v >= 0]
            assert sorted(r[k][b].keys()) == ids, k
            for i, kk in enumerate(ids):
                want = float(g[k + '__angles'][b, i])
                got = r[k][b][kk]
                assert abs(got - want) < 1e-4 or abs(abs(got - want) - 2 * np.pi) < 1e-4, (k, kk)
