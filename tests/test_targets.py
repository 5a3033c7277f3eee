"""
GPU tier (pytest -m gpu): ground-truth target generation on the device
through the C ABI (nmsa_instance_clear_stuff / nmsa_instance_targets / nmsa_panoptic_targets /
nmsa_dve_targets) and the batch-level mirrors of the reference classes, against the reference's
own numpy generators (tests/golden/target_cases.npz) and the C oracle.  Heat-maps, offsets,
masks, panoptic ids and id dicts are bit-exact; the normalised embedding LUT is fp32 within
rtol 1e-5 (numpy's pairwise float32 norm vs a wave reduction).
"""
import numpy as np
import pytest
import torch

from _golden import load, ids_from_arrays
from nicr_mt_scene_analysis_amd.testing import synthetic as syn

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _stuff_lut(is_thing):
    st = np.zeros((len(is_thing),), np.uint8)
    st[np.where(~is_thing)[0][1:]] = 1
    return st


def test_targets_vs_golden():
    from nicr_mt_scene_analysis_amd.data.preprocessing import (
        DenseVisualEmbeddingTargetGenerator, InstanceClearStuffIDs, InstanceTargetGenerator,
        PanopticTargetGenerator)
    g = load('target_cases')
    is_thing = tuple(bool(x) for x in g['in_is_thing'])
    batch = {'semantic': dev(g['in_semantic']), 'instance': dev(g['in_instance'])}
    with pytest.raises(AssertionError):            # like the reference on an uncleared map
        InstanceTargetGenerator(sigma=8, semantic_classes_is_thing=is_thing)(dict(batch))
    batch = InstanceClearStuffIDs(semantic_classes_is_thing=is_thing)(batch)
    assert np.array_equal(batch['instance'].cpu().numpy(), g['cleared_instance'])
    for name, gen in (
            ('s8n', InstanceTargetGenerator(sigma=8, semantic_classes_is_thing=is_thing)),
            ('s3u', InstanceTargetGenerator(sigma=3, semantic_classes_is_thing=is_thing,
                                            normalized_offset=False)),
            ('s5nothing', InstanceTargetGenerator(sigma=5))):
        r = gen(dict(batch), n_classes=len(is_thing))
        assert np.array_equal(r['instance_center'].cpu().numpy(), g[f'{name}__center']), name
        off = r['instance_offset'].cpu().numpy()
        assert off.dtype == g[f'{name}__offset'].dtype and np.array_equal(off, g[f'{name}__offset'])
        assert r['instance_foreground'].dtype == torch.bool
        assert np.array_equal(r['instance_foreground'].cpu().numpy(), g[f'{name}__foreground'])
        assert np.array_equal(r['instance_center_mask'].cpu().numpy(), g[f'{name}__center_mask'])
        enc = gen.last_dynamic_parameters['encoded_instances']
        for b in range(len(enc)):
            assert enc[b] == sorted(int(v) for v in np.unique(g['cleared_instance'][b]) if v != 0)
    r = PanopticTargetGenerator(semantic_classes_is_thing=is_thing)(dict(batch))
    assert r['panoptic'].dtype == torch.int64
    assert np.array_equal(r['panoptic'].cpu().numpy(), g['panoptic'])
    want = ids_from_arrays(g['pan_ids_n'], g['pan_ids_pan'], g['pan_ids_ins'])
    assert [list(d.items()) for d in r['panoptic_ids_to_instance_dict']] == \
        [list(d.items()) for d in want]
    db = {'panoptic': r['panoptic'], 'panoptic_embedding_keys': dev(g['dve_keys']),
          'panoptic_embedding_n': dev(g['dve_n']), 'panoptic_embedding': dev(g['dve_emb']),
          'image_embedding': dev(g['dve_img'])}
    rd = DenseVisualEmbeddingTargetGenerator(diff_factor=0.65)(db)
    assert np.array_equal(rd['dense_visual_embedding_indices'].cpu().numpy(), g['dve_indices'])
    lut = rd['dense_visual_embedding_lut'].cpu().numpy()
    for b in range(lut.shape[0]):
        n = int(g['dve_n'][b])
        np.testing.assert_allclose(lut[b, :n], g['dve_lut'][b, :n], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize('cfg', [
    dict(B=2, NC=41, H=480, W=640, n=60, seed=4, sigma=8),       # dataset-sized
    dict(B=3, NC=5, H=37, W=53, n=9, seed=5, sigma=2),           # ragged, patches clipped everywhere
    dict(B=1, NC=151, H=192, W=256, n=3000, seed=6, sigma=4, max_radius=5),   # > 1024 ids
])
def test_targets_vs_oracle(oracle, cfg):
    from nicr_mt_scene_analysis_amd import ops
    m = syn.make_label_maps(cfg['B'], cfg['NC'], cfg['H'], cfg['W'], cfg['n'], seed=cfg['seed'],
                            max_radius=cfg.get('max_radius'))
    is_thing = m['semantic_classes_is_thing']
    sem = m['semantic']
    ins = m['instance'].copy()
    ins[~is_thing[sem]] = 0                                       # InstanceClearStuffIDs
    d_ins = dev(m['instance'])
    stuff_incl_void = dev((~is_thing).astype(np.uint8))
    ops.instance_clear_stuff(dev(sem), d_ins, stuff_incl_void)
    assert np.array_equal(d_ins.cpu().numpy(), ins)
    max_inst = 4096 if cfg['n'] > 1000 else 1024
    r = ops.instance_targets(dev(sem), d_ins, cfg['NC'], dev(is_thing.astype(np.uint8)),
                             dev(_stuff_lut(is_thing)), cfg['sigma'], True, max_instances=max_inst)
    torch.cuda.synchronize()
    assert int(r['status'].item()) == 0
    o = oracle.instance_targets(sem, ins, cfg['NC'], is_thing, _stuff_lut(is_thing), cfg['sigma'], True)
    assert np.array_equal(r['center'].cpu().numpy(), o['center'])
    assert np.array_equal(r['offset'].cpu().numpy(), o['offset'])
    assert np.array_equal(r['foreground'].cpu().numpy(), o['foreground'])
    assert np.array_equal(r['center_mask'].cpu().numpy(), o['center_mask'])
    n_enc = r['n_encoded'].cpu().numpy()
    for b in range(cfg['B']):
        assert r['encoded_ids'][b, :n_enc[b]].cpu().tolist() == o['encoded'][b]
    if cfg['n'] > 1000:
        small = ops.instance_targets(dev(sem), d_ins, cfg['NC'], None, None, cfg['sigma'], True,
                                     max_instances=1024)
        n_distinct = max(len(np.unique(ins[b])) - 1 for b in range(cfg['B']))
        assert bool(int(small['status'].item()) & 1) == (n_distinct > 1024)   # table too small
    p = ops.panoptic_targets(dev(sem), d_ins, cfg['NC'], dev(is_thing.astype(np.uint8)), 1 << 16, 0,
                             max_instances=max_inst, max_segments=8192)
    assert int(p['status'].item()) == 0
    pan, dicts = oracle.naive_merge(sem, ins, 1 << 16, np.where(is_thing)[0], 0, cap=8192)
    assert np.array_equal(p['panoptic'].cpu().numpy(), pan)
    got = ids_from_arrays(p['n_ids'].cpu().numpy(), p['ids_pan'].cpu().numpy(), p['ids_ins'].cpu().numpy())
    assert [list(d.items()) for d in got] == [list(d.items()) for d in dicts]
    # naive merge on the UNcleared map (instances over stuff / void, mixed labels)
    p = ops.panoptic_targets(dev(sem), dev(m['instance']), cfg['NC'], dev(is_thing.astype(np.uint8)),
                             1 << 16, 0, max_instances=max_inst, max_segments=8192)
    pan, dicts = oracle.naive_merge(sem, m['instance'], 1 << 16, np.where(is_thing)[0], 0, cap=8192)
    assert np.array_equal(p['panoptic'].cpu().numpy(), pan)
    got = ids_from_arrays(p['n_ids'].cpu().numpy(), p['ids_pan'].cpu().numpy(), p['ids_ins'].cpu().numpy())
    assert [list(d.items()) for d in got] == [list(d.items()) for d in dicts]


def test_targets_status_bits():
    from nicr_mt_scene_analysis_amd import ops
    sem = torch.zeros((1, 8, 8), dtype=torch.uint8, device='cuda')
    ins = torch.zeros((1, 8, 8), dtype=torch.int32, device='cuda')
    ins[0, 0, 0] = 70000
    r = ops.instance_targets(sem, ins, 4, None, None, 2)
    assert int(r['status'].item()) & 32
    ins[0, 0, 0] = 5
    sem[0, 0, 0] = 9
    r = ops.instance_targets(sem, ins, 4, None, None, 2)
    assert int(r['status'].item()) == 64            # SET by the call: the 32 of the call before is gone
    # the persistent workspace is clean again after calls that raised status bits: the next call
    # (which skips its memset) sees none of their ids
    sem[0, 0, 0] = 1
    r = ops.instance_targets(sem, ins, 4, None, None, 2)
    assert int(r['status'].item()) == 0
    assert r['n_encoded'].tolist() == [1] and r['encoded_ids'][0, 0].item() == 5
    p = ops.panoptic_targets(sem, ins, 4, None, 1 << 16)
    assert int(p['status'].item()) == 0 and p['n_ids'].tolist() == [1]
    assert int(p['panoptic'][0, 0, 0]) == (1 << 16) + 1


def test_reference_property_checks():
    """the properties the reference's own tests assert (tests/test_preprocessing.py:145-216):
    after InstanceClearStuffIDs every stuff pixel has id 0; the target generator's foreground
    is exactly the set of thing-class pixels that carry an instance."""
    from nicr_mt_scene_analysis_amd.data.preprocessing import InstanceClearStuffIDs, InstanceTargetGenerator
    m = syn.make_label_maps(2, 41, 120, 160, 25, seed=8)
    thing_classes = [1, 2, 3]
    is_thing = tuple(i in thing_classes for i in range(41))
    m['semantic'][m['semantic'] > 6] = 0          # keep a few classes, lots of void
    batch = {'semantic': dev(m['semantic']), 'instance': dev(m['instance'])}
    batch = InstanceClearStuffIDs(semantic_classes_is_thing=is_thing)(batch)
    sem, ins = batch['semantic'].cpu().numpy(), batch['instance'].cpu().numpy()
    for c in range(41):
        if not is_thing[c]:
            assert (ins[sem == c] == 0).all()
    assert (ins[np.isin(sem, thing_classes)] == m['instance'][np.isin(sem, thing_classes)]).all()
    r = InstanceTargetGenerator(sigma=8, semantic_classes_is_thing=is_thing)(batch)
    for k in ('instance_center', 'instance_offset', 'instance_foreground', 'instance_center_mask'):
        assert k in r
    fg = r['instance_foreground'].cpu().numpy()
    assert np.array_equal(fg, np.isin(sem, thing_classes) & (ins > 0))
    assert float(r['instance_center'].max()) == 1.0 and float(r['instance_center'].min()) >= 0.0


def test_naive_and_deeplab_merge_agree_on_gt_style_maps():
    """the invariant of the reference's tests/test_merge.py:27-102 on the HIP path: on ground-truth
    style maps (every instance lies inside ONE thing class, stuff pixels carry id 0) the naive merge
    (target generation) and the deeplab merge (evaluation side) paint the same panoptic map."""
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.utils.panoptic_merge import deeplab_merge_batch
    m = syn.make_label_maps(2, 21, 120, 160, 40, seed=12, mixed_fraction=0.0, max_radius=14)
    is_thing = m['semantic_classes_is_thing']
    sem = m['semantic'].copy()
    ins = m['instance'].copy()
    ins[~is_thing[sem]] = 0                         # stuff: no instance
    # pixels of a thing class without an instance would stay void in both merges; make every
    # instance single-class (mixed_fraction = 0 paints the class in) and drop overpainted rims
    for b in range(sem.shape[0]):
        for iid in np.unique(ins[b]):
            if iid == 0:
                continue
            mask = ins[b] == iid
            cls = np.bincount(sem[b][mask]).argmax()
            sem[b][mask] = cls
    thing_ids = np.where(is_thing)[0]
    naive = ops.panoptic_targets(dev(sem), dev(ins), len(is_thing), dev(is_thing.astype(np.uint8)),
                                 1 << 16, 0)
    pan_d, ids_d = deeplab_merge_batch(dev(sem.astype(np.int64)), dev(ins), dev(ins > 0), 1 << 16,
                                       thing_ids, 0, n_classes=len(is_thing))
    torch.cuda.synchronize()
    got_n, got_d = naive['panoptic'].cpu().numpy(), pan_d.cpu().numpy()
    has_inst_or_stuff = (ins > 0) | ~is_thing[sem]
    assert np.array_equal(got_n[has_inst_or_stuff], got_d[has_inst_or_stuff])
    ids_n = ids_from_arrays(naive['n_ids'].cpu().numpy(), naive['ids_pan'].cpu().numpy(),
                            naive['ids_ins'].cpu().numpy())
    assert [sorted(d.items()) for d in ids_n] == [sorted(d.items()) for d in ids_d]


def test_numpy_merge_entry_points(oracle):
    """the reference's numpy twins (utils/panoptic_merge.py:43-169): same signatures, dtypes
    and dict order, computed by the HIP kernels"""
    from nicr_mt_scene_analysis_amd.utils.panoptic_merge import (
        deeplab_merge_semantic_and_instance_np, naive_merge_semantic_and_instance_np)
    m = syn.make_label_maps(1, 9, 60, 80, 11, seed=21)
    sem = m['semantic'][0].astype(np.uint8)
    ins = m['instance'][0].astype(np.uint16)
    thing_ids = np.where(m['semantic_classes_is_thing'])[0]
    pan, ids = naive_merge_semantic_and_instance_np(sem, ins, 1 << 16, thing_ids, 0)
    w_pan, w_ids = oracle.naive_merge(sem[None], ins[None], 1 << 16, thing_ids, 0)
    assert pan.dtype == np.uint32 and np.array_equal(pan, w_pan[0])
    assert list(ids.items()) == list(w_ids[0].items())
    thing_seg = m['semantic_classes_is_thing'][sem]
    pan, ids = deeplab_merge_semantic_and_instance_np(sem, ins, thing_seg, 1 << 16, thing_ids, 0)
    w_pan, w_ids = oracle.deeplab_merge(sem[None], ins[None], thing_seg[None], 1 << 16, thing_ids, 0)
    assert pan.dtype == np.uint32 and np.array_equal(pan, w_pan[0])
    assert list(ids.items()) == list(w_ids[0].items())
