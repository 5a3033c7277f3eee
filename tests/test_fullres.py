"""
GPU tier (pytest -m gpu): the full-resolution step — crop to the
valid region + resize to the dataset resolution, reference dense_base.py:15-58 and
semantic.py:61-80 — through the C ABI (nmsa_resize_nearest / nmsa_resize_bilinear /
nmsa_semantic_argmax_resized) against the reference's golden vectors and the C oracle.
Resized maps and logits are bit-exact; the softmax score is fp32 within rtol 1e-5.
"""
import numpy as np
import pytest
import torch

from _golden import load, jload
from nicr_mt_scene_analysis_amd.testing import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'needs the MI355X'
    from nicr_mt_scene_analysis_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _geoms(g):
    return {name: ((slice(c[0], c[1]), slice(c[2], c[3])), tuple(size))
            for name, (c, size) in jload(g['geoms']).items()}


def test_resize_vs_golden(ops):
    g = load('fullres_cases')
    n = 0
    for name, (crop, size) in _geoms(g).items():
        for k in ('u8', 'bool', 'i32', 'i64', 'f32'):
            key = f'{name}__nearest_{k}'
            if key not in g:
                continue
            got = ops.resize_nearest(dev(g[f'in__{k}']), size, crop).cpu().numpy()
            assert got.dtype == g[key].dtype and np.array_equal(got, g[key]), key
            n += 1
        got = ops.resize_bilinear(dev(g['in__logits']), size, crop).cpu().numpy()
        assert np.array_equal(got, g[f'{name}__bilinear']), name          # bit-exact
        n += 1
    assert n >= 20


@pytest.mark.parametrize('geom', [
    # (Hs, Ws, crop, size): ragged widths (scalar-store path), down / up, 1-px planes
    (33, 41, (slice(0, 33), slice(0, 41)), (67, 83)),
    (33, 41, (slice(2, 30), slice(3, 40)), (64, 128)),
    (120, 160, (slice(0, 120), slice(0, 160)), (75, 101)),
    (8, 8, (slice(3, 4), slice(5, 6)), (70, 90)),
    (480, 640, (slice(0, 480), slice(0, 640)), (530, 730)),
])
def test_resize_vs_oracle(ops, oracle, geom):
    Hs, Ws, crop, size = geom
    rng = np.random.default_rng(Hs * 1000 + size[1])
    for dt in (np.uint8, np.int16, np.int32, np.int64, np.float32, np.bool_):
        if dt == np.bool_:
            a = rng.random((2, 3, Hs, Ws)) < 0.5
        elif dt == np.float32:
            a = rng.standard_normal((2, 3, Hs, Ws)).astype(np.float32)
        else:
            hi = min(np.iinfo(dt).max, 1 << 27)
            a = rng.integers(0, hi, (2, 3, Hs, Ws)).astype(dt)
        got = ops.resize_nearest(dev(a), size, crop).cpu().numpy()
        assert np.array_equal(got, oracle.resize_nearest(a, size, crop)), dt
    x = (rng.standard_normal((2, 5, Hs, Ws)) * 4).astype(np.float32)
    want = oracle.resize_bilinear(x, size, crop)
    got = ops.resize_bilinear(dev(x), size, crop).cpu().numpy()
    assert np.array_equal(got, want)
    # bf16 storage: fp32 arithmetic on the bf16 values, rounded once to bf16
    xb = torch.from_numpy(x).to(torch.bfloat16)
    want_b = torch.from_numpy(oracle.resize_bilinear(xb.float().numpy(), size, crop)).to(torch.bfloat16)
    got_b = ops.resize_bilinear(xb.cuda(), size, crop).cpu()
    assert got_b.dtype == torch.bfloat16 and torch.equal(got_b.view(torch.int16), want_b.view(torch.int16))


@pytest.mark.parametrize('geom', [
    (96, 128, (slice(4, 92), slice(0, 128)), (150, 200), 40),
    (33, 41, (slice(2, 30), slice(3, 40)), (67, 83), 7),
    (120, 160, (slice(0, 120), slice(0, 160)), (75, 101), 19),
])
def test_argmax_resized_vs_oracle(ops, oracle, geom):
    """fused crop + bilinear + argmax + score == argmax over the materialised fullres logits"""
    Hs, Ws, crop, size, C = geom
    rng = np.random.default_rng(C)
    x = (rng.standard_normal((2, C, Hs, Ws)) * 3).astype(np.float32)
    x[0, 1, 5:9, 4:12] = -np.inf                         # ordinary columns with -inf entries
    x[1, :, 10:12, 10:12] = -np.inf                      # all -inf -> index 0, NaN score
    x[1, 2, 20, 20] = np.nan
    x[0, 3, 25, 30] = np.inf
    lf = oracle.resize_bilinear(x, size, crop)
    with np.errstate(all='ignore'):
        idx, score = oracle.semantic_argmax(lf)
    r = ops.semantic_argmax_resized(dev(x), size, crop, want_u8=True, want_i64=True, want_score=True)
    torch.cuda.synchronize()
    assert np.array_equal(r['idx'].cpu().numpy(), idx)
    assert np.array_equal(r['idx_u8'].cpu().numpy(), idx.astype(np.uint8))
    np.testing.assert_allclose(r['score'].cpu().numpy(), score, rtol=1e-5, atol=1e-7, equal_nan=True)
    # and the materialised HIP path agrees with itself
    lf_hip = ops.resize_bilinear(dev(x), size, crop)
    r2 = ops.semantic_argmax(lf_hip, want_u8=False, want_i64=True, want_score=True)
    assert torch.equal(r2['idx'], r['idx'])
    # bf16 logits: fused == materialise-then-argmax on the HIP path
    xb = torch.from_numpy(np.nan_to_num(x, nan=0.0, posinf=50.0, neginf=-50.0)).to(torch.bfloat16).cuda()
    rb = ops.semantic_argmax_resized(xb, size, crop)
    rb2 = ops.semantic_argmax(ops.resize_bilinear(xb, size, crop))
    assert torch.equal(rb['idx'], rb2['idx'])
    np.testing.assert_allclose(rb['score'].cpu().numpy(), rb2['score'].cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
@pytest.mark.parametrize('geom', [
    # (Hs, Ws, crop, size): rows of whole 16-byte pieces -> the row-aligned window shape
    (60, 128, (slice(3, 57), slice(5, 123)), (97, 211)),      # crop starts off a 16-byte boundary
    (48, 96, (slice(0, 48), slice(0, 96)), (55, 110)),        # last tile's window pulled back to the row end
    (40, 256, (slice(1, 40), slice(130, 256)), (64, 300)),    # window start = the crop's first column, 2.4x
])
def test_upscaling_window_shapes_agree(ops, oracle, monkeypatch, geom, dtype):
    """k_resized_tile stages a tile's source window either row-aligned (16-byte aligned rows of
    8 / 16 pieces; plan_tiles picks it when the source rows allow it) or packed
    (`NMSA_RESIZE_PACKED_STAGING=1`, read per call): same bits from both, equal to the oracle"""
    Hs, Ws, crop, size = geom
    rng = np.random.default_rng(Hs * Ws)
    x = (rng.integers(-8, 9, (2, 9, Hs, Ws)) * 0.375).astype(np.float32)      # exact in bf16; ties
    xd = dev(x).to(getattr(torch, dtype))
    want = oracle.resize_bilinear(x, size, crop)
    if dtype != 'float32':
        want = torch.from_numpy(want).to(torch.bfloat16).float().numpy()
    idx, score = oracle.semantic_argmax(want)
    got = {}
    for packed in ('', '1'):
        if packed:
            monkeypatch.setenv('NMSA_RESIZE_PACKED_STAGING', packed)
        else:
            monkeypatch.delenv('NMSA_RESIZE_PACKED_STAGING', raising=False)
        lf = ops.resize_bilinear(xd, size, crop)
        r = ops.semantic_argmax_resized(xd, size, crop, want_u8=True, want_i64=True, want_score=True)
        torch.cuda.synchronize()
        assert np.array_equal(lf.float().cpu().numpy(), want)
        assert np.array_equal(r['idx'].cpu().numpy(), idx)
        np.testing.assert_allclose(r['score'].cpu().numpy(), score, rtol=1e-5, atol=1e-7)
        got[packed] = (lf, r['idx'], r['idx_u8'], r['score'])
    for a, b in zip(got[''], got['1']):
        assert torch.equal(a, b)


def test_panoptic_postprocess_fullres_vs_golden():
    """PanopticPostprocessing.postprocess with a real crop + upscale, every *_fullres entry
    against the reference's own output (oracle/gen_golden.py::gen_fullres)."""
    from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    g = load('fullres_panoptic')
    inp = syn.make_panoptic_inputs(2, n_classes=40, height=96, width=128, n_centers=9, seed=77)
    if syn.input_digest(inp['semantic_logits'], inp['instance_center'],
                        inp['instance_offset']) != jload(g['input_digest']):
        pytest.skip('synthetic inputs differ bit-wise on this host (numpy/libm)')
    c = [int(v) for v in g['crop']]
    size = tuple(int(v) for v in g['size'])
    is_thing = tuple(bool(x) for x in inp['semantic_classes_is_thing'])
    post = get_postprocessing_class('panoptic')(
        semantic_postprocessing=get_postprocessing_class('semantic')(),
        instance_postprocessing=get_postprocessing_class('instance')(),
        semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing)
    batch = {
        'rgb_fullres': torch.zeros((2, 3) + size),
        APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize',
                                      'valid_region_slice_y': slice(c[0], c[1]),
                                      'valid_region_slice_x': slice(c[2], c[3])}]] * 2,
    }
    data = ((dev(inp['semantic_logits']),
             (dev(inp['instance_center']), dev(inp['instance_offset']))), (None, None))
    r = post.postprocess(data, batch, is_training=False)
    for k in ('semantic_segmentation_idx_fullres', 'panoptic_segmentation_deeplab_fullres',
              'panoptic_segmentation_deeplab_instance_idx_fullres',
              'panoptic_segmentation_deeplab_semantic_idx_fullres'):
        got = r[k].cpu().numpy()
        assert got.shape == g[k].shape and np.array_equal(got, g[k]), k
    assert r['semantic_segmentation_idx_fullres'].dtype == torch.int64
    np.testing.assert_allclose(r['semantic_segmentation_score_fullres'].cpu().numpy(),
                               g['semantic_segmentation_score_fullres'], rtol=1e-5, atol=1e-7)
    out_fullres = r['semantic_output_fullres']                  # lazy: materialised on read
    assert np.array_equal(out_fullres[:, ::13].cpu().numpy(), g['semantic_output_fullres'])
    probs = r['semantic_softmax_scores_fullres'].cpu().numpy()
    want = torch.softmax(out_fullres.cpu(), dim=1).numpy()
    np.testing.assert_allclose(probs, want, rtol=1e-5, atol=1e-7)


def test_resize_errors(ops):
    x = torch.zeros((1, 4, 4), dtype=torch.uint8)
    with pytest.raises(Exception):
        ops.resize_nearest(x, (8, 8))                            # CPU tensor: no fallback
    xd = x.cuda()
    with pytest.raises(ValueError):
        ops.resize_nearest(xd, (8, 8), (slice(0, 4, 2), slice(0, 4)))
    with pytest.raises(TypeError):
        ops.resize_nearest(xd.double(), (8, 8))
