"""
GPU tier (pytest -m gpu): the HIP path, called through the C ABI, against
 (1) the committed golden vectors from the reference's own Python and
 (2) the C oracle on the same seeded inputs.
Integer / id outputs must be bit-identical; the softmax score is fp32 within
rtol 1e-5 (north_star tolerance).
"""
import numpy as np
import pytest
import torch

from _golden import load, jload, ids_from_arrays
from nicr_mt_scene_analysis_amd.testing import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'needs the MI355X'
    from nicr_mt_scene_analysis_amd import ops as o
    return o


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def run_hip_pipeline(ops, logits, center, offset, is_thing, kw=None, **extra):
    kw = dict(kw or {})
    r = ops.panoptic_pipeline(
        dev(logits), dev(center), dev(offset), dev(is_thing),
        threshold=kw.get('heatmap_threshold', 0.1),
        kernel_size=kw.get('heatmap_nms_kernel_size', 3),
        top_k=kw.get('top_k_instances', 64),
        apply_foreground_mask=kw.get('heatmap_apply_foreground_mask', False),
        distance_threshold=kw.get('offset_distance_threshold'),
        want_score=True, want_panoptic_semantic=True, **extra)
    torch.cuda.synchronize()
    return {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in r.items()}


def check_against_golden(g, r):
    assert (r['semantic_idx_u8'] == g['semantic_idx']).all()
    st = int(g['semantic_score_stride']) if 'semantic_score_stride' in g else 1
    np.testing.assert_allclose(r['semantic_score'][:, ::st, ::st], g['semantic_score'],
                               rtol=1e-5, atol=1e-7)
    assert (r['foreground'] == g['foreground']).all()
    assert (r['n_centers'] == g['meta_n']).all()
    for b in range(len(r['n_centers'])):
        nb = int(r['n_centers'][b])
        assert (r['centers_yx'][b, :nb] == g['meta_center_yx'][b, :nb]).all()
        assert (r['center_scores'][b, :nb] == g['meta_score'][b, :nb]).all()
        assert (r['area'][b, 1:nb + 1] == g['meta_area'][b, :nb]).all()
    assert (r['instance'] == g['instance']).all()
    assert (r['panoptic'] == g['panoptic']).all()
    assert (r['panoptic_semantic'] == g['panoptic_semantic']).all()
    want = ids_from_arrays(g['ids_n'], g['ids_pan'], g['ids_ins'])
    got = ids_from_arrays(r['n_ids'], r['ids_pan'], r['ids_ins'])
    for a, b in zip(got, want):
        assert list(a.items()) == list(b.items())


@pytest.mark.parametrize('name', ['panoptic_small', 'panoptic_small_kwargs',
                                  'panoptic_edges_plain', 'panoptic_edges_thr'])
def test_pipeline_small_vs_golden(ops, name):
    g = load(name)
    kw = jload(g['kwargs']) if 'kwargs' in g else None
    r = run_hip_pipeline(ops, g['in_semantic_logits'], g['in_instance_center'],
                         g['in_instance_offset'], g['in_semantic_classes_is_thing'], kw)
    check_against_golden(g, r)


@pytest.mark.parametrize('name', ['panoptic_cfg1_q', 'panoptic_cfg1_r'])
def test_pipeline_cfg1_vs_golden(ops, name):
    g = load(name)
    p = jload(g['params'])
    inp = syn.make_panoptic_inputs(p['batch_size'], seed=p['seed'],
                                   quantize_offsets=p['quantize_offsets'])
    digest = syn.input_digest(inp['semantic_logits'], inp['instance_center'],
                              inp['instance_offset'])
    if digest != jload(g['digest']):
        pytest.skip('synthetic inputs differ bit-wise on this host (numpy/libm)')
    r = run_hip_pipeline(ops, inp['semantic_logits'], inp['instance_center'],
                         inp['instance_offset'], inp['semantic_classes_is_thing'])
    check_against_golden(g, r)


@pytest.mark.parametrize('shape', [(3, 7, 50, 37), (2, 40, 120, 160), (1, 150, 96, 128)])
@pytest.mark.parametrize('quant', [True, False])
def test_pipeline_vs_oracle(ops, oracle, shape, quant):
    """odd shapes (scalar path: H*W % 4 != 0), C=150, vs the C oracle."""
    B, C, H, W = shape
    inp = syn.make_panoptic_inputs(B, C, H, W, n_centers=9, seed=21, quantize_offsets=quant)
    r = run_hip_pipeline(ops, inp['semantic_logits'], inp['instance_center'],
                         inp['instance_offset'], inp['semantic_classes_is_thing'])
    is_thing = inp['semantic_classes_is_thing']
    idx, score = oracle.semantic_argmax(inp['semantic_logits'])
    fg = is_thing[idx]
    cyx, n, scores, _ = oracle.center_nms_topk(inp['instance_center'], max_centers=256)
    inst, area = oracle.group_offsets(inp['instance_offset'], fg, cyx, n, scale_y=H, scale_x=W)
    pan, ids = oracle.deeplab_merge(idx + 1, inst, fg, 1 << 16, np.where(is_thing)[0] + 1, 0)
    assert (r['semantic_idx_u8'] == idx).all()
    np.testing.assert_allclose(r['semantic_score'], score, rtol=1e-5, atol=1e-7)
    assert (r['n_centers'] == n).all()
    for b in range(B):
        assert (r['centers_yx'][b, :n[b]] == cyx[b, :n[b]]).all()
    assert (r['instance'] == inst).all()
    assert (r['panoptic'] == pan).all()
    got = ids_from_arrays(r['n_ids'], r['ids_pan'], r['ids_ins'])
    for a, b in zip(got, ids):
        assert list(a.items()) == list(b.items())


def test_cfg5_shape_vs_oracle(ops, oracle):
    """BASELINE configs[4] shape: 1024x768, 150 classes, 48 centers, bf16 logits — one full-size
    image end to end against the C oracle (ids bit-exact)."""
    B, C, H, W = 1, 150, 768, 1024
    dinp = syn.make_panoptic_inputs_torch(B, C, H, W, n_centers=48, seed=31, device='cuda',
                                          logits_dtype=torch.bfloat16)
    lb = dinp['semantic_logits']
    assert lb.dtype == torch.bfloat16
    r = ops.panoptic_pipeline(lb, dinp['instance_center'], dinp['instance_offset'],
                              dinp['semantic_classes_is_thing'], want_score=True,
                              want_panoptic_semantic=True)
    torch.cuda.synchronize()
    r = {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in r.items()}
    inp = {k: v.float().cpu().numpy() if v.is_floating_point() else v.cpu().numpy()
           for k, v in dinp.items() if isinstance(v, torch.Tensor)}
    is_thing = inp['semantic_classes_is_thing'].astype(bool)
    idx, score = oracle.semantic_argmax(inp['semantic_logits'])
    fg = is_thing[idx]
    cyx, n, _, _ = oracle.center_nms_topk(inp['instance_center'], max_centers=256)
    inst, area = oracle.group_offsets(inp['instance_offset'], fg, cyx, n, scale_y=H, scale_x=W)
    pan, ids = oracle.deeplab_merge(idx + 1, inst, fg, 1 << 16, np.where(is_thing)[0] + 1, 0)
    assert (r['semantic_idx_u8'] == idx).all()
    np.testing.assert_allclose(r['semantic_score'], score, rtol=1e-5, atol=1e-7)
    assert (r['n_centers'] == n).all() and (r['centers_yx'][0, :n[0]] == cyx[0, :n[0]]).all()
    assert (r['instance'] == inst).all()
    assert (r['panoptic'] == pan).all()
    assert (r['panoptic_semantic'] == pan // 65536).all()
    got = ids_from_arrays(r['n_ids'], r['ids_pan'], r['ids_ins'])
    assert list(got[0].items()) == list(ids[0].items())


def test_bf16_logits_vs_oracle(ops, oracle):
    inp = syn.make_panoptic_inputs(2, 40, 96, 128, n_centers=7, seed=5)
    lb = torch.from_numpy(inp['semantic_logits']).to(torch.bfloat16)
    r = ops.panoptic_pipeline(lb.cuda(), dev(inp['instance_center']), dev(inp['instance_offset']),
                              dev(inp['semantic_classes_is_thing']), want_score=True)
    torch.cuda.synchronize()
    idx, score = oracle.semantic_argmax(lb.float().numpy())
    assert (r['semantic_idx_u8'].cpu().numpy() == idx).all()
    np.testing.assert_allclose(r['semantic_score'].cpu().numpy(), score, rtol=1e-5, atol=1e-7)


def test_nonfinite_logits_vs_oracle(ops, oracle):
    """NaN / +inf / -inf logits: softmax-then-max semantics (index 0 for degenerate columns,
    -inf entries alone are ordinary), checked against the oracle; f32 and bf16."""
    inp = syn.make_panoptic_inputs(2, 9, 32, 48, n_centers=4, seed=8)
    lg = inp['semantic_logits'].copy()
    lg[0, 3, 2, 2] = np.nan                  # NaN -> index 0
    lg[0, 5, 4, 7] = np.inf                  # +inf -> index 0
    lg[0, 2, 6, 1] = -np.inf                 # a single -inf: ordinary column
    lg[0, :, 8, 8] = -np.inf                 # all -inf -> index 0
    lg[1, 0, 3, 3] = -np.inf
    lg[1, 8, 3, 3] = np.inf                  # both infinities -> index 0
    lg[1, :, 10, 10] = -np.inf
    lg[1, 4, 10, 10] = 1.0                   # all but one -inf: that one wins
    for dt in (torch.float32, torch.bfloat16):
        x = torch.from_numpy(lg).to(dt)
        idx, score = oracle.semantic_argmax(x.float().numpy())
        r = ops.semantic_argmax(x.cuda(), want_u8=True, want_i64=True, want_score=True)
        torch.cuda.synchronize()
        assert (r['idx'].cpu().numpy() == idx).all()
        assert (r['idx_u8'].cpu().numpy() == idx).all()
        np.testing.assert_allclose(r['score'].cpu().numpy(), score, rtol=1e-5, atol=1e-7,
                                   equal_nan=True)
        p = ops.panoptic_pipeline(x.cuda(), dev(inp['instance_center']), dev(inp['instance_offset']),
                                  dev(inp['semantic_classes_is_thing']))
        torch.cuda.synchronize()
        assert (p['semantic_idx_u8'].cpu().numpy() == idx).all()
    assert idx[0, 2, 2] == 0 and idx[0, 4, 7] == 0 and idx[0, 8, 8] == 0 and idx[1, 3, 3] == 0
    assert idx[1, 10, 10] == 4 and idx[0, 6, 1] != 2


def test_centers_adversarial(ops):
    g = load('centers_adversarial')
    for name in jload(g['names']):
        kw = jload(g[f'{name}__kwargs'])
        fg = dev(g[f'{name}__fg']) if f'{name}__fg' in g else None
        r = ops.center_nms_topk(
            dev(g[f'{name}__heat']), fg,
            threshold=kw.get('heatmap_threshold', 0.1),
            kernel_size=kw.get('heatmap_nms_kernel_size', 3),
            top_k=kw.get('top_k_instances', 64),
            apply_foreground_mask=kw.get('heatmap_apply_foreground_mask', False),
            max_centers=1024, want_mask=True)
        torch.cuda.synchronize()
        n = r['n_centers'].cpu().numpy()
        assert (n == g[f'{name}__n']).all(), name
        assert (r['center_mask'].cpu().numpy() == g[f'{name}__mask']).all(), name
        cyx = r['centers_yx'].cpu().numpy()
        for b in range(len(n)):
            assert (cyx[b, :n[b]] == g[f'{name}__centers'][b, :n[b]]).all(), name


def test_centers_many_candidates_vs_oracle(ops, oracle):
    """radix select with > 30k candidates per image and heavy ties."""
    rng = np.random.default_rng(3)
    heat = rng.random((2, 1, 480, 640)).astype(np.float32)
    heat[1] = np.round(heat[1] * 64) / 64
    for k, ks in ((64, 3), (1, 3), (254, 1), (17, 5)):
        cyx, n, scores, mask = oracle.center_nms_topk(heat, ksize=ks, topk=k, max_centers=8192)
        r = ops.center_nms_topk(dev(heat), kernel_size=ks, top_k=k, max_centers=8192,
                                want_mask=True)
        torch.cuda.synchronize()
        assert (r['n_centers'].cpu().numpy() == n).all(), (k, ks)
        assert (r['center_mask'].cpu().numpy() == mask).all(), (k, ks)
        for b in range(2):
            assert (r['centers_yx'].cpu().numpy()[b, :n[b]] == cyx[b, :n[b]]).all()
            assert (r['scores'].cpu().numpy()[b, :n[b]] == scores[b, :n[b]]).all()


def test_centers_candidate_list_sizes_vs_oracle(ops, oracle):
    """k_select_compact keeps up to 2048 candidates per image as a list in LDS (raster order, keys
    gathered once) and walks the mask words beyond that: candidate counts on both sides of the
    capacity, exactly at it, heavy ties, top-k below / at / above the count, with the foreground
    filter, at sizes whose mask is not a whole number of 16-byte pieces"""
    rng = np.random.default_rng(11)
    for (H, W), n_peaks in (((480, 640), 1500), ((480, 640), 2048), ((480, 640), 2049), ((480, 640), 2600),
                            ((96, 200), 300), ((37, 101), 90), ((768, 1024), 2000)):
        heat = np.zeros((2, H, W), np.float32)
        for b in range(2):
            # isolated peaks on a 3-px lattice: every one of them survives the 3x3 NMS
            ys, xs = np.meshgrid(np.arange(1, H - 1, 3), np.arange(1, W - 1, 3), indexing='ij')
            pick = rng.choice(ys.size, size=min(n_peaks, ys.size), replace=False)
            vals = 0.2 + 0.8 * rng.random(pick.size)
            if b == 1:
                vals = np.round(vals * 8) / 8                       # heavy ties around the k-th value
            heat[b, ys.ravel()[pick], xs.ravel()[pick]] = vals.astype(np.float32)
        fg = (rng.random((2, H, W)) < 0.7).astype(np.uint8)
        total = int((heat[0] > 0).sum())
        for k in (1, 64, total - 1, total, total + 5):
            for apply_fg in (False, True):
                cyx, n, scores, mask = oracle.center_nms_topk(heat, fg=fg, ksize=3, topk=k, apply_fg=apply_fg,
                                                              max_centers=4096)
                r = ops.center_nms_topk(dev(heat), dev(fg), kernel_size=3, top_k=k,
                                        apply_foreground_mask=apply_fg, max_centers=4096, want_mask=True)
                torch.cuda.synchronize()
                what = (H, W, n_peaks, k, apply_fg)
                assert (r['n_centers'].cpu().numpy() == n).all(), what
                assert (r['center_mask'].cpu().numpy() == mask).all(), what
                for b in range(2):
                    assert (r['centers_yx'].cpu().numpy()[b, :n[b]] == cyx[b, :n[b]]).all(), what
                    assert (r['scores'].cpu().numpy()[b, :n[b]] == scores[b, :n[b]]).all(), what


def test_grouping_adversarial(ops):
    g = load('grouping_adversarial')
    for name in jload(g['names']):
        kw = jload(g[f'{name}__kwargs'])
        cen = ops.center_nms_topk(dev(g[f'{name}__heat']),
                                  top_k=kw.get('top_k_instances', 64), max_centers=512)
        r = ops.group_offsets(dev(g[f'{name}__offset']), dev(g[f'{name}__fg']),
                              cen['centers_yx'], cen['n_centers'],
                              distance_threshold=kw.get('offset_distance_threshold'))
        torch.cuda.synchronize()
        assert (r['instance'].cpu().numpy() == g[f'{name}__inst']).all(), name
        n = cen['n_centers'].cpu().numpy()
        assert (n == g[f'{name}__meta_n']).all(), name
        area = r['area'].cpu().numpy()
        for b in range(len(n)):
            nb = int(n[b])
            got = np.array([area[b, i] if i <= 255 else 0 for i in range(1, nb + 1)])
            assert (got == g[f'{name}__meta_area'][b, :nb]).all(), name


def test_merge_cases(ops):
    g = load('merge_cases')
    for name in jload(g['names']):
        p = jload(g[f'{name}__params'])
        sem, ins = g[f'{name}__sem'], g[f'{name}__ins']
        n_classes = int(sem.max()) + 1
        lut = np.zeros((n_classes,), np.uint8)
        for t in p['thing_ids']:
            if t < n_classes:
                lut[t] = 1
        if ins.dtype == np.uint8:
            r = ops.panoptic_merge(dev(sem), dev(ins), dev(g[f'{name}__thing']), dev(lut),
                                   p['max_inst'], p['void'])
        else:       # ground-truth style ids (int32, up to 65535): ranked on the device
            r = ops.panoptic_merge_wide(dev(sem), dev(ins), dev(g[f'{name}__thing']), dev(lut),
                                        p['max_inst'], p['void'],
                                        max_segments=4096 if name == 'many_ids' else 1024)
            assert int(r['status'].item()) == 0
        torch.cuda.synchronize()
        assert (r['panoptic'].cpu().numpy() == g[f'{name}__pan']).all(), name
        got = ids_from_arrays(r['n_ids'].cpu().numpy(), r['ids_pan'].cpu().numpy(),
                              r['ids_ins'].cpu().numpy())
        want = ids_from_arrays(g[f'{name}__ids_n'], g[f'{name}__ids_pan'], g[f'{name}__ids_ins'])
        for a, b in zip(got, want):
            assert list(a.items()) == list(b.items()), name


def test_orientation_cases(ops):
    g = load('orientation_cases')
    for name, mask in (('masked', g['mask']), ('nomask', None)):
        r = ops.instance_orientation_sums(dev(g['orientation']), dev(g['inst']),
                                          None if mask is None else dev(mask))
        torch.cuda.synchronize()
        sums = r['sums'].cpu().numpy()
        cnt = r['count'].cpu().numpy()
        want = g[f'{name}__angle']
        assert ((cnt > 0) == ~np.isnan(want)).all()
        ang = np.arctan2(sums[..., 1].astype(np.float32), sums[..., 0].astype(np.float32))
        np.testing.assert_allclose(ang[cnt > 0], want[cnt > 0], rtol=1e-5, atol=1e-5)


def test_full_size_properties(ops):
    """B=32 640x480 (BASELINE cfg2): size-independent properties of the merge."""
    B, C, H, W = 32, 40, 480, 640
    g = torch.Generator(device='cuda').manual_seed(0)
    coarse = torch.randn((B, C, H // 32, W // 32), device='cuda', generator=g)
    logits = 4 * torch.nn.functional.interpolate(coarse, size=(H, W), mode='bilinear')
    center = torch.zeros((B, 1, H, W), device='cuda')
    cy = torch.randint(8, H - 8, (B, 24), device='cuda', generator=g)
    cx = torch.randint(8, W - 8, (B, 24), device='cuda', generator=g)
    yy = torch.arange(H, device='cuda').view(1, 1, H, 1)
    xx = torch.arange(W, device='cuda').view(1, 1, 1, W)
    d2 = (yy - cy.view(B, 24, 1, 1)) ** 2 + (xx - cx.view(B, 24, 1, 1)) ** 2
    center[:, 0] = torch.exp(-d2 / 128.0).amax(dim=1)
    nearest = d2.argmin(dim=1)
    oy = (torch.gather(cy, 1, nearest.view(B, -1)).view(B, H, W) - yy[0]) / H
    ox = (torch.gather(cx, 1, nearest.view(B, -1)).view(B, H, W) - xx[0]) / W
    offset = torch.stack([oy, ox], 1).float().contiguous()
    is_thing = (torch.arange(C, device='cuda') >= C // 2)
    r = ops.panoptic_pipeline(logits, center, offset, is_thing, want_panoptic_semantic=True)
    torch.cuda.synchronize()
    pan, inst, sem = r['panoptic'], r['instance'], r['semantic_idx_u8'].long()
    fg = r['foreground']
    # argmax agrees with torch on the device
    assert (sem == logits.argmax(dim=1)).all()
    assert (fg == is_thing[sem]).all()
    # instances only on foreground, every fg pixel got one (no threshold)
    assert ((inst > 0) == fg).all()
    # stuff pixels carry class*65536; thing pixels carry a per-class running number
    stuff = ~fg
    assert (pan[stuff] == (sem[stuff] + 1) * 65536).all()
    assert (pan[fg] % 65536 > 0).all()
    assert (r['panoptic_semantic'] == pan // 65536).all()
    # the id dict is a bijection between used instance ids and panoptic ids
    n_ids = r['n_ids'].cpu().numpy()
    for b in range(B):
        used = torch.unique(inst[b][inst[b] > 0]).cpu().numpy()
        ids_ins = r['ids_ins'][b, :n_ids[b]].cpu().numpy()
        assert (np.sort(ids_ins) == used).all()
        pans = torch.unique(pan[b][fg[b]]).cpu().numpy()
        assert (np.sort(r['ids_pan'][b, :n_ids[b]].cpu().numpy()) == pans).all()
    # area sums to the number of foreground pixels
    assert (r['area'].sum(dim=1).cpu() == fg.flatten(1).sum(dim=1).cpu()).all()
    # idempotence / determinism: a second run is bit-identical
    r2 = ops.panoptic_pipeline(logits, center, offset, is_thing)
    assert (r2['panoptic'] == pan).all() and (r2['instance'] == inst).all()


def test_graphed_pipeline_matches_eager(ops):
    """HIP-graph replay of the hot path (launch-bound small batches) == eager launches, also
    after the bound inputs were overwritten in place."""
    from nicr_mt_scene_analysis_amd.graph import GraphedPanopticPipeline
    a = syn.make_panoptic_inputs(1, 12, 96, 128, n_centers=5, seed=41)
    b = syn.make_panoptic_inputs(1, 12, 96, 128, n_centers=7, seed=42)
    lg, ce, of = dev(a['semantic_logits']), dev(a['instance_center']), dev(a['instance_offset'])
    th = dev(a['semantic_classes_is_thing'])
    pipe = GraphedPanopticPipeline(lg, ce, of, th, want_score=True)
    for inp in (a, b, a):
        lg.copy_(dev(inp['semantic_logits']))
        ce.copy_(dev(inp['instance_center']))
        of.copy_(dev(inp['instance_offset']))
        out = pipe.replay()
        torch.cuda.synchronize()
        got = {k: v.clone() for k, v in out.items() if isinstance(v, torch.Tensor)}
        ref = ops.panoptic_pipeline(lg, ce, of, th, want_score=True)
        torch.cuda.synchronize()
        for k in ('panoptic', 'instance', 'semantic_idx_u8', 'semantic_score', 'n_centers',
                  'centers_yx', 'ids_pan', 'ids_ins', 'n_ids', 'area'):
            n = None
            if k in ('centers_yx',):
                n = int(ref['n_centers'][0])
                assert torch.equal(got[k][:, :n], ref[k][:, :n]), k
            elif k in ('ids_pan', 'ids_ins'):
                n = int(ref['n_ids'][0])
                assert torch.equal(got[k][:, :n], ref[k][:, :n]), k
            else:
                assert torch.equal(got[k], ref[k]), k


def test_argmax_tie_band_boundary_hip(oracle):
    """the HIP argmax on the reference-run near-tie fixture (CPU-tier twin with the details:
    test_oracle_vs_golden.py): EVERY column and pixel as the reference returned it — the kernels
    decide candidate columns with ATen's own softmax arithmetic (argmax_state.hpp) — for the
    stand-alone argmax, the with-score variant, the fused kernel and the full-resolution path"""
    from _golden import aten_softmax_argmax
    from nicr_mt_scene_analysis_amd import ops
    g = load('argmax_ties')

    def all_paths(x):
        out = [ops.semantic_argmax(x, want_u8=True, want_i64=False, want_score=False)['idx_u8'],
               ops.semantic_argmax(x, want_u8=True, want_i64=False, want_score=True)['idx_u8']]
        B, C, H, W = x.shape
        zeros = torch.zeros((B, 1, H, W), device='cuda')
        for want_score in (False, True):                  # both instantiations of the fused kernel
            r = ops.panoptic_pipeline(x, zeros, torch.zeros((B, 2, H, W), device='cuda'),
                                      torch.zeros((C,), dtype=torch.bool, device='cuda'),
                                      want_score=want_score)
            out.append(r['semantic_idx_u8'])
        # foreground-masked heat-map: the argmax runs first as its own kernel
        r = ops.panoptic_pipeline(x, zeros, torch.zeros((B, 2, H, W), device='cuda'),
                                  torch.ones((C,), dtype=torch.bool, device='cuda'),
                                  apply_foreground_mask=True)
        out.append(r['semantic_idx_u8'])
        return [o.cpu().numpy() for o in out]

    x = torch.from_numpy(g['logits']).cuda()
    ref = g['ref_idx'].reshape(-1)
    delta = g['delta']
    in_between = (delta > 2.0 ** -25) & (delta <= 2.0 ** -23)
    assert in_between.sum() == 229 and (ref[in_between] == g['c1'][in_between]).sum() == 38
    for idx in all_paths(x):
        assert (idx.reshape(-1) == ref).all()                 # all 1680 columns
    # the score of a candidate column is the reference's own maximum probability
    sc = ops.semantic_argmax(x, want_u8=True, want_i64=False, want_score=True)['score'].cpu().numpy()
    twin_idx, twin_p = aten_softmax_argmax(g['logits'])
    np.testing.assert_allclose(sc, twin_p, rtol=1e-5)
    for name in ('tiny', 'small'):
        xn = g[f'{name}_logits']
        for idx in all_paths(torch.from_numpy(xn).cuda()):
            assert (idx == g[f'{name}_ref_idx']).all(), name
        # full resolution: the same rule on the interpolated logits
        ref_full = g[f'{name}_ref_idx_fullres']
        size = ref_full.shape[-2:]
        for want_score in (False, True):
            got = ops.semantic_argmax_resized(torch.from_numpy(xn).cuda(), size, None, want_u8=True,
                                              want_i64=False, want_score=want_score
                                              )['idx_u8'].cpu().numpy()
            assert (got == ref_full).all(), (name, want_score)
    # 16-bit logits (softmax of the upcast values): gaps that small exist only below 2^-16 (bf16) /
    # 2^-13 (f16)
    for dt, scale in ((torch.bfloat16, 2.0 ** -40), (torch.bfloat16, 2.0 ** -17), (torch.float16, 2.0 ** -14)):
        xb = (torch.randn((1, 12, 16, 24), device='cuda') * scale).to(dt)
        want_b, _ = aten_softmax_argmax(xb.float().cpu().numpy())
        for idx in all_paths(xb):
            assert (idx == want_b).all(), (dt, scale)
    # the edge of the trigger: a maximum of exactly 1.0 (f32) / 2^-16 (bf16) / 2^-13 (f16) has its
    # lower neighbour 2^-24 away — e = 1 - 2^-24, a tie or not with the column's sum —; one
    # binade up the neighbour is 2^-23 away and can never tie
    for dt, top in ((torch.float32, 1.0), (torch.bfloat16, 2.0 ** -16), (torch.float16, 2.0 ** -13),
                    (torch.float32, 2.0)):
        xe = torch.full((1, 6, 8, 16), -1.0, dtype=torch.float32)
        xe[:, 4] = top
        gap = 2.0 ** -24 * (2.0 if top == 2.0 else 1.0)
        xe[:, 1, :, ::2] = top - gap
        xe[:, 0, 1] = -top                                    # (a negative twin of the maximum: no tie)
        xe[:, 2] = torch.linspace(-3.0, top - 0.25, 8 * 16).reshape(8, 16)       # a different sum per pixel
        xe = xe.to(dt)
        assert (xe[:, 1, :, ::2].float() == top - gap).all()               # representable
        want_e, _ = aten_softmax_argmax(xe.float().numpy())
        if top != 2.0:
            assert 0 < (want_e == 1).sum()                                 # some pixels do tie
        else:
            assert (want_e == 4).all()
        for idx in all_paths(xe.cuda()):
            assert (idx == want_e).all(), (dt, top)


def test_argmax_on_natural_logits_vs_the_restated_softmax():
    """a1 at full size: the bench's logits (blobby segments, B=32 640x480 C=40: 9.8 M columns) and
    the same maps scaled to |x| < 1, where fp32 is fine enough for probability ties to exist.
    Checked against the numpy twin of ATen's softmax -> max on every column that has a lower class
    within 2^-22 of its maximum (everywhere else the answer is the plain argmax, asserted too),
    and — informational — against torch's own softmax -> max on this machine's CPU, whose scalar
    tail per thread chunk / SIMD width may differ from the build container's."""
    from _golden import aten_softmax_argmax
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn
    inp = syn.make_panoptic_inputs_torch(32, 40, 480, 640, n_centers=24, seed=1234, device='cuda')
    base = inp['semantic_logits']
    report = {}
    for name, x in (('bench', base), ('scaled_to_unit', base / base.abs().max())):
        x = x.contiguous()
        got = ops.semantic_argmax(x, want_u8=True, want_i64=True, want_score=False)['idx']
        m, am = x.max(dim=1, keepdim=True)
        d = x - m                                                       # fp32, like ATen's x - max
        cls = torch.arange(x.shape[1], device='cuda').view(1, -1, 1, 1)
        near = ((d >= -2.0 ** -22) & (cls < am)).any(dim=1)             # a lower class comes close
        assert torch.equal(got[~near], am[:, 0][~near])                 # plain first-index argmax
        cols = x.permute(0, 2, 3, 1)[near].cpu().numpy()                # [n, C]
        n_near = cols.shape[0]
        differing = 0
        if n_near:
            want, _ = aten_softmax_argmax(cols.T[None, :, :, None])     # [1, C, n, 1]
            want = want.reshape(-1)
            differing = int((want != am[:, 0][near].cpu().numpy()).sum())
            assert (got[near].cpu().numpy() == want).all()
        ref = torch.softmax(x.cpu(), dim=1).max(dim=1)[1].cuda()        # this host's torch
        report[name] = {'columns': near.numel(), 'columns_with_a_close_lower_class': n_near,
                        'of_those_not_the_plain_argmax': differing,
                        'differing_from_this_hosts_torch': int((ref != got).sum())}
        assert n_near <= near.numel() * 1e-4, report
    print('a1 natural logits:', report)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_pipeline_image_side_beyond_16_bits(ops, oracle, dtype):
    """an image side > 65535: the lane-held center table of the fused kernel packs (y, x) in 32
    bits, so it must fall back to the LDS table — ids still bit-exact vs the oracle"""
    B, C, H, W = 1, 3, 2, 70000
    rng = np.random.default_rng(5)
    logits = rng.integers(0, 5, (B, C, H, W)).astype(np.float32)
    heat = np.zeros((B, 1, H, W), np.float32)
    xs = np.array([100, 30000, 65600, 69990])
    heat[0, 0, 1, xs] = [0.9, 0.8, 0.7, 0.6]
    off_px = rng.integers(-6, 7, (B, 2, H, W)).astype(np.float32)
    offset = np.stack([off_px[:, 0] / H, off_px[:, 1] / W], 1).astype(np.float32)
    is_thing = np.array([False, True, True])
    x = dev(logits).to(getattr(torch, dtype))
    r = ops.panoptic_pipeline(x, dev(heat), dev(offset), dev(is_thing), kernel_size=1)
    torch.cuda.synchronize()
    idx, _ = oracle.semantic_argmax(x.float().cpu().numpy())
    fg = is_thing[idx]
    cyx, n, _, _ = oracle.center_nms_topk(heat, ksize=1, max_centers=256)
    inst, _ = oracle.group_offsets(offset, fg, cyx, n, scale_y=H, scale_x=W)
    pan, _ = oracle.deeplab_merge(idx + 1, inst, fg, 1 << 16, np.where(is_thing)[0] + 1, 0)
    assert int(n[0]) == 4 and (r['n_centers'].cpu().numpy() == n).all()
    assert (r['instance'].cpu().numpy() == inst).all()
    assert (r['panoptic'].cpu().numpy() == pan).all()


def test_deeplab_merge_more_than_4096_instance_ids(oracle):
    """a ground-truth style map with more distinct instance ids than the 4096 the ranked merge
    held until round 4 (reference utils/panoptic_merge.py:172-225 takes any number): 4608 ids on
    one image, classes and thing mask disagreeing here and there — panoptic map and the id dict in
    the reference's insertion order, against the C oracle"""
    from nicr_mt_scene_analysis_amd.utils.panoptic_merge import deeplab_merge_batch
    rng = np.random.default_rng(65)
    H = W = 96
    n_cells = H * (W // 2)
    ids = rng.permutation(np.arange(1, 65536))[:n_cells]
    ins = np.repeat(ids.reshape(1, H, W // 2), 2, axis=2).astype(np.int64)          # 1 x 2 px per instance
    ins[0, rng.random((H, W)) < 0.05] = 0
    sem = np.repeat(np.repeat(rng.integers(0, 6, (1, H // 4, W // 4)), 4, 1), 4, 2).astype(np.int64)
    sem[0, rng.random((H, W)) < 0.1] = rng.integers(0, 6)
    thing_ids = [2, 3, 4]
    thing_seg = np.isin(sem, thing_ids) ^ (rng.random((1, H, W)) < 0.05)
    assert len(np.unique(ins)) - 1 > 4096
    want_pan, want_ids = oracle.deeplab_merge(sem, ins, thing_seg, 1 << 16, thing_ids, 0, cap=8192)
    pan, dicts = deeplab_merge_batch(torch.from_numpy(sem).cuda(), torch.from_numpy(ins.astype(np.int32)).cuda(),
                                     torch.from_numpy(thing_seg).cuda(), 1 << 16, thing_ids, 0)
    assert np.array_equal(pan.cpu().numpy(), want_pan)
    assert len(want_ids[0]) > 1000
    assert [list(d.items()) for d in dicts] == [list(d.items()) for d in want_ids]
