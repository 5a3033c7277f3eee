"""
GPU tier: the whole-column losses at the configs[4] shapes.

 * `k_ce_split` (csrc/losses.hip): cross entropy for 49..256 classes, forward + gradient in one
   pass with the class column split over the four waves of a workgroup (reference
   loss/ce.py:40-68) — against torch's fp64 `F.cross_entropy` (the op the reference calls), the
   two-kernel path of this library, the confirm / recompute protocol of the speculative gradient,
   -inf logits, a full-size 150-class image against the C oracle;
 * the dense cosine-embedding loss (reference loss/cos_emb.py:21-56 +
   task_helper/dense_visual_embedding.py:110-171) on index maps with aligned segments, segment
   boundaries anywhere, per-pixel noise and no targets at all, embedding sizes around the LDS
   chunking of `k_cos_emb_lds`, all three dtypes — against torch's fp64 op.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _gen(seed=0):
    return torch.Generator(device='cuda').manual_seed(seed)


def _grad_tol(dtype):
    return {torch.float32: 2e-5, torch.bfloat16: 2 ** -7, torch.float16: 2 ** -9}[dtype]


def _stats():
    from nicr_mt_scene_analysis_amd.loss import speculation_stats
    return speculation_stats()


def _cos_reference(x, idx, lut):
    """torch.nn.functional.cosine_embedding_loss on the gathered rows, fp64 (what the
    reference's task helper computes), + autograd gradient of sum / n"""
    xr = x.double().requires_grad_(True)
    B, D, H, W = x.shape
    valid = idx != 0
    rows = xr.permute(0, 2, 3, 1)[valid]
    b_idx = torch.where(valid)[0]
    tgt = lut.double()[b_idx, (idx[valid] - 1).long()]
    n = int(valid.sum())
    if n == 0:
        return 0.0, 0, torch.zeros_like(xr)
    loss = torch.nn.functional.cosine_embedding_loss(
        rows, tgt, torch.ones(n, device=x.device, dtype=torch.float64), reduction='sum')
    (loss / n).backward()
    return float(loss), n, xr.grad


def _index_map(kind, B, H, W, L, g):
    if kind == 'blocks':                       # 8-px aligned segments: every lane uniform
        idx = torch.randint(0, L + 1, (B, (H + 7) // 8, (W + 7) // 8), device='cuda', generator=g,
                            dtype=torch.int32)
        return idx.repeat_interleave(8, 1).repeat_interleave(8, 2)[:, :H, :W].contiguous()
    if kind == 'segments':                     # boundaries anywhere: two rows inside a lane
        idx = torch.randint(0, L + 1, (B, (H + 4) // 5, (W + 10) // 11), device='cuda', generator=g,
                            dtype=torch.int32)
        return idx.repeat_interleave(5, 1).repeat_interleave(11, 2)[:, :H, :W].contiguous()
    if kind == 'noise':                        # a different row per pixel
        return torch.randint(0, L + 1, (B, H, W), device='cuda', generator=g, dtype=torch.int32)
    if kind == 'empty':
        return torch.zeros((B, H, W), device='cuda', dtype=torch.int32)
    raise ValueError(kind)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize('kind', ['blocks', 'segments', 'noise', 'empty'])
@pytest.mark.parametrize('shape', [(2, 64, 8, 24), (1, 100, 12, 20), (3, 512, 4, 40), (1, 770, 6, 28),
                                   (2, 33, 16, 16)])
def test_cos_emb_vs_torch_fp64(dtype, kind, shape):
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss
    B, D, H, W = shape
    L = 7
    g = _gen(D + H)
    x = torch.randn((B, D, H, W), device='cuda', generator=g).to(dtype)
    lut = torch.nn.functional.normalize(torch.randn((B, L, D), device='cuda', generator=g), dim=-1)
    idx = _index_map(kind, B, H, W, L, g)
    ref_loss, ref_n, ref_grad = _cos_reference(x, idx, lut)

    xs = x.clone().requires_grad_(True)
    loss, n = CosineEmbeddingLoss().lut_sum(xs, idx, lut)
    (loss / n.clamp(min=1)).backward()
    assert int(n) == ref_n
    np.testing.assert_allclose(float(loss), ref_loss, rtol=RTOL, atol=1e-6)
    tol = _grad_tol(dtype)
    atol = tol * float(ref_grad.abs().max()) * 0.05 + (6e-8 if dtype == torch.float16 else 1e-12)   # f16 subnormals
    np.testing.assert_allclose(xs.grad.double().cpu().numpy(), ref_grad.cpu().numpy(), rtol=tol, atol=atol)


def test_cos_emb_out_of_range_index_sets_status():
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss, check_loss_status
    g = _gen(5)
    x = torch.randn((1, 64, 8, 16), device='cuda', generator=g).requires_grad_(True)
    lut = torch.randn((1, 3, 64), device='cuda', generator=g)
    idx = torch.randint(0, 4, (1, 8, 16), device='cuda', generator=g, dtype=torch.int32)
    check_loss_status()
    idx[0, 3, 5] = 9
    loss, n = CosineEmbeddingLoss().lut_sum(x, idx, lut)
    with pytest.raises(IndexError):
        check_loss_status()
    assert int(n) == int(((idx > 0) & (idx <= 3)).sum())


def _ce_case(B, C, H, W, dtype, seed, void_frac=0.2):
    g = _gen(seed)
    x = (torch.randn((B, C, H, W), device='cuda', generator=g) * 3).to(dtype)
    t = torch.randint(1, C + 1, (B, H, W), device='cuda', generator=g)
    t[torch.rand((B, H, W), device='cuda', generator=g) < void_frac] = 0
    w = torch.rand(C, device='cuda', generator=g) + 0.5
    return x, t.to(torch.uint8), w


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize('C', [49, 64, 150, 151, 255])
@pytest.mark.parametrize('label_smoothing', [0.0, 0.1])
@pytest.mark.parametrize('shape', [(2, 24, 36), (1, 8, 1000), (3, 5, 8), (2, 7, 9)])    # last: rows not readable as 8-byte pieces
def test_ce_split_vs_torch_fp64(dtype, C, label_smoothing, shape):
    from nicr_mt_scene_analysis_amd.loss import _functional as _F
    if not _F.speculation_enabled():
        pytest.skip('NMSA_SPECULATIVE_GRAD=0: forward-written gradients are switched off')
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    B, H, W = shape
    x, t, w = _ce_case(B, C, H, W, dtype, seed=C + H)
    n = F_.count_u8(t, 1, C)
    scale = F_.expected_scale(n)

    xs = x.clone().requires_grad_(True)
    before = _stats()
    loss, n_el, wsum = F_.cross_entropy_sum(xs, t, w, label_smoothing, expected_scale=scale)
    (loss / n_el).backward()
    after = _stats()
    assert (after['confirmed'] - before['confirmed'], after['recomputed'] - before['recomputed']) == (1, 0)
    assert int(n_el) == int(n) == int((t != 0).sum())
    np.testing.assert_allclose(float(wsum), float(w[(t[t != 0] - 1).long()].double().sum()), rtol=1e-6)

    xr = x.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t.long() - 1, weight=w.double(), reduction='sum',
                                            ignore_index=-1, label_smoothing=label_smoothing)
    (ref / int(n)).backward()
    np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL)
    tol = _grad_tol(dtype)
    atol = max(tol * float(xr.grad.abs().max()) * 0.05, 4e-6 * float(w.max()) / int(n),
               6e-8 if dtype == torch.float16 else 0.0)                      # f16 subnormals
    np.testing.assert_allclose(xs.grad.double().cpu().numpy(), xr.grad.cpu().numpy(), rtol=tol, atol=atol)

    # wrong expectation -> recomputed by the same single-pass kernel, identical gradient
    xw = x.clone().requires_grad_(True)
    loss_w, n_w, _ = F_.cross_entropy_sum(xw, t, w, label_smoothing,
                                          expected_scale=torch.full((1,), 3.0, device='cuda'))
    (loss_w / n_w).backward()
    assert torch.equal(xw.grad, xs.grad)
    np.testing.assert_allclose(float(loss_w), float(loss), rtol=1e-7)

    # two-kernel path (forward + saved log-sum-exp, backward)
    xu = x.clone().requires_grad_(True)
    loss_u, n_u, _ = F_.cross_entropy_sum(xu, t, w, label_smoothing)
    (loss_u / n_u).backward()
    np.testing.assert_allclose(float(loss_u), float(ref), rtol=RTOL)
    np.testing.assert_allclose(xu.grad.double().cpu().numpy(), xs.grad.double().cpu().numpy(),
                               rtol=tol, atol=atol)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize('C,hw', [(150, (40, 96)), (64, (33, 64)), (255, (16, 128))])
def test_ce_split_piecewise_constant_labels(dtype, C, hw):
    """label maps of real scenes are piecewise constant: most class planes of a wave's 64 x PXT pixels
    have no target pixel and k_ce_split skips their target selects (the wave-uniform `present`
    mask); blocks of 8 x 16 pixels, a void stripe, one block row of a single class"""
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    H, W = hw
    B = 2
    g = _gen(C + H)
    x = (torch.randn((B, C, H, W), device='cuda', generator=g) * 3).to(dtype)
    coarse = torch.randint(1, C + 1, (B, (H + 7) // 8, (W + 15) // 16), device='cuda', generator=g)
    t = coarse.repeat_interleave(8, 1).repeat_interleave(16, 2)[:, :H, :W].contiguous()
    t[:, :, 5:9] = 0
    t[:, 8:16] = C                                       # the last class: held by the last wave only
    t = t.to(torch.uint8)
    w = torch.rand(C, device='cuda', generator=g) + 0.5
    n = F_.count_u8(t, 1, C)
    xs = x.clone().requires_grad_(True)
    loss, n_el, wsum = F_.cross_entropy_sum(xs, t, w, 0.0, expected_scale=F_.expected_scale(n))
    (loss / n_el).backward()
    xr = x.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t.long() - 1, weight=w.double(), reduction='sum', ignore_index=-1)
    (ref / int(n)).backward()
    assert int(n_el) == int(n)
    np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL)
    tol = _grad_tol(dtype)
    atol = max(tol * float(xr.grad.abs().max()) * 0.05, 4e-6 * float(w.max()) / int(n),
               6e-8 if dtype == torch.float16 else 0.0)
    np.testing.assert_allclose(xs.grad.double().cpu().numpy(), xr.grad.cpu().numpy(), rtol=tol, atol=atol)
    # the target elements themselves (p - 1 cancels there): taken from the logit, tight
    tgt = (t.long() - 1).clamp(min=0).unsqueeze(1)
    got_t = xs.grad.double().gather(1, tgt)[t.unsqueeze(1) != 0]
    want_t = xr.grad.gather(1, tgt)[t.unsqueeze(1) != 0]
    np.testing.assert_allclose(got_t.cpu().numpy(), want_t.cpu().numpy(), rtol=tol, atol=atol)


def test_ce_split_bf16_far_classes_keep_their_gradient():
    """the wide kernel keeps the exponentials of its sum walk in the register tile as well (bf16:
    fp16 pairs of e * 2^14): classes 12 ... 20 below the maximum still get torch's fp32 softmax
    gradient rounded to bf16, element-wise, at 150 classes"""
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    B, C, H, W = 1, 150, 8, 64
    g = _gen(151)
    gaps = 12.0 + 8.0 * torch.rand((B, C, H, W), device='cuda', generator=g)
    x = -gaps
    top = torch.randint(0, C, (B, 1, H, W), device='cuda', generator=g)
    x.scatter_(1, top, 0.0)
    x = (x + torch.randn((B, 1, H, W), device='cuda', generator=g)).to(torch.bfloat16)
    t = torch.randint(1, C + 1, (B, H, W), device='cuda', generator=g).to(torch.uint8)
    n = F_.count_u8(t, 1, C)
    xs = x.clone().requires_grad_(True)
    loss, n_el, _ = F_.cross_entropy_sum(xs, t, None, 0.0, expected_scale=F_.expected_scale(n))
    (loss / n_el).backward()
    xr = x.float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t.long() - 1, reduction='sum')
    (ref / int(n)).backward()
    want = xr.grad.to(torch.bfloat16).float()
    got = xs.grad.float()
    np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2 ** -7, atol=1e-9 / int(n) * 0.1)
    small = (xr.grad.abs() < 1e-5 / int(n)) & (xr.grad != 0)
    assert int(small.sum()) > 1000 and bool((got[small] != 0).all())



def test_ce_split_minus_infinity_logits():
    """classes at -inf (masked logits): the waves whose rows hold only -inf must contribute
    nothing instead of NaN"""
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    C = 150
    x, t, w = _ce_case(1, C, 8, 32, torch.float32, seed=3)
    x[:, 40:130] = -float('inf')
    t[(t > 40) & (t <= 130)] = 0
    n = F_.count_u8(t, 1, C)
    xs = x.clone().requires_grad_(True)
    loss, n_el, _ = F_.cross_entropy_sum(xs, t, w, 0.0, expected_scale=F_.expected_scale(n))
    (loss / n_el).backward()
    xr = x.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t.long() - 1, weight=w.double(), reduction='sum',
                                            ignore_index=-1)
    (ref / int(n)).backward()
    np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL)
    np.testing.assert_allclose(xs.grad.double().cpu().numpy(), xr.grad.cpu().numpy(), rtol=2e-5, atol=1e-9)


def test_ce_c150_full_size_vs_oracle():
    """configs[4] shape (one 768x1024 image, 150 classes, bf16): loss sum and sampled gradient
    against the C oracle / fp64 torch"""
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    from oracle import oracle as orc
    B, C, H, W = 1, 150, 768, 1024
    x, t, w = _ce_case(B, C, H, W, torch.bfloat16, seed=150)
    n = F_.count_u8(t, 1, C)
    xs = x.clone().requires_grad_(True)
    loss, n_el, wsum = F_.cross_entropy_sum(xs, t, w, 0.0, expected_scale=F_.expected_scale(n))
    (loss / n_el).backward()
    want, want_n, want_w, _ = orc.loss_ce(x.float().cpu().numpy(), t.cpu().numpy(), w.cpu().numpy(), 0.0)
    assert int(n_el) == int(want_n)
    np.testing.assert_allclose(float(loss), want, rtol=RTOL)
    np.testing.assert_allclose(float(wsum), want_w, rtol=RTOL)
    # gradient on a strip of rows against fp64 autograd
    rows = slice(300, 304)
    xr = x[:, :, rows].double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t[:, rows].long() - 1, weight=w.double(),
                                            reduction='sum', ignore_index=-1)
    (ref / int(n)).backward()
    got = xs.grad[:, :, rows].double()
    atol = max(2 ** -7 * float(xr.grad.abs().max()) * 0.05, 4e-6 * float(w.max()) / int(n))
    np.testing.assert_allclose(got.cpu().numpy(), xr.grad.cpu().numpy(), rtol=2 ** -7, atol=atol)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize('D,L', [(64, 1), (128, 64), (320, 9), (512, 64), (576, 5), (768, 64), (1024, 33),
                                 (96, 5), (300, 17), (700, 64), (1001, 8)])        # D % 64 != 0: a ragged last wave
@pytest.mark.parametrize('hw', [(4, 68), (24, 44), (3, 1000)])          # last tile ragged / several tiles
def test_cos_split_one_pass_confirms_and_recomputes(dtype, D, L, hw):
    """k_cos_split (forward + gradient in one pass, csrc/losses_cos.hip): the shapes it takes go
    through it (one backward check confirmed, none recomputed), loss / count / gradient equal
    torch in fp64; an upstream factor it did not expect is recomputed by the same kernel with the
    identical result; the forward-only call and the two-kernel path agree with it"""
    from nicr_mt_scene_analysis_amd.loss import _functional as _F
    if not _F.speculation_enabled():
        pytest.skip('NMSA_SPECULATIVE_GRAD=0: forward-written gradients are switched off')
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss, _multi
    H, W = hw
    B = 2
    g = _gen(D + W)
    x = torch.randn((B, D, H, W), device='cuda', generator=g).to(dtype)
    lut = torch.nn.functional.normalize(torch.randn((B, L, D), device='cuda', generator=g), dim=-1)
    idx = _index_map('noise' if W == 44 else 'segments', B, H, W, L, g)
    assert _multi.cos_supported(x, lut)
    ref_loss, ref_n, ref_grad = _cos_reference(x, idx, lut)
    cos = CosineEmbeddingLoss()
    xs = x.clone().requires_grad_(True)
    before = _stats()
    loss, n = cos.lut_sum(xs, idx, lut)
    (loss / n.clamp(min=1)).backward()
    after = _stats()
    assert (after['confirmed'] - before['confirmed'], after['recomputed'] - before['recomputed']) == (1, 0)
    assert int(n) == ref_n
    np.testing.assert_allclose(float(loss), ref_loss, rtol=RTOL, atol=1e-6)
    tol = _grad_tol(dtype)
    atol = tol * float(ref_grad.abs().max()) * 0.05 + (6e-8 if dtype == torch.float16 else 1e-12)
    np.testing.assert_allclose(xs.grad.double().cpu().numpy(), ref_grad.cpu().numpy(), rtol=tol, atol=atol)
    # an unannounced factor: one miss (recomputed, same bits up to the factor), then learned
    xw = x.clone().requires_grad_(True)
    lw, nw = cos.lut_sum(xw, idx, lut)
    (4.0 * (lw / nw.clamp(min=1))).backward()
    assert _stats()['recomputed'] - after['recomputed'] == 1
    np.testing.assert_allclose(xw.grad.double().cpu().numpy(), 4.0 * ref_grad.cpu().numpy(), rtol=tol, atol=4 * atol)
    xw2 = x.clone().requires_grad_(True)
    lw2, nw2 = cos.lut_sum(xw2, idx, lut)
    (4.0 * (lw2 / nw2.clamp(min=1))).backward()
    assert _stats()['recomputed'] - after['recomputed'] == 1               # confirmed this time
    assert torch.equal(xw2.grad, xw.grad)
    # forward only (two-kernel forward) and the two-kernel backward agree
    with torch.no_grad():
        lf, nf = cos.lut_sum(x, idx, lut)
    assert int(nf) == ref_n
    np.testing.assert_allclose(float(lf), float(loss), rtol=1e-6, atol=1e-6)


def test_cos_split_falls_back_where_it_cannot_run():
    """D > 1024, a LUT beyond the LDS, a pixel count that is no multiple of 4: the two-kernel path
    answers (same results, no expectation involved)"""
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss, _multi
    g = _gen(77)
    for (B, D, H, W, L) in ((1, 1030, 8, 16, 5), (1, 1088, 4, 16, 5), (1, 512, 4, 16, 200), (1, 128, 3, 7, 4)):
        x = torch.randn((B, D, H, W), device='cuda', generator=g).to(torch.bfloat16)
        lut = torch.nn.functional.normalize(torch.randn((B, L, D), device='cuda', generator=g), dim=-1)
        idx = _index_map('segments', B, H, W, L, g)
        assert not _multi.cos_supported(x, lut), (D, H, W, L)
        ref_loss, ref_n, ref_grad = _cos_reference(x, idx, lut)
        xs = x.clone().requires_grad_(True)
        before = _stats()
        loss, n = CosineEmbeddingLoss().lut_sum(xs, idx, lut)
        (loss / n.clamp(min=1)).backward()
        assert _stats() == before
        assert int(n) == ref_n
        np.testing.assert_allclose(float(loss), ref_loss, rtol=RTOL, atol=1e-6)
        tol = _grad_tol(torch.bfloat16)
        np.testing.assert_allclose(xs.grad.double().cpu().numpy(), ref_grad.cpu().numpy(), rtol=tol,
                                   atol=tol * float(ref_grad.abs().max()) * 0.05 + 1e-12)


@pytest.fixture
def cos_split_run():
    """NMSA_COS_SPLIT_RUN (tiles per workgroup of k_cos_split, read by the library at every call)"""
    import os
    prev = os.environ.get('NMSA_COS_SPLIT_RUN')

    def set_run(k):
        if k is None:
            os.environ.pop('NMSA_COS_SPLIT_RUN', None)
        else:
            os.environ['NMSA_COS_SPLIT_RUN'] = str(k)
    yield set_run
    set_run(prev)


@pytest.fixture
def cos_parts_env():
    """NMSA_COS_PARTS (read by the library at every call): 1 = k_cos_parts wherever it can run"""
    import os
    prev = os.environ.get('NMSA_COS_PARTS')

    def set_parts(v):
        if v is None:
            os.environ.pop('NMSA_COS_PARTS', None)
        else:
            os.environ['NMSA_COS_PARTS'] = str(v)
    yield set_parts
    set_parts(prev)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize('D,L,parts', [(64, 3, None), (256, 17, None), (512, 64, None), (768, 64, None),
                                       (640, 9, None), (1024, 21, None), (200, 9, None), (833, 30, None),
                                       (128, 5, 1), (320, 17, 1), (512, 64, 1)])
@pytest.mark.parametrize('hw,run', [((8, 200), 3), ((9, 444), 5), ((33, 100), 4), ((6, 1000), 64)])
def test_cos_split_runs_of_several_tiles_per_workgroup(dtype, D, L, parts, hw, run, cos_split_run, cos_parts_env):
    """the production geometry of k_cos_split on small shapes: every workgroup walks a RUN of
    tiles (at configs[4] 96 of them; by default small images get one tile per workgroup), so the
    register hand-over to the next tile inside the gradient walk, the ragged last tile inside a
    run, the clamped lane offsets past the image and the last run being shorter are all compared
    with torch's fp64 op — and bit for bit with the one-tile-per-workgroup geometry"""
    from nicr_mt_scene_analysis_amd.loss import _functional as _F
    if not _F.speculation_enabled():
        pytest.skip('NMSA_SPECULATIVE_GRAD=0: forward-written gradients are switched off')
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss, _multi
    # parts = 1: the cooperating-workgroups kernel (k_cos_parts: columns beyond 512 planes) also on
    # columns one workgroup could hold — one part, a part with idle waves (320 = 256 + 64), two parts
    cos_parts_env(parts)
    H, W = hw
    B = 2
    g = _gen(D + W + run)
    x = torch.randn((B, D, H, W), device='cuda', generator=g).to(dtype)
    lut = torch.nn.functional.normalize(torch.randn((B, L, D), device='cuda', generator=g), dim=-1)
    idx = _index_map('noise' if W == 444 else 'segments', B, H, W, L, g)
    assert _multi.cos_supported(x, lut)
    tile = 128 if dtype == torch.float32 else 256
    n_tiles = -(-H * W // tile)
    assert n_tiles >= 3 and (run >= n_tiles or n_tiles % run != 0 or (H * W) % tile != 0)
    ref_loss, ref_n, ref_grad = _cos_reference(x, idx, lut)
    results = []
    for k in (run, 1):
        cos_split_run(k)
        cos = CosineEmbeddingLoss()
        xs = x.clone().requires_grad_(True)
        before = _stats()
        loss, n = cos.lut_sum(xs, idx, lut)
        (loss / n.clamp(min=1)).backward()
        after = _stats()
        assert (after['confirmed'] - before['confirmed'], after['recomputed'] - before['recomputed']) == (1, 0)
        assert int(n) == ref_n
        np.testing.assert_allclose(float(loss), ref_loss, rtol=RTOL, atol=1e-6)
        tol = _grad_tol(dtype)
        atol = tol * float(ref_grad.abs().max()) * 0.05 + (6e-8 if dtype == torch.float16 else 1e-12)
        np.testing.assert_allclose(xs.grad.double().cpu().numpy(), ref_grad.cpu().numpy(), rtol=tol, atol=atol)
        # the recomputing launch (gradient only) walks the same runs
        xw = x.clone().requires_grad_(True)
        lw, nw = cos.lut_sum(xw, idx, lut)
        (0.5 * (lw / nw.clamp(min=1))).backward()
        np.testing.assert_allclose(xw.grad.double().cpu().numpy(), 0.5 * ref_grad.cpu().numpy(), rtol=tol, atol=atol)
        results.append((xs.grad.clone(), xw.grad.clone()))
    # per-pixel arithmetic does not depend on the geometry: identical bits
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])


def _segment_indices(B, H, W, L, g, cell=(37, 53)):
    """segment-style index map: blobs of ~cell pixels with an index in [0, L] (0 = no target)"""
    ch, cw = cell
    idx = torch.randint(0, L + 1, (B, -(-H // ch), -(-W // cw)), device='cuda', generator=g, dtype=torch.int32)
    return idx.repeat_interleave(ch, 1).repeat_interleave(cw, 2)[:, :H, :W].contiguous()


@pytest.mark.parametrize('hw,D', [((768, 1024), 512), ((765, 1020), 512), ((768, 1024), 768), ((765, 1020), 768)])
def test_cos_split_full_size_image_vs_oracle(hw, D):
    """configs[4]: ONE full-size image (D x 768 x 1024 bf16, D = 512: k_cos_split, D = 768: the
    cooperating workgroups of k_cos_parts; L = 64, segment-style indices; the
    second shape has H*W % 256 != 0: a ragged last tile at the end of the last run) through
    k_cos_split at the geometry it was built for (runs of ~12 tiles per workgroup at B = 1, of 96
    at B = 16) against the C oracle (reference loss/cos_emb.py:21-56 +
    task_helper/dense_visual_embedding.py:110-171): sum rtol 1e-5, n exact, the gradient on 4096
    sampled pixel columns (incl. the first and last pixels) element-wise to bf16 rounding, and
    EVERY element through three f64 projections: per-plane sums, per-pixel sums, a signed sum"""
    from nicr_mt_scene_analysis_amd.loss import _functional as _F
    if not _F.speculation_enabled():
        pytest.skip('NMSA_SPECULATIVE_GRAD=0')
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss, _multi
    from oracle import oracle as orc
    H, W = hw
    B, L = 1, 64
    g = _gen(H)
    x = torch.randn((B, D, H, W), device='cuda', generator=g).to(torch.bfloat16)
    lut = torch.nn.functional.normalize(torch.randn((B, L, D), device='cuda', generator=g), dim=-1)
    idx = _segment_indices(B, H, W, L, g)
    assert _multi.cos_supported(x, lut)
    cos = CosineEmbeddingLoss()
    xs = x.clone().requires_grad_(True)
    before = _stats()
    loss, n = cos.lut_sum(xs, idx, lut)
    (loss / n.clamp(min=1)).backward()
    after = _stats()
    assert (after['confirmed'] - before['confirmed'], after['recomputed'] - before['recomputed']) == (1, 0)
    want, want_n, want_grad = orc.loss_cosine_embedding(x.float().cpu().numpy(), idx.cpu().numpy(),
                                                        lut.cpu().numpy(), want_grad=True)
    assert int(n) == want_n and want_n > 0.9 * H * W
    np.testing.assert_allclose(float(loss), want, rtol=RTOL)
    want_grad = want_grad.reshape(D, H * W) / want_n             # the oracle's gradient is d sum / d x
    got = xs.grad.float().cpu().numpy().reshape(D, H * W)
    rs = np.random.default_rng(H)
    cols = np.unique(np.concatenate([rs.integers(0, H * W, 4096), [0, 1, 2, 3, 255, 256, H * W - 1, H * W - 4]]))
    scale = float(np.abs(want_grad[:, cols]).max())
    np.testing.assert_allclose(got[:, cols], want_grad[:, cols], rtol=2 ** -7, atol=2 ** -8 * 0.05 * scale)
    # pixels without a target: exactly zero, everywhere
    off = (idx.reshape(-1) == 0).cpu().numpy()
    assert off.any() and not got[:, off].any()
    # projections over all 4e8 elements (f64 accumulation; bf16 rounding errors average out)
    g64 = got.astype(np.float64)
    w64 = want_grad.astype(np.float64)
    def close(a, b, what):
        err = np.abs(a - b).max()
        ref = np.abs(b).max()
        assert err <= 2e-3 * ref, (what, err, ref)
    close(g64.sum(axis=1), w64.sum(axis=1), 'per-plane sums')
    close(np.abs(g64).sum(axis=0), np.abs(w64).sum(axis=0), 'per-pixel |.| sums')
    sign = np.where((np.arange(D)[:, None] + np.arange(H * W)[None, :]) % 3 == 0, 1.0, -0.5)
    # a signed sum cancels to ~1e-3 of its terms: bound its error by the bf16 rounding noise of
    # the 4e8 independent roundings (std <= 2^-9 |g| each), six sigma
    noise = 6 * 2 ** -9 * float(np.sqrt((w64 * w64).sum()))
    assert abs(float((g64 * sign).sum()) - float((w64 * sign).sum())) <= noise


def test_cos_split_non_finite_prediction_without_target_gets_zero_gradient():
    """a pixel without a target (index 0) gets exactly 0 — also when the prediction there is
    inf / NaN (the reference gathers the valid rows only; 0 * inf must not leak a NaN)"""
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss, _multi
    g = _gen(5)
    B, D, H, W, L = 1, 128, 4, 64, 5
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn((B, D, H, W), device='cuda', generator=g).to(dtype)
        lut = torch.nn.functional.normalize(torch.randn((B, L, D), device='cuda', generator=g), dim=-1)
        idx = _index_map('segments', B, H, W, L, g)
        idx[0, 1, 3] = 0
        idx[0, 2, 10] = 0
        x[0, 7, 1, 3] = float('inf')
        x[0, 9, 2, 10] = float('nan')
        assert _multi.cos_supported(x, lut)
        xs = x.clone().requires_grad_(True)
        loss, n = CosineEmbeddingLoss().lut_sum(xs, idx, lut)
        (loss / n.clamp(min=1)).backward()
        assert torch.isfinite(loss) and torch.isfinite(xs.grad).all()
        assert not xs.grad[0, :, 1, 3].any() and not xs.grad[0, :, 2, 10].any()
        clean = torch.nan_to_num(x, nan=0.0, posinf=0.0)
        _, _, ref_grad = _cos_reference(clean, idx, lut)
        tol = _grad_tol(dtype)
        np.testing.assert_allclose(xs.grad.double().cpu().numpy(), ref_grad.cpu().numpy(), rtol=tol,
                                   atol=tol * float(ref_grad.abs().max()) * 0.05 + 1e-12)


def test_ce_bf16_far_classes_keep_their_gradient():
    """bf16 logits with classes 12 ... 20 below the maximum: their exponentials (1e-9 ... 6e-6)
    are kept as fp16 in the register tile — scaled, so they are fp16 normals; the gradient of
    every class equals torch's fp32 softmax gradient rounded to bf16 (element-wise)"""
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    B, C, H, W = 1, 40, 8, 64
    g = _gen(40)
    gaps = 12.0 + 8.0 * torch.rand((B, C, H, W), device='cuda', generator=g)
    x = -gaps
    top = torch.randint(0, C, (B, 1, H, W), device='cuda', generator=g)
    x.scatter_(1, top, 0.0)
    x = (x + torch.randn((B, 1, H, W), device='cuda', generator=g)).to(torch.bfloat16)
    t = torch.randint(1, C + 1, (B, H, W), device='cuda', generator=g).to(torch.uint8)
    n = F_.count_u8(t, 1, C)
    xs = x.clone().requires_grad_(True)
    loss, n_el, _ = F_.cross_entropy_sum(xs, t, None, 0.0, expected_scale=F_.expected_scale(n))
    (loss / n_el).backward()
    xr = x.float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t.long() - 1, reduction='sum')
    (ref / int(n)).backward()
    want = xr.grad.to(torch.bfloat16).float()
    got = xs.grad.float()
    np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL)
    # one bf16 ulp (2^-8 relative) — rounding of a value computed two ways —, no absolute slack
    # beyond the smallest gradients that exist here (1e-9 / n)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=2 ** -7, atol=1e-9 / int(n) * 0.1)
    small = (xr.grad.abs() < 1e-5 / int(n)) & (xr.grad != 0)
    assert int(small.sum()) > 1000 and bool((got[small] != 0).all())


@pytest.fixture
def lib_env():
    """environment variables the library reads at every call; restored afterwards"""
    import os
    saved = {}

    def set_env(**kw):
        for k, v in kw.items():
            if k not in saved:
                saved[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
    yield set_env
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _cos_parts_case(B, D, H, W, L, seed):
    g = _gen(seed)
    x = torch.randn((B, D, H, W), device='cuda', generator=g).to(torch.bfloat16)
    lut = torch.nn.functional.normalize(torch.randn((B, L, D), device='cuda', generator=g), dim=-1)
    idx = _segment_indices(B, H, W, L, g)
    return x, lut, idx


def test_cos_parts_grid_follows_the_device_geometry(lib_env):
    """the cooperating-workgroup cosine kernel sizes its grid from the compute units the HIP
    runtime reports (csrc/api.hip device_geometry), not from a literal 256: with a quarter of the
    device assumed (NMSA_ASSUME_CUS, e.g. a partition or a CU mask) the grid shrinks — fewer,
    longer runs of tiles — and loss, count and every gradient bit stay the same"""
    from nicr_mt_scene_analysis_amd import _lib as L_
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss, _multi, check_loss_status
    from nicr_mt_scene_analysis_amd.loss import _functional as _F
    if not _F.speculation_enabled():
        pytest.skip('NMSA_SPECULATIVE_GRAD=0')
    lib_env(NMSA_ASSUME_CUS=None, NMSA_ASSUME_XCDS=None)
    cus, xcds, lds = L_.device_geometry()
    props = torch.cuda.get_device_properties(0)
    assert cus == props.multi_processor_count and xcds >= 1 and lds >= 64 * 1024
    B, D, H, W, L = 2, 768, 128, 512, 64
    x, lut, idx = _cos_parts_case(B, D, H, W, L, 11)
    assert _multi.cos_supported(x, lut)
    ws_bytes = lambda: L_.lib().nmsa_loss_cos_emb_fwd_grad_workspace_bytes(B, D, H, W, L)
    ref_loss, ref_n, ref_grad = _cos_reference(x, idx, lut)
    out = []
    for assumed in (None, max(1, cus // 4)):
        lib_env(NMSA_ASSUME_CUS=assumed)
        assert L_.device_geometry()[0] == (assumed or cus)
        out.append([ws_bytes()])
        for scale in (1.0, 0.5):               # confirmed forward-written gradient / the recomputing launch
            xs = x.clone().requires_grad_(True)
            loss, n = CosineEmbeddingLoss().lut_sum(xs, idx, lut)
            (scale * (loss / n.clamp(min=1))).backward()
            assert int(n) == ref_n
            np.testing.assert_allclose(float(loss), ref_loss, rtol=RTOL)
            out[-1] += [float(loss), xs.grad.clone()]
    check_loss_status()                         # nothing gave up, nothing out of range
    (ws_a, loss_a, g_a, _, gh_a), (ws_b, loss_b, g_b, _, gh_b) = out
    assert ws_b < ws_a, 'a smaller device must get a smaller grid (fewer exchange slots)'
    assert abs(loss_a - loss_b) <= 1e-9 * abs(loss_a)        # (f64 sums of the same f32 tile sums, other grouping)
    assert torch.equal(g_a, g_b) and torch.equal(gh_a, gh_b)
    tol = _grad_tol(torch.bfloat16)
    np.testing.assert_allclose(g_b.double().cpu().numpy(), ref_grad.cpu().numpy(), rtol=tol,
                               atol=tol * float(ref_grad.abs().max()) * 0.05 + 1e-12)


def test_cos_parts_stranded_partners_take_the_fallback(lib_env):
    """a grid that is NOT resident as a whole: four times the device's compute units assumed and
    the parts of a group a third of the grid apart in the workgroup order (NMSA_COS_PARTS_ORDER=1)
    instead of neighbours — the first workgroups fill the device and wait for partners that cannot
    be dispatched.  Their waits time out (20 ms here), the call's gave-up word gates the two-walk
    kernels in ON THE DEVICE, and the call returns the right loss, count and gradient (forward-
    written and recomputed); the status word reports it as a warning"""
    from nicr_mt_scene_analysis_amd import _lib as L_
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss, _multi, check_loss_status
    from nicr_mt_scene_analysis_amd.loss import _functional as _F
    if not _F.speculation_enabled():
        pytest.skip('NMSA_SPECULATIVE_GRAD=0')
    import warnings
    lib_env(NMSA_ASSUME_CUS=None)
    cus = L_.device_geometry()[0]
    B, D, L = 2, 768, 64
    H, W = 192, 512 * max(1, cus // 256)        # 768 tiles on a whole MI355X: more than the 682 groups asked for
    x, lut, idx = _cos_parts_case(B, D, H, W, L, 23)
    assert _multi.cos_supported(x, lut)
    ref_loss, ref_n, ref_grad = _cos_reference(x, idx, lut)
    tol = _grad_tol(torch.bfloat16)
    atol = tol * float(ref_grad.abs().max()) * 0.05 + 1e-12
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        check_loss_status()                     # clean before
    lib_env(NMSA_ASSUME_CUS=4 * cus, NMSA_COS_PARTS_ORDER=1, NMSA_COS_PARTS_TIMEOUT_MS=20)
    for scale in (1.0, 0.5):
        xs = x.clone().requires_grad_(True)
        loss, n = CosineEmbeddingLoss().lut_sum(xs, idx, lut)
        (scale * (loss / n.clamp(min=1))).backward()
        torch.cuda.synchronize()
        assert int(n) == ref_n
        assert np.isfinite(float(loss))
        np.testing.assert_allclose(float(loss), ref_loss, rtol=RTOL)
        assert torch.isfinite(xs.grad).all()
        np.testing.assert_allclose(xs.grad.double().cpu().numpy(), scale * ref_grad.cpu().numpy(), rtol=tol, atol=atol)
        with pytest.warns(RuntimeWarning, match='k_cos_parts'):
            check_loss_status()
    # the neighbours order survives the same over-subscription (partners are dispatched together) —
    # and whatever happens, the result is right
    lib_env(NMSA_COS_PARTS_ORDER=None)
    xs = x.clone().requires_grad_(True)
    loss, n = CosineEmbeddingLoss().lut_sum(xs, idx, lut)
    (loss / n.clamp(min=1)).backward()
    np.testing.assert_allclose(float(loss), ref_loss, rtol=RTOL)
    np.testing.assert_allclose(xs.grad.double().cpu().numpy(), ref_grad.cpu().numpy(), rtol=tol, atol=atol)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        check_loss_status()
