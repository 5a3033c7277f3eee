"""
GPU tier: HIP loss kernels (forward + backward through autograd) against
 * golden values + autograd gradients from the reference's loss classes with the
   task helpers' masking (tests/golden/loss_cases.npz, made by oracle/gen_golden.py),
 * torch's own fp32 ops on the same tensors at full size (floating-point kernels keep a
   torch reference; tolerance: relative 1e-5 on the scalars, as north_star states),
 * the properties the reference's tests/test_loss_functions.py:37-170 assert
   (loss != 0, n_elements, one result per scale).
"""
import numpy as np
import pytest
import torch

from _golden import load

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def dev(a, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.requires_grad_(True) if grad else t


def test_ce_vs_reference_golden():
    from nicr_mt_scene_analysis_amd.loss import CrossEntropyLossSemantic
    g = load('loss_cases')
    w = dev(g['in_class_weights'])
    tgt = dev(g['in_semantic_target'])
    for name, kw in (('plain', {}), ('weighted', dict(weights=w)),
                     ('smooth', dict(weights=w, label_smoothing=0.25)),
                     ('smooth_nw', dict(label_smoothing=0.5)),
                     ('wred', dict(weights=w, weighted_reduction=True))):
        x = dev(g['in_semantic_logits'], grad=True)
        (loss, n), = CrossEntropyLossSemantic(**kw)([x], [tgt])
        np.testing.assert_allclose(float(loss), g[f'ce_{name}__loss'], rtol=RTOL)
        assert int(n) == int(g[f'ce_{name}__n'])
        loss.backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[f'ce_{name}__grad'],
                                   rtol=1e-4, atol=1e-6)


def test_instance_losses_vs_reference_golden():
    from nicr_mt_scene_analysis_amd.loss import L1Loss, MSELoss, VonMisesLossBiternion
    g = load('loss_cases')
    for kind, cls in (('mse', MSELoss), ('l1', L1Loss)):
        x = dev(g['in_center_pred'], grad=True)
        loss, n = cls().masked_sum(x, dev(g['in_center_target']), dev(g['in_center_mask']))
        np.testing.assert_allclose(float(loss), g[f'center_{kind}__loss'], rtol=RTOL)
        assert int(n) == int(g[f'center_{kind}__n_mask'])
        loss.backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[f'center_{kind}__grad'],
                                   rtol=1e-5, atol=1e-7)
    x = dev(g['in_offset_pred'], grad=True)
    loss, n = L1Loss().masked_sum(x, dev(g['in_offset_target']), dev(g['in_offset_mask']))
    np.testing.assert_allclose(float(loss), g['offset_l1__loss'], rtol=RTOL)
    assert int(n) == int(g['offset_l1__n_mask'])
    loss.backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g['offset_l1__grad'], rtol=1e-5, atol=1e-7)

    for kappa in (1.0, 2.5):
        x = dev(g['in_orientation_pred'], grad=True)
        loss, n = VonMisesLossBiternion(kappa=kappa).masked_sum(
            x, dev(g['in_orientation_target']), dev(g['in_orientation_mask']))
        np.testing.assert_allclose(float(loss), g[f'vonmises_{kappa}__loss'], rtol=RTOL)
        assert int(n) == int(g[f'vonmises_{kappa}__n'])
        loss.backward()
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[f'vonmises_{kappa}__grad'],
                                   rtol=1e-5, atol=1e-7)


def test_cos_emb_vs_reference_golden():
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss
    g = load('loss_cases')
    x = dev(g['in_embedding_pred'], grad=True)
    loss, n = CosineEmbeddingLoss().lut_sum(x, dev(g['in_embedding_indices']),
                                            dev(g['in_embedding_lut']))
    np.testing.assert_allclose(float(loss), g['cos_emb__loss'], rtol=RTOL)
    assert int(n) == int(g['cos_emb__n'])
    loss.backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g['cos_emb__grad'], rtol=1e-4, atol=1e-6)
    # rows API of the reference (_compute_loss(input_[N,D], target[N,D]))
    rows = torch.randn((300, 16), device='cuda', requires_grad=True)
    tg = torch.randn((300, 16), device='cuda')
    (l2, n2), = CosineEmbeddingLoss()([rows], [tg])
    ref = torch.nn.functional.cosine_embedding_loss(rows.detach(), tg, torch.ones(300, device='cuda'),
                                                    reduction='sum')
    np.testing.assert_allclose(float(l2), float(ref), rtol=RTOL)
    assert n2 == 300
    l2.backward()
    assert rows.grad.shape == rows.shape


# ---- reference tests/test_loss_functions.py restated ------------------------------------
def _scaled(t, use_scales):
    ts = [t]
    if use_scales:
        h, w = t.shape[-2:]
        for e in range(3):
            s = 2 ** (e + 1)
            ts.append(t[..., :h // s, :w // s].contiguous())
    return ts


@pytest.mark.parametrize('batch_size', (1, 8))
@pytest.mark.parametrize('use_weights', (False, True))
@pytest.mark.parametrize('use_scales', (False, True))
@pytest.mark.parametrize('label_smoothing', (0.0, 0.5))
def test_ce_full_size(batch_size, use_weights, use_scales, label_smoothing):
    from nicr_mt_scene_analysis_amd.loss import CrossEntropyLossSemantic
    h, w, n_classes = 480, 640, 40
    g = torch.Generator(device='cuda').manual_seed(batch_size)
    x = torch.rand((batch_size, n_classes, h, w), device='cuda', generator=g)
    t = (torch.rand((batch_size, h, w), device='cuda', generator=g) * n_classes).long()
    inputs, targets = _scaled(x, use_scales), _scaled(t, use_scales)
    weights = torch.rand(n_classes, device='cuda', generator=g) if use_weights else None
    fn = CrossEntropyLossSemantic(weights=weights, label_smoothing=label_smoothing)
    outs = fn(inputs, targets)
    assert len(outs) == len(targets)
    ref_fn = torch.nn.CrossEntropyLoss(weight=None if weights is None else weights.double(),
                                       reduction='sum', ignore_index=-1,
                                       label_smoothing=label_smoothing)
    for inp, tgt, (loss, n) in zip(inputs, targets, outs):
        assert loss != 0
        assert n == (tgt > 0).sum()
        ref = ref_fn(inp.double(), tgt - 1)            # fp64 torch reference on the device
        np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL)


@pytest.mark.parametrize('cls_name', ('L1Loss', 'MSELoss'))
@pytest.mark.parametrize('batch_size', (1, 8))
@pytest.mark.parametrize('use_scales', (False, True))
@pytest.mark.parametrize('reduction', ('none', 'mean', 'sum'))
def test_l1_mse_full_size(cls_name, batch_size, use_scales, reduction):
    from nicr_mt_scene_analysis_amd import loss as L_
    h, w = 480, 640
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.rand((batch_size, 2, h, w), device='cuda', generator=g)
    y = torch.rand((batch_size, 2, h, w), device='cuda', generator=g)
    inputs, targets = _scaled(x, use_scales), _scaled(y, use_scales)
    outs = getattr(L_, cls_name)(reduction)(inputs, targets)
    assert len(outs) == len(targets)
    for inp, tgt, (loss, n) in zip(inputs, targets, outs):
        d = (inp.double() - tgt.double())
        e = d * d if cls_name == 'MSELoss' else d.abs()
        if reduction == 'none':
            assert loss.shape == inp.shape and n == inp.numel()
        elif reduction == 'mean':
            assert loss.shape == () and loss != 0 and n == 1
        else:
            assert loss.shape == () and loss != 0
            b, c, h_, w_ = inp.shape
            assert n == b * h_ * w_
            np.testing.assert_allclose(float(loss), float(e.mean(dim=1).sum()), rtol=RTOL)


@pytest.mark.parametrize('batch_size', (1, 8))
@pytest.mark.parametrize('with_random_masks', (False, True))
def test_vonmises_full_size(batch_size, with_random_masks):
    from nicr_mt_scene_analysis_amd.loss import VonMisesLossBiternion
    h, w = 480, 640
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.rand((batch_size, 2, h, w), device='cuda', generator=g)
    y = torch.rand((batch_size, 2, h, w), device='cuda', generator=g)
    rows_x = x.permute(0, 2, 3, 1).reshape(-1, 2)
    rows_y = y.permute(0, 2, 3, 1).reshape(-1, 2)
    mask = None
    if with_random_masks:
        mask = torch.rand((batch_size, h, w), device='cuda', generator=g) > 0.5
        rows_x, rows_y = rows_x[mask.flatten()], rows_y[mask.flatten()]
    fn = VonMisesLossBiternion()
    (loss_rows, n_rows), = fn([rows_x], [rows_y])                 # the reference's rows API
    loss_planar, n_planar = fn.masked_sum(x, y, mask)             # fused masking
    ref = (1 - torch.exp((rows_x.double() * rows_y.double()).sum(1) - 1)).sum()
    assert loss_rows != 0
    assert n_rows == rows_x.shape[0] and int(n_planar) == rows_x.shape[0]
    np.testing.assert_allclose(float(loss_rows), float(ref), rtol=RTOL)
    np.testing.assert_allclose(float(loss_planar), float(ref), rtol=RTOL)
    with pytest.raises(ValueError):
        fn([x], [y])                                              # 4-D input is rejected


def test_bf16_predictions_and_grads():
    """cfg3 dtype: bf16 predictions, fp32 accumulate, bf16 gradients."""
    from nicr_mt_scene_analysis_amd.loss import CrossEntropyLossSemantic, L1Loss
    g = torch.Generator(device='cuda').manual_seed(7)
    x = (torch.randn((2, 40, 96, 128), device='cuda', generator=g) * 3).to(torch.bfloat16)
    t = torch.randint(0, 41, (2, 96, 128), device='cuda', generator=g).to(torch.uint8)
    w = torch.rand(40, device='cuda', generator=g) + 0.5
    xb = x.clone().requires_grad_(True)
    (loss, n), = CrossEntropyLossSemantic(weights=w)([xb], [t])
    xr = x.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t.long() - 1, weight=w.double(), reduction='sum',
                                            ignore_index=-1)
    np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL)
    (loss / n).backward()
    (ref / n).backward()
    assert xb.grad.dtype == torch.bfloat16
    np.testing.assert_allclose(xb.grad.float().cpu().numpy(), xr.grad.float().cpu().numpy(),
                               rtol=1e-2, atol=1e-7)                # bf16 rounding of the output
    p = torch.randn((2, 2, 96, 128), device='cuda', generator=g).to(torch.bfloat16)
    y = torch.randn((2, 2, 96, 128), device='cuda', generator=g)
    m = torch.rand((2, 96, 128), device='cuda', generator=g) > 0.5
    loss, n = L1Loss().masked_sum(p, y, m)
    ref = ((p.double() * m.unsqueeze(1)) - y.double()).abs().mean(dim=1).sum()
    np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL)
    assert int(n) == int(m.sum())


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_center_focal_loss_extension_vs_torch(dtype):
    """`CenterFocalLoss` is an extension (the reference has MSE / L1 only): parity unpinned, checked
    against a plain PyTorch fp32 implementation of the documented formula, values and gradient"""
    from nicr_mt_scene_analysis_amd.loss import CenterFocalLoss
    g = torch.Generator(device='cuda').manual_seed(3)
    B, H, W = 3, 37, 53
    pred = torch.rand((B, H, W), device='cuda', generator=g)
    pred[0, 0, :4] = torch.tensor([0.0, 1.0, 1e-6, 1 - 1e-6], device='cuda')      # clamped ends
    target = torch.rand((B, H, W), device='cuda', generator=g) ** 4
    target[torch.rand((B, H, W), device='cuda', generator=g) < 0.02] = 1.0
    mask = torch.rand((B, H, W), device='cuda', generator=g) < 0.7
    x = pred.to(dtype).requires_grad_(True)
    loss, n = CenterFocalLoss().masked_sum(x, target, mask)
    loss.backward()

    xr = x.detach().float().requires_grad_(True)
    p = xr.clamp(1e-4, 1 - 1e-4)
    pos = target == 1
    val = torch.where(pos, -(1 - p) ** 2 * torch.log(p), -(1 - target) ** 4 * p ** 2 * torch.log(1 - p))
    ref = (val * mask).sum()
    ref.backward()
    assert int(n) == max(int((pos & mask).sum()), 1)
    torch.testing.assert_close(loss.detach(), ref.detach(), rtol=2e-5, atol=1e-5)
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(x.grad.float(), xr.grad, **tol)


def test_instance_task_helper_accepts_focal_center_loss():
    from nicr_mt_scene_analysis_amd.task_helper import InstanceTaskHelper
    h = InstanceTaskHelper(semantic_n_classes=5, semantic_classes_is_thing=(False, True, True, False, True),
                           loss_name_instance_center='focal', disable_multiscale_supervision=True)
    h.initialize(torch.device('cuda'))
    B, H, W = 2, 16, 24
    g = torch.Generator(device='cuda').manual_seed(1)
    center = torch.rand((B, 1, H, W), device='cuda', generator=g, requires_grad=True)
    offset = torch.zeros((B, 2, H, W), device='cuda', requires_grad=True)
    tgt = torch.rand((B, H, W), device='cuda', generator=g)
    tgt[:, 4, 4] = 1.0
    batch = {'instance_center': tgt, 'instance_center_mask': torch.ones((B, H, W), dtype=torch.bool, device='cuda'),
             'instance_offset': torch.zeros((B, 2, H, W), device='cuda'),
             'instance_foreground': torch.ones((B, H, W), dtype=torch.bool, device='cuda')}
    losses, _ = h.training_step(batch, 0, {'instance_output': (center, offset), 'instance_side_outputs': (None,)})
    total = losses['instance_center_total_loss']
    total.backward()
    assert torch.isfinite(total) and float(total) > 0 and center.grad.abs().sum() > 0


def test_off_path_reductions_on_device():
    """'mean' / 'none' reductions, 2-D rows and labelled cosine pairs (reference loss/mse.py:21-41,
    l1.py:21-41, cos_emb.py:29-56): results on the device, equal to torch's formulas"""
    from nicr_mt_scene_analysis_amd.loss import (CosineEmbeddingLoss, L1Loss, MSELoss,
                                                 VonMisesLossBiternion)
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.rand((2, 2, 5, 7), device='cuda', generator=g)
    y = torch.rand((2, 2, 5, 7), device='cuda', generator=g)
    for cls, f in ((MSELoss, lambda d: d * d), (L1Loss, torch.abs)):
        (l, n), = cls('none')([x], [y])
        assert l.shape == x.shape and n == x.numel() and torch.allclose(l, f(x - y))
        (l, n), = cls('mean')([x], [y])
        assert n == 1 and float(l) == pytest.approx(float(f(x - y).mean()), rel=1e-6)
        (l, n), = cls('sum')([x], [y])
        assert n == 2 * 5 * 7 and float(l) == pytest.approx(float(f(x - y).mean(1).sum()), rel=1e-6)
        x2, y2 = x.reshape(-1, 10), y.reshape(-1, 10)          # [N, C] rows
        (l, n), = cls('sum')([x2], [y2])
        assert n == x2.shape[0] and float(l) == pytest.approx(float(f(x2 - y2).mean(1).sum()), rel=1e-6)
    rows = torch.rand((9, 2), device='cuda', generator=g)
    tg = torch.rand((9, 2), device='cuda', generator=g)
    (l, n), = VonMisesLossBiternion()([rows], [tg])
    assert n == 9 and float(l) == pytest.approx(float((1 - torch.exp((rows * tg).sum(1) - 1)).sum()), rel=1e-6)
    (l, n), = VonMisesLossBiternion(reduction='none')([rows], [tg])
    assert l.shape == (9, 1) and n == 9
    minus = -torch.ones(9, device='cuda')
    l, n = CosineEmbeddingLoss()._compute_loss(rows, tg, target_similarity=minus)
    want = torch.nn.functional.cosine_embedding_loss(rows, tg, minus, reduction='none')
    assert n == 9 and float(l) == pytest.approx(float(want.sum()), rel=1e-6)
    want = torch.nn.functional.cosine_embedding_loss(rows, tg, torch.ones(9, device='cuda'),
                                                     reduction='none')
    (l, n), = CosineEmbeddingLoss('mean')([rows], [tg])
    assert n == 1 and float(l) == pytest.approx(float(want.mean()), rel=1e-5)
    (l, n), = CosineEmbeddingLoss('none')([rows], [tg])
    assert n == rows.numel() and torch.allclose(l, want)
    # a target that asks for a gradient gets one (the reference's op differentiates both sides)
    tg2 = tg.clone().requires_grad_(True)
    (l, n), = CosineEmbeddingLoss()([rows], [tg2])
    l.backward()
    assert tg2.grad is not None and float(tg2.grad.abs().sum()) > 0


@pytest.mark.parametrize('as_bf16', [False, True])
def test_cos_emb_large_dims_vs_reference_golden(as_bf16):
    """BASELINE configs[4] sizes: D = 512 (one 131 KB LDS chunk), D = 768 (197 KB as fp32 ->
    two chunks), ragged P, D % chunk != 0, and a LUT too tall for LDS (generic kernel);
    forward and gradient against the reference's CosineEmbeddingLoss + autograd.  bf16: the
    golden's prediction holds bf16-representable values, so the bf16 tensor is the same
    input and the loss must agree to 1e-5; its gradient is rounded to bf16 once (2^-8)."""
    from _golden import cos_emb_large_cases, check_cos_emb_large_grad
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss
    ran = 0
    for name, p, inp, g in cos_emb_large_cases():
        if as_bf16 and not p['bf16']:
            continue
        x = dev(inp['embedding_pred'])
        if as_bf16:
            x = x.to(torch.bfloat16)
            assert torch.equal(x.float().cpu(), torch.from_numpy(inp['embedding_pred']))
        x.requires_grad_(True)
        loss, n = CosineEmbeddingLoss().lut_sum(x, dev(inp['embedding_indices']),
                                                dev(inp['embedding_lut']))
        np.testing.assert_allclose(float(loss), g[f'{name}__loss'], rtol=RTOL, err_msg=name)
        assert int(n) == int(g[f'{name}__n']), name
        loss.backward()
        grad = x.grad.float().cpu().numpy()
        if as_bf16:
            check_cos_emb_large_grad(name, p, g, grad, rtol=2 ** -7, atol=1e-6, sums_rtol=2e-3)
        else:
            check_cos_emb_large_grad(name, p, g, grad, rtol=1e-4, atol=1e-7, sums_rtol=1e-5)
        ran += 1
    assert ran >= 3


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,label_smoothing', [(256, 0.0), (549, 0.1), (1000, 0.0)])
def test_ce_with_more_than_255_classes_vs_torch_fp64(dtype, C, label_smoothing):
    """more classes than a uint8 label holds (the reference's CrossEntropyLoss takes any number,
    ce.py:40-68; e.g. ScanNet's 549 classes): the labels travel as int16 through the two-kernel
    path — sum, n, weight sum and gradient against torch's fp64 op, weighted reduction included"""
    from nicr_mt_scene_analysis_amd.loss import CrossEntropyLossSemantic, check_loss_status
    g = torch.Generator(device='cuda').manual_seed(C)
    B, H, W = 2, 9, 20
    x = (torch.randn((B, C, H, W), device='cuda', generator=g) * 3).to(dtype)
    t = torch.randint(0, C + 1, (B, H, W), device='cuda', generator=g)
    t[0, 0, :4] = torch.tensor([C, C - 1, 256, 0], device='cuda')        # labels beyond uint8, and void
    w = torch.rand(C, device='cuda', generator=g) + 0.5
    xr = x.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t - 1, weight=w.double(), reduction='sum', ignore_index=-1,
                                            label_smoothing=label_smoothing)
    n_ref = int((t != 0).sum())
    (ref / n_ref).backward()
    for target in (t, t.to(torch.int16), t.to(torch.int32)):
        xs = x.clone().requires_grad_(True)
        (loss, n), = CrossEntropyLossSemantic(weights=w, label_smoothing=label_smoothing)([xs], [target])
        (loss / n).backward()
        assert int(n) == n_ref
        np.testing.assert_allclose(float(loss), float(ref), rtol=1e-5)
        tol = 2e-5 if dtype == torch.float32 else 2 ** -7
        # (with label smoothing the target class' element (a + bsum) p_t - a - b_t cancels in fp32:
        # a handful of elements are off by up to 5e-5 of the largest gradient entry)
        np.testing.assert_allclose(xs.grad.double().cpu().numpy(), xr.grad.cpu().numpy(), rtol=tol,
                                   atol=max(tol * 0.05, 5e-5) * float(xr.grad.abs().max()))
    (lw, nw), = CrossEntropyLossSemantic(weights=w, weighted_reduction=True)([x], [t])
    ref_w = torch.nn.functional.cross_entropy(x.double(), t - 1, weight=w.double(), reduction='mean', ignore_index=-1)
    np.testing.assert_allclose(float(lw), float(ref_w), rtol=1e-5)
    check_loss_status()


def test_ce_rejects_more_than_4096_classes_and_flags_bad_labels():
    from nicr_mt_scene_analysis_amd.loss import CrossEntropyLossSemantic, check_loss_status
    x = torch.zeros((1, 4097, 2, 2), device='cuda')
    with pytest.raises(ValueError, match='4096'):
        CrossEntropyLossSemantic()([x], [torch.zeros((1, 2, 2), dtype=torch.int64, device='cuda')])
    check_loss_status()                                  # clean so far
    # a label beyond C in the int16 path is flagged like in the uint8 path
    x = torch.randn((1, 300, 4, 8), device='cuda')
    t = torch.randint(0, 301, (1, 4, 8), device='cuda')
    t[0, 2, 2] = 301
    try:
        CrossEntropyLossSemantic()([x], [t])
        raised_at_call = False
    except IndexError:
        raised_at_call = True
    if not raised_at_call:
        with pytest.raises(IndexError, match='out of range'):
            check_loss_status()
    check_loss_status()
    x = torch.randn((1, 5, 4, 8), device='cuda')
    t = torch.randint(0, 6, (1, 4, 8), device='cuda')
    (l0, n0), = CrossEntropyLossSemantic()([x], [t])
    check_loss_status()
    t[0, 0, 0] = 300                                     # wraps to 44 as uint8: must be flagged
    t[0, 1, 1] = 7                                       # > C
    try:
        CrossEntropyLossSemantic()([x], [t])
        raised_at_call = False
    except IndexError:                                   # NMSA_CHECK_STATUS=1 checks per call
        raised_at_call = True
    if not raised_at_call:
        with pytest.raises(IndexError, match='out of range'):
            check_loss_status()
    check_loss_status()                                  # the word was cleared


def test_loss_forms_vs_golden(oracle, monkeypatch):
    """the forms of MSELoss / L1Loss / CosineEmbeddingLoss no task helper calls — reduction none /
    sum / mean on 2-D, 3-D and 4-D inputs (reference mse.py:21-41, l1.py:21-41), labelled cosine
    pairs (cos_emb.py:21-56) — against the reference-run fixture and the fp64 oracle, through the
    library (ATen's cosine op and the torch subtraction of the old fallback are made to raise)"""
    from nicr_mt_scene_analysis_amd import loss as L_
    from nicr_mt_scene_analysis_amd.loss import _elementwise
    g = load('loss_forms')

    def no_torch(*a, **k):
        raise AssertionError('this form must run in the library')
    monkeypatch.setattr(torch.nn.functional, 'cosine_embedding_loss', no_torch)
    monkeypatch.setattr(_elementwise._ElementwiseLoss, '_pointwise', no_torch)
    for kind, cls in (('mse', L_.MSELoss), ('l1', L_.L1Loss)):
        for rk in ('r2', 'r3', 'r4'):
            for red in ('none', 'sum', 'mean'):
                key = f'{kind}_{rk}_{red}'
                x = dev(g[f'{rk}__x']).requires_grad_(True)
                (loss, n), = cls(reduction=red)([x], [dev(g[f'{rk}__t'])])
                ((loss * dev(g[f'{rk}__w'])).sum() if red == 'none' else loss).backward()
                np.testing.assert_allclose(loss.detach().cpu().numpy(), g[key + '__loss'], rtol=1e-5,
                                           atol=1e-7, err_msg=key)
                assert int(n) == int(g[key + '__n']), key
                np.testing.assert_allclose(x.grad.cpu().numpy(), g[key + '__grad'], rtol=1e-5, atol=1e-7,
                                           err_msg=key)
                o_loss, o_n, o_grad = oracle.loss_elementwise_form(g[f'{rk}__x'], g[f'{rk}__t'], kind, red,
                                                                   g[f'{rk}__w'])
                np.testing.assert_allclose(loss.detach().cpu().numpy(), o_loss, rtol=1e-5, atol=1e-7)
                np.testing.assert_allclose(x.grad.cpu().numpy(), o_grad, rtol=1e-5, atol=1e-7)
    for lab in ('labelled', 'plain'):
        for red in ('none', 'sum', 'mean'):
            key = f'cos_{lab}_{red}'
            x = dev(g['cos__x']).requires_grad_(True)
            fn = L_.CosineEmbeddingLoss(reduction=red)
            args = (x, dev(g['cos__t'])) + ((dev(g['cos__labels']),) if lab == 'labelled' else ())
            loss, n = fn._compute_loss(*args)
            ((loss * dev(g['cos__w'])).sum() if red == 'none' else loss).backward()
            np.testing.assert_allclose(loss.detach().cpu().numpy(), g[key + '__loss'], rtol=1e-5, atol=1e-6,
                                       err_msg=key)
            assert int(n) == int(g[key + '__n']), key
            np.testing.assert_allclose(x.grad.cpu().numpy(), g[key + '__grad'], rtol=1e-4, atol=1e-6,
                                       err_msg=key)


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_loss_forms_16_bit_inputs(dtype):
    """the same forms on 16-bit predictions: op-math in fp32, one rounding to the promoted type
    (what ATen does), gradients in the prediction's type — against torch's own ops on the device"""
    from nicr_mt_scene_analysis_amd import loss as L_
    g = torch.Generator(device='cuda').manual_seed(3)
    x0 = torch.randn((5, 3, 9, 7), device='cuda', generator=g).to(dtype)
    for t_dtype in (torch.float32, dtype):
        t = torch.randn((5, 3, 9, 7), device='cuda', generator=g).to(t_dtype)
        w = torch.randn((5, 3, 9, 7), device='cuda', generator=g)
        for kind, cls, ref in (('mse', L_.MSELoss, torch.nn.functional.mse_loss),
                               ('l1', L_.L1Loss, torch.nn.functional.l1_loss)):
            x = x0.clone().requires_grad_(True)
            (loss, n), = cls(reduction='none')([x], [t])
            xr = x0.clone().requires_grad_(True)
            want = ref(xr, t, reduction='none') if t_dtype == dtype else ref(xr.float(), t, reduction='none')
            assert loss.dtype == want.dtype and n == x0.numel()
            # (ATen's 16-bit kernel rounds the difference before it squares it: two roundings)
            torch.testing.assert_close(loss, want, rtol=2e-2 if t_dtype == dtype else 1e-6, atol=1e-3 if t_dtype == dtype else 1e-6)
            (loss.float() * w).sum().backward()
            (want.float() * w).sum().backward()
            torch.testing.assert_close(x.grad.float(), xr.grad.float(), rtol=2e-2, atol=1e-3)
    rows = torch.randn((33, 48), device='cuda', generator=g).to(dtype)
    tgt = torch.randn((33, 48), device='cuda', generator=g)
    lab = torch.where(torch.rand(33, device='cuda', generator=g) < 0.5, 1.0, -1.0)
    x = rows.clone().requires_grad_(True)
    loss, n = L_.CosineEmbeddingLoss(reduction='sum')._compute_loss(x, tgt, lab)
    loss.backward()
    xr = rows.float().clone().requires_grad_(True)
    want = torch.nn.functional.cosine_embedding_loss(xr, tgt, lab, reduction='sum')
    want.backward()
    assert n == 33
    torch.testing.assert_close(loss.float(), want, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.grad.float(), xr.grad, rtol=2e-2, atol=2e-3)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_vonmises_rows_none_vs_torch(dtype, monkeypatch):
    """VonMisesLossBiternion(reduction='none') on biternion rows (reference vonmises.py:27-51): the
    row kernel against the reference's three torch lines, loss [n, 1] and gradient"""
    from nicr_mt_scene_analysis_amd.loss import VonMisesLossBiternion
    g = torch.Generator(device='cuda').manual_seed(11)
    ang = torch.rand((1000, 2), device='cuda', generator=g) * 6.28
    rows = torch.stack([torch.cos(ang[:, 0]), torch.sin(ang[:, 0])], 1).to(dtype)
    tgt = torch.stack([torch.cos(ang[:, 1]), torch.sin(ang[:, 1])], 1)
    w = torch.randn((1000, 1), device='cuda', generator=g)
    for kappa in (1.0, 2.5):
        xr = rows.float().clone().requires_grad_(True)
        want = 1 - torch.exp(kappa * ((xr * tgt).sum(dim=1, keepdim=True) - 1))
        (want * w).sum().backward()
        monkeypatch.setattr(torch, 'exp', None)          # the library's path calls no torch.exp
        x = rows.clone().requires_grad_(True)
        (loss, n), = VonMisesLossBiternion(kappa=kappa, reduction='none')([x], [tgt])
        monkeypatch.undo()
        assert loss.shape == (1000, 1) and n == 1000 and loss.dtype == torch.float32
        (loss * w).sum().backward()
        torch.testing.assert_close(loss, want.detach(), rtol=1e-5, atol=1e-6)
        tol = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-3)
        torch.testing.assert_close(x.grad.float(), xr.grad, **tol)
