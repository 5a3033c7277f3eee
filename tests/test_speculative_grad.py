"""
GPU tier: forward kernels that also write the gradient for an EXPECTED upstream scale
(csrc/losses.hip k_ce_fused / k_elem_fused / k_vm_fused, loss/_functional.py).

 * confirmed expectation: loss sum and gradient equal torch's fp32 autograd on the same
   tensors (relative 1e-5 on the scalars; gradients to fp32 / 16-bit rounding) and the
   two-kernel path of this library, and the backward launch only confirms;
 * wrong expectation: the backward launch recomputes — the gradient is the one the
   unspeculated path gives, whatever the expectation was;
 * the default expectation of the loss classes (`loss_sum / n` of the same call) is confirmed
   by autograd's own division, and the task helpers' sum-over-scales reduction is too.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(autouse=True)
def fresh_speculation_state():
    """the default expectation switches itself off after a history of misses (other test files
    call backward on bare sums): every test here starts from a clean history"""
    from nicr_mt_scene_analysis_amd.loss import reset_speculation_state
    reset_speculation_state()
    yield


def _gen(seed=0):
    return torch.Generator(device='cuda').manual_seed(seed)


def _stats():
    from nicr_mt_scene_analysis_amd.loss import speculation_stats
    return speculation_stats()


def _delta(before):
    now = _stats()
    return now['confirmed'] - before['confirmed'], now['recomputed'] - before['recomputed']


def _grad_tol(dtype):
    # outputs are rounded to the prediction dtype: half an ulp of 8 / 11 significand bits
    return {torch.float32: 2e-5, torch.bfloat16: 2 ** -7, torch.float16: 2 ** -9}[dtype]


def _ce_case(B, C, H, W, dtype, seed, void_frac=0.2):
    g = _gen(seed)
    x = (torch.randn((B, C, H, W), device='cuda', generator=g) * 3).to(dtype)
    t = torch.randint(1, C + 1, (B, H, W), device='cuda', generator=g)
    t[torch.rand((B, H, W), device='cuda', generator=g) < void_frac] = 0
    w = torch.rand(C, device='cuda', generator=g) + 0.5
    return x, t.to(torch.uint8), w


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize('C', [1, 5, 19, 24, 25, 40, 41, 48])
@pytest.mark.parametrize('label_smoothing', [0.0, 0.1])
def test_ce_forward_writes_gradient(dtype, C, label_smoothing):
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    x, t, w = _ce_case(2, C, 24, 36, dtype, seed=C)
    n = F_.count_u8(t, 1, C)
    assert int(n) == int((t != 0).sum())
    scale = F_.expected_scale(n)

    xs = x.clone().requires_grad_(True)
    before = _stats()
    loss, n_el, _ = F_.cross_entropy_sum(xs, t, w, label_smoothing, expected_scale=scale)
    (loss / n_el).backward()
    assert _delta(before) == (1, 0)                     # the forward's gradient was the right one

    xu = x.clone().requires_grad_(True)                 # two-kernel path of this library
    loss_u, n_u, _ = F_.cross_entropy_sum(xu, t, w, label_smoothing)
    (loss_u / n_u).backward()
    assert int(n_el) == int(n_u) == int(n)
    # C = 1: the true loss is 0 and what is left is the fp32 rounding of x*log2(e) per pixel
    atol = 1e-4 if C == 1 else 0.0
    np.testing.assert_allclose(float(loss), float(loss_u), rtol=RTOL, atol=atol)

    xr = x.double().requires_grad_(True)                # torch autograd, fp64
    ref = torch.nn.functional.cross_entropy(xr, t.long() - 1, weight=w.double(), reduction='sum',
                                            ignore_index=-1, label_smoothing=label_smoothing)
    (ref / int(n)).backward()
    np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL, atol=atol)
    tol = _grad_tol(dtype)
    # p - 1 at the target class cancels in fp32: absolute error ~ 1e-6 of the term g * w
    atol = max(tol * float(xr.grad.abs().max()) * 0.05, 4e-6 * float(w.max()) / int(n))
    np.testing.assert_allclose(xs.grad.double().cpu().numpy(), xr.grad.cpu().numpy(),
                               rtol=tol, atol=atol)
    np.testing.assert_allclose(xs.grad.double().cpu().numpy(), xu.grad.double().cpu().numpy(),
                               rtol=tol, atol=atol)


@pytest.mark.parametrize('shape', [(1, 7, 9), (3, 5, 11), (2, 16, 18), (1, 1, 3)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_ce_ragged_sizes_and_unaligned_views(shape, dtype):
    """pixel counts that are no multiple of the 8-B lane tile, and a base pointer that is not
    8-B aligned (the scalar tail path of k_ce_fused)"""
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    B, H, W = shape
    C = 13
    x, t, w = _ce_case(B, C, H, W, dtype, seed=H * W)
    flat = torch.zeros(x.numel() + 1, device='cuda', dtype=dtype)
    flat[1:] = x.flatten()
    x_off = flat[1:].view(B, C, H, W)                   # element-aligned only
    assert x_off.is_contiguous()
    for src in (x, x_off):
        xs = src.detach().clone().requires_grad_(True) if src is x else src.detach().requires_grad_(True)
        n = F_.count_u8(t, 1, C)
        loss, n_el, _ = F_.cross_entropy_sum(xs, t, w, 0.0, expected_scale=F_.expected_scale(n))
        (loss / n_el).backward()
        xr = x.double().requires_grad_(True)
        ref = torch.nn.functional.cross_entropy(xr, t.long() - 1, weight=w.double(),
                                                reduction='sum', ignore_index=-1)
        (ref / max(int(n), 1)).backward()
        np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL, atol=1e-6)
        if int(n):
            tol = _grad_tol(dtype)
            np.testing.assert_allclose(xs.grad.double().cpu().numpy(), xr.grad.cpu().numpy(),
                                       rtol=tol, atol=tol * float(xr.grad.abs().max()) * 0.05)


def test_wrong_expectation_is_recomputed():
    """the result never depends on the expectation: a wrong one costs a recomputation"""
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    x, t, w = _ce_case(2, 40, 48, 64, torch.bfloat16, seed=3)
    wrong = torch.full((1,), 0.125, device='cuda')
    xs = x.clone().requires_grad_(True)
    before = _stats()
    loss, n_el, _ = F_.cross_entropy_sum(xs, t, w, 0.0, expected_scale=wrong)
    (3.0 * loss / n_el).backward()
    assert _delta(before) == (0, 1)
    xu = x.clone().requires_grad_(True)
    loss_u, n_u, _ = F_.cross_entropy_sum(xu, t, w, 0.0)
    (3.0 * loss_u / n_u).backward()
    np.testing.assert_allclose(float(loss), float(loss_u), rtol=RTOL)
    # same formula, same upstream scale; the log-sum-exp is summed in another order
    np.testing.assert_allclose(xs.grad.float().cpu().numpy(), xu.grad.float().cpu().numpy(),
                               rtol=2 ** -7, atol=1e-9)
    # a second backward through the same graph recomputes as well
    xs2 = x.clone().requires_grad_(True)
    loss2, n2, _ = F_.cross_entropy_sum(xs2, t, w, 0.0, expected_scale=F_.expected_scale(n_u))
    out = loss2 / n2
    out.backward(retain_graph=True)
    first = xs2.grad.clone()
    xs2.grad = None
    out.backward()
    np.testing.assert_allclose(xs2.grad.float().cpu().numpy(), first.float().cpu().numpy(),
                               rtol=2 ** -7, atol=1e-9)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('C,label_smoothing', [(49, 0.0), (150, 0.0), (150, 0.1), (255, 0.0)])
def test_more_classes_than_registers_take_two_walks(dtype, C, label_smoothing):
    """beyond 48 classes the column does not fit the registers of k_ce_fused: the backward
    kernel's two walks also produce the forward sum — still one launch, still confirmed"""
    from nicr_mt_scene_analysis_amd import _lib as L
    from nicr_mt_scene_analysis_amd.loss import CrossEntropyLossSemantic
    assert L.lib().nmsa_loss_ce_fwd_grad_supported(L.float_dtype_code(torch.zeros(1, dtype=dtype)), C)
    x, t, w = _ce_case(2, C, 24, 36, dtype, seed=C)
    xs = x.clone().requires_grad_(True)
    before = _stats()
    (loss, n), = CrossEntropyLossSemantic(weights=w, label_smoothing=label_smoothing)([xs], [t])
    (loss / n).backward()
    assert _delta(before) == (1, 0)
    xr = x.double().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, t.long() - 1, weight=w.double(), reduction='sum',
                                            ignore_index=-1, label_smoothing=label_smoothing)
    (ref / int(n)).backward()
    assert int(n) == int((t != 0).sum())
    np.testing.assert_allclose(float(loss), float(ref), rtol=RTOL)
    tol = _grad_tol(dtype)
    atol = max(tol * float(xr.grad.abs().max()) * 0.05, 4e-6 * float(w.max()) / int(n))
    np.testing.assert_allclose(xs.grad.double().cpu().numpy(), xr.grad.cpu().numpy(),
                               rtol=tol, atol=atol)
    # a wrong expectation is recomputed here too
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    xs = x.clone().requires_grad_(True)
    l2, n2, _ = F_.cross_entropy_sum(xs, t, w, label_smoothing,
                                     expected_scale=torch.full((1,), 0.25, device='cuda'))
    (l2 / n2).backward()
    assert _delta(before) == (1, 1)
    np.testing.assert_allclose(xs.grad.double().cpu().numpy(), xr.grad.cpu().numpy(),
                               rtol=tol, atol=atol)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_loss_classes_expect_the_mean_by_default(dtype):
    """`loss / n` of every loss class confirms its forward-written gradient; gradients equal
    torch's on the same tensors"""
    from nicr_mt_scene_analysis_amd.loss import (CrossEntropyLossSemantic, L1Loss, MSELoss,
                                                 VonMisesLossBiternion)
    g = _gen(11)
    B, H, W = 2, 40, 52
    x, t, w = _ce_case(B, 40, H, W, dtype, seed=1)
    center = torch.rand((B, H, W), device='cuda', generator=g).to(dtype)
    center_t = torch.rand((B, H, W), device='cuda', generator=g)
    offset = torch.randn((B, 2, H, W), device='cuda', generator=g).to(dtype)
    offset_t = torch.randn((B, 2, H, W), device='cuda', generator=g)
    ori = torch.randn((B, 2, H, W), device='cuda', generator=g).to(dtype)
    ori_t = torch.nn.functional.normalize(torch.randn((B, 2, H, W), device='cuda', generator=g), dim=1)
    m1 = torch.rand((B, H, W), device='cuda', generator=g) < 0.7
    m2 = torch.rand((B, H, W), device='cuda', generator=g) < 0.5
    m3 = torch.rand((B, H, W), device='cuda', generator=g) < 0.3

    leaves = [v.clone().requires_grad_(True) for v in (x, center, offset, ori)]
    before = _stats()
    (lc, n), = CrossEntropyLossSemantic(weights=w)([leaves[0]], [t])
    a = MSELoss().masked_sum(leaves[1], center_t, m1)
    b = L1Loss().masked_sum(leaves[2], offset_t, m2)
    c = VonMisesLossBiternion().masked_sum(leaves[3], ori_t, m3)
    total = lc / n + a[0] / a[1] + b[0] / b[1] + c[0] / c[1]
    total.backward()
    assert _delta(before) == (4, 0)

    ref_leaves = [v.double().requires_grad_(True) for v in (x, center, offset, ori)]
    r_ce = torch.nn.functional.cross_entropy(ref_leaves[0], t.long() - 1, weight=w.double(),
                                             reduction='sum', ignore_index=-1) / int(n)
    r_a = ((ref_leaves[1] * m1) - center_t.double()).pow(2).sum() / int(m1.sum())
    r_b = ((ref_leaves[2] * m2.unsqueeze(1)) - offset_t.double()).abs().mean(dim=1).sum() / int(m2.sum())
    dot = (ref_leaves[3] * ori_t.double()).sum(dim=1)
    r_c = (1 - torch.exp(dot - 1))[m3].sum() / int(m3.sum())
    ref_total = r_ce + r_a + r_b + r_c
    ref_total.backward()
    np.testing.assert_allclose(float(total), float(ref_total), rtol=RTOL)
    tol = _grad_tol(dtype)
    for got, ref in zip(leaves, ref_leaves):
        np.testing.assert_allclose(got.grad.double().cpu().numpy(), ref.grad.cpu().numpy(),
                                   rtol=tol, atol=tol * float(ref.grad.abs().max()) * 0.05)


@pytest.mark.parametrize('cls_name', ['MSELoss', 'L1Loss'])
def test_unmasked_sum_and_mean_reductions(cls_name):
    """`loss / n_px` (python int) and reduction='mean' are expected too"""
    from nicr_mt_scene_analysis_amd import loss as losses
    g = _gen(2)
    x = torch.randn((2, 2, 20, 28), device='cuda', generator=g)
    y = torch.randn((2, 2, 20, 28), device='cuda', generator=g)
    for reduction in ('sum', 'mean'):
        xs = x.clone().requires_grad_(True)
        before = _stats()
        (l, n), = getattr(losses, cls_name)(reduction=reduction)([xs], [y])
        (l / n).backward()
        assert _delta(before) == (1, 0), reduction
        xr = x.double().requires_grad_(True)
        d = xr - y.double()
        ref = (d * d if cls_name == 'MSELoss' else d.abs()).mean(dim=1).sum() / (2 * 20 * 28)
        ref.backward()
        np.testing.assert_allclose(float(l / n), float(ref), rtol=RTOL)
        np.testing.assert_allclose(xs.grad.double().cpu().numpy(), xr.grad.cpu().numpy(),
                                   rtol=2e-5, atol=1e-9)


def test_sum_without_division_recomputes_and_is_right():
    """backward on the bare sum (upstream gradient 1.0) misses the default expectation"""
    from nicr_mt_scene_analysis_amd.loss import L1Loss, VonMisesLossBiternion
    g = _gen(4)
    p = torch.randn((2, 2, 24, 32), device='cuda', generator=g)
    y = torch.nn.functional.normalize(torch.randn((2, 2, 24, 32), device='cuda', generator=g), dim=1)
    m = torch.rand((2, 24, 32), device='cuda', generator=g) > 0.5
    for loss_obj, ref_fn in (
            (L1Loss(), lambda q: ((q * m.unsqueeze(1)) - y.double()).abs().mean(dim=1).sum()),
            (VonMisesLossBiternion(),
             lambda q: (1 - torch.exp((q * y.double()).sum(dim=1) - 1))[m].sum())):
        ps = p.clone().requires_grad_(True)
        before = _stats()
        l, n = loss_obj.masked_sum(ps, y, m)
        l.backward()
        assert _delta(before) == (0, 1)
        pr = p.double().requires_grad_(True)
        ref = ref_fn(pr)
        ref.backward()
        np.testing.assert_allclose(float(l), float(ref), rtol=RTOL)
        np.testing.assert_allclose(ps.grad.double().cpu().numpy(), pr.grad.cpu().numpy(),
                                   rtol=2e-5, atol=1e-9)


def test_no_gradient_is_written_without_autograd():
    """under no_grad / for predictions that do not require a gradient the plain forward runs"""
    from nicr_mt_scene_analysis_amd.loss import CrossEntropyLossSemantic
    x, t, w = _ce_case(1, 40, 24, 32, torch.bfloat16, seed=8)
    before = _stats()
    with torch.no_grad():
        (l0, n0), = CrossEntropyLossSemantic(weights=w)([x.clone().requires_grad_(True)], [t])
    (l1, n1), = CrossEntropyLossSemantic(weights=w)([x], [t])
    assert not l0.requires_grad and not l1.requires_grad
    np.testing.assert_allclose(float(l0), float(l1), rtol=1e-7)
    assert _delta(before) == (0, 0)


def test_expectation_is_learned_per_instance_and_switches_off_when_unstable():
    """no process-global policy: every loss instance (and task helper) owns the upstream factor it
    expects and learns it ON THE DEVICE.  (1) A constant factor the caller never told anybody
    about (`3 * loss / n`) misses once and is confirmed from then on; (2) an instance whose
    upstream gradient is no `w / n` at all (backward on the bare sum, with a varying count)
    switches ITS expectation off after 8 misses in a row — other instances are unaffected — and
    gradients stay right throughout"""
    from nicr_mt_scene_analysis_amd.loss import L1Loss
    g = _gen(9)
    p = torch.randn((1, 2, 16, 20), device='cuda', generator=g)
    y = torch.randn((1, 2, 16, 20), device='cuda', generator=g)

    def reference(m, scale):
        pr = p.double().requires_grad_(True)
        ((pr * m.unsqueeze(1) - y.double()).abs().mean(dim=1).sum() * scale).backward()
        return pr.grad

    steady, erratic = L1Loss(), L1Loss()
    before = _stats()
    for step in range(12):
        m = torch.rand((1, 16, 20), device='cuda', generator=g) > 0.5       # a new count every step
        ps = p.clone().requires_grad_(True)
        l, n = steady.masked_sum(ps, y, m)
        (3.0 * (l / n)).backward()                                          # weight x normalised loss
        np.testing.assert_allclose(ps.grad.double().cpu().numpy(), reference(m, 3.0 / int(m.sum())).cpu().numpy(),
                                   rtol=2e-5, atol=1e-9)
        ps = p.clone().requires_grad_(True)
        erratic.masked_sum(ps, y, m)[0].backward()                          # upstream gradient 1.0, not w / n
        np.testing.assert_allclose(ps.grad.double().cpu().numpy(), reference(m, 1.0).cpu().numpy(),
                                   rtol=2e-5, atol=1e-9)
    s = steady._spec.stats()
    assert s['recomputed'] == 1 and s['confirmed'] == 11, s                 # one miss, then learned
    assert steady._spec.weights('cuda')[0] == 3.0
    e = erratic._spec.records('cuda').tolist()[0]
    assert e[0] == 0 and e[1] == 12 and (e[5] & 1) == 1, e                   # never confirmed: switched off
    assert _delta(before) == (11, 13)


def test_constant_loss_weights_confirm_without_backward_scale():
    """the reference's FixedLossWeighting (loss_weighting/fixed.py:28-37) multiplies the task
    totals with constant weights; nobody sets `backward_scale`: the helpers learn the factors from
    the first backward pass and >= 90 % of the totals' backward passes are confirmed"""
    from test_task_helpers import make_loss_batch
    from nicr_mt_scene_analysis_amd.task_helper import InstanceTaskHelper, SemanticTaskHelper
    weights = {'semantic': 2.0, 'instance_center': 0.5, 'instance_offset': 0.1,
               'instance_orientation': 3.0}
    batch, preds, t = make_loss_batch()
    sem = SemanticTaskHelper(n_classes=9, class_weights=t['class_weights'].cpu().numpy())
    ins = InstanceTaskHelper(semantic_n_classes=10, semantic_classes_is_thing=(False,) * 5 + (True,) * 5)
    sem.initialize(torch.device('cuda'))
    ins.initialize(torch.device('cuda'))
    leaves = [preds['semantic_output'], *preds['instance_output']]
    before = _stats()
    steps = 12
    for _ in range(steps):
        for x in leaves:
            x.grad = None
        losses = {}
        for helper in (sem, ins):
            losses.update(helper.training_step(batch, 0, preds)[0])
        sum(w * losses[f'{k}_total_loss'] for k, w in weights.items()).backward()
    confirmed, recomputed = _delta(before)
    assert confirmed + recomputed == 4 * steps
    assert recomputed == 4 and confirmed >= 0.9 * 4 * steps, (confirmed, recomputed)
    learned = dict(zip(('semantic',), sem.spec_state(('semantic',)).weights('cuda')))
    learned.update(zip(('instance_center', 'instance_offset', 'instance_orientation'),
                       ins.spec_state(('instance_center', 'instance_offset', 'instance_orientation')).weights('cuda')))
    for k, w in weights.items():
        assert learned[k] == np.float32(w), (k, learned[k])


def test_count_u8_ranges_sizes_and_alignments():
    """k_count_u8: empty input, sizes around the 16-B vector width and the per-block tile,
    unaligned base pointers (scalar path), every kind of [lo, hi] range, bool masks"""
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    g = _gen(21)
    base = torch.randint(0, 256, (300_000,), device='cuda', generator=g, dtype=torch.int32).to(torch.uint8)
    host = base.cpu().numpy()
    for n in (0, 1, 15, 16, 17, 255, 4096, 16 * 1024 + 3, 262_144, 299_999):
        for off in (0, 1, 7, 16):
            if off + n > base.numel():
                continue
            view = base[off:off + n]
            for lo, hi in ((1, 255), (1, 40), (0, 0), (255, 255), (7, 7), (0, 255)):
                want = int(((host[off:off + n] >= lo) & (host[off:off + n] <= hi)).sum())
                assert int(F_.count_u8(view, lo, hi)) == want, (n, off, lo, hi)
    mask = torch.rand((3, 37, 53), device='cuda', generator=g) < 0.3
    cnt, inv = F_.count_u8(mask, with_mean_scale=True)
    assert int(cnt) == int(mask.sum())
    assert torch.equal(inv, torch.ones(1, device='cuda') / cnt)        # the division autograd does
    with pytest.raises(Exception):
        F_.count_u8(mask.cpu())


def test_nothing_selected_gives_zero_gradients():
    """all labels void / all mask bytes 0: n = 0, `loss / n` sends inf back — and the gradient is
    all zeros, not NaN"""
    from nicr_mt_scene_analysis_amd.loss import CrossEntropyLossSemantic, L1Loss, VonMisesLossBiternion
    g = _gen(5)
    x = torch.randn((2, 7, 12, 20), device='cuda', generator=g).requires_grad_(True)
    t = torch.zeros((2, 12, 20), dtype=torch.uint8, device='cuda')
    before = _stats()
    (l, n), = CrossEntropyLossSemantic()([x], [t])
    assert int(n) == 0 and float(l) == 0.0
    (l / n).backward()
    # (the expectation divides by max(n, 1) as the task helpers do; `l / 0` sends inf: recomputed)
    assert sum(_delta(before)) == 1
    assert torch.count_nonzero(x.grad) == 0
    p = torch.randn((2, 2, 12, 20), device='cuda', generator=g).requires_grad_(True)
    y = torch.randn((2, 2, 12, 20), device='cuda', generator=g)
    m = torch.zeros((2, 12, 20), dtype=torch.bool, device='cuda')
    for loss in (L1Loss(), VonMisesLossBiternion()):
        p.grad = None
        l, n = loss.masked_sum(p, y, m)
        assert int(n) == 0
        (l / n).backward()
        assert torch.count_nonzero(p.grad) == 0 and torch.isfinite(p.grad).all()
    # the task helper's guarded reduction (count.clamp(min=1)) sends 1.0 instead of inf
    from nicr_mt_scene_analysis_amd.task_helper.base import TaskHelperBase
    helper = TaskHelperBase()
    p.grad = None
    l, n = L1Loss().masked_sum(p, y, m)
    total = helper.accumulate_losses([l], [n])
    assert float(total) == float(l)          # masked-out pixels still count |0 - target| (instance.py:129-139)
    total.backward()
    assert torch.count_nonzero(p.grad) == 0


# ---- the multi-loss call itself (loss/_multi.py, csrc k_multi_*) -----------------------------------
def _multi_case(seed=21, B=2, H=24, W=40, dtype=torch.bfloat16):
    g = _gen(seed)

    def rnd(*s):
        return torch.randn(s, device='cuda', generator=g)
    case = {
        'ce40': ((rnd(B, 40, H, W) * 3).to(dtype), torch.randint(0, 41, (B, H, W), device='cuda', generator=g).to(torch.uint8),
                 torch.rand(40, device='cuda', generator=g) + 0.5),
        'ce19': ((rnd(B, 19, H // 2, W // 2) * 3).to(dtype),
                 torch.randint(0, 20, (B, H // 2, W // 2), device='cuda', generator=g).to(torch.uint8), None),
        'ce150': ((rnd(B, 150, H // 2, W // 2) * 3).to(dtype),
                  torch.randint(0, 151, (B, H // 2, W // 2), device='cuda', generator=g).to(torch.uint8),
                  torch.rand(150, device='cuda', generator=g) + 0.5),
        'mse': (torch.rand((B, H, W), device='cuda', generator=g).to(dtype), torch.rand((B, H, W), device='cuda', generator=g),
                torch.rand((B, H, W), device='cuda', generator=g) < 0.7),
        'l1': (rnd(B, 2, H, W).to(dtype), rnd(B, 2, H, W), torch.rand((B, H, W), device='cuda', generator=g) < 0.5),
        'vm': (rnd(B, 2, H, W).to(dtype), torch.nn.functional.normalize(rnd(B, 2, H, W), dim=1),
               torch.zeros((B, H, W), dtype=torch.bool, device='cuda')),      # empty mask: clamp
    }
    peaks = torch.rand((B, H, W), device='cuda', generator=g)
    peaks = torch.where(peaks > 0.97, torch.ones((), device='cuda'), peaks * 0.9)
    case['focal'] = (torch.rand((B, H, W), device='cuda', generator=g).clamp(0.02, 0.98).to(dtype), peaks,
                     torch.rand((B, H, W), device='cuda', generator=g) < 0.8)
    return case


def _torch_sums(case):
    """fp64 torch restatement of the item sums / counts (the loss definitions of loss/*.py)"""
    out = {}
    for k in ('ce40', 'ce19', 'ce150'):
        x, t, w = case[k]
        tl = t.long() - 1
        out[k] = (torch.nn.functional.cross_entropy(x.double(), tl, weight=None if w is None else w.double(),
                                                    ignore_index=-1, reduction='sum'), int((t != 0).sum()))
    x, y, m = case['mse']
    out['mse'] = (((x.double() * m - y.double()) ** 2).sum(), int(m.sum()))
    x, y, m = case['l1']
    out['l1'] = ((x.double() * m.unsqueeze(1) - y.double()).abs().mean(dim=1).sum(), int(m.sum()))
    x, y, m = case['vm']
    out['vm'] = ((1 - torch.exp(1.0 * ((x.double() * y.double()).sum(dim=1) - 1)))[m].sum(), int(m.sum()))
    return out


@pytest.mark.parametrize('with_grad', [True, False])
def test_multi_loss_mixed_items_one_call(with_grad):
    """one call with everything the dispatcher has: two register-resident CE variants (40 and 19
    classes: the second runs as a launch of its own), a 150-class CE (k_ce_split), MSE, L1 and a
    von Mises item whose mask is EMPTY (its count enters as max(0, 1)); with gradients and
    forward only (no count pass, divisors from the finalized counts) — sums, counts, per-item
    losses, totals and gradients against torch in fp64"""
    from nicr_mt_scene_analysis_amd.loss import _multi
    case = _multi_case()
    leaves = {k: v[0].clone().requires_grad_(with_grad) for k, v in case.items()}
    items = [{'kind': 'ce', 'pred': leaves['ce40'], 'mask': case['ce40'][1], 'weights': case['ce40'][2], 'total': 0},
             {'kind': 'ce', 'pred': leaves['ce19'], 'mask': case['ce19'][1], 'total': 0},
             {'kind': 'ce', 'pred': leaves['ce150'], 'mask': case['ce150'][1], 'weights': case['ce150'][2], 'total': 1},
             {'kind': 'mse', 'pred': leaves['mse'], 'target': case['mse'][1], 'mask': case['mse'][2], 'total': 2},
             {'kind': 'l1', 'pred': leaves['l1'], 'target': case['l1'][1], 'mask': case['l1'][2], 'total': 3},
             {'kind': 'vonmises', 'pred': leaves['vm'], 'target': case['vm'][1], 'mask': case['vm'][2], 'param': 1.0,
              'total': 3, 'clamp': True}]
    names = ['ce40', 'ce19', 'ce150', 'mse', 'l1', 'vm']
    ref = _torch_sums(case)
    spec = _multi.SpecState(4)
    for step in range(2):                                     # second step: the expectation (w = 1) is confirmed
        for v in leaves.values():
            v.grad = None
        if with_grad:
            res = _multi.multi_loss(items, 4, spec)
        else:
            with torch.no_grad():
                res = _multi.multi_loss(items, 4, spec)
        counts = res.counts.tolist()
        for i, k in enumerate(names):
            assert counts[i] == ref[k][1], (k, counts[i], ref[k][1])
            np.testing.assert_allclose(float(res.sums[i]), float(ref[k][0]), rtol=RTOL, atol=1e-6)
            np.testing.assert_allclose(float(res.item_losses[i]), float(ref[k][0]) / max(ref[k][1], 1),
                                       rtol=RTOL, atol=1e-6)
        div = [ref['ce40'][1] + ref['ce19'][1], ref['ce150'][1], ref['mse'][1], ref['l1'][1] + max(ref['vm'][1], 1)]
        tot = [float(ref['ce40'][0] + ref['ce19'][0]), float(ref['ce150'][0]), float(ref['mse'][0]),
               float(ref['l1'][0] + ref['vm'][0])]
        assert res.divisors.tolist() == [float(max(d, 1)) for d in div]
        np.testing.assert_allclose(res.total_losses.tolist(), [t / max(d, 1) for t, d in zip(tot, div)], rtol=RTOL)
        if not with_grad:
            continue
        (res.total_losses * torch.tensor([1.0, 1.0, 1.0, 1.0], device='cuda')).sum().backward()
        # torch fp64 gradients of the same totals
        dbl = {k: case[k][0].double().requires_grad_(True) for k in names}
        c2 = {k: (dbl[k],) + tuple(case[k][1:]) for k in names}
        r2 = _torch_sums(c2)
        ((r2['ce40'][0] + r2['ce19'][0]) / div[0] + r2['ce150'][0] / div[1] + r2['mse'][0] / div[2]
         + (r2['l1'][0] + r2['vm'][0]) / div[3]).backward()
        for k in names:
            gd = dbl[k].grad if dbl[k].grad is not None else torch.zeros_like(dbl[k])
            got = leaves[k].grad.double()
            tol = _grad_tol(torch.bfloat16)
            err = (got - gd).abs() - tol * gd.abs()
            assert float(err.max()) <= 1e-7, (k, step, float(err.max()))
    if with_grad:
        assert spec.stats() == {'confirmed': 8, 'recomputed': 0}


def test_multi_loss_focal_total_takes_its_divisor_from_the_loss_kernel():
    """the center-focal extension divides by the number of heat-map PEAKS, which only the loss
    kernel counts: such a total gets no expectation (backward recomputes) and its divisor is
    filled in by the finalize step — with and without gradients"""
    from nicr_mt_scene_analysis_amd.loss import CenterFocalLoss, _multi
    case = _multi_case(seed=5, dtype=torch.float32)
    x0, peaks, m = case['focal']
    focal = CenterFocalLoss()
    xs = x0.clone().requires_grad_(True)
    ls, n = focal.masked_sum(xs, peaks, m)                     # the loss-by-loss path
    (ls / n).backward()
    spec = _multi.SpecState(1)
    xm = x0.clone().requires_grad_(True)
    res = _multi.multi_loss([{'kind': 'focal', 'pred': xm, 'target': peaks, 'mask': m, 'total': 0}], 1, spec)
    assert int(res.counts[0]) == int(n) == int(((peaks == 1) & m).sum())
    assert float(res.divisors[0]) == float(max(int(n), 1))
    np.testing.assert_allclose(float(res.total_losses[0]), float(ls / n), rtol=1e-6)
    res.total_losses[0].backward()
    np.testing.assert_allclose(xm.grad.cpu().numpy(), xs.grad.cpu().numpy(), rtol=2e-5, atol=1e-9)
    assert spec.stats() == {'confirmed': 0, 'recomputed': 1}
    with torch.no_grad():
        res = _multi.multi_loss([{'kind': 'focal', 'pred': xm, 'target': peaks, 'mask': m, 'total': 0}], 1, spec)
    np.testing.assert_allclose(float(res.total_losses[0]), float(ls / n), rtol=1e-6)


def test_multi_loss_sixteen_items_and_recompute_walk():
    """the largest call (16 items, 8 totals) with upstream factors that are NOT the expected ones:
    the recomputing launch walks all block ranges with its small grid; gradients equal the ones
    of sixteen single calls"""
    from nicr_mt_scene_analysis_amd.loss import L1Loss, _multi
    g = _gen(31)
    preds, tgts, masks = [], [], []
    for i in range(16):
        H, W = 8 + 4 * (i % 5), 20 + 4 * (i % 3)
        preds.append(torch.randn((2, 2, H, W), device='cuda', generator=g))
        tgts.append(torch.randn((2, 2, H, W), device='cuda', generator=g))
        masks.append(torch.rand((2, H, W), device='cuda', generator=g) < 0.6)
    factors = torch.tensor([0.3, 1.7, 2.0, 0.5, 1.0, 4.0, 0.25, 3.0], device='cuda')
    leaves = [p.clone().requires_grad_(True) for p in preds]
    items = [{'kind': 'l1', 'pred': leaves[i], 'target': tgts[i], 'mask': masks[i], 'total': i // 2}
             for i in range(16)]
    spec = _multi.SpecState(8)
    res = _multi.multi_loss(items, 8, spec)
    (res.total_losses * factors).sum().backward()
    assert spec.stats() == {'confirmed': 1, 'recomputed': 7}            # only the factor 1.0 was expected
    single = L1Loss()
    for i in range(16):
        p = preds[i].clone().requires_grad_(True)
        j = i ^ 1
        l, n = single.masked_sum(p, tgts[i], masks[i])
        n_total = n + masks[j].sum()
        (factors[i // 2] * l / n_total).backward()
        np.testing.assert_allclose(leaves[i].grad.cpu().numpy(), p.grad.cpu().numpy(), rtol=2e-5, atol=1e-9)
    assert not _multi.supported(items + [items[0]])                       # 17 items: loss by loss


def test_multi_loss_second_backward_through_a_retained_graph():
    """backward twice through one graph (retain_graph=True, e.g. per-task gradient norms): the
    second pass recomputes into fresh buffers and accumulates like autograd does"""
    from nicr_mt_scene_analysis_amd.loss import L1Loss
    g = _gen(3)
    p = torch.randn((2, 2, 12, 20), device='cuda', generator=g).requires_grad_(True)
    y = torch.randn((2, 2, 12, 20), device='cuda', generator=g)
    m = torch.rand((2, 12, 20), device='cuda', generator=g) > 0.4
    l, n = L1Loss().masked_sum(p, y, m)
    loss = l / n
    loss.backward(retain_graph=True)
    g1 = p.grad.clone()
    (2.0 * loss).backward()
    np.testing.assert_allclose(p.grad.cpu().numpy(), (3.0 * g1).cpu().numpy(), rtol=1e-6, atol=1e-9)
    pr = p.detach().double().requires_grad_(True)
    ((pr * m.unsqueeze(1) - y.double()).abs().mean(dim=1).sum() / int(m.sum())).backward()
    np.testing.assert_allclose(g1.double().cpu().numpy(), pr.grad.cpu().numpy(), rtol=2e-5, atol=1e-9)


@pytest.mark.parametrize('seed', range(10))
def test_multi_loss_fuzz_random_item_mixes(seed):
    """random mixes of items — kinds, class counts (register-resident, split and mixed variants in
    one call), dtypes, shapes, totals, cosine items, with random upstream factors per total:
    sums, counts, totals and gradients of ONE call against torch in fp64 item by item"""
    from nicr_mt_scene_analysis_amd.loss import _multi
    rng = np.random.default_rng(1000 + seed)
    g = _gen(500 + seed)
    n_items = int(rng.integers(2, 9))
    n_totals = int(rng.integers(1, min(n_items, 4) + 1))
    items, refs = [], []
    for i in range(n_items):
        kind = ['ce', 'mse', 'l1', 'vonmises', 'cos'][int(rng.integers(0, 5))]
        dtype = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(0, 3))]
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(1, 7)) * 4, int(rng.integers(1, 12)) * 4
        total = i if i < n_totals else int(rng.integers(0, n_totals))
        it = {'kind': kind, 'total': total}
        if kind == 'ce':
            C = int(rng.choice([2, 7, 19, 24, 40, 48, 49, 70, 150]))
            x = (torch.randn((B, C, H, W), device='cuda', generator=g) * 3).to(dtype)
            t = torch.randint(0, C + 1, (B, H, W), device='cuda', generator=g).to(torch.uint8)
            w = torch.rand(C, device='cuda', generator=g) + 0.5 if rng.random() < 0.7 else None
            it.update(pred=x, mask=t, weights=w)

            def ref(xd, t=t, w=w):
                return torch.nn.functional.cross_entropy(xd, t.long() - 1, weight=None if w is None else w.double(),
                                                         ignore_index=-1, reduction='sum'), int((t != 0).sum())
        elif kind in ('mse', 'l1'):
            Cc = 1 if kind == 'mse' else 2
            shape = (B, H, W) if Cc == 1 else (B, Cc, H, W)
            x = torch.randn(shape, device='cuda', generator=g).to(dtype)
            y = torch.randn(shape, device='cuda', generator=g)
            m = torch.rand((B, H, W), device='cuda', generator=g) < 0.6 if rng.random() < 0.8 else None
            it.update(pred=x, target=y, mask=m)

            def ref(xd, y=y, m=m, kind=kind, Cc=Cc, B=B, H=H, W=W):
                mm = torch.ones((B, H, W), device='cuda', dtype=torch.bool) if m is None else m
                xm = xd * (mm if Cc == 1 else mm.unsqueeze(1))
                d = xm - y.double()
                v = (d ** 2) if kind == 'mse' else d.abs()
                return (v if Cc == 1 else v.mean(dim=1)).sum(), int(mm.sum())
        elif kind == 'vonmises':
            x = torch.randn((B, 2, H, W), device='cuda', generator=g).to(dtype)
            y = torch.nn.functional.normalize(torch.randn((B, 2, H, W), device='cuda', generator=g), dim=1)
            m = torch.rand((B, H, W), device='cuda', generator=g) < 0.4
            kappa = float(rng.choice([1.0, 2.5]))
            it.update(pred=x, target=y, mask=m, param=kappa, clamp=True)

            def ref(xd, y=y, m=m, kappa=kappa):
                return (1 - torch.exp(kappa * ((xd * y.double()).sum(dim=1) - 1)))[m].sum(), max(int(m.sum()), 1)
        else:
            D, NL = int(rng.choice([64, 128])), int(rng.integers(1, 9))
            x = torch.randn((B, D, H, W), device='cuda', generator=g).to(dtype)
            lut = torch.nn.functional.normalize(torch.randn((B, NL, D), device='cuda', generator=g), dim=-1)
            ix = torch.randint(0, NL + 1, (B, H, W), device='cuda', generator=g, dtype=torch.int32)
            it.update(pred=x, target=lut, mask=ix, clamp=2)

            def ref(xd, lut=lut, ix=ix):
                valid = ix != 0
                rows = xd.permute(0, 2, 3, 1)[valid]
                tgt = lut[torch.where(valid)[0], (ix[valid] - 1).long()].double()
                if len(rows) == 0:
                    return xd.sum() * 0, 0
                return torch.nn.functional.cosine_embedding_loss(rows, tgt, torch.ones(len(rows), device='cuda'),
                                                                 reduction='sum'), len(rows)
        items.append(it)
        refs.append(ref)
    assert _multi.supported(items)
    leaves = [it['pred'].clone().requires_grad_(True) for it in items]
    for it, lf in zip(items, leaves):
        it['pred'] = lf
    factors = torch.tensor(rng.choice([1.0, 0.5, 2.0, 3.0], size=n_totals), dtype=torch.float32, device='cuda')
    spec = _multi.SpecState(n_totals)
    res = _multi.multi_loss(items, n_totals, spec)
    (res.total_losses * factors).sum().backward()
    dbl = [lf.detach().double().requires_grad_(True) for lf in leaves]
    vals = [r(d) for r, d in zip(refs, dbl)]
    div = [max(sum(n for (_, n), it in zip(vals, items) if it['total'] == t), 1) for t in range(n_totals)]
    sum(float(factors[t]) * sum(l for (l, _), it in zip(vals, items) if it['total'] == t) / div[t]
        for t in range(n_totals)).backward()
    counts = res.counts.tolist()
    for i, ((l, n), it) in enumerate(zip(vals, items)):
        n_raw = n if it['kind'] != 'vonmises' else int(it['mask'].sum())
        assert counts[i] == n_raw, (i, it['kind'], counts[i], n_raw)
        np.testing.assert_allclose(float(res.sums[i]), float(l), rtol=2e-5, atol=1e-5)
    np.testing.assert_allclose(res.divisors.tolist(), [float(d) for d in div])
    for i, (lf, d, it) in enumerate(zip(leaves, dbl, items)):
        gd = d.grad if d.grad is not None else torch.zeros_like(d)
        tol = _grad_tol(lf.dtype)
        err = (lf.grad.double() - gd).abs() - tol * gd.abs()
        assert float(err.max()) <= max(1e-6, 0.05 * tol * float(gd.abs().max())), (i, it['kind'], float(err.max()))


def test_nan_upstream_gradient_is_never_taken_for_the_missing_expectation():
    """"no expectation" is a NaN in the expectation slot.  A real upstream gradient that is that
    NaN (NaN loss weight) must not be confirmed against it: the recompute runs and the gradients
    are NaN — what autograd gives for the reference — instead of the uninitialised buffer"""
    from nicr_mt_scene_analysis_amd.loss import _multi
    g = _gen(3)
    x = torch.rand((2, 1, 16, 32), device='cuda', generator=g)
    tgt = torch.rand((2, 1, 16, 32), device='cuda', generator=g)
    tgt[:, :, 4, 4] = 1.0
    logits, labels, w = _ce_case(2, 7, 16, 32, torch.float32, seed=9)
    for kind in ('focal', 'mse'):
        spec = _multi.SpecState(2)
        if kind == 'mse':
            spec.records('cuda')[0, 5] |= 1                 # the total's expectation is switched off
        xs = x.clone().requires_grad_(True)
        ls = logits.clone().requires_grad_(True)
        res = _multi.multi_loss([{'kind': kind, 'pred': xs, 'target': tgt, 'mask': None, 'total': 0},
                                 {'kind': 'ce', 'pred': ls, 'mask': labels, 'weights': w, 'total': 1}], 2, spec)
        nan = torch.tensor(float('nan'), device='cuda')
        (res.total_losses[0] * nan + res.total_losses[1]).backward()
        assert torch.isnan(xs.grad).all(), kind
        assert torch.isfinite(ls.grad).all() and ls.grad.abs().sum() > 0


def test_second_backward_through_a_retained_graph_leaves_the_record_alone():
    from nicr_mt_scene_analysis_amd.loss import _multi
    logits, labels, w = _ce_case(2, 7, 16, 32, torch.float32, seed=11)
    spec = _multi.SpecState(1)
    ls = logits.clone().requires_grad_(True)
    res = _multi.multi_loss([{'kind': 'ce', 'pred': ls, 'mask': labels, 'weights': w, 'total': 0}], 1, spec)
    before = _stats()
    res.total_losses[0].backward(retain_graph=True)
    first = ls.grad.clone()
    assert _delta(before) == (1, 0) and spec.stats() == {'confirmed': 1, 'recomputed': 0}
    rec = spec.records('cuda').clone()
    for _ in range(10):                                   # e.g. a gradient penalty loop
        ls.grad = None
        res.total_losses[0].backward(retain_graph=True)
        assert torch.equal(ls.grad, first)
    assert torch.equal(spec.records('cuda'), rec)         # not a miss: nothing was predicted
    assert _delta(before) == (1, 0)
    # an in-place change of the prediction between forward and a recomputing backward is caught
    with torch.no_grad():
        ls.add_(1.0)
    with pytest.raises(RuntimeError, match='modified by an inplace operation'):
        res.total_losses[0].backward()


def test_lists_of_several_scales_take_the_plain_path():
    """LossBase.forward over several scales (outside the task helpers' one-call path): the sums
    are divided by the SUMMED counts later, which one record per instance cannot predict per
    scale -> no forward-written gradient, no misses, results as ever"""
    from nicr_mt_scene_analysis_amd.loss import CrossEntropyLossSemantic, MSELoss
    logits, labels, w = _ce_case(2, 7, 16, 32, torch.float32, seed=13)
    small = logits[:, :, ::2, ::2].contiguous()
    small_labels = labels[:, ::2, ::2].contiguous()
    ce = CrossEntropyLossSemantic(weights=w)
    for _ in range(3):
        a = logits.clone().requires_grad_(True)
        b = small.clone().requires_grad_(True)
        before = _stats()
        (l0, n0), (l1, n1) = ce([a, b], [labels, small_labels])
        ((l0 + l1) / (n0 + n1)).backward()
        assert _delta(before) == (0, 0)
    ra = logits.double().requires_grad_(True)
    rb = small.double().requires_grad_(True)
    ref = sum(torch.nn.functional.cross_entropy(p, t.long() - 1, weight=w.double(), reduction='sum',
                                                ignore_index=-1) for p, t in ((ra, labels), (rb, small_labels)))
    (ref / int(n0 + n1)).backward()
    np.testing.assert_allclose(a.grad.double().cpu().numpy(), ra.grad.cpu().numpy(), rtol=2e-5,
                               atol=2e-6 * float(ra.grad.abs().max()))        # fp32 softmax next to p - 1
    # a single-scale call still speculates (and is confirmed)
    a = logits.clone().requires_grad_(True)
    before = _stats()
    (l0, n0), = ce([a], [labels])
    (l0 / n0).backward()
    assert _delta(before) == (1, 0)
    # the ESANet reduction divides by the weight sum: no expectation either
    cew = CrossEntropyLossSemantic(weights=w, weighted_reduction=True)
    a = logits.clone().requires_grad_(True)
    before = _stats()
    (l0, n0), = cew([a], [labels])
    l0.backward()
    assert _delta(before) == (0, 0)
    del MSELoss
