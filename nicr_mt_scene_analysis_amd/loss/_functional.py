"""autograd Functions over the HIP loss kernels (csrc/losses.hip).

Forward returns device scalars (loss sum as fp32 0-d tensor, counts as int64 0-d
tensors: no `.item()` host sync, unlike reference loss/ce.py:50 and
task_helper/instance.py:138-139); backward recomputes from the saved inputs and
scales by the upstream gradient on the device.

Speculative gradients.  Every loss here is a SUM the caller divides by a count that depends
on the targets only (`loss / n`; task_helper/base.py:161-182 divides the sum over the scales
by the summed counts), so the gradient that autograd is going to hand to the loss sum is
known before the forward kernel runs: `count_u8` counts the labels / mask bytes (1 B/px) and
`expected_scale` forms the very division autograd will do.  Given that expectation the forward
kernel also writes the finished gradient (prediction read once, gradient written once); the
backward launch compares the real upstream gradient with the expectation ON THE DEVICE and
recomputes only when they differ bit-wise, so results never depend on the expectation being
right.  `speculation_stats()` counts both outcomes.  NMSA_SPECULATIVE_GRAD=0 turns it off.
"""
import os
import warnings
from typing import Dict, Optional, Tuple

import ctypes as C

import torch

from .. import _lib as L


def _workspace(B, H, W, dev):
    nbytes = L.lib().nmsa_loss_workspace_bytes(B, H, W)
    return torch.empty((nbytes,), dtype=torch.uint8, device=dev), nbytes


def _u8(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)
    if t.dtype == torch.uint8:
        return t.contiguous()
    return (t != 0).contiguous().view(torch.uint8)


def _scalar_outputs(dev):
    return (torch.empty((1,), dtype=torch.float64, device=dev),
            torch.empty((1,), dtype=torch.int64, device=dev))


# Out-of-range labels / LUT indices are a device assert in PyTorch.  The kernels skip such
# pixels and OR a bit into ONE persistent status word per device; `check_loss_status()` reads
# it (a host sync, so not per call): the task helpers call it at `validation_epoch_end`, and
# NMSA_CHECK_STATUS=1 checks after every loss call.
_STATUS: Dict[torch.device, torch.Tensor] = {}      # int32 [status, confirmed, recomputed, -]
_CHECK_EVERY_CALL = bool(int(os.environ.get('NMSA_CHECK_STATUS', '0') or 0))
_SPECULATE = bool(int(os.environ.get('NMSA_SPECULATIVE_GRAD', '1') or 0))


def _status_word(dev: torch.device) -> torch.Tensor:
    st = _STATUS.get(dev)
    if st is None:
        st = _STATUS[dev] = torch.zeros((4,), dtype=torch.int32, device=dev)
    return st


def _counters_ptr(dev: torch.device):
    return C.c_void_p(_status_word(dev).data_ptr() + 4)


def check_loss_status() -> None:
    """raise IndexError if any loss kernel since the last check saw a label >= C + 1 or a LUT
    index outside [0, L]; warn when a wide-column cosine call took its fallback (host sync)"""
    for dev, st in _STATUS.items():
        v = int(st[0].item())
        if v & 32:
            # results are right (the two-walk kernels recomputed that call on the device, gated by the
            # kernel's own gave-up word): this only says that time was lost
            warnings.warn(f'loss kernels on {dev}: cooperating workgroups of the wide-column cosine loss '
                          '(k_cos_parts) did not all become resident within the time-out; those calls were '
                          'recomputed by the two-walk kernels (correct, slower). Another stream holding '
                          'compute units for long, or a device that schedules on fewer compute units than '
                          'it reports, causes this.', RuntimeWarning, stacklevel=2)
            v &= ~32
        if v:
            st[0] = 0
            raise IndexError(f'loss kernels on {dev}: target label / LUT index out of range '
                             '(PyTorch raises a device-side assert for these)')
        st[0] = 0


def speculation_stats() -> Dict[str, int]:
    """backward passes that found their gradient already written by the forward kernel
    ('confirmed') / had to recompute it ('recomputed') since the process started or the last
    reset: a per-device tally next to the status word that every backward launch adds to (explicit
    `expected_scale=` calls count per loss, the learned expectations of loss instances and task
    helpers — loss/_multi.py — per total).  Statistics only: no policy reads them.  Host sync."""
    out = {'confirmed': 0, 'recomputed': 0}
    for st in _STATUS.values():
        _, confirmed, recomputed, _ = (int(v) for v in st.tolist())
        out['confirmed'] += confirmed
        out['recomputed'] += recomputed
    return out


def reset_speculation_state() -> None:
    """forget the confirmed / recomputed history and the learned upstream factors of every live
    loss instance / task helper (host sync) — a new training run, a test, a benchmark leg"""
    from . import _multi
    _multi.reset_all()
    for st in _STATUS.values():
        st[1:3] = 0


def speculation_enabled() -> bool:
    return _SPECULATE


def mean_speculation_enabled() -> bool:
    """kept for callers of round 2: the default expectation of the loss classes is a learned,
    per-instance state now (loss/_multi.py); this only says whether speculation is compiled in"""
    return _SPECULATE


def count_u8(values: torch.Tensor, lo: int = 1, hi: int = 255, with_mean_scale: bool = False):
    """number of bytes in [lo, hi] as an int64 0-d device tensor (labels 1..C, mask != 0);
    `with_mean_scale`: also 1 / count as the fp32 [1] tensor `expected_scale(count)` gives"""
    v = L.require_device_tensor(values, 'values')
    if v.dtype == torch.bool:
        v = v.contiguous().view(torch.uint8)
    assert v.dtype == torch.uint8
    v = v.contiguous()
    dev = v.device
    out = torch.empty((1,), dtype=torch.int64, device=dev)
    scale = torch.empty((1,), dtype=torch.float32, device=dev) if with_mean_scale else None
    nbytes = L.lib().nmsa_count_workspace_bytes()
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    L.check(L.lib().nmsa_count_u8(L.ptr(v), v.numel(), int(lo), int(hi), L.ptr(out), L.ptr(scale),
                                  1.0, L.ptr(ws), nbytes, L.stream_ptr(dev)), 'nmsa_count_u8')
    return (out[0], scale) if with_mean_scale else out[0]


_ONES: Dict[Tuple[torch.device, float], torch.Tensor] = {}


def expected_scale(count, weight: float = 1.0, device=None) -> torch.Tensor:
    """the gradient autograd hands to `loss_sum` in `weight * loss_sum / count`: the same ATen
    division, so the value is bit-equal to the one backward receives"""
    dev = count.device if isinstance(count, torch.Tensor) else torch.device(device)
    key = (dev, float(weight))
    one = _ONES.get(key)
    if one is None:
        one = _ONES[key] = torch.full((), float(weight), dtype=torch.float32, device=dev)
    return (one / count).reshape(1)


def labels_u8(target: torch.Tensor, dev) -> torch.Tensor:
    """labels 0..C (0 = void) as contiguous uint8 on `dev`; out-of-range values must not wrap
    into valid labels: they become 255 (> C) and set the status bit in the kernel"""
    t = target.to(dev)
    if t.dtype != torch.uint8:
        t = torch.where((t < 0) | (t > 255), 255, t).to(torch.uint8) \
            if t.dtype != torch.bool else t.to(torch.uint8)
    return t.contiguous()


def labels_i16(target: torch.Tensor, dev) -> torch.Tensor:
    """labels 0..C (0 = void) as contiguous int16 on `dev` (more than 255 classes); values that
    do not fit become 32767 (> C: the kernel skips the pixel and sets the status bit)"""
    t = target.to(dev)
    if t.dtype != torch.int16:
        t = torch.where((t < 0) | (t > 32767), 32767, t).to(torch.int16) \
            if t.dtype not in (torch.bool, torch.uint8) else t.to(torch.int16)
    return t.contiguous()


def _expected(hint, dev) -> Optional[torch.Tensor]:
    if hint is None or not _SPECULATE:
        return None
    return hint.detach().to(dev, torch.float32).reshape(1).contiguous()


def _grad_scale(g: torch.Tensor) -> torch.Tensor:
    return g.detach().to(torch.float32).reshape(1).contiguous()


class CrossEntropyFunction(torch.autograd.Function):
    """sum over non-void px of the (weighted, label-smoothed) CE; also n and sum w[label]."""

    @staticmethod
    def forward(ctx, logits, target, weights, label_smoothing, expected=None):
        x = L.require_device_tensor(logits, 'input_')
        B, C_, H, W = x.shape
        dev = x.device
        wide = C_ > 255
        if C_ > 4096:
            raise ValueError(f'{C_} classes: the CE kernels take up to 4096')
        # labels travel as uint8 (0 = void, 1..C; ToTorchTensors keeps semantic uint8); with more
        # than 255 classes as int16 through the two-kernel path (nmsa_loss_ce_*_i16)
        t = labels_i16(target, dev) if wide else labels_u8(target, dev)
        w = None if weights is None else weights.to(dev, torch.float32).contiguous()
        s, n = _scalar_outputs(dev)
        wsum = torch.empty((1,), dtype=torch.float64, device=dev)
        status = _status_word(dev)
        ws, nbytes = _workspace(B, H, W, dev)
        code = L.float_dtype_code(x)
        exp = _expected(expected, dev) if ctx.needs_input_grad[0] else None
        if exp is not None and (wide or not L.lib().nmsa_loss_ce_fwd_grad_supported(code, C_)):
            exp = None                               # no such kernel for this dtype / C
        lse2 = grad = None
        if exp is not None:
            # forward sum + the gradient for the expected upstream scale in one pass
            grad = torch.empty_like(x)
            L.check(L.lib().nmsa_loss_ce_fwd_grad(
                L.ptr(x), code, L.ptr(t), L.ptr(w), B, C_, H, W, float(label_smoothing),
                L.ptr(exp), L.ptr(s), L.ptr(n), L.ptr(wsum), L.ptr(grad), L.ptr(status),
                L.ptr(ws), nbytes, L.stream_ptr(dev)), 'nmsa_loss_ce_fwd_grad')
        else:
            # per-pixel log-sum-exp for the backward pass (4 B/px instead of a second read of
            # the logits), only when a gradient can be asked for
            lse2 = torch.empty((B, H, W), dtype=torch.float32, device=dev) \
                if ctx.needs_input_grad[0] else None
            fwd = L.lib().nmsa_loss_ce_fwd_i16 if wide else L.lib().nmsa_loss_ce_fwd
            L.check(fwd(
                L.ptr(x), code, L.ptr(t), L.ptr(w), B, C_, H, W,
                float(label_smoothing), L.ptr(s), L.ptr(n), L.ptr(wsum), L.ptr(lse2),
                L.ptr(status), L.ptr(ws), nbytes, L.stream_ptr(dev)), 'nmsa_loss_ce_fwd')
        if _CHECK_EVERY_CALL:
            check_loss_status()
        ctx.save_for_backward(x, t, w if w is not None else torch.empty(0, device=dev),
                              lse2 if lse2 is not None else torch.empty(0, device=dev))
        ctx.has_lse = lse2 is not None
        ctx.has_w = w is not None
        ctx.ls = float(label_smoothing)
        ctx.spec = (grad, exp) if grad is not None else None
        loss = s[0].to(torch.float32)
        n_el = n[0]
        wsum_ = wsum[0]
        ctx.mark_non_differentiable(n_el, wsum_)
        return loss, n_el, wsum_

    @staticmethod
    def backward(ctx, g_loss, g_n, g_w):
        x, t, w, lse2 = ctx.saved_tensors
        B, C_, H, W = x.shape
        gs = _grad_scale(g_loss)
        spec, ctx.spec = ctx.spec, None          # a second backward recomputes
        if spec is not None:
            grad, exp = spec
            L.check(L.lib().nmsa_loss_ce_bwd_unless(
                L.ptr(x), L.float_dtype_code(x), L.ptr(t), L.ptr(w) if ctx.has_w else None,
                B, C_, H, W, ctx.ls, L.ptr(gs), L.ptr(grad), L.ptr(exp),
                _counters_ptr(x.device), L.stream_ptr(x.device)), 'nmsa_loss_ce_bwd_unless')
            return grad, None, None, None, None
        grad = torch.empty_like(x)
        bwd = L.lib().nmsa_loss_ce_bwd_i16 if t.dtype == torch.int16 else L.lib().nmsa_loss_ce_bwd
        L.check(bwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(t), L.ptr(w) if ctx.has_w else None,
            B, C_, H, W, ctx.ls, L.ptr(gs), L.ptr(lse2) if ctx.has_lse else None, L.ptr(grad),
            L.stream_ptr(x.device)), 'nmsa_loss_ce_bwd')
        return grad, None, None, None, None


class MaskedElementwiseFunction(torch.autograd.Function):
    """sum_px mean_c f(pred*mask - target) and n = sum(mask); kind 0 = MSE, 1 = L1."""

    @staticmethod
    def forward(ctx, pred, target, mask, kind, expected=None):
        x = L.require_device_tensor(pred, 'input_')
        dev = x.device
        if x.ndim == 3:
            B, H, W = x.shape
            C_ = 1
        else:
            B, C_, H, W = x.shape
        y = target.to(dev, torch.float32).contiguous()
        m = _u8(None if mask is None else mask.to(dev))
        s, n = _scalar_outputs(dev)
        ws, nbytes = _workspace(B, H, W, dev)
        exp = _expected(expected, dev) if ctx.needs_input_grad[0] else None
        grad = None
        if exp is not None:
            grad = torch.empty_like(x)
            L.check(L.lib().nmsa_loss_masked_fwd_grad(
                L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m), B, C_, H, W, int(kind),
                L.ptr(exp), L.ptr(s), L.ptr(n), L.ptr(grad), L.ptr(ws), nbytes,
                L.stream_ptr(dev)), 'nmsa_loss_masked_fwd_grad')
        else:
            L.check(L.lib().nmsa_loss_masked_fwd(
                L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m), B, C_, H, W, int(kind),
                L.ptr(s), L.ptr(n), L.ptr(ws), nbytes, L.stream_ptr(dev)), 'nmsa_loss_masked_fwd')
        ctx.save_for_backward(x, y, m if m is not None else torch.empty(0, device=dev))
        ctx.has_m = m is not None
        ctx.kind = int(kind)
        ctx.dims = (B, C_, H, W)
        ctx.spec = (grad, exp) if grad is not None else None
        loss = s[0].to(torch.float32)
        n_el = n[0]
        ctx.mark_non_differentiable(n_el)
        return loss, n_el

    @staticmethod
    def backward(ctx, g_loss, g_n):
        x, y, m = ctx.saved_tensors
        B, C_, H, W = ctx.dims
        gs = _grad_scale(g_loss)
        spec, ctx.spec = ctx.spec, None
        if spec is not None:
            grad, exp = spec
            L.check(L.lib().nmsa_loss_masked_bwd_unless(
                L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m) if ctx.has_m else None,
                B, C_, H, W, ctx.kind, L.ptr(gs), L.ptr(grad), L.ptr(exp),
                _counters_ptr(x.device), L.stream_ptr(x.device)), 'nmsa_loss_masked_bwd_unless')
            return grad, None, None, None, None
        grad = torch.empty_like(x)
        L.check(L.lib().nmsa_loss_masked_bwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m) if ctx.has_m else None,
            B, C_, H, W, ctx.kind, L.ptr(gs), L.ptr(grad), L.stream_ptr(x.device)),
            'nmsa_loss_masked_bwd')
        return grad, None, None, None, None


class VonMisesFunction(torch.autograd.Function):
    """sum over masked px of 1 - exp(kappa (x.y - 1)); pred/target planar [B,2,H,W]."""

    @staticmethod
    def forward(ctx, pred, target, mask, kappa, expected=None):
        x = L.require_device_tensor(pred, 'input_')
        dev = x.device
        B, two, H, W = x.shape
        assert two == 2
        y = target.to(dev, torch.float32).contiguous()
        m = _u8(None if mask is None else mask.to(dev))
        s, n = _scalar_outputs(dev)
        ws, nbytes = _workspace(B, H, W, dev)
        exp = _expected(expected, dev) if ctx.needs_input_grad[0] else None
        grad = None
        if exp is not None:
            grad = torch.empty_like(x)
            L.check(L.lib().nmsa_loss_vonmises_fwd_grad(
                L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m), B, H, W, float(kappa),
                L.ptr(exp), L.ptr(s), L.ptr(n), L.ptr(grad), L.ptr(ws), nbytes,
                L.stream_ptr(dev)), 'nmsa_loss_vonmises_fwd_grad')
        else:
            L.check(L.lib().nmsa_loss_vonmises_fwd(
                L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m), B, H, W, float(kappa),
                L.ptr(s), L.ptr(n), L.ptr(ws), nbytes, L.stream_ptr(dev)), 'nmsa_loss_vonmises_fwd')
        ctx.save_for_backward(x, y, m if m is not None else torch.empty(0, device=dev))
        ctx.has_m = m is not None
        ctx.kappa = float(kappa)
        ctx.spec = (grad, exp) if grad is not None else None
        loss = s[0].to(torch.float32)
        n_el = n[0]
        ctx.mark_non_differentiable(n_el)
        return loss, n_el

    @staticmethod
    def backward(ctx, g_loss, g_n):
        x, y, m = ctx.saved_tensors
        B, _, H, W = x.shape
        gs = _grad_scale(g_loss)
        spec, ctx.spec = ctx.spec, None
        if spec is not None:
            grad, exp = spec
            L.check(L.lib().nmsa_loss_vonmises_bwd_unless(
                L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m) if ctx.has_m else None,
                B, H, W, ctx.kappa, L.ptr(gs), L.ptr(grad), L.ptr(exp),
                _counters_ptr(x.device), L.stream_ptr(x.device)), 'nmsa_loss_vonmises_bwd_unless')
            return grad, None, None, None, None
        grad = torch.empty_like(x)
        L.check(L.lib().nmsa_loss_vonmises_bwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m) if ctx.has_m else None,
            B, H, W, ctx.kappa, L.ptr(gs), L.ptr(grad), L.stream_ptr(x.device)),
            'nmsa_loss_vonmises_bwd')
        return grad, None, None, None, None


class CosineEmbeddingLutFunction(torch.autograd.Function):
    """sum over px with index != 0 of 1 - cos(pred[:, px], lut[b, index-1])."""

    @staticmethod
    def forward(ctx, pred, indices, lut):
        x = L.require_device_tensor(pred, 'input_')
        dev = x.device
        B, D, H, W = x.shape
        idx = indices.to(dev, torch.int32).contiguous()
        lt = lut.to(dev, torch.float32).contiguous()
        Lr = lt.shape[1]
        s, n = _scalar_outputs(dev)
        status = _status_word(dev)
        ws, nbytes = _workspace(B, H, W, dev)
        # x.y and |x|^2 per pixel for the backward pass (8 B/px instead of a second read of the
        # 2D B/px prediction), only when a gradient can be asked for
        dots = None
        if ctx.needs_input_grad[0] and x.data_ptr() % 16 == 0 and \
                L.lib().nmsa_loss_cos_emb_can_keep_dots(D, H, W, Lr):
            dots = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
        L.check(L.lib().nmsa_loss_cos_emb_fwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(idx), L.ptr(lt), B, D, H, W, Lr,
            L.ptr(s), L.ptr(n), L.ptr(dots), L.ptr(status), L.ptr(ws), nbytes,
            L.stream_ptr(dev)), 'nmsa_loss_cos_emb_fwd')
        if _CHECK_EVERY_CALL:
            check_loss_status()
        ctx.save_for_backward(x, idx, lt, dots if dots is not None else torch.empty(0, device=dev))
        ctx.has_dots = dots is not None
        loss = s[0].to(torch.float32)
        n_el = n[0]
        ctx.mark_non_differentiable(n_el)
        return loss, n_el

    @staticmethod
    def backward(ctx, g_loss, g_n):
        x, idx, lt, dots = ctx.saved_tensors
        B, D, H, W = x.shape
        grad = torch.empty_like(x)
        gs = _grad_scale(g_loss)
        L.check(L.lib().nmsa_loss_cos_emb_bwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(idx), L.ptr(lt), B, D, H, W, lt.shape[1],
            L.ptr(gs), L.ptr(dots) if ctx.has_dots and grad.data_ptr() % 16 == 0 else None,
            L.ptr(grad), L.stream_ptr(x.device)), 'nmsa_loss_cos_emb_bwd')
        return grad, None, None


def cross_entropy_sum(logits, target, weights=None, label_smoothing=0.0, expected_scale=None
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    return CrossEntropyFunction.apply(logits, target, weights, label_smoothing, expected_scale)


_KINDS = {'mse': 0, 'l1': 1, 'focal': 2}


class ElementwiseNoneFunction(torch.autograd.Function):
    """per-element (pred - target)^2 / |pred - target| (reduction='none', reference
    loss/mse.py:21-41, l1.py:21-41): csrc/losses_forms.hip k_elem_none"""

    @staticmethod
    def forward(ctx, pred, target, kind):
        x = L.require_device_tensor(pred, 'input_')
        dev = x.device
        y = target.to(dev, torch.float32).contiguous()
        if y.shape != x.shape:
            y = y.expand_as(x).contiguous()
        # the reference's result has the promoted type of its two operands
        out_dtype = torch.result_type(pred, target)
        if out_dtype not in (torch.float32, x.dtype):
            out_dtype = torch.float32
        out = torch.empty(x.shape, dtype=out_dtype, device=dev)
        L.check(L.lib().nmsa_loss_elementwise_none_fwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), x.numel(), int(kind), L.float_dtype_code(out),
            L.ptr(out), L.stream_ptr(dev)), 'nmsa_loss_elementwise_none_fwd')
        ctx.save_for_backward(x, y)
        ctx.kind = int(kind)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = g.contiguous()
        grad = torch.empty_like(x)
        L.check(L.lib().nmsa_loss_elementwise_none_bwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), x.numel(), ctx.kind, L.float_dtype_code(g),
            L.ptr(g), L.ptr(grad), L.stream_ptr(x.device)), 'nmsa_loss_elementwise_none_bwd')
        return grad, None, None


class CosineRowsFunction(torch.autograd.Function):
    """per-row cosine embedding loss of [N, D] rows with labels +1 / -1 (reference
    loss/cos_emb.py:21-56 with `target_similarity`; margin 0): csrc/losses_forms.hip k_cos_rows"""

    @staticmethod
    def forward(ctx, input_, target, labels):
        x = L.require_device_tensor(input_, 'input_')
        dev = x.device
        n, d = x.shape
        y = target.to(dev, torch.float32).contiguous()
        lab = None if labels is None else labels.to(dev, torch.float32).contiguous()
        if lab is not None and lab.numel() != n:
            raise ValueError(f'target_similarity holds {lab.numel()} labels for {n} rows')
        rows = torch.empty((n,), dtype=torch.float32, device=dev)
        L.check(L.lib().nmsa_loss_cos_rows_fwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), None if lab is None else L.ptr(lab), n, d, 0.0,
            L.ptr(rows), L.stream_ptr(dev)), 'nmsa_loss_cos_rows_fwd')
        ctx.save_for_backward(x, y, lab if lab is not None else torch.empty(0, device=dev))
        ctx.has_labels = lab is not None
        return rows

    @staticmethod
    def backward(ctx, g):
        x, y, lab = ctx.saved_tensors
        n, d = x.shape
        # `rows.sum()` hands over one value for every row (a stride-0 view): passed as a scalar
        scalar = g.numel() > 0 and g.stride(0) == 0
        up = (g.reshape(-1)[:1] if scalar else g).to(torch.float32).contiguous()
        grad = torch.empty_like(x)
        L.check(L.lib().nmsa_loss_cos_rows_bwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(lab) if ctx.has_labels else None, n, d, 0.0,
            L.ptr(up), int(scalar), L.ptr(grad), L.stream_ptr(x.device)), 'nmsa_loss_cos_rows_bwd')
        return grad, None, None


class VonMisesRowsFunction(torch.autograd.Function):
    """per-row 1 - exp(kappa (x.y - 1)) of biternion rows [N, 2] (reduction='none', reference
    loss/vonmises.py:27-51): csrc/losses_forms.hip k_vm_rows"""

    @staticmethod
    def forward(ctx, input_, target, kappa):
        x = L.require_device_tensor(input_, 'input_')
        dev = x.device
        n = x.shape[0]
        y = target.to(dev, torch.float32).contiguous()
        rows = torch.empty((n, 1), dtype=torch.float32, device=dev)
        L.check(L.lib().nmsa_loss_vonmises_rows_fwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), n, float(kappa), L.ptr(rows), L.stream_ptr(dev)),
            'nmsa_loss_vonmises_rows_fwd')
        ctx.save_for_backward(x, y)
        ctx.kappa = float(kappa)
        return rows

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        up = g.to(torch.float32).contiguous()
        grad = torch.empty_like(x)
        L.check(L.lib().nmsa_loss_vonmises_rows_bwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), x.shape[0], ctx.kappa, L.ptr(up), L.ptr(grad),
            L.stream_ptr(x.device)), 'nmsa_loss_vonmises_rows_bwd')
        return grad, None, None


def vonmises_rows(input_, target, kappa: float) -> torch.Tensor:
    return VonMisesRowsFunction.apply(input_, target, kappa)


def elementwise_none(pred, target, kind: str) -> torch.Tensor:
    return ElementwiseNoneFunction.apply(pred, target, _KINDS[kind])


def cosine_embedding_rows(input_, target, labels=None) -> torch.Tensor:
    return CosineRowsFunction.apply(input_, target, labels)


def masked_elementwise_sum(pred, target, mask, kind: str, expected_scale=None
                           ) -> Tuple[torch.Tensor, torch.Tensor]:
    return MaskedElementwiseFunction.apply(pred, target, mask, _KINDS[kind], expected_scale)


def vonmises_sum(pred, target, mask, kappa: float = 1.0, expected_scale=None
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    return VonMisesFunction.apply(pred, target, mask, kappa, expected_scale)


def wants_gradient(pred: torch.Tensor) -> bool:
    return torch.is_grad_enabled() and pred.requires_grad


def ce_forward_can_write_gradient(logits: torch.Tensor) -> bool:
    """the library has a forward kernel that also writes the gradient for these logits
    (register-resident column up to C = 48, two walks in one launch above)"""
    return logits.is_cuda and logits.ndim == 4 and bool(
        L.lib().nmsa_loss_ce_fwd_grad_supported(L.float_dtype_code(logits), logits.shape[1]))


def cosine_embedding_lut_sum(pred, indices, lut) -> Tuple[torch.Tensor, torch.Tensor]:
    return CosineEmbeddingLutFunction.apply(pred, indices, lut)
