"""autograd Functions over the HIP loss kernels (csrc/losses.hip).

Forward returns device scalars (loss sum as fp32 0-d tensor, counts as int64 0-d
tensors: no `.item()` host sync, unlike reference loss/ce.py:50 and
task_helper/instance.py:138-139); backward recomputes from the saved inputs and
scales by the upstream gradient on the device.
"""
import os
from typing import Dict, Optional, Tuple

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
_STATUS: Dict[torch.device, torch.Tensor] = {}
_CHECK_EVERY_CALL = bool(int(os.environ.get('NMSA_CHECK_STATUS', '0') or 0))


def _status_word(dev: torch.device) -> torch.Tensor:
    st = _STATUS.get(dev)
    if st is None:
        st = _STATUS[dev] = torch.zeros((1,), dtype=torch.int32, device=dev)
    return st


def check_loss_status() -> None:
    """raise IndexError if any loss kernel since the last check saw a label >= C + 1 or a LUT
    index outside [0, L] (host sync)"""
    for dev, st in _STATUS.items():
        v = int(st.item())
        if v:
            st.zero_()
            raise IndexError(f'loss kernels on {dev}: target label / LUT index out of range '
                             '(PyTorch raises a device-side assert for these)')


def _grad_scale(g: torch.Tensor) -> torch.Tensor:
    return g.detach().to(torch.float32).reshape(1).contiguous()


class CrossEntropyFunction(torch.autograd.Function):
    """sum over non-void px of the (weighted, label-smoothed) CE; also n and sum w[label]."""

    @staticmethod
    def forward(ctx, logits, target, weights, label_smoothing):
        x = L.require_device_tensor(logits, 'input_')
        B, C, H, W = x.shape
        dev = x.device
        if C > 255:
            # labels travel as uint8 (0 = void, 1..C; ToTorchTensors keeps semantic uint8):
            # a wider label would wrap silently
            raise ValueError(f'{C} classes: the CE kernel takes uint8 labels (C <= 255)')
        t = target.to(dev)
        if t.dtype != torch.uint8:
            # labels 0..C (0 = void), reference: target.long() - 1.  Out-of-range values must
            # not wrap into valid labels: they become 255 (> C) and set the status bit
            t = torch.where((t < 0) | (t > 255), 255, t).to(torch.uint8) \
                if t.dtype != torch.bool else t.to(torch.uint8)
        t = t.contiguous()
        w = None if weights is None else weights.to(dev, torch.float32).contiguous()
        s, n = _scalar_outputs(dev)
        wsum = torch.empty((1,), dtype=torch.float64, device=dev)
        status = _status_word(dev)
        ws, nbytes = _workspace(B, H, W, dev)
        # per-pixel log-sum-exp for the backward pass (4 B/px instead of a second read of the
        # logits), only when a gradient can be asked for
        lse2 = torch.empty((B, H, W), dtype=torch.float32, device=dev) \
            if ctx.needs_input_grad[0] else None
        L.check(L.lib().nmsa_loss_ce_fwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(t), L.ptr(w), B, C, H, W,
            float(label_smoothing), L.ptr(s), L.ptr(n), L.ptr(wsum), L.ptr(lse2), L.ptr(status),
            L.ptr(ws), nbytes, L.stream_ptr(dev)), 'nmsa_loss_ce_fwd')
        if _CHECK_EVERY_CALL:
            check_loss_status()
        ctx.save_for_backward(x, t, w if w is not None else torch.empty(0, device=dev),
                              lse2 if lse2 is not None else torch.empty(0, device=dev))
        ctx.has_lse = lse2 is not None
        ctx.has_w = w is not None
        ctx.ls = float(label_smoothing)
        loss = s[0].to(torch.float32)
        n_el = n[0]
        wsum_ = wsum[0]
        ctx.mark_non_differentiable(n_el, wsum_)
        return loss, n_el, wsum_

    @staticmethod
    def backward(ctx, g_loss, g_n, g_w):
        x, t, w, lse2 = ctx.saved_tensors
        B, C, H, W = x.shape
        grad = torch.empty_like(x)
        gs = _grad_scale(g_loss)
        L.check(L.lib().nmsa_loss_ce_bwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(t), L.ptr(w) if ctx.has_w else None,
            B, C, H, W, ctx.ls, L.ptr(gs), L.ptr(lse2) if ctx.has_lse else None, L.ptr(grad),
            L.stream_ptr(x.device)), 'nmsa_loss_ce_bwd')
        return grad, None, None, None


class MaskedElementwiseFunction(torch.autograd.Function):
    """sum_px mean_c f(pred*mask - target) and n = sum(mask); kind 0 = MSE, 1 = L1."""

    @staticmethod
    def forward(ctx, pred, target, mask, kind):
        x = L.require_device_tensor(pred, 'input_')
        dev = x.device
        if x.ndim == 3:
            B, H, W = x.shape
            C = 1
        else:
            B, C, H, W = x.shape
        y = target.to(dev, torch.float32).contiguous()
        m = _u8(None if mask is None else mask.to(dev))
        s, n = _scalar_outputs(dev)
        ws, nbytes = _workspace(B, H, W, dev)
        L.check(L.lib().nmsa_loss_masked_fwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m), B, C, H, W, int(kind),
            L.ptr(s), L.ptr(n), L.ptr(ws), nbytes, L.stream_ptr(dev)), 'nmsa_loss_masked_fwd')
        ctx.save_for_backward(x, y, m if m is not None else torch.empty(0, device=dev))
        ctx.has_m = m is not None
        ctx.kind = int(kind)
        ctx.dims = (B, C, H, W)
        loss = s[0].to(torch.float32)
        n_el = n[0]
        ctx.mark_non_differentiable(n_el)
        return loss, n_el

    @staticmethod
    def backward(ctx, g_loss, g_n):
        x, y, m = ctx.saved_tensors
        B, C, H, W = ctx.dims
        grad = torch.empty_like(x)
        gs = _grad_scale(g_loss)
        L.check(L.lib().nmsa_loss_masked_bwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m) if ctx.has_m else None,
            B, C, H, W, ctx.kind, L.ptr(gs), L.ptr(grad), L.stream_ptr(x.device)),
            'nmsa_loss_masked_bwd')
        return grad, None, None, None


class VonMisesFunction(torch.autograd.Function):
    """sum over masked px of 1 - exp(kappa (x.y - 1)); pred/target planar [B,2,H,W]."""

    @staticmethod
    def forward(ctx, pred, target, mask, kappa):
        x = L.require_device_tensor(pred, 'input_')
        dev = x.device
        B, two, H, W = x.shape
        assert two == 2
        y = target.to(dev, torch.float32).contiguous()
        m = _u8(None if mask is None else mask.to(dev))
        s, n = _scalar_outputs(dev)
        ws, nbytes = _workspace(B, H, W, dev)
        L.check(L.lib().nmsa_loss_vonmises_fwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m), B, H, W, float(kappa),
            L.ptr(s), L.ptr(n), L.ptr(ws), nbytes, L.stream_ptr(dev)), 'nmsa_loss_vonmises_fwd')
        ctx.save_for_backward(x, y, m if m is not None else torch.empty(0, device=dev))
        ctx.has_m = m is not None
        ctx.kappa = float(kappa)
        loss = s[0].to(torch.float32)
        n_el = n[0]
        ctx.mark_non_differentiable(n_el)
        return loss, n_el

    @staticmethod
    def backward(ctx, g_loss, g_n):
        x, y, m = ctx.saved_tensors
        B, _, H, W = x.shape
        grad = torch.empty_like(x)
        gs = _grad_scale(g_loss)
        L.check(L.lib().nmsa_loss_vonmises_bwd(
            L.ptr(x), L.float_dtype_code(x), L.ptr(y), L.ptr(m) if ctx.has_m else None,
            B, H, W, ctx.kappa, L.ptr(gs), L.ptr(grad), L.stream_ptr(x.device)),
            'nmsa_loss_vonmises_bwd')
        return grad, None, None, None


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


def cross_entropy_sum(logits, target, weights=None, label_smoothing=0.0
                      ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    return CrossEntropyFunction.apply(logits, target, weights, label_smoothing)


def masked_elementwise_sum(pred, target, mask, kind: str) -> Tuple[torch.Tensor, torch.Tensor]:
    return MaskedElementwiseFunction.apply(pred, target, mask, {'mse': 0, 'l1': 1, 'focal': 2}[kind])


def vonmises_sum(pred, target, mask, kappa: float = 1.0) -> Tuple[torch.Tensor, torch.Tensor]:
    return VonMisesFunction.apply(pred, target, mask, kappa)


def cosine_embedding_lut_sum(pred, indices, lut) -> Tuple[torch.Tensor, torch.Tensor]:
    return CosineEmbeddingLutFunction.apply(pred, indices, lut)
