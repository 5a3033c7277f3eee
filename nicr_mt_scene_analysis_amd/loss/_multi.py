"""All losses of a task helper in ONE call of the C ABI (`nmsa_multitask_loss_fwd_grad`,
csrc/losses.hip k_multi_*; reference task_helper/instance.py:92-269, semantic.py:57-90,
base.py:161-182).

Every (loss, supervision scale) pair is an ITEM, the items whose sums the caller adds and divides
by their summed element counts form a TOTAL.  The forward call counts the labels / mask bytes,
forms per total the divisor and the EXPECTED upstream gradient `w / n` of its loss sums, computes
all sums and writes all gradients for that expectation (3 launches whatever the number of items:
the count's last workgroup forms the expectation; 2 without gradients);
backward compares the real upstream gradients with the expectation ON THE DEVICE and recomputes
only what differs (1 launch: workgroup 0 of the recomputing walk keeps the records).  `w` — the factor the trainer multiplies the total with before
`backward()`: loss weights (reference loss_weighting/fixed.py:28-37), an AMP scale — lives in a
`SpecState` owned by the caller (a task helper, a loss instance) and is LEARNED on the device from
the upstream gradients it sees: constant factors are confirmed from the second step on without any
hint, factors that keep changing switch the expectation of that total off (the forward pass then
writes no gradient and backward recomputes, as without speculation).  There is no process-global
policy state.
"""
import ctypes as C
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import _lib as L

KIND = {'ce': 0, 'mse': 1, 'l1': 2, 'focal': 3, 'vonmises': 4, 'cos': 5}
MAX_ITEMS, MAX_TOTALS = 16, 8


class _Item(C.Structure):
    _fields_ = [('kind', C.c_int32), ('dtype', C.c_int32), ('B', C.c_int32), ('C', C.c_int32),
                ('H', C.c_int32), ('W', C.c_int32), ('total', C.c_int32), ('clamp_count', C.c_int32),
                ('param', C.c_float), ('reserved', C.c_int32),
                ('pred', C.c_void_p), ('target', C.c_void_p), ('mask', C.c_void_p),
                ('weights', C.c_void_p), ('grad', C.c_void_p)]


_STATES: 'weakref.WeakSet[SpecState]' = weakref.WeakSet()


class SpecState:
    """device-resident spec records of the totals of one caller: int32 [n_totals + 1, 8]
    ([0] confirmed, [1] recomputed, [2] w as fp32 bits, [5] flags; see csrc/losses_multi.hip);
    the last row is scratch of the calls (the tickets of their launches), zero between calls"""

    def __init__(self, n_totals: int, initial_weight=1.0) -> None:
        assert 1 <= n_totals <= MAX_TOTALS
        self.n_totals = n_totals
        self._w0 = [float(initial_weight)] * n_totals if not isinstance(initial_weight, (list, tuple)) \
            else [float(v) for v in initial_weight]
        self._rec: Dict[torch.device, torch.Tensor] = {}
        _STATES.add(self)

    def records(self, dev) -> torch.Tensor:
        dev = torch.device(dev)
        if dev.type == 'cuda' and dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        r = self._rec.get(dev)
        if r is None:
            host = torch.zeros((self.n_totals + 1, 8), dtype=torch.int32)
            host.view(torch.float32)[:self.n_totals, 2] = torch.tensor(self._w0, dtype=torch.float32)
            r = self._rec[dev] = host.to(dev)
        return r

    def reset(self) -> None:
        for dev in list(self._rec):
            del self._rec[dev]

    def stats(self) -> Dict[str, int]:
        """host sync"""
        out = {'confirmed': 0, 'recomputed': 0}
        for r in self._rec.values():
            v = r[:self.n_totals, :2].sum(dim=0).tolist()
            out['confirmed'] += int(v[0])
            out['recomputed'] += int(v[1])
        return out

    def weights(self, dev) -> List[float]:
        """the learned upstream factors (host sync; tests, diagnostics)"""
        return self.records(dev).view(torch.float32)[:self.n_totals, 2].tolist()

# where a forward-written gradient lives (tools/diag_cos_relalign.py swaps it to place the buffer)
_alloc_grad = torch.empty_like


def reset_all() -> None:
    """forget what every live SpecState has learned (new records on next use)"""
    for st in list(_STATES):
        st.reset()


def supported(items: Sequence[dict]) -> bool:
    """every item can go through the multi-loss call (device tensors, C <= 256 for CE)"""
    if not 1 <= len(items) <= MAX_ITEMS:
        return False
    for it in items:
        p = it['pred']
        if not (isinstance(p, torch.Tensor) and p.is_cuda and p.dtype in
                (torch.float32, torch.bfloat16, torch.float16) and p.numel() > 0):
            return False
        if it['kind'] == 'ce' and p.shape[1] > 255:
            return False
        if it['kind'] == 'cos' and not cos_supported(p, it['target']):
            return False
    return True


def cos_supported(pred: torch.Tensor, lut: torch.Tensor) -> bool:
    """a one-pass cosine kernel (csrc/losses_cos.hip) takes this prediction / LUT: whole groups of
    4 (f32: 2) pixels, D <= 1024 (k_cos_parts: the column over cooperating workgroups; any D, a
    ragged last wave when D % 64 != 0) or D % 64 == 0, D <= 512 with the image's LUT within the LDS
    of a CU (k_cos_split)"""
    if not (pred.is_cuda and pred.ndim == 4 and lut.ndim == 3 and pred.dtype in
            (torch.float32, torch.bfloat16, torch.float16)):
        return False
    B, D, H, W = pred.shape
    pxt = 2 if pred.dtype == torch.float32 else 4
    if (H * W) % pxt or lut.shape[0] != B or lut.shape[2] != D:
        return False
    return bool(L.lib().nmsa_loss_cos_emb_fwd_grad_supported(L.float_dtype_code(pred), D, H, W, lut.shape[1]))


def _u8(t: Optional[torch.Tensor], dev) -> Optional[torch.Tensor]:
    if t is None:
        return None
    t = t.to(dev)
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)
    if t.dtype == torch.uint8:
        return t.contiguous()
    return (t != 0).contiguous().view(torch.uint8)


class MultiLossFunction(torch.autograd.Function):
    """apply(desc, *preds) -> (sums [n], item_losses [n], total_losses [T]) float32: the loss sums,
    sum / count per item and sum(sums) / divisor per total (csrc k_multi_finalize), three views
    of one vector; counts / aux / divisors are left on `desc`"""

    @staticmethod
    def forward(ctx, desc, *preds):
        items, n_totals, spec = desc['items'], desc['n_totals'], desc['spec']
        dev = preds[0].device
        n = len(items)
        ctx.set_materialize_grads(False)       # unused outputs come back as None, not as zero tensors
        want_grad = desc.get('grad_enabled', True)
        arr = (_Item * n)()
        keep = []
        grads: List[Optional[torch.Tensor]] = []
        from ._functional import _status_word, labels_u8
        for i, (it, x) in enumerate(zip(items, preds)):
            x = L.require_device_tensor(x, 'input_')
            kind = it['kind']
            if x.ndim == 3:
                B, H, W = x.shape
                Cc = 1
            else:
                B, Cc, H, W = x.shape
            a = arr[i]
            a.kind, a.dtype = KIND[kind], L.float_dtype_code(x)
            a.B, a.C, a.H, a.W = B, Cc, H, W
            a.total, a.clamp_count = int(it['total']), int(it.get('clamp', 0))      # 0 | 1 (True) | 2: item loss only
            a.param = float(it.get('param', 0.0))
            tgt, msk, wts = it.get('target'), it.get('mask'), it.get('weights')
            if kind == 'ce':
                msk = labels_u8(msk, dev)
                tgt = None
                wts = None if wts is None else wts.to(dev, torch.float32).contiguous()
            elif kind == 'cos':
                # target = per-image LUT [B, L, D], mask = indices [B, H, W] (0 = no target)
                tgt = tgt.to(dev, torch.float32).contiguous()
                msk = msk.to(dev, torch.int32).contiguous()
                a.reserved = int(tgt.shape[1])
                wts = None
            else:
                tgt = tgt.to(dev, torch.float32).contiguous()
                msk = _u8(msk, dev)
            # (grad mode is always off inside forward(): desc['grad_enabled'] is the caller's grad
            # mode, needs_input_grad says which inputs autograd is going to ask a gradient for)
            g = _alloc_grad(x) if want_grad and ctx.needs_input_grad[i + 1] else None
            a.pred, a.target = x.data_ptr(), (tgt.data_ptr() if tgt is not None else None)
            a.mask = msk.data_ptr() if msk is not None else None
            a.weights = wts.data_ptr() if wts is not None else None
            a.grad = g.data_ptr() if g is not None else None
            keep.append((x, tgt, msk, wts))
            grads.append(g)
        rec = spec.records(dev)
        # one allocation for the small outputs: sums f64 [n] | aux f64 [n] | counts i64 [n]
        small = torch.empty((3 * n,), dtype=torch.float64, device=dev)
        sums, aux, counts = small[:n], small[n:2 * n], small[2 * n:].view(torch.int64)
        expect = torch.empty((n_totals, 2), dtype=torch.float32, device=dev)
        out = torch.empty((2 * n + n_totals,), dtype=torch.float32, device=dev)
        status = _status_word(dev)
        lib = L.lib()
        nbytes = lib.nmsa_multitask_loss_workspace_bytes(arr, n)
        if nbytes == 0:
            raise L.NmsaError('nmsa_multitask_loss_workspace_bytes: invalid items')
        ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
        L.check(lib.nmsa_multitask_loss_fwd_grad(arr, n, n_totals, L.ptr(rec), L.ptr(expect), L.ptr(sums),
                                                 L.ptr(counts), L.ptr(aux), L.ptr(out), L.ptr(status),
                                                 L.ptr(ws), nbytes, L.stream_ptr(dev)),
                'nmsa_multitask_loss_fwd_grad')
        # the predictions through save_for_backward: an in-place change between forward and a
        # recomputing backward is caught by autograd's version check (the kernels read them again)
        ctx.save_for_backward(*preds)
        ctx.arr, ctx.keep, ctx.grads = arr, keep, grads
        ctx.has_grad = [g is not None for g in grads]
        ctx.n_totals, ctx.rec, ctx.expect, ctx.counts = n_totals, rec, expect, counts
        # backward borrows the workspace (the granule exchange of wide cosine columns); only kept
        # alive when a cosine item exists
        ctx.ws = ws if any(it['kind'] == 'cos' for it in items) else None
        desc['counts'], desc['aux'], desc['divisors'], desc['packed'] = counts, aux, expect[:, 1], out
        return out[:n], out[n:2 * n], out[2 * n:]

    @staticmethod
    def backward(ctx, g_sums, g_items, g_totals):
        dev = ctx.expect.device
        # (dropping the context's reference lets autograd take the buffers as .grad instead of
        # cloning them: a logits-sized copy per prediction otherwise)
        grads, ctx.grads = ctx.grads, None
        expect = ctx.expect
        rec = ctx.rec
        ctx.saved_tensors                      # version check of the predictions
        if grads is None:
            # a second backward pass through the same graph (retain_graph=True): the buffers of
            # the first one belong to autograd now -> fresh ones, and no expectation to confirm
            grads = [torch.empty_like(k[0]) if has else None for k, has in zip(ctx.keep, ctx.has_grad)]
            for a, gbuf in zip(ctx.arr, grads):
                a.grad = gbuf.data_ptr() if gbuf is not None else None
            # (and nothing was predicted: the caller's record and the tally are left alone)
            expect = ctx.expect.clone()
            expect[:, 0] = float('nan')
            rec = None
        n = len(grads)
        up = [None if g is None else g.detach().to(torch.float32).contiguous()
              for g in (g_sums, g_items, g_totals)]
        gs = torch.empty((n,), dtype=torch.float32, device=dev)
        from ._functional import _counters_ptr
        L.check(L.lib().nmsa_multitask_loss_bwd_unless(
            ctx.arr, n, ctx.n_totals, *(None if g is None else L.ptr(g) for g in up), L.ptr(ctx.counts),
            L.ptr(expect), L.ptr(rec), L.ptr(gs), None if rec is None else _counters_ptr(dev),
            L.ptr(ctx.ws), 0 if ctx.ws is None else ctx.ws.numel(), L.stream_ptr(dev)),
            'nmsa_multitask_loss_bwd_unless')
        return (None, *grads)


class MultiLossResult:
    """what one call returns: `sums` [n] float32 loss sums, `item_losses` [n] = sum / count,
    `total_losses` [T] = sum over the total's items / divisor (all three differentiable),
    `counts` int64 [n], `aux` float64 [n], `divisors` float32 [T], `packed`: the detached
    [2 n + T] vector the three are views of"""
    __slots__ = ('sums', 'item_losses', 'total_losses', 'counts', 'aux', 'divisors', 'packed')

    def __init__(self, outs, desc):
        self.sums, self.item_losses, self.total_losses = outs
        self.counts, self.aux, self.divisors = desc['counts'], desc['aux'], desc['divisors']
        self.packed = desc['packed']


def multi_loss(items: Sequence[dict], n_totals: int, spec: SpecState) -> MultiLossResult:
    """items: dicts with kind ('ce' | 'mse' | 'l1' | 'focal' | 'vonmises' | 'cos'), pred, target (not
    CE; the LUT [B, L, D] for 'cos'), mask (labels for CE, int indices for 'cos'), weights (CE),
    param (label smoothing | kappa), total, clamp (True / 1: max(count, 1) divides the item's loss
    and enters the total's divisor; 2: the item's loss only)."""
    desc = {'items': list(items), 'n_totals': n_totals, 'spec': spec, 'grad_enabled': torch.is_grad_enabled()}
    outs = MultiLossFunction.apply(desc, *[it['pred'] for it in items])
    return MultiLossResult(outs, desc)
