"""The C ABI never throws and never launches on bad arguments: every entry point answers a
negative NMSA_ERR_* code (include/nmsa.h) for null pointers, impossible shapes, unknown dtype
codes and short workspaces — the Python wrappers turn those into the reference's exception
types.  Runs on the GPU box (valid calls in between prove that the process stays healthy)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ERR_ARG, ERR_WORKSPACE = -1, -3


@pytest.fixture(scope='module')
def env():
    from nicr_mt_scene_analysis_amd import _lib as L
    dev = torch.device('cuda')
    B, C, H, W = 2, 5, 16, 32
    t = {
        'logits': torch.randn((B, C, H, W), device=dev),
        'center': torch.rand((B, H, W), device=dev),
        'offset': torch.zeros((B, 2, H, W), device=dev),
        'fg': torch.ones((B, H, W), dtype=torch.uint8, device=dev),
        'u8': torch.zeros((B, H, W), dtype=torch.uint8, device=dev),
        'u8b': torch.zeros((B, H, W), dtype=torch.uint8, device=dev),
        'i64': torch.zeros((B, H, W), dtype=torch.int64, device=dev),
        'i64b': torch.zeros((B, H, W), dtype=torch.int64, device=dev),
        'cyx': torch.zeros((B, 256, 2), dtype=torch.int32, device=dev),
        'n': torch.zeros((B,), dtype=torch.int32, device=dev),
        'scores': torch.zeros((B, 256), dtype=torch.float32, device=dev),
        'thing': torch.ones((C,), dtype=torch.uint8, device=dev),
        'votes': torch.zeros((B, 256, C + 1), dtype=torch.int32, device=dev),
        'f64': torch.zeros((4, 8), dtype=torch.float64, device=dev),
        'status': torch.zeros((1,), dtype=torch.int32, device=dev),
        'ws': torch.zeros((1 << 20,), dtype=torch.uint8, device=dev),
    }
    return L, L.lib(), dev, (B, C, H, W), t


def test_strerror_names_every_code(env):
    L, lib, *_ = env
    for code in (0, -1, -2, -3, -4):
        assert lib.nmsa_strerror(code).decode()
    assert lib.nmsa_version() > 0


def test_bad_arguments_return_codes(env):
    L, lib, dev, (B, C, H, W), t = env
    p, st = L.ptr, L.stream_ptr(dev)
    ws_nms = lib.nmsa_center_nms_workspace_bytes(B, H, W)

    def nms(center=t['center'], ksize=3, topk=4, h=H, w=W, ws_bytes=ws_nms, max_centers=256):
        return lib.nmsa_center_nms_topk(p(center), None, B, h, w, 0.1, ksize, topk, 0, max_centers,
                                        p(t['cyx']), p(t['n']), p(t['scores']), None, p(t['ws']),
                                        ws_bytes, st)
    assert nms() == 0
    assert nms(center=None) == ERR_ARG
    assert nms(ksize=4) == ERR_ARG                       # instance.py:37: odd kernel sizes only
    assert nms(topk=0) == ERR_ARG
    assert nms(topk=H * W + 1) == ERR_ARG                # torch.topk would raise
    assert nms(h=0) == ERR_ARG
    assert nms(ws_bytes=8) == ERR_WORKSPACE

    def argmax(logits=t['logits'], dtype=0, c=C):
        return lib.nmsa_semantic_argmax(p(logits), dtype, B, c, H, W, p(t['u8']), None, None, st)
    assert argmax() == 0
    assert argmax(logits=None) == ERR_ARG
    assert argmax(dtype=7) == ERR_ARG
    assert argmax(c=0) == ERR_ARG
    assert argmax(c=300) == ERR_ARG                      # u8 class map: at most 256 classes

    def fused(max_centers=256, logits=t['logits']):
        return lib.nmsa_panoptic_fused(p(logits), 0, p(t['offset']), p(t['cyx']), p(t['n']),
                                       p(t['thing']), B, C, H, W, max_centers, float(H), float(W), 0,
                                       0.0, p(t['u8']), p(t['u8b']), None, None, p(t['votes']), 0, 0, st)
    assert fused() == 0
    assert fused(logits=None) == ERR_ARG
    assert fused(max_centers=1 << 16) == ERR_ARG         # center table would not fit the LDS

    def nearest(h=H, w=W, ho=8, wo=8, elem=3):
        return lib.nmsa_resize_nearest(p(t['i64']), elem, B, H, W, 0, 0, h, w, ho, wo, p(t['i64b']), st)
    assert nearest() == 0
    assert nearest(h=H + 1) == ERR_ARG                   # crop outside the plane
    assert nearest(ho=0) == ERR_ARG
    assert nearest(elem=42) == ERR_ARG

    cm = torch.zeros((C, C), dtype=torch.int64, device=dev)
    ws_cm = lib.nmsa_confmat_workspace_bytes(C)
    cm_ws = torch.zeros((ws_cm,), dtype=torch.uint8, device=dev)

    def confmat(n_classes=C, ws_bytes=ws_cm, preds=t['u8']):
        return lib.nmsa_confmat_update(p(preds), 0, 1, p(t['u8b']), 0, B * H * W, n_classes, 0,
                                       p(cm), p(t['status']), p(cm_ws), ws_bytes, st)
    assert confmat() == 0
    assert confmat(preds=None) == ERR_ARG
    assert confmat(n_classes=0) == ERR_ARG
    assert confmat(ws_bytes=8) == ERR_WORKSPACE

    ws_pq = lib.nmsa_pq_workspace_bytes(B, H, W, C)
    pq_ws = torch.zeros((ws_pq,), dtype=torch.uint8, device=dev)

    def pq(num_categories=C, ws_bytes=ws_pq, offset=256 ** 3):
        f = t['f64']
        return lib.nmsa_pq_update(p(t['i64']), p(t['i64b']), B, H, W, num_categories, 0, 1 << 16, offset,
                                  0, p(f[0]), p(f[1]), p(f[2]), p(f[3]), None, 0, None, p(t['status']),
                                  p(pq_ws), ws_bytes, 0, st)
    assert pq() == 0
    assert pq(num_categories=0) == ERR_ARG
    assert pq(offset=0) == ERR_ARG
    assert pq(ws_bytes=16) == ERR_WORKSPACE
    torch.cuda.synchronize()                             # nothing faulted on the way


def test_wrappers_raise_the_reference_exception_types(env):
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    L, lib, dev, (B, C, H, W), t = env
    with pytest.raises(L.NmsaError):                     # CPU tensors: no silent CPU fallback
        ops.semantic_argmax(t['logits'].cpu())
    with pytest.raises(L.NmsaError):
        ops.center_nms_topk(t['center'], top_k=H * W + 1)
    with pytest.raises(AssertionError):
        get_postprocessing_class('instance')(heatmap_nms_kernel_size=2)
    with pytest.raises(ValueError):
        get_postprocessing_class('nope')


def test_forward_written_gradient_entry_points_return_codes(env):
    """nmsa_count_u8 / nmsa_loss_*_fwd_grad / nmsa_loss_*_bwd_unless: null pointers, unknown
    dtype codes, short workspaces, impossible class counts"""
    L, lib, dev, (B, C, H, W), t = env
    p, st = L.ptr, L.stream_ptr(dev)
    x = t['logits']
    grad = torch.empty_like(x)
    labels = torch.randint(0, C + 1, (B, H, W), dtype=torch.uint8, device=dev)
    one = torch.ones((1,), device=dev)
    s = torch.zeros((1,), dtype=torch.float64, device=dev)
    n = torch.zeros((1,), dtype=torch.int64, device=dev)
    ws_b = lib.nmsa_loss_workspace_bytes(B, H, W)
    cnt_b = lib.nmsa_count_workspace_bytes()
    assert cnt_b > 0

    def count(values=labels, nbytes=cnt_b, lo=1, hi=C, out=n):
        return lib.nmsa_count_u8(p(values), values.numel() if values is not None else 0, lo, hi,
                                 p(out), None, 1.0, p(t['ws']), nbytes, st)
    assert count() == 0 and int(n) == int((labels != 0).sum())
    assert lib.nmsa_count_u8(None, 5, 1, C, p(n), None, 1.0, p(t['ws']), cnt_b, st) == ERR_ARG
    assert lib.nmsa_count_u8(None, 0, 1, C, p(n), None, 1.0, p(t['ws']), cnt_b, st) == 0 and int(n) == 0
    assert count(out=None) == ERR_ARG
    assert count(lo=3, hi=2) == ERR_ARG
    assert count(hi=256) == ERR_ARG
    assert count(nbytes=cnt_b - 1) == ERR_WORKSPACE

    def ce(logits=x, dtype=0, c=C, expected=one, g=grad, nbytes=ws_b):
        return lib.nmsa_loss_ce_fwd_grad(p(logits), dtype, p(labels), None, B, c, H, W, 0.0,
                                         p(expected), p(s), p(n), None, p(g), p(t['status']),
                                         p(t['ws']), nbytes, st)
    assert ce() == 0
    assert ce(logits=None) == ERR_ARG
    assert ce(expected=None) == ERR_ARG
    assert ce(g=None) == ERR_ARG
    assert ce(dtype=7) == ERR_ARG
    assert ce(c=0) == ERR_ARG
    assert ce(c=4097) == ERR_ARG
    assert ce(nbytes=ws_b - 1) == ERR_WORKSPACE
    assert lib.nmsa_loss_ce_fwd_grad_supported(0, 48) == 1
    assert lib.nmsa_loss_ce_fwd_grad_supported(0, 49) == 1     # two walks in one launch
    assert lib.nmsa_loss_ce_fwd_grad_supported(0, 4097) == 0
    assert lib.nmsa_loss_ce_fwd_grad_supported(9, 8) == 0

    def ce_bwd(computed_for=one, g=grad):
        return lib.nmsa_loss_ce_bwd_unless(p(x), 0, p(labels), None, B, C, H, W, 0.0, p(one), p(g),
                                           p(computed_for), None, st)
    assert ce_bwd() == 0
    assert ce_bwd(computed_for=None) == ERR_ARG
    assert ce_bwd(g=None) == ERR_ARG

    pred, tgt = t['offset'], torch.zeros_like(t['offset'])
    gp = torch.empty_like(pred)

    def masked(kind=1, expected=one, g=gp, nbytes=ws_b):
        return lib.nmsa_loss_masked_fwd_grad(p(pred), 0, p(tgt), p(t['fg']), B, 2, H, W, kind,
                                             p(expected), p(s), p(n), p(g), p(t['ws']), nbytes, st)
    assert masked() == 0
    assert masked(kind=3) == ERR_ARG
    assert masked(expected=None) == ERR_ARG
    assert masked(g=None) == ERR_ARG
    assert masked(nbytes=8) == ERR_WORKSPACE
    assert lib.nmsa_loss_masked_bwd_unless(p(pred), 0, p(tgt), p(t['fg']), B, 2, H, W, 1, p(one),
                                           p(gp), None, None, st) == ERR_ARG

    def vm(expected=one, g=gp, nbytes=ws_b):
        return lib.nmsa_loss_vonmises_fwd_grad(p(pred), 0, p(tgt), p(t['fg']), B, H, W, 1.0,
                                               p(expected), p(s), p(n), p(g), p(t['ws']), nbytes, st)
    assert vm() == 0
    assert vm(expected=None) == ERR_ARG
    assert vm(g=None) == ERR_ARG
    assert vm(nbytes=8) == ERR_WORKSPACE
    assert lib.nmsa_loss_vonmises_bwd_unless(p(pred), 0, p(tgt), p(t['fg']), B, H, W, 1.0, p(one),
                                             p(gp), None, None, st) == ERR_ARG
    torch.cuda.synchronize()


def test_multitask_loss_bad_arguments(env):
    """nmsa_multitask_loss_*: item lists the dispatcher cannot take answer a code, a valid call
    in between still works"""
    L, lib, dev, (B, C, H, W), t = env
    from nicr_mt_scene_analysis_amd.loss._multi import _Item
    p, st = L.ptr, L.stream_ptr(dev)
    labels = torch.randint(0, C + 1, (B, H, W), device=dev).to(torch.uint8)
    grad = torch.empty_like(t['logits'])
    spec = torch.zeros((2 + 1, 8), dtype=torch.int32, device=dev)        # (last row: the calls' tickets)
    spec.view(torch.float32)[:2, 2] = 1.0
    expect = torch.empty((2, 2), dtype=torch.float32, device=dev)
    small = torch.empty((6,), dtype=torch.float64, device=dev)
    out = torch.empty((5,), dtype=torch.float32, device=dev)

    def items(n=1, **over):
        arr = (_Item * n)()
        for a in arr:
            a.kind, a.dtype, a.B, a.C, a.H, a.W, a.total = 0, 0, B, C, H, W, 0
            a.pred, a.mask, a.grad = t['logits'].data_ptr(), labels.data_ptr(), grad.data_ptr()
            for k, v in over.items():
                setattr(a, k, v)
        return arr

    def fwd(arr, n=1, n_totals=1, ws_bytes=None, spec_=spec):
        need = lib.nmsa_multitask_loss_workspace_bytes(arr, n)
        return lib.nmsa_multitask_loss_fwd_grad(arr, n, n_totals, p(spec_), p(expect), p(small[:2]),
                                                p(small[4:].view(torch.int64)), p(small[2:4]), p(out),
                                                p(t['status']), p(t['ws']),
                                                need if ws_bytes is None else ws_bytes, st)
    good = items()
    assert fwd(good) == 0
    assert fwd(good, spec_=None) == ERR_ARG
    assert fwd(good, ws_bytes=16) == ERR_WORKSPACE
    assert fwd(good, n=0) == ERR_ARG
    assert fwd(items(17), n=17) == ERR_ARG                          # NMSA_MULTI_MAX_ITEMS = 16
    assert lib.nmsa_multitask_loss_workspace_bytes(items(17), 17) == 0
    assert fwd(items(total=1)) == ERR_ARG                           # total index outside n_totals
    assert fwd(items(total=-1)) == ERR_ARG
    assert fwd(items(dtype=7)) == ERR_ARG
    assert fwd(items(kind=9)) == ERR_ARG
    assert fwd(items(mask=None)) == ERR_ARG                         # a cross entropy needs its labels
    assert fwd(items(kind=1)) == ERR_ARG                            # MSE without a target
    assert fwd(items(kind=4, target=t['logits'].data_ptr())) == ERR_ARG     # von Mises needs 2 channels
    assert fwd(items(C=300)) != 0                                   # beyond the 256-class column split
    assert fwd(items(B=0)) == ERR_ARG
    gs = torch.empty((1,), dtype=torch.float32, device=dev)
    g_tot = torch.ones((1,), dtype=torch.float32, device=dev)
    counts = small[4:].view(torch.int64)

    def bwd(arr, counts_=counts, gs_=gs):
        return lib.nmsa_multitask_loss_bwd_unless(arr, 1, 1, None, None, p(g_tot), p(counts_), p(expect),
                                                  p(spec), p(gs_), None, None, 0, st)
    assert fwd(good) == 0 and bwd(good) == 0
    assert bwd(good, counts_=None) == ERR_ARG
    assert bwd(good, gs_=None) == ERR_ARG
    torch.cuda.synchronize()
    assert int(spec[0, 0]) + int(spec[0, 1]) == 1                   # exactly one backward pass was judged
    assert not spec[2].any()                                        # the tickets are zero again after every call
