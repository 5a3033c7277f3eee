"""Tensors with more than 2^31 elements: every plane offset in the kernels has to be 64-bit.

Size-independent property used here: images are independent units, so the results for the LAST
image of a > 2^31-element batch (whose offsets are beyond the int32 range) must equal, bit for
bit, the results of the same image processed alone (the small-size parity of that single-image
call against the oracle is covered by the other test files).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

B, C, H, W = 14, 160, 1024, 1024            # 2.35e9 logits


@pytest.fixture(scope='module')
def big():
    dev = torch.device('cuda')
    assert B * C * H * W > 2 ** 31
    g = torch.Generator(device=dev).manual_seed(7)
    coarse = torch.randn((B, C, H // 32, W // 32), device=dev, generator=g)
    logits = torch.empty((B, C, H, W), dtype=torch.bfloat16, device=dev)
    for b in range(B):
        up = torch.nn.functional.interpolate(coarse[b:b + 1], size=(H, W), mode='bilinear')
        logits[b] = (4 * up[0]).to(torch.bfloat16)
        del up
    yield logits
    del logits
    torch.cuda.empty_cache()


def test_argmax_and_softmax_beyond_int32(big):
    from nicr_mt_scene_analysis_amd import ops
    full = ops.semantic_argmax(big, want_u8=True, want_i64=True, want_score=True)
    for b in (0, B - 1):
        one = ops.semantic_argmax(big[b:b + 1], want_u8=True, want_i64=True, want_score=True)
        assert torch.equal(full['idx_u8'][b], one['idx_u8'][0])
        assert torch.equal(full['idx'][b], one['idx'][0])
        assert torch.equal(full['score'][b], one['score'][0])
    # spot check against ATen on the last image
    assert torch.equal(full['idx'][B - 1], big[B - 1].float().argmax(0))
    del full
    probs = ops.semantic_softmax(big)                       # [B,C,H,W] f32: 9.4 GB
    last = ops.semantic_softmax(big[B - 1:])
    assert torch.equal(probs[B - 1], last[0])
    del probs, last
    torch.cuda.empty_cache()


def test_pipeline_beyond_int32(big):
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn
    dev = big.device
    small = syn.make_panoptic_inputs_torch(B, 4, H, W, n_centers=32, seed=3, device=dev)
    is_thing = (torch.arange(C, device=dev) >= C // 2).to(torch.uint8)
    a = (small['instance_center'], small['instance_offset'])
    full = ops.panoptic_pipeline(big, a[0], a[1], is_thing, want_score=True)
    b = B - 1
    one = ops.panoptic_pipeline(big[b:], a[0][b:], a[1][b:], is_thing, want_score=True)
    for k in ('panoptic', 'instance', 'semantic_idx_u8', 'semantic_score', 'foreground'):
        assert torch.equal(full[k][b], one[k][0]), k
    n = int(one['n_ids'][0])
    assert n > 0 and int(full['n_ids'][b]) == n
    assert torch.equal(full['ids_pan'][b, :n], one['ids_pan'][0, :n])


def test_cross_entropy_beyond_int32(big):
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    dev = big.device
    g = torch.Generator(device=dev).manual_seed(11)
    target = torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8)
    x = big.detach().clone().requires_grad_(True)
    loss, n, _ = F_.cross_entropy_sum(x, target)
    loss.backward()
    b = B - 1
    x1 = big[b:].detach().clone().requires_grad_(True)
    loss1, _, _ = F_.cross_entropy_sum(x1, target[b:])
    loss1.backward()
    assert torch.equal(x.grad[b], x1.grad[0])               # 'sum' reduction: same scale
    assert int(n) == int((target != 0).sum())
    # the batch sum equals the sum of the per-image sums up to fp32 summation order
    parts = sum(float(F_.cross_entropy_sum(big[i:i + 1], target[i:i + 1])[0]) for i in range(B))
    assert abs(float(loss.detach()) - parts) <= 1e-5 * abs(parts)
    del x, x1
    torch.cuda.empty_cache()


def test_resize_beyond_int32():
    from nicr_mt_scene_analysis_amd import ops
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(5)
    src = torch.randn((4, 150, 480, 640), device=dev, generator=g)
    size = (1920, 2560)                                     # 4*150*1920*2560 = 2.9e9 outputs
    out = ops.resize_bilinear(src, size)
    last = ops.resize_bilinear(src[3:], size)
    assert out.numel() > 2 ** 31
    assert torch.equal(out[3], last[0])
    del out, last
    am = ops.semantic_argmax_resized(src, size, None, want_u8=False, want_i64=True, want_score=True)
    am1 = ops.semantic_argmax_resized(src[3:], size, None, want_u8=False, want_i64=True,
                                      want_score=True)
    assert torch.equal(am['idx'][3], am1['idx'][0])
    assert torch.equal(am['score'][3], am1['score'][0])
    ids = torch.randint(0, 1 << 20, (40, 1, 4096, 4096), device=dev, generator=g)   # i64 maps
    near = ops.resize_nearest(ids, (8192, 8192))            # 2.7e9 outputs
    assert near.numel() > 2 ** 31
    assert torch.equal(near[39], ops.resize_nearest(ids[39:], (8192, 8192))[0])
    del near, ids
    torch.cuda.empty_cache()


def test_metrics_beyond_int32():
    """2112 images of 1024x1024 (2.2e9 pixels): accumulators equal those of two half updates"""
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality
    dev = torch.device('cuda')
    n_img, n_cls, max_inst = 2112, 41, 1 << 16
    assert n_img * H * W > 2 ** 31
    g = torch.Generator(device=dev).manual_seed(21)

    def blocky(shift):
        cls = torch.randint(0, n_cls, (n_img, 16, 16), device=dev, generator=g)
        ins = torch.randint(0, 6, (n_img, 16, 16), device=dev, generator=g) * (cls >= 20)
        coarse = cls * max_inst + ins
        full = coarse.repeat_interleave(64, dim=1).repeat_interleave(64, dim=2)
        return torch.roll(full, shifts=(shift, shift), dims=(1, 2)).contiguous()

    pred, target = blocky(0), blocky(5)
    sem_target = (target // max_inst).to(torch.uint8)
    is_thing = [c >= 20 for c in range(n_cls)]
    half = n_img // 2

    def run(chunks, fused):
        pq = PanopticQuality(n_cls, 0, max_inst, 256 ** 3, is_thing, device=dev)
        miou = MeanIntersectionOverUnion(n_cls, ignore_first_class=True, device=dev)
        for sl in chunks:
            if fused:
                pq.update_with_miou(pred[sl], target[sl], miou, sem_target[sl], max_inst)
            else:
                pq.update(pred[sl], target[sl])
                miou.update_from_panoptic(pred[sl], sem_target[sl], max_inst)
        pq._check_status()
        miou._check_status()
        return [pq.iou_per_class, pq.tp_per_class, pq.fn_per_class, pq.fp_per_class, miou.confmat]

    whole = run([slice(0, n_img)], fused=False)
    halves = run([slice(0, half), slice(half, n_img)], fused=False)
    whole_fused = run([slice(0, n_img)], fused=True)
    assert int(whole[4].sum()) == n_img * H * W
    for a, b, c in zip(whole, halves, whole_fused):
        assert torch.equal(a, b)
        assert torch.equal(a, c)
    del pred, target, sem_target
    torch.cuda.empty_cache()


def test_elementwise_losses_beyond_int32():
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    dev = torch.device('cuda')
    n_img = 1100                                             # 1100*2*1024*1024 = 2.3e9
    g = torch.Generator(device=dev).manual_seed(31)
    pred = torch.randn((n_img, 2, H, W), device=dev, generator=g).requires_grad_(True)
    target = torch.randn((n_img, 2, H, W), device=dev, generator=g)
    mask = torch.rand((n_img, H, W), device=dev, generator=g) < 0.5
    assert pred.numel() > 2 ** 31
    loss, n = F_.masked_elementwise_sum(pred, target, mask, 'mse')
    loss.backward()
    assert int(n) == int(mask.sum())
    b = n_img - 1
    p1 = pred[b:].detach().clone().requires_grad_(True)
    l1, _ = F_.masked_elementwise_sum(p1, target[b:], mask[b:], 'mse')
    l1.backward()
    assert torch.equal(pred.grad[b], p1.grad[0])
    ref = 0.0
    for i in range(0, n_img, 100):
        d = (pred[i:i + 100].detach() * mask[i:i + 100, None] - target[i:i + 100]).double() ** 2
        ref += float(d.mean(dim=1).sum())
    assert abs(float(loss.detach()) - ref) <= 1e-5 * ref
    lv, nv = F_.vonmises_sum(pred.detach(), target, mask, 1.0)
    l1v, _ = F_.vonmises_sum(pred[b:].detach(), target[b:], mask[b:], 1.0)
    tot = sum(float(F_.vonmises_sum(pred[i:i + 100].detach(), target[i:i + 100], mask[i:i + 100], 1.0)[0])
              for i in range(0, n_img, 100))
    assert abs(float(lv) - tot) <= 1e-5 * abs(tot)
    assert int(nv) == int(mask.sum())
    del pred, target, mask
    torch.cuda.empty_cache()
