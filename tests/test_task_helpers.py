"""
GPU tier: the task helpers (training_step / validation_step / validation_epoch_end) on a
synthetic batch with multi-scale side outputs.  Loss-dict values are checked against the
reference's formulas written with plain torch ops (fp64) in this file
(task_helper/semantic.py:57-90, task_helper/instance.py:92-269, task_helper/base.py:161-182),
the metric logs against the oracle.
"""
import numpy as np
import pytest
import torch

from nicr_mt_scene_analysis_amd.testing import synthetic as syn

pytestmark = pytest.mark.gpu
RTOL = 1e-5


def _down(t, s):
    return t[..., ::s, ::s].contiguous()


def make_loss_batch(B=2, C=9, H=64, W=96, seed=0, with_orientation=True):
    d = syn.make_loss_inputs(B, C, H, W, seed=seed)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in d.items()}
    batch = {
        'semantic': t['semantic_target'], 'instance_center': t['center_target'],
        'instance_center_mask': t['center_mask'], 'instance_offset': t['offset_target'],
        'instance_foreground': t['offset_mask'], 'orientation': t['orientation_target'],
        'orientation_foreground': t['orientation_mask'],
    }
    for s in (2, 4):
        batch[f'_down_{s}'] = {k: _down(v, s) for k, v in batch.items() if isinstance(v, torch.Tensor)}
    main = (t['center_pred'].unsqueeze(1), t['offset_pred'], t['orientation_pred'])
    if not with_orientation:
        main = main[:2]
    side = tuple(tuple(_down(x, s) * 0.9 for x in main) for s in (2, 4))
    sem_side = tuple(_down(t['semantic_logits'], s) * 0.9 for s in (2, 4))
    preds = {'semantic_output': t['semantic_logits'].requires_grad_(True),
             'semantic_side_outputs': sem_side,
             'instance_output': tuple(x.requires_grad_(True) for x in main),
             'instance_side_outputs': side}
    return batch, preds, t


def ref_ce(x, t, w):
    return torch.nn.functional.cross_entropy(x.double(), t.long() - 1, weight=w.double(),
                                             reduction='sum', ignore_index=-1), (t > 0).sum()


def test_semantic_task_helper_losses():
    from nicr_mt_scene_analysis_amd.task_helper import SemanticTaskHelper
    batch, preds, t = make_loss_batch()
    helper = SemanticTaskHelper(n_classes=9, class_weights=t['class_weights'].cpu().numpy())
    helper.initialize(torch.device('cuda'))
    losses, logs = helper.training_step(batch, 0, preds)
    assert set(losses) == {'semantic_loss_main', 'semantic_loss_down_2', 'semantic_loss_down_4',
                           'semantic_total_loss'}
    assert 'semantic_step_time' in logs and 'semantic_total_loss' in logs
    tot_l, tot_n = 0, 0
    for key, x, tg in (('main', preds['semantic_output'], batch['semantic']),
                       ('down_2', preds['semantic_side_outputs'][0], batch['_down_2']['semantic']),
                       ('down_4', preds['semantic_side_outputs'][1], batch['_down_4']['semantic'])):
        l, n = ref_ce(x.detach(), tg, t['class_weights'])
        np.testing.assert_allclose(float(losses[f'semantic_loss_{key}']), float(l / n), rtol=RTOL)
        tot_l, tot_n = tot_l + l, tot_n + n
    np.testing.assert_allclose(float(losses['semantic_total_loss']), float(tot_l / tot_n), rtol=RTOL)
    losses['semantic_total_loss'].backward()
    assert preds['semantic_output'].grad is not None


def test_instance_task_helper_losses():
    from nicr_mt_scene_analysis_amd.task_helper import InstanceTaskHelper
    batch, preds, t = make_loss_batch()
    helper = InstanceTaskHelper(semantic_n_classes=10, semantic_classes_is_thing=(False,) * 5 + (True,) * 5)
    helper.initialize(torch.device('cuda'))
    losses, _ = helper.training_step(batch, 0, preds)
    want_keys = {f'instance_{k}_loss_{s}' for k in ('center', 'offset', 'orientation')
                 for s in ('main', 'down_2', 'down_4')}
    want_keys |= {f'instance_{k}_total_loss' for k in ('center', 'offset', 'orientation')}
    assert set(losses) == want_keys

    def scales(key_main, side_idx):
        yield 'main', preds['instance_output'][side_idx], batch
        for j, s in enumerate((2, 4)):
            yield f'down_{s}', preds['instance_side_outputs'][j][side_idx], batch[f'_down_{s}']

    sums = {k: [0, 0] for k in ('center', 'offset', 'orientation')}
    for key, p, bt in scales('center', 0):
        m = bt['instance_center_mask']
        l = ((p.detach()[:, 0].double() * m) - bt['instance_center'].double()).pow(2).sum()
        np.testing.assert_allclose(float(losses[f'instance_center_loss_{key}']), float(l / m.sum()), rtol=RTOL)
        sums['center'][0] += l; sums['center'][1] += m.sum()
    for key, p, bt in scales('offset', 1):
        m = bt['instance_foreground']
        l = ((p.detach().double() * m.unsqueeze(1)) - bt['instance_offset'].double()).abs().mean(1).sum()
        np.testing.assert_allclose(float(losses[f'instance_offset_loss_{key}']), float(l / m.sum()), rtol=RTOL)
        sums['offset'][0] += l; sums['offset'][1] += m.sum()
    for key, p, bt in scales('orientation', 2):
        m = bt['orientation_foreground']
        dot = (p.detach().double() * bt['orientation'].double()).sum(1)
        l = (1 - torch.exp(dot - 1))[m].sum()
        n = max(int(m.sum()), 1)
        np.testing.assert_allclose(float(losses[f'instance_orientation_loss_{key}']), float(l / n), rtol=RTOL)
        sums['orientation'][0] += l; sums['orientation'][1] += n
    for k, (l, n) in sums.items():
        np.testing.assert_allclose(float(losses[f'instance_{k}_total_loss']), float(l / n), rtol=RTOL)
    total = sum(losses[f'instance_{k}_total_loss'] for k in sums)
    total.backward()
    for x in preds['instance_output']:
        assert x.grad is not None and torch.isfinite(x.grad).all()


def test_task_helper_totals_confirm_forward_written_gradients(monkeypatch):
    """with `backward_scale` as the starting value of the learned upstream factors every total's
    backward pass confirms from the first step (one forward call per helper for all scales), and
    the gradients are those of the two-kernel path"""
    from nicr_mt_scene_analysis_amd.loss import _functional as _F
    if not _F.speculation_enabled():
        pytest.skip('NMSA_SPECULATIVE_GRAD=0: forward-written gradients are switched off')
    from nicr_mt_scene_analysis_amd.loss import _functional as F_
    from nicr_mt_scene_analysis_amd.loss import reset_speculation_state, speculation_stats
    from nicr_mt_scene_analysis_amd.task_helper import InstanceTaskHelper, SemanticTaskHelper
    reset_speculation_state()
    weights = {'semantic': 0.75, 'instance_center': 2.0, 'instance_offset': 1.0,
               'instance_orientation': 0.5}

    def run():
        batch, preds, t = make_loss_batch()
        preds['semantic_side_outputs'] = tuple(
            x.detach().requires_grad_(True) for x in preds['semantic_side_outputs'])
        preds['instance_side_outputs'] = tuple(
            tuple(x.detach().requires_grad_(True) for x in side)
            for side in preds['instance_side_outputs'])
        sem = SemanticTaskHelper(n_classes=9, class_weights=t['class_weights'].cpu().numpy())
        ins = InstanceTaskHelper(semantic_n_classes=10,
                                 semantic_classes_is_thing=(False,) * 5 + (True,) * 5)
        sem.backward_scale = weights['semantic']
        ins.backward_scale = weights
        losses = {}
        for helper in (sem, ins):
            helper.initialize(torch.device('cuda'))
            losses.update(helper.training_step(batch, 0, preds)[0])
        total = sum(w * losses[f'{k}_total_loss'] for k, w in weights.items())
        total.backward()
        leaves = [preds['semantic_output'], *preds['semantic_side_outputs'],
                  *preds['instance_output'], *(x for s in preds['instance_side_outputs'] for x in s)]
        return float(total), [x.grad.clone() for x in leaves]

    before = speculation_stats()
    total, grads = run()
    after = speculation_stats()
    assert after['recomputed'] == before['recomputed']
    assert after['confirmed'] - before['confirmed'] == 1 + 3           # one per total: CE + 3 instance
    monkeypatch.setattr(F_, '_SPECULATE', False)
    total_plain, grads_plain = run()
    assert speculation_stats() == after
    np.testing.assert_allclose(total, total_plain, rtol=1e-6)
    for a, b in zip(grads, grads_plain):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=1e-9)


def test_panoptic_and_semantic_validation(oracle):
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    from nicr_mt_scene_analysis_amd.task_helper import PanopticTaskHelper, SemanticTaskHelper
    from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY
    B, C, H, W = 3, 8, 96, 128
    inp = syn.make_panoptic_inputs(B, C, H, W, n_centers=6, seed=2)
    is_thing = tuple(bool(x) for x in inp['semantic_classes_is_thing'])
    post = get_postprocessing_class('panoptic')(
        semantic_postprocessing=get_postprocessing_class('semantic')(),
        instance_postprocessing=get_postprocessing_class('instance')(),
        semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    rng = np.random.default_rng(0)
    sem_gt = rng.integers(0, C + 1, (B, H, W)).astype(np.uint8)
    batch = {'rgb_fullres': torch.zeros((B, 3, H, W)),
             'semantic': dev(sem_gt), 'semantic_fullres': dev(sem_gt),
             APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize', 'valid_region_slice_y': slice(0, H),
                                           'valid_region_slice_x': slice(0, W)}]] * B}
    data = ((dev(inp['semantic_logits']), (dev(inp['instance_center']), dev(inp['instance_offset']))),
            (None, None))
    r = post.postprocess(data, batch, is_training=False)
    pan = r['panoptic_segmentation_deeplab'].cpu().numpy()
    pan_gt = np.roll(pan, (2, 3), axis=(1, 2))
    pan_gt[:, :4] = 0
    batch['panoptic_fullres'] = dev(pan_gt)
    batch['panoptic_ids_to_instance_dict'] = [{} for _ in range(B)]

    ph = PanopticTaskHelper(C + 1, (False,) + is_thing)
    ph.initialize(torch.device('cuda'))
    assert ph.training_step(batch, 0, r)[0] == {}
    ph.validation_step(batch, 0, r)
    artifacts, examples, logs = ph.validation_epoch_end()
    for k in ('panoptic_all_deeplab_pq', 'panoptic_all_with_gt_deeplab_pq', 'panoptic_things_deeplab_rq',
              'panoptic_stuff_deeplab_sq', 'panoptic_deeplab_semantic_miou', 'panoptic_epoch_end_time'):
        assert k in logs, k
    for k in ('panoptic_pq_per_class', 'panoptic_deeplab_semantic_cm',
              'panoptic_deeplab_semantic_ious_per_class'):
        assert k in artifacts, k
    # against the oracle
    state = None
    cm = None
    for b in range(B):
        *state, _ = oracle.pq_compare_and_accumulate(pan[b], pan_gt[b], C + 1, 0, 1 << 16, 256 ** 3,
                                                     state=state)
        cm = oracle.confmat_update(pan[b] // 65536, sem_gt[b], C + 1, cm)
    assert (artifacts['panoptic_deeplab_semantic_cm'].cpu().numpy() == cm).all()
    miou, _ = oracle.miou_compute(cm, True)
    np.testing.assert_allclose(float(logs['panoptic_deeplab_semantic_miou']), miou, rtol=1e-5)
    sq = np.where(state[1] > 0, state[0] / np.maximum(state[1], 1), 0)
    rq = np.where(state[1] + state[2] + state[3] > 0,
                  state[1] / np.maximum(state[1] + .5 * state[2] + .5 * state[3], 1e-30), 0)
    valid = (state[1] + state[2] + state[3]) != 0
    valid[0] = False
    np.testing.assert_allclose(float(logs['panoptic_all_deeplab_pq']), (sq * rq)[valid].mean(), rtol=1e-12)
    # after epoch end the states are reset
    assert float(ph._mae_pq_deeplab.tp_per_class.sum()) == 0

    sh = SemanticTaskHelper(n_classes=C, disable_multiscale_supervision=True)
    sh.initialize(torch.device('cuda'))
    sh.validation_step(batch, 0, r)
    art, _, logs = sh.validation_epoch_end()
    idx = r['semantic_segmentation_idx_fullres'].cpu().numpy()
    m = sem_gt != 0
    cm2 = oracle.confmat_update(idx[m], sem_gt[m] - 1, C)
    assert (art['semantic_cm'].cpu().numpy() == cm2).all()
    np.testing.assert_allclose(float(logs['semantic_miou']), oracle.miou_compute(cm2)[0], rtol=1e-5)


def test_instance_validation_step():
    """instance quality with GT semantics (task_helper/instance.py:289-357): a perfect
    prediction of GT-derived centers/offsets yields PQ == 1 for the thing classes."""
    from nicr_mt_scene_analysis_amd.model.postprocessing import InstancePostprocessing
    from nicr_mt_scene_analysis_amd.task_helper import InstanceTaskHelper
    from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY
    B, H, W = 2, 64, 96
    inst = np.zeros((B, H, W), np.int32)
    sem = np.ones((B, H, W), np.uint8)                 # class 1 = stuff
    heat = np.zeros((B, 1, H, W), np.float32)
    off = np.zeros((B, 2, H, W), np.float32)
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    boxes = [(8, 8, 24, 30, 2), (30, 40, 60, 90, 3), (10, 50, 26, 80, 2)]
    for b in range(B):
        for k, (y0, x0, y1, x1, cls) in enumerate(boxes[:2 + b]):
            inst[b, y0:y1, x0:x1] = k + 1
            sem[b, y0:y1, x0:x1] = cls
            cy, cx = (y0 + y1) // 2, (x0 + x1) // 2
            heat[b, 0, cy, cx] = 1.0
            off[b, 0, y0:y1, x0:x1] = (cy - yy[y0:y1, x0:x1]) / H
            off[b, 1, y0:y1, x0:x1] = (cx - xx[y0:y1, x0:x1]) / W
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()      # noqa: E731
    pan_gt = sem.astype(np.int64) * 65536 + inst
    fg = inst != 0
    batch = {
        'instance_foreground': dev(fg), 'instance_fullres': dev(inst), 'semantic_fullres': dev(sem),
        'panoptic_fullres': dev(pan_gt), 'panoptic_ids_to_instance_dict': [{} for _ in range(B)],
        'instance_center': dev(heat[:, 0]), 'instance_center_mask': dev(np.ones((B, H, W), bool)),
        'instance_offset': dev(off),
        APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize', 'valid_region_slice_y': slice(0, H),
                                      'valid_region_slice_x': slice(0, W)}]] * B}
    post = InstancePostprocessing()
    preds = post.postprocess(((dev(heat), dev(off)), (None,)), batch, is_training=False)
    helper = InstanceTaskHelper(4, (False, False, True, True), disable_multiscale_supervision=True)
    helper.initialize(torch.device('cuda'))
    losses, _ = helper.validation_step(batch, 0, preds)
    assert float(losses['instance_center_total_loss']) == 0.0
    assert float(losses['instance_offset_total_loss']) == 0.0
    _, _, logs = helper.validation_epoch_end()
    assert float(logs['instance_things_deeplab_pq']) == 1.0
    assert float(logs['instance_all_deeplab_rq']) == 1.0
    assert int(logs['instance_things_deeplab_num_categories']) == 2


def test_dve_task_helper_loss():
    from nicr_mt_scene_analysis_amd.task_helper import DenseVisualEmbeddingTaskHelper
    d = syn.make_loss_inputs(2, 5, 24, 32, seed=3, embedding_dim=16, n_lut=7)
    pred = torch.from_numpy(d['embedding_pred']).cuda().requires_grad_(True)
    idx = torch.from_numpy(d['embedding_indices']).cuda()
    luts = [torch.from_numpy(d['embedding_lut'][b]).cuda() for b in range(2)]
    batch = {'dense_visual_embedding_lut': luts, 'dense_visual_embedding_indices': idx}
    helper = DenseVisualEmbeddingTaskHelper(n_classes=5, disable_multiscale_supervision=True)
    helper.initialize(torch.device('cuda'))
    losses, _ = helper.training_step(batch, 0, {'dense_visual_embedding_output': pred})
    valid = idx != 0
    rows = pred.detach().permute(0, 2, 3, 1)[valid].double()
    b_idx = torch.where(valid)[0]
    tgt = torch.stack(luts)[b_idx, (idx[valid] - 1).long()].double()
    ref = torch.nn.functional.cosine_embedding_loss(rows, tgt, torch.ones(len(rows), device='cuda'),
                                                    reduction='sum') / len(rows)
    np.testing.assert_allclose(float(losses['dense_visual_embedding_total_loss']), float(ref), rtol=RTOL)
    np.testing.assert_allclose(float(losses['dense_visual_embedding_loss_main']), float(ref), rtol=RTOL)
    losses['dense_visual_embedding_total_loss'].backward()
    assert torch.isfinite(pred.grad).all()


def test_dve_task_helper_multiscale_one_call():
    """embedding dimensions the one-pass cosine kernel takes (D = 64): main + two side outputs go
    through ONE multi-loss call; per-scale losses and the total (sum of sums / sum of counts)
    against torch in fp64, the trainer's loss weight is learned after one step"""
    from nicr_mt_scene_analysis_amd.loss import _functional as _F
    if not _F.speculation_enabled():
        pytest.skip('NMSA_SPECULATIVE_GRAD=0: forward-written gradients are switched off')
    from nicr_mt_scene_analysis_amd.task_helper import DenseVisualEmbeddingTaskHelper
    from nicr_mt_scene_analysis_amd.loss import reset_speculation_state, speculation_stats
    reset_speculation_state()
    g = torch.Generator(device='cuda').manual_seed(4)
    B, D, H, W, NL = 2, 64, 16, 32, 6
    luts = torch.nn.functional.normalize(torch.randn((B, NL, D), device='cuda', generator=g), dim=-1)
    idx = torch.randint(0, NL + 1, (B, H // 4, W // 4), device='cuda', generator=g, dtype=torch.int32)
    idx = idx.repeat_interleave(4, 1).repeat_interleave(4, 2).contiguous()
    batch = {'dense_visual_embedding_lut': luts, 'dense_visual_embedding_indices': idx}
    for sc in (2, 4):
        batch[f'_down_{sc}'] = {'dense_visual_embedding_indices': idx[:, ::sc, ::sc].contiguous(),
                               'dense_visual_embedding_lut': luts}
    preds = [torch.randn((B, D, H // sc, W // sc), device='cuda', generator=g).requires_grad_(True)
             for sc in (1, 2, 4)]
    post = {'dense_visual_embedding_output': preds[0], 'dense_visual_embedding_side_outputs': tuple(preds[1:])}
    helper = DenseVisualEmbeddingTaskHelper(n_classes=5)
    helper.initialize(torch.device('cuda'))

    def reference(p, ix):
        valid = ix != 0
        rows = p.detach().permute(0, 2, 3, 1)[valid].double()
        tgt = luts[torch.where(valid)[0], (ix[valid] - 1).long()].double()
        return torch.nn.functional.cosine_embedding_loss(rows, tgt, torch.ones(len(rows), device='cuda'),
                                                         reduction='sum'), len(rows)
    refs = [reference(p, idx[:, ::sc, ::sc]) for p, sc in zip(preds, (1, 2, 4))]
    s0 = speculation_stats()
    for step in range(3):
        for p in preds:
            p.grad = None
        losses, _ = helper.training_step(batch, step, post)
        assert list(losses) == ['dense_visual_embedding_loss_main', 'dense_visual_embedding_loss_down_2',
                                'dense_visual_embedding_loss_down_4', 'dense_visual_embedding_total_loss']
        for (l, n), k in zip(refs, ('main', 'down_2', 'down_4')):
            np.testing.assert_allclose(float(losses[f'dense_visual_embedding_loss_{k}']), float(l) / n, rtol=RTOL)
        np.testing.assert_allclose(float(losses['dense_visual_embedding_total_loss']),
                                   float(sum(l for l, _ in refs)) / sum(n for _, n in refs), rtol=RTOL)
        (0.25 * losses['dense_visual_embedding_total_loss']).backward()
    s1 = speculation_stats()
    assert (s1['confirmed'] - s0['confirmed'], s1['recomputed'] - s0['recomputed']) == (2, 1)
    # gradient of the main scale against autograd on the gathered rows
    pr = preds[0].detach().double().requires_grad_(True)
    valid = idx != 0
    rows = pr.permute(0, 2, 3, 1)[valid]
    tgt = luts[torch.where(valid)[0], (idx[valid] - 1).long()].double()
    (0.25 * torch.nn.functional.cosine_embedding_loss(rows, tgt, torch.ones(len(rows), device='cuda'),
                                                      reduction='sum') / sum(n for _, n in refs)).backward()
    np.testing.assert_allclose(preds[0].grad.double().cpu().numpy(), pr.grad.cpu().numpy(), rtol=2e-5, atol=1e-10)


def test_accumulate_losses_zero_elements_warns():
    from nicr_mt_scene_analysis_amd.task_helper import SemanticTaskHelper
    h = SemanticTaskHelper(3)
    with pytest.warns(UserWarning):
        out = h.accumulate_losses([torch.zeros(())], [0])
    assert float(out) == 0.0
    out = h.accumulate_losses([torch.zeros((), device='cuda')], [torch.zeros((), dtype=torch.int64, device='cuda')])
    assert float(out) == 0.0


def test_four_losses_cfg3_bf16_full_batch(oracle):
    """BASELINE configs[2]: CE + center (MSE) + offset (L1) + von Mises orientation through
    SemanticTaskHelper + InstanceTaskHelper.training_step at B=64, 640x480, C=40, bf16
    predictions.  (a) a 4-image slice equals the C oracle on the same values (1e-5);
    (b) the full-batch loss values equal the count-weighted combination of the sixteen 4-image
    slices (additivity of sums and counts over images); (c) backward fills bf16 gradients."""
    from nicr_mt_scene_analysis_amd.task_helper import InstanceTaskHelper, SemanticTaskHelper
    B, C, H, W, S = 64, 40, 480, 640, 4
    dev = torch.device('cuda')
    g = torch.Generator(device=dev).manual_seed(42)

    def rnd(*shape):
        return torch.randn(shape, device=dev, generator=g)
    bf = torch.bfloat16
    logits = (rnd(B, C, H, W) * 3).to(bf)
    center = torch.rand((B, 1, H, W), device=dev, generator=g).to(bf)
    offset = (rnd(B, 2, H, W) * 0.05).to(bf)
    ori = torch.nn.functional.normalize(rnd(B, 2, H, W), dim=1).to(bf)
    batch = {
        'semantic': torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8),
        'instance_center': torch.rand((B, H, W), device=dev, generator=g),
        'instance_center_mask': torch.rand((B, H, W), device=dev, generator=g) < 0.7,
        'instance_offset': rnd(B, 2, H, W) * 0.05,
        'instance_foreground': torch.rand((B, H, W), device=dev, generator=g) < 0.5,
        'orientation': torch.nn.functional.normalize(rnd(B, 2, H, W), dim=1),
        'orientation_foreground': torch.rand((B, H, W), device=dev, generator=g) < 0.3,
    }
    weights = (torch.rand(C, device=dev, generator=g) + 0.5)
    sem = SemanticTaskHelper(n_classes=C, class_weights=weights.cpu().numpy())
    ins = InstanceTaskHelper(semantic_n_classes=C + 1,
                             semantic_classes_is_thing=(False,) * 21 + (True,) * 20)
    sem.initialize(dev)
    ins.initialize(dev)

    def run(lo, hi, grad=False):
        sl = slice(lo, hi)
        bt = {k: v[sl] for k, v in batch.items()}
        x = [t[sl].clone().requires_grad_(grad) for t in (logits, center, offset, ori)]
        preds = {'semantic_output': x[0], 'semantic_side_outputs': (None, None),
                 'instance_output': (x[1], x[2], x[3]), 'instance_side_outputs': (None, None)}
        losses = dict(sem.training_step(bt, 0, preds)[0])
        losses.update(ins.training_step(bt, 0, preds)[0])
        return losses, x

    keys = ('semantic_loss_main', 'instance_center_loss_main', 'instance_offset_loss_main',
            'instance_orientation_loss_main')
    count_of = {'semantic_loss_main': lambda sl: (batch['semantic'][sl] > 0).sum(),
                'instance_center_loss_main': lambda sl: batch['instance_center_mask'][sl].sum(),
                'instance_offset_loss_main': lambda sl: batch['instance_foreground'][sl].sum(),
                'instance_orientation_loss_main': lambda sl: batch['orientation_foreground'][sl].sum()}

    # (c) + full batch
    full, x = run(0, B, grad=True)
    total = sum(full[k.replace('_loss_main', '_total_loss')] for k in keys)
    total.backward()
    for t in x:
        assert t.grad is not None and t.grad.dtype == bf and bool(torch.isfinite(t.grad).all())
    for k in keys:        # one scale: total == main
        np.testing.assert_allclose(float(full[k.replace('_loss_main', '_total_loss')]),
                                   float(full[k]), rtol=1e-6)

    # (b) additivity over sixteen slices of four images
    acc = {k: [0.0, 0] for k in keys}
    for lo in range(0, B, S):
        part, _ = run(lo, lo + S)
        for k in keys:
            n = int(count_of[k](slice(lo, lo + S)))
            acc[k][0] += float(part[k]) * n
            acc[k][1] += n
    for k in keys:
        np.testing.assert_allclose(float(full[k]), acc[k][0] / acc[k][1], rtol=RTOL, err_msg=k)
        assert acc[k][1] == int(count_of[k](slice(0, B)))

    # (a) first slice against the C oracle (bf16 values as float32: the same numbers)
    part, _ = run(0, S)
    f32 = lambda t: t[:S].float().cpu().numpy()                        # noqa: E731
    npy = lambda t: t[:S].cpu().numpy()                                # noqa: E731
    s_, n_, _, _ = oracle.loss_ce(f32(logits), npy(batch['semantic']), weights.cpu().numpy())
    np.testing.assert_allclose(float(part['semantic_loss_main']), s_ / n_, rtol=RTOL)
    s_, n_, _ = oracle.loss_masked_elementwise(f32(center)[:, 0], npy(batch['instance_center']),
                                               npy(batch['instance_center_mask']), 'mse')
    np.testing.assert_allclose(float(part['instance_center_loss_main']), s_ / n_, rtol=RTOL)
    s_, n_, _ = oracle.loss_masked_elementwise(f32(offset), npy(batch['instance_offset']),
                                               npy(batch['instance_foreground']), 'l1')
    np.testing.assert_allclose(float(part['instance_offset_loss_main']), s_ / n_, rtol=RTOL)
    s_, n_, _ = oracle.loss_vonmises(f32(ori), npy(batch['orientation']),
                                     npy(batch['orientation_foreground']), 1.0)
    np.testing.assert_allclose(float(part['instance_orientation_loss_main']), s_ / max(n_, 1),
                               rtol=RTOL)


# ---- pinned by the reference's own task helpers (tests/golden/task_helper_cases.npz) -------------
def _to_cuda(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x)).cuda()
    if isinstance(x, dict):
        return {k: _to_cuda(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return type(x)(_to_cuda(v) for v in x)
    return x


def _check_scalars(got: dict, keys, values, what, rtol=RTOL):
    from _golden import jload
    keys = jload(keys)
    assert list(got.keys()) == keys, (what, list(got.keys()), keys)          # same keys, same order
    for k, v in zip(keys, values):
        np.testing.assert_allclose(float(got[k]), v, rtol=rtol, atol=1e-9, err_msg=f'{what}: {k}')


def test_training_loss_dicts_vs_reference_task_helpers():
    """a10: loss dicts (every key, in order, and every value) of the reference's
    SemanticTaskHelper / InstanceTaskHelper / DenseVisualEmbeddingTaskHelper.training_step on a
    batch with main + 2 side outputs (reference task_helper/semantic.py:57-90,
    instance.py:92-269, dense_visual_embedding.py:70-193, base.py:161-182)"""
    from _golden import load, jload
    from nicr_mt_scene_analysis_amd.task_helper import (DenseVisualEmbeddingTaskHelper,
                                                        InstanceTaskHelper, SemanticTaskHelper)
    g = load('task_helper_cases')
    batch_np, preds_np, weights = syn.make_training_case()
    assert syn.input_digest(preds_np['semantic_output'], preds_np['instance_output'][1],
                            preds_np['dense_visual_embedding_output'], batch_np['semantic'],
                            batch_np['dense_visual_embedding_indices']) == jload(g['train__digest'])
    batch, preds = _to_cuda(batch_np), _to_cuda(preds_np)
    C = preds_np['semantic_output'].shape[1]
    is_thing = (False,) * (C // 2 + 1) + (True,) * (C - C // 2)
    cases = {
        'sem_plain': SemanticTaskHelper(n_classes=C),
        'sem_weighted_smooth': SemanticTaskHelper(n_classes=C, class_weights=weights,
                                                  label_smoothing=0.1),
        'sem_single_scale': SemanticTaskHelper(n_classes=C, disable_multiscale_supervision=True),
        'ins_mse': InstanceTaskHelper(C + 1, is_thing),
        'ins_l1': InstanceTaskHelper(C + 1, is_thing, loss_name_instance_center='l1'),
        'dve_cos': DenseVisualEmbeddingTaskHelper(n_classes=C, loss_name='cos_emb'),
    }
    dev = torch.device('cuda')
    for name, helper in cases.items():
        helper.initialize(dev)
        losses, logs = helper.training_step(batch, 0, preds)
        _check_scalars(losses, g[f'train__{name}__keys'], g[f'train__{name}__values'], name)
        assert sorted(logs.keys()) == jload(g[f'train__{name}__log_keys']), name
    preds2 = dict(preds, instance_output=preds['instance_output'][:2],
                  instance_side_outputs=tuple(p[:2] for p in preds['instance_side_outputs']))
    helper = InstanceTaskHelper(C + 1, is_thing)
    helper.initialize(dev)
    losses, _ = helper.training_step(batch, 0, preds2)
    _check_scalars(losses, g['train__ins_no_orientation__keys'],
                   g['train__ins_no_orientation__values'], 'ins_no_orientation')


def test_validation_chain_vs_reference_task_helpers():
    """a14: two validation steps + validation_epoch_end of the reference's Semantic / Instance /
    Panoptic task helpers behind the reference's PanopticPostprocessing, on ground truth made by
    the reference's target generators: loss dicts per step, log keys, epoch logs (PQ / SQ / RQ,
    mIoU, MAE) and artifacts (confusion matrices exactly, per-class vectors)."""
    from _golden import load, jload, ids_from_arrays
    from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    from nicr_mt_scene_analysis_amd.task_helper import (InstanceTaskHelper, PanopticTaskHelper,
                                                        SemanticTaskHelper)
    g = load('task_helper_cases')
    gt = {k[len('val__gt_'):]: g[k] for k in g.files if k.startswith('val__gt_')}
    B, H, W = gt['semantic'].shape
    is_thing_nc = tuple(bool(x) for x in g['val__is_thing_with_void'])
    C = len(is_thing_nc) - 1
    batch = _to_cuda(gt)
    for k in ('semantic', 'instance', 'panoptic'):
        batch[f'{k}_fullres'] = batch[k]
    batch['panoptic_ids_to_instance_dict'] = ids_from_arrays(
        g['val__pan_ids_n'], g['val__pan_ids_pan'], g['val__pan_ids_ins'])
    batch['orientations_present'] = [{int(k): v for k, v in d.items()}
                                     for d in jload(g['val__orientations_present'])]
    batch['rgb_fullres'] = torch.zeros((B, 3, H, W))
    batch[APPLIED_PREPROCESSING_KEY] = [[{'type': 'Resize', 'valid_region_slice_y': slice(0, H),
                                          'valid_region_slice_x': slice(0, W)}]] * B
    post = get_postprocessing_class('panoptic')(
        semantic_postprocessing=get_postprocessing_class('semantic')(),
        instance_postprocessing=get_postprocessing_class('instance')(),
        semantic_classes_is_thing=is_thing_nc[1:], semantic_class_has_orientation=is_thing_nc[1:])
    dev = torch.device('cuda')
    sem = SemanticTaskHelper(n_classes=C)
    ins = InstanceTaskHelper(C + 1, is_thing_nc)
    pan = PanopticTaskHelper(C + 1, is_thing_nc, None)
    for h in (sem, ins, pan):
        h.initialize(dev)
    digests = jload(g['val__pred_digests'])
    for step in range(2):
        logits, center, offset, ori = syn.make_predictions_from_targets(
            gt['semantic'], gt['instance_center'], gt['instance_offset'], gt['orientation'], C,
            seed=step)
        assert syn.input_digest(logits, center, offset, ori) == digests[step]
        data = ((_to_cuda(logits), (_to_cuda(center), _to_cuda(offset), _to_cuda(ori))),
                ((None, None), (None, None)))
        r = post.postprocess(data, batch, is_training=False)
        if step == 0:
            assert (r['panoptic_segmentation_deeplab_fullres'].cpu().numpy()
                    == g['val__pred_panoptic_step0']).all()
        for name, h in (('sem', sem), ('ins', ins), ('pan', pan)):
            losses, logs = h.validation_step(batch, step, r)
            _check_scalars(losses, g[f'val__{name}__step{step}__loss_keys'],
                           g[f'val__{name}__step{step}__loss_values'], f'{name} step {step}')
            assert sorted(logs.keys()) == jload(g[f'val__{name}__step{step}__log_keys']), name
    for name, h in (('sem', sem), ('ins', ins), ('pan', pan)):
        artifacts, examples, logs = h.validation_epoch_end()
        scal = {k: v for k, v in logs.items() if not k.endswith('_time')}
        _check_scalars(scal, g[f'val__{name}__log_keys'], g[f'val__{name}__log_values'],
                       f'{name} epoch logs')
        assert len([k for k in logs if k.endswith('_time')]) == 1        # the profiling decorator's key
        want_keys = jload(g[f'val__{name}__artifact_keys'])
        assert list(artifacts.keys()) == want_keys, (name, list(artifacts.keys()), want_keys)
        for k in want_keys:
            got, want = artifacts[k].cpu().numpy(), g[f'val__{name}__artifact__{k}']
            assert got.shape == want.shape and got.dtype == want.dtype, (k, got.dtype, want.dtype)
            if np.issubdtype(want.dtype, np.integer):
                assert (got == want).all(), k
            else:
                np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-12, equal_nan=True, err_msg=k)
