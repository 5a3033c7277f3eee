"""
CPU tier, world_size 2 over gloo: the N>1 path of the metric accumulators.

Each rank accumulates the metric states of ITS shard of the images (here with
the C oracle standing in for the HIP update kernels, which need a GPU), then
`Metric.sync()` sums the states with one all-reduce per dtype.  The result must
equal the single-process accumulation over the whole batch: exactly for the
integer-valued states, and within 1e-12 for the fp64 IoU sums (reduction order).
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_CAT = 9
PQ_ARGS = dict(num_categories=N_CAT, ignored_label=0, max_instances_per_category=65536,
               offset=256 ** 3)


def _maps(seed, B=4, H=48, W=64):
    r = np.random.default_rng(seed)
    cls = np.repeat(np.repeat(r.integers(0, N_CAT, (B, H // 8, W // 8)), 8, 1), 8, 2)
    ins = np.repeat(np.repeat(r.integers(0, 4, (B, H // 4, W // 4)), 4, 1), 4, 2)
    pred = (cls * 65536 + ins * (cls >= N_CAT // 2)).astype(np.int64)
    tgt = np.roll(pred, 2, axis=2)
    tgt[:, :5] = 0
    sem_tgt = r.integers(0, N_CAT, pred.shape).astype(np.uint8)
    return pred, tgt, sem_tgt


def _accumulate(images, pred, tgt, sem_tgt):
    from oracle import oracle as orc
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion, PanopticQuality
    miou = MeanIntersectionOverUnion(N_CAT, ignore_first_class=True, device='cpu')
    pq = PanopticQuality(is_thing=[c >= N_CAT // 2 for c in range(N_CAT)], device='cpu',
                         **PQ_ARGS)
    state = None
    cm = np.zeros((N_CAT, N_CAT), np.int64)
    for b in images:
        *state, _ = orc.pq_compare_and_accumulate(pred[b], tgt[b], N_CAT, 0, 65536, 256 ** 3,
                                                  state=state)
        cm = orc.confmat_update(pred[b] // 65536, sem_tgt[b], N_CAT, cm)
    miou.confmat += torch.from_numpy(cm)
    for name, s in zip(('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class'), state or ()):
        getattr(pq, name).add_(torch.from_numpy(s))          # (a rank without images: zero states)
    return miou, pq


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    pred, tgt, sem_tgt = _maps(7)
    shard = list(range(rank, pred.shape[0], world))          # images are independent units
    miou, pq = _accumulate(shard, pred, tgt, sem_tgt)
    miou.sync()
    pq.sync()
    res = pq.compute(suffix='_deeplab')
    np.savez(os.path.join(out_dir, f'rank{rank}.npz'),
             confmat=miou.confmat.numpy(),
             state=np.stack([pq.iou_per_class.numpy(), pq.tp_per_class.numpy(),
                             pq.fn_per_class.numpy(), pq.fp_per_class.numpy()]),
             miou=float(miou.compute()), pq=float(res['all_deeplab_pq']))
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('world', [2, 8])
def test_metric_sync_world2(tmp_path, world):
    """world 2, and the 8 ranks of a whole node (each with one image or none of the four: ranks
    without any image take part in the all-reduce with zero states)"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    pred, tgt, sem_tgt = _maps(7)
    miou, pq = _accumulate(range(pred.shape[0]), pred, tgt, sem_tgt)
    want_state = np.stack([pq.iou_per_class.numpy(), pq.tp_per_class.numpy(),
                           pq.fn_per_class.numpy(), pq.fp_per_class.numpy()])
    for rank in range(world):
        got = np.load(tmp_path / f'rank{rank}.npz')
        assert (got['confmat'] == miou.confmat.numpy()).all()
        assert (got['state'][1:] == want_state[1:]).all()
        np.testing.assert_allclose(got['state'][0], want_state[0], rtol=1e-12)
        np.testing.assert_allclose(got['miou'], float(miou.compute()), rtol=1e-6)
        np.testing.assert_allclose(got['pq'], float(pq.compute('_deeplab')['all_deeplab_pq']),
                                   rtol=1e-12)


def _fill_helpers(images, pred, tgt, sem_tgt):
    """task helpers with CPU-resident metric states filled from the oracle (the update kernels
    need a GPU; the epoch-end arithmetic and the rank sync do not)"""
    from nicr_mt_scene_analysis_amd.task_helper import PanopticTaskHelper, SemanticTaskHelper
    miou, pq = _accumulate(images, pred, tgt, sem_tgt)
    is_thing = [c >= N_CAT // 2 for c in range(N_CAT)]
    pan = PanopticTaskHelper(semantic_n_classes=N_CAT, semantic_classes_is_thing=is_thing)
    pan.initialize(torch.device('cpu'))
    pan._metric_iou.confmat += miou.confmat
    for name in ('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class'):
        getattr(pan._mae_pq_deeplab, name).add_(getattr(pq, name))
    pan._mae_pq_deeplab.sum_angular_error += 0.25 * (1 + len(list(images)))
    pan._mae_pq_deeplab.n_elements += len(list(images))
    sem = SemanticTaskHelper(n_classes=N_CAT - 1)
    sem.initialize(torch.device('cpu'))
    sem._metric_iou.confmat += miou.confmat[1:, 1:]
    return pan, sem, miou


def _epoch_end(pan, sem):
    out = {}
    artifacts, _, logs = pan.validation_epoch_end()
    out.update({f'log_{k}': v.numpy() for k, v in logs.items() if 'time' not in k})
    out.update({f'art_{k}': v.numpy() for k, v in artifacts.items()})
    artifacts, _, logs = sem.validation_epoch_end()
    out.update({f'log_{k}': v.numpy() for k, v in logs.items() if 'time' not in k})
    out.update({f'art_{k}': v.numpy() for k, v in artifacts.items()})
    return out


def _helper_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    pred, tgt, sem_tgt = _maps(11, B=6)
    shard = list(range(rank, pred.shape[0], world))
    pan, sem, miou_local = _fill_helpers(shard, pred, tgt, sem_tgt)
    # compute() twice: the second call must see the local states again (no double sum)
    first = pan._mae_pq_deeplab.compute(suffix='_deeplab')['all_deeplab_pq'].clone()
    assert torch.equal(pan._metric_iou.confmat, miou_local.confmat)
    second = pan._mae_pq_deeplab.compute(suffix='_deeplab')['all_deeplab_pq']
    assert torch.equal(first, second)
    # rank-local view on request
    pan._mae_pq_deeplab.sync_on_compute = False
    local = pan._mae_pq_deeplab.compute(suffix='_deeplab')['all_deeplab_pq'].clone()
    pan._mae_pq_deeplab.sync_on_compute = True
    out = _epoch_end(pan, sem)
    out['local_pq'] = local.numpy()
    out['local_cm'] = miou_local.confmat.numpy()
    assert int(pan._metric_iou.confmat.sum()) == 0 and float(pan._mae_pq_deeplab.tp_per_class.sum()) == 0
    assert not pan._mae_pq_deeplab._is_synced and pan._mae_pq_deeplab._cache is None
    np.savez(os.path.join(out_dir, f'helper_rank{rank}.npz'), **out)
    dist.destroy_process_group()


def test_task_helper_epoch_end_sums_the_ranks(tmp_path):
    """`validation_epoch_end` of the Panoptic / Semantic task helpers under an initialised process
    group: every rank logs the metrics of ALL shards (torchmetrics' `dist_reduce_fx='sum'` inside
    `compute()`, reference metric/miou.py:21-25, pq.py:228-246, mae.py:39-44,79-82), the
    confusion-matrix artifact — read outside `compute()` — stays rank-local as in the reference
    (task_helper/panoptic.py:202-204, semantic.py:152-155), and the states are reset"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_helper_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    pred, tgt, sem_tgt = _maps(11, B=6)
    pan, sem, _ = _fill_helpers(range(pred.shape[0]), pred, tgt, sem_tgt)
    # single process: MAE states of the two shards add up to (0.25*4 + 0.25*4, 3 + 3)
    pan._mae_pq_deeplab.sum_angular_error.fill_(2.0)
    pan._mae_pq_deeplab.n_elements.fill_(6)
    want = _epoch_end(pan, sem)
    got = [np.load(tmp_path / f'helper_rank{r}.npz') for r in range(2)]
    log_keys = [k for k in want if k.startswith('log_')]
    assert 'log_panoptic_all_with_gt_deeplab_pq' in log_keys and 'log_semantic_miou' in log_keys \
        and 'log_panoptic_deeplab_semantic_miou' in log_keys and 'log_panoptic_mae_deeplab_deg' in log_keys
    for g in got:
        assert sorted(g.files) == sorted(list(want) + ['local_pq', 'local_cm'])
        for k in want:
            if k.endswith('_cm'):
                continue
            np.testing.assert_allclose(g[k], want[k], rtol=1e-12, atol=0, equal_nan=True, err_msg=k)
    # the artifacts read outside compute() are the shards' own matrices and add up to the whole
    assert (got[0]['art_panoptic_deeplab_semantic_cm'] == got[0]['local_cm']).all()
    assert (got[0]['art_panoptic_deeplab_semantic_cm'] + got[1]['art_panoptic_deeplab_semantic_cm']
            == want['art_panoptic_deeplab_semantic_cm']).all()
    assert (got[0]['art_semantic_cm'] + got[1]['art_semantic_cm'] == want['art_semantic_cm']).all()
    # sync_on_compute=False had given the shard's own PQ
    assert got[0]['local_pq'] != got[0]['log_panoptic_all_deeplab_pq'] or \
        got[1]['local_pq'] != got[1]['log_panoptic_all_deeplab_pq']


def _synced_update_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion
    miou = MeanIntersectionOverUnion(N_CAT, device='cpu')
    miou.confmat += 1
    miou.sync()
    try:
        miou.update(torch.zeros((1, 2, 2), dtype=torch.int64), torch.zeros((1, 2, 2), dtype=torch.int64))
        outcome = 'no error'
    except RuntimeError as e:
        outcome = 'raised' if 'synced' in str(e) else f'other: {e}'
    miou.unsync()
    assert int(miou.confmat.sum()) == N_CAT * N_CAT           # the rank-local states are back
    with open(os.path.join(out_dir, f'rank{rank}.txt'), 'w') as f:
        f.write(outcome)
    dist.destroy_process_group()


def test_update_on_synced_states_raises(tmp_path):
    """an update between sync() and unsync() would land in the rank-summed states and be discarded
    by unsync(): it raises, like torchmetrics (checked before the device is even looked at)"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_synced_update_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        assert (tmp_path / f'rank{rank}.txt').read_text() == 'raised'


def test_sync_without_process_group_is_noop():
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion
    m = MeanIntersectionOverUnion(3, device='cpu')
    m.confmat += 1
    m.sync()
    assert int(m.confmat.sum()) == 9 and not m._is_synced
    with m.sync_context():
        assert int(m.confmat.sum()) == 9
    m.compute()
    m.reset()
    assert int(m.confmat.sum()) == 0
