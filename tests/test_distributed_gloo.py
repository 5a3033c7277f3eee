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
    for name, s in zip(('iou_per_class', 'tp_per_class', 'fn_per_class', 'fp_per_class'), state):
        getattr(pq, name).add_(torch.from_numpy(s))
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


def test_metric_sync_world2(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    pred, tgt, sem_tgt = _maps(7)
    miou, pq = _accumulate(range(pred.shape[0]), pred, tgt, sem_tgt)
    want_state = np.stack([pq.iou_per_class.numpy(), pq.tp_per_class.numpy(),
                           pq.fn_per_class.numpy(), pq.fp_per_class.numpy()])
    for rank in range(2):
        got = np.load(tmp_path / f'rank{rank}.npz')
        assert (got['confmat'] == miou.confmat.numpy()).all()
        assert (got['state'][1:] == want_state[1:]).all()
        np.testing.assert_allclose(got['state'][0], want_state[0], rtol=1e-12)
        np.testing.assert_allclose(got['miou'], float(miou.compute()), rtol=1e-6)
        np.testing.assert_allclose(got['pq'], float(pq.compute('_deeplab')['all_deeplab_pq']),
                                   rtol=1e-12)


def test_sync_without_process_group_is_noop():
    from nicr_mt_scene_analysis_amd.metric import MeanIntersectionOverUnion
    m = MeanIntersectionOverUnion(3, device='cpu')
    m.confmat += 1
    m.sync()
    assert int(m.confmat.sum()) == 9
    m.reset()
    assert int(m.confmat.sum()) == 0
