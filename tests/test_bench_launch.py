"""
GPU tier: the launch shapes of bench.py (contract: `python bench.py --gpus N` and the same
under torch.distributed.run must both work).

* `python bench.py --gpus 2` WITHOUT a launcher starts two ranks itself (here over gloo, both
  ranks sharing the one GPU of the test box: RCCL refuses shared devices) and prints one JSON
  line for the whole job;
* under torch.distributed.run with one rank the collective leg runs on RCCL: process group
  'nccl' with device_id, an actual all-reduce (`rccl_ranks`), the packed accumulator
  all-reduce on the metric side stream.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ['--steps', '3', '--warmup', '2', '--batch-per-gpu', '2', '--height', '96', '--width', '128',
         '--classes', '8', '--centers', '4', '--no-secondary', '--no-cpu-baseline']


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    env = dict(os.environ, NMSA_BENCH_BACKEND='gloo')
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + SMALL,
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 4 and d['scaling'] == 'weak'
    assert d['collective']['backend'] == 'gloo' and d['collective']['payload_bytes'] > 0
    assert d['collective']['totals_identical_on_all_ranks'] is True
    assert d['value'] > 0 and d['cpu_baseline'] is None


def test_bench_names_the_rank_that_failed():
    """a rank that dies takes the job down with a non-zero code, and the launcher says WHICH rank
    and shows the tail of its stderr (NMSA_BENCH_FAIL_RANK: a test hook that makes that rank raise
    after the process group is up, while the others wait in a collective)"""
    env = dict(os.environ, NMSA_BENCH_BACKEND='gloo', NMSA_BENCH_FAIL_RANK='1')
    env.pop('RANK', None)
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + SMALL,
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert 'rank 1 of 2 exited with code' in r.stderr and 'NMSA_BENCH_FAIL_RANK' in r.stderr, r.stderr[-3000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]


def test_bench_under_launcher_runs_the_rccl_leg():
    env = dict(os.environ)
    env.pop('NMSA_BENCH_BACKEND', None)
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                        '--nproc-per-node=1', '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + SMALL,
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    c = d['collective']
    assert c['backend'] == 'rccl' and c['rccl_ranks'] == 1
    assert c['payload_bytes'] == (9 * 9 + 4 * 9) * 8 and c['totals_identical_on_all_ranks'] is True


def test_bench_falls_back_to_gloo_when_rccl_cannot_start():
    """a node whose RCCL does not come up (NMSA_BENCH_FORCE_RCCL_FAIL: the hook raises where
    init_process_group('nccl') would) still yields the scaling figures: the accumulator all-reduce
    goes through gloo, and the line says so"""
    env = dict(os.environ, NMSA_BENCH_FORCE_RCCL_FAIL='1')
    for k in ('NMSA_BENCH_BACKEND', 'RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + SMALL,
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    c = d['collective']
    assert d['n_gpus'] == 2 and c['backend'] == 'gloo' and c['rccl_ranks'] is None
    assert 'NMSA_BENCH_FORCE_RCCL_FAIL' in c['rccl_error']
    assert c['totals_identical_on_all_ranks'] is True


def test_bench_rejects_a_world_size_mismatch():
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + SMALL,
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE' in (r.stderr + r.stdout)


def test_metric_sync_on_rccl():
    """`Metric.sync()` (dist_reduce_fx='sum' of reference metric/miou.py:21-25, pq.py:228-246) on
    the RCCL backend with device-resident int64 / float64 states, one rank"""
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.pop('NMSA_BENCH_BACKEND', None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                        '--nproc-per-node=1', '--master-addr', '127.0.0.1', '--master-port', str(port),
                        os.path.join(ROOT, 'tests', '_nccl_sync_worker.py')],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    assert 'RCCL_SYNC_OK rank 0 of 1' in r.stdout, r.stdout[-2000:]


def test_bench_secondary_legs_run_on_small_shapes():
    """every leg of bench.py's `secondary` object (the other BASELINE configurations and the
    "next" rows) on tiny shapes: no exception, the fields the driver reads are there"""
    import importlib
    import torch
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn
    dev = torch.device('cuda', 0)
    legs = {
        'pipeline_bf16': bench.secondary_pipeline(ops, syn, dev, 2, 8, 96, 128, 4, torch.bfloat16),
        'pipeline_f32_serial': bench.secondary_pipeline(ops, syn, dev, 2, 150, 64, 128, 4, None,
                                                        overlap=False),
        'losses': bench.secondary_losses(dev, B=2, C=8, H=64, W=96),
        'ce_many_classes': bench.secondary_ce(dev, B=1, C=60, H=32, W=48),
        'cos': bench.secondary_cos_emb(dev, B=1, D=64, H=32, W=32, L=8),
        'next_rows': bench.secondary_next_rows(ops, syn, dev, B=2, C=8, H=96, W=128),
    }
    n = 0
    for name, leg in legs.items():
        for key, entry in leg.items():
            if key == 'backward_launches':
                # the loss leg's forward-written gradients must all have been confirmed
                assert entry['confirmed'] > 0 and entry['recomputed'] == 0, entry
            elif key == 'oracle_check':
                # the timed cosine legs compare image 0 with the C oracle after the timing
                assert entry['matches_oracle'] and entry['count_exact'], entry
            elif isinstance(entry, dict):
                n += 1
                assert entry['ms'] == entry['ms'] or key == 'step_two_batches_in_flight', (name, key)
                assert entry['algorithmic_bytes'] > 0 and 'frac' in entry, (name, key)
    assert n >= 19
