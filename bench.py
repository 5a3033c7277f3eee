#!/usr/bin/env python3
"""
bench.py — Mpix/s of the panoptic hot path on synthetic 640x480 maps.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Both launch shapes work for every N: without RANK in the environment `--gpus N>1`
starts N fresh child ranks itself (torch.distributed.run as a child process, before
anything in this process touches the GPU) and exits with their return code.

A "step" is one pass of the hot path over one batch that is already resident
in HBM: center-NMS/top-k -> fused semantic-argmax + offset grouping + class
votes -> per-instance class/rank -> panoptic paint (BASELINE.json configs[1]:
B=32 per GPU, C=40, 640x480), followed — when the metric accumulators are
enabled — by the mIoU confusion-matrix + PQ updates of configs[3].  Images are
independent units, so ranks shard the batch (weak scaling: 32 images per GPU)
with no data-path collective; only the few-KB metric accumulators are
all-reduced (RCCL) inside the timed region.  Behind the warm-up the step is captured into
hipGraphs (pipeline per batch stream + metric chain) and replayed; `--no-graph` launches every
kernel from Python.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").  At N=1 the line also
carries `secondary`: the other BASELINE.json configurations (bf16 logits, configs[2] losses,
configs[4] shapes + dense cosine loss), measured after the timed region with HIP events.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--batch-per-gpu', type=int, default=32)
    ap.add_argument('--classes', type=int, default=40)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--centers', type=int, default=24)
    ap.add_argument('--dtype', choices=('f32', 'bf16', 'f16'), default='f32',
                    help='dtype of the semantic logits (the arithmetic is f32 throughout)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true',
                    help='skip the other BASELINE configs measured after the timed region (N=1)')
    ap.add_argument('--no-metrics', action='store_true')
    ap.add_argument('--no-side-stream', action='store_true',
                    help='enqueue the metric kernels on the main stream (no overlap)')
    ap.add_argument('--cpu-sample-images', type=int, default=32)
    ap.add_argument('--metric-sync', choices=('step', 'end'), default='end',
                    help='N>1: ranks accumulate locally and the accumulators are all-reduced ONCE '
                         'after the last step, inside the timed region (default: what the '
                         "reference's torchmetrics states do at compute()), or after every step")
    ap.add_argument('--no-graph', action='store_true',
                    help='launch every step kernel by kernel from Python (default: the steps that carry no '
                         'event records replay two hipGraphs — the pipeline of their batch stream and the '
                         'metric chain — so the host queues a step in ~50 us instead of ~170)')
    ap.add_argument('--streams', type=int, default=2,
                    help='batches in flight: consecutive steps alternate over this many HIP streams')
    return ap.parse_args()


def host_cores():
    """nproc of the node, the cores this process may run on, its cgroup CPU quota, and the
    thread count the CPU baseline uses: every core the process is actually given"""
    nproc = os.cpu_count() or 1
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else nproc
    quota = None
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            q, period = f.read().split()
            if q != 'max':
                quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    usable = affinity if quota is None else max(1, min(affinity, int(round(quota))))
    return {'nproc': nproc, 'affinity': affinity, 'cgroup_quota': quota, 'usable': usable}


def host_threads_per_rank(n_ranks):
    return max(1, host_cores()['usable'] // max(n_ranks, 1))


def cpu_baseline(inp_dev, n_images, C, H, W, metrics=None, reps=3):
    """The C oracle ("port" of the reference's CPU path) timed on the host, single
    thread, on the first `n_images` images of the very batch the GPU processes."""
    from oracle import oracle as orc
    orc.build()
    n = min(n_images, inp_dev['semantic_logits'].shape[0])
    if metrics is not None:
        tgt_pan = metrics.target_panoptic[:n].cpu().numpy()
        tgt_sem = metrics.target_semantic[:n].cpu().numpy()
    logits = inp_dev['semantic_logits'][:n].float().cpu().numpy()
    center = inp_dev['instance_center'][:n].cpu().numpy()
    offset = inp_dev['instance_offset'][:n].cpu().numpy()
    is_thing = inp_dev['semantic_classes_is_thing'].cpu().numpy().astype(bool)
    t0 = time.perf_counter()
    for _ in range(max(1, reps)):           # bounded sample: ~10 s of single-core work
        idx, _ = orc.semantic_argmax(logits)
        fg = is_thing[idx]
        cyx, nc, _, _ = orc.center_nms_topk(center, max_centers=256)
        inst, _ = orc.group_offsets(offset, fg, cyx, nc, scale_y=H, scale_x=W)
        pan, _ = orc.deeplab_merge(idx + 1, inst, fg, 1 << 16, np.where(is_thing)[0] + 1, 0)
        what = 'argmax+NMS+grouping+merge'
        if metrics is not None:
            state = None
            cm = None
            for b in range(n):
                *state, _ = orc.pq_compare_and_accumulate(pan[b], tgt_pan[b], C + 1, 0, 1 << 16,
                                                          256 ** 3, state=state)
                cm = orc.confmat_update(pan[b] // 65536, tgt_sem[b], C + 1, cm)
            what += '+confmat+PQ'
    dt = time.perf_counter() - t0
    n_total = n * max(1, reps)
    one_core = n_total * H * W / dt / 1e6

    # the same chain, one image per task, on the host cores this process may use (the C
    # functions are re-entrant and ctypes releases the GIL): the stronger CPU baseline
    from concurrent.futures import ThreadPoolExecutor
    cores = host_cores()
    threads = cores['usable']

    def one_image(b):
        i_, _ = orc.semantic_argmax(logits[b:b + 1])
        f_ = is_thing[i_]
        c_, n_, _, _ = orc.center_nms_topk(center[b:b + 1], max_centers=256)
        s_, _ = orc.group_offsets(offset[b:b + 1], f_, c_, n_, scale_y=H, scale_x=W)
        p_, _ = orc.deeplab_merge(i_ + 1, s_, f_, 1 << 16, np.where(is_thing)[0] + 1, 0)
        if metrics is not None:
            orc.pq_compare_and_accumulate(p_[0], tgt_pan[b], C + 1, 0, 1 << 16, 256 ** 3)
            orc.confmat_update(p_[0] // 65536, tgt_sem[b], C + 1, None)
        return p_[0]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as pool:
        pans = list(pool.map(one_image, [b for _ in range(max(1, reps)) for b in range(n)]))
    dt_mt = time.perf_counter() - t0
    assert all(np.array_equal(pans[b], pan[b]) for b in range(n))
    return {'value': round(n_total * H * W / dt_mt / 1e6, 3), 'unit': 'Mpix/s', 'cores': threads,
            'nproc': cores['nproc'], 'affinity': cores['affinity'],
            'cgroup_cpu_quota': cores['cgroup_quota'],
            'kind': 'port', 'value_1core': round(one_core, 3),
            # stated context, not measured here: the reference's own Python / torch-CPU path on the
            # build container (8 vCPU; /root/reference does not exist on this box)
            'reference_python_mpix_s': 0.33, 'reference_python_cores': 8,
            'sample': f'{max(1, reps)} passes over {n} images {W}x{H}x{C} of the bench batch, C oracle '
                      f'({what}): {dt:.2f} s on 1 core, {dt_mt:.2f} s with {threads} threads '
                      '(one image per task)'}, (idx, inst, pan)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) as CHILD
    processes — this process has not touched the GPU and never execs — with the environment a
    launcher gives them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), forward rank 0's stdout (the
    JSON line) and return 0 when every rank did.  A rank that fails is named, with the tail of its
    stderr, before the others are stopped and a non-zero code is returned."""
    n = args.gpus
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    base = dict(os.environ)
    # the host driver of this pool only supports dmabuf IPC: without this RCCL's (and torch's)
    # cross-process sharing of device memory fails with `hipIpcGetMemHandle: invalid argument`
    # (it is exported on the GPU boxes already; kept in the environment we build for the ranks)
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # the ranks share the job's CPU quota (16 cores for the whole GPU box): bound every rank's host
    # thread pools, or 8 ranks x (intra-op pool + RCCL proxy + HSA threads) oversubscribe it
    base.setdefault('OMP_NUM_THREADS', str(host_threads_per_rank(n)))
    base.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    import tempfile
    procs = []
    with tempfile.TemporaryDirectory(prefix='nmsa_bench_ranks_') as logdir:
        for rank in range(n):
            env = dict(base, RANK=str(rank), LOCAL_RANK=str(rank))
            err = open(os.path.join(logdir, f'rank{rank}.stderr'), 'w+')
            out = None if rank == 0 else subprocess.DEVNULL        # rank 0 prints the one JSON line
            procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                           env=env, stdout=out, stderr=err), err))
        failed = None
        pending = set(range(n))
        while pending and failed is None:
            for rank in sorted(pending):
                rc = procs[rank][0].poll()
                if rc is None:
                    continue
                pending.discard(rank)
                if rc != 0:
                    failed = (rank, rc)
                    break
            else:
                time.sleep(0.05)
        if failed is not None:
            for rank in pending:                                   # the others wait in a collective
                procs[rank][0].terminate()
            for rank in pending:
                try:
                    procs[rank][0].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[rank][0].kill()
        code = 0
        for rank, (proc, err) in enumerate(procs):
            err.seek(0)
            text = err.read()
            err.close()
            if failed is not None and rank == failed[0]:
                tail = ''.join(text.splitlines(True)[-25:])
                print(f'bench.py: rank {rank} of {n} exited with code {failed[1]}; the tail of its stderr:\n'
                      f'{tail}', file=sys.stderr, flush=True)
                code = failed[1] if 0 < failed[1] < 256 else 1
            elif rank == 0 and failed is None and text:
                sys.stderr.write(text)                             # warnings of a healthy run
        return code


def hip_timed(fn, reps, warm, min_region_ms=5.0, max_reps=400):
    """mean milliseconds per call, HIP events on the stream the kernels are launched on
    (the ops enqueue on torch's current stream).  The GPU idles between the start event and the
    first kernel for as long as the host needs to enqueue one call (~35 us for the target
    generators): a region shorter than `min_region_ms` is measured again with as many calls as
    fill it (10 calls of a 65 us op read 72 us, 50 calls 64 us)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    while True:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        total = e0.elapsed_time(e1)
        if total >= min_region_ms or reps >= max_reps:
            return total / reps
        reps = min(max_reps, max(reps + 1, int(reps * min_region_ms / max(total, 1e-3) * 1.1)))


def _leg(ms, n_px, bytes_px, **extra):
    gbs = n_px * bytes_px / (ms * 1e-3) / 1e9
    d = {'ms': round(ms, 4), 'Mpix_s': round(n_px / (ms * 1e-3) / 1e6, 1),
         'algorithmic_bytes_per_px': bytes_px, 'algorithmic_bytes': n_px * bytes_px,
         'achieved_GBs': round(gbs, 1), 'frac': round(gbs / HBM_PEAK_GBS, 4)}
    d.update(extra)
    return d


def secondary_pipeline(ops, syn, dev, B, C, H, W, K, dtype, with_metrics=True, overlap=True):
    """the panoptic step on another configuration: whole step (HIP events around the five
    launches + metric updates) and the fused kernel alone (events around that launch)"""
    from tools import bench_support
    inp = syn.make_panoptic_inputs_torch(B, C, H, W, n_centers=K, seed=4321, device=dev,
                                         logits_dtype=dtype)
    a = (inp['semantic_logits'], inp['instance_center'], inp['instance_offset'],
         inp['semantic_classes_is_thing'])
    es = a[0].element_size()
    m = bench_support.MetricAccumulators(C + 1, dev, inp, 0, side_stream=False) \
        if with_metrics else None
    ev = []

    def step():
        r = ops.panoptic_pipeline(*a, want_foreground=False, fused_kernel_events=ev)
        if m is not None:
            m.update_and_reduce(r)
    ms = hip_timed(step, reps=20, warm=5)
    # the headline's schedule: consecutive batches alternate over two streams, metric kernels
    # on a side stream (the one-workgroup-per-image kernels hide behind the other batch)
    m2 = bench_support.MetricAccumulators(C + 1, dev, inp, 0, side_stream=True) \
        if with_metrics and overlap else None
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def step2(i):
        with torch.cuda.stream(streams[i % 2]):
            r = ops.panoptic_pipeline(*a, want_foreground=False)
            if m2 is not None:
                m2.update_and_reduce(r)
    ms2 = float('nan')
    if overlap:                                 # (rocprofv3 leg profiles skip it: clean per-kernel averages)
        for i in range(6):
            step2(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_rep = 30
        for i in range(n_rep):
            step2(i)
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) / n_rep * 1e3
    for mm in (m, m2):
        if mm is not None:
            mm.pq._check_status()               # table overflow etc. would void the timing
            mm.miou._check_status()
    fused_ms = float(np.mean([x.elapsed_time(y) for x, y in ev[5:]]))
    n_px = B * H * W
    out = {'shape': f'B={B} C={C} {W}x{H}', 'logits_dtype': str(a[0].dtype).replace('torch.', ''),
           'step_serial': _leg(ms, n_px, es * C + 21 + (17 if with_metrics else 0),
                               what='one stream, no batch overlap; pipeline'
                                    + (' + mIoU/PQ updates' if with_metrics else '')),
           'step_two_batches_in_flight': _leg(ms2, n_px, es * C + 21 + (17 if with_metrics else 0),
                                              what='wall clock over 30 steps, two streams + metric '
                                                   'side stream (the headline schedule)'),
           'k_panoptic_fused': _leg(fused_ms, n_px, es * C + 9)}
    del inp, a, m, m2
    torch.cuda.empty_cache()
    return out


def secondary_losses(dev, B=64, C=40, H=480, W=640):
    """BASELINE configs[2]: CE + center (MSE) + offset (L1) + von Mises orientation on bf16
    predictions, f32 targets, u8 labels / masks (114 B/px forward, 204 B/px with
    the backward's gradient writes)"""
    from nicr_mt_scene_analysis_amd.loss import (CenterFocalLoss, CrossEntropyLossSemantic, L1Loss,
                                                 MSELoss, VonMisesLossBiternion)
    g = torch.Generator(device=dev).manual_seed(7)
    dt = torch.bfloat16

    def rnd(*shape):
        return torch.randn(shape, device=dev, generator=g)
    logits = (rnd(B, C, H, W) * 3).to(dt).requires_grad_(True)
    labels = torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8)
    w = torch.rand(C, device=dev, generator=g) + 0.5
    center = torch.rand((B, H, W), device=dev, generator=g).to(dt).requires_grad_(True)
    center_t = torch.rand((B, H, W), device=dev, generator=g)
    offset = rnd(B, 2, H, W).to(dt).requires_grad_(True)
    offset_t = rnd(B, 2, H, W)
    ori = rnd(B, 2, H, W).to(dt).requires_grad_(True)
    ori_t = torch.nn.functional.normalize(rnd(B, 2, H, W), dim=1)
    m1 = torch.rand((B, H, W), device=dev, generator=g) < 0.7
    m2 = torch.rand((B, H, W), device=dev, generator=g) < 0.5
    m3 = torch.rand((B, H, W), device=dev, generator=g) < 0.3
    ce = CrossEntropyLossSemantic(weights=w)
    mse, l1, vm = MSELoss(), L1Loss(), VonMisesLossBiternion()
    # configs[2] names a center-FOCAL loss: not in the reference (its center loss is MSE | L1),
    # an extension here (loss/focal.py); heat-map peaks are exactly 1 in ~0.1 % of the pixels
    focal = CenterFocalLoss()
    center_t_peaks = torch.where(torch.rand((B, H, W), device=dev, generator=g) < 1e-3,
                                 torch.ones((), device=dev), center_t * 0.98)

    from nicr_mt_scene_analysis_amd.loss import _multi
    spec4, spec4f = _multi.SpecState(4), _multi.SpecState(4)

    def fwd(center_loss=mse, center_target=center_t):
        # configs[2] as ONE call (what the task helpers issue): count, expectation, one launch for
        # the four forward sums + gradients, finalize.  With the focal extension as center loss
        # the center total's divisor (the number of heat-map peaks) is only known after the loss
        # kernel ran: that total gets no expectation and its gradient is computed in backward
        items = [{'kind': 'ce', 'pred': logits, 'mask': labels, 'weights': w, 'total': 0},
                 {'kind': 'mse' if center_loss is mse else 'focal', 'pred': center, 'target': center_target,
                  'mask': m1, 'total': 1, 'clamp': center_loss is not mse},
                 {'kind': 'l1', 'pred': offset, 'target': offset_t, 'mask': m2, 'total': 2},
                 {'kind': 'vonmises', 'pred': ori, 'target': ori_t, 'mask': m3, 'param': 1.0, 'total': 3,
                  'clamp': True}]
        return _multi.multi_loss(items, 4, spec4 if center_loss is mse else spec4f).total_losses.sum()

    def fwd_bwd(center_loss=mse, center_target=center_t):
        for t in (logits, center, offset, ori):
            t.grad = None
        fwd(center_loss, center_target).backward()

    def ce_fwd_bwd():
        logits.grad = None
        (lc, n), = ce([logits], [labels])
        (lc / n).backward()
    n_px = B * H * W
    from nicr_mt_scene_analysis_amd.loss import (_functional as loss_f, reset_speculation_state,
                                                 speculation_stats)
    reset_speculation_state()                  # history of whatever ran before in this process
    if loss_f.mean_speculation_enabled():
        # forward kernels that also write the gradient: per loss the element count
        # (labels / mask, 1 B/px) + inputs once + gradient once; backward launches only confirm
        moved = (1 + 2 * C + 1 + 2 * C) + (1 + 7 + 2) + 2 * (1 + 13 + 4)
        how = ('ONE multi-loss call: count of labels / masks (1 B/px per loss; its last workgroup forms the '
               'expectation), one launch for the four forward sums + gradients, finalize; backward: one '
               'launch that compares on the device and recomputes nothing when the expectation held')
    else:
        # two-kernel path: forward inputs (2C+34) + log-sum-exp write 4; CE backward logits 2C +
        # label 1 + lse 4 + gradient 2C; element-wise backward pred + target + mask + gradient
        moved = (2 * C + 34) + 4 + (2 * C + 1 + 4 + 2 * C) + 9 + 17 + 17
        how = ('the backward kernels re-read their inputs (CE: logits + saved log-sum-exp)')
    spec0 = speculation_stats()
    with torch.no_grad():
        ms_f = hip_timed(fwd, reps=10, warm=3)
    ms_fb = hip_timed(fwd_bwd, reps=10, warm=3)
    ms_ce = hip_timed(ce_fwd_bwd, reps=10, warm=2)
    spec_mse = speculation_stats()
    ms_focal = hip_timed(lambda: fwd_bwd(focal, center_t_peaks), reps=10, warm=3)
    out = {'shape': f'B={B} C={C} {W}x{H}', 'pred_dtype': 'bfloat16',
           'four_losses_fwd': _leg(ms_f, n_px, 2 * C + 34),
           'four_losses_fwd_bwd': _leg(
               ms_fb, n_px, 2 * C + 34 + 2 * C + 2 + 4 + 4,
               moved_bytes_per_px=moved,
               frac_of_moved_bytes=round(n_px * moved / (ms_fb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               note='algorithmic (SURVEY 8d): inputs once + gradient writes = 204 B/px at C=40; '
                    + how + '; moved_bytes_per_px is what crosses HBM'),
           'four_losses_fwd_bwd_center_focal': _leg(
               ms_focal, n_px, 2 * C + 34 + 2 * C + 2 + 4 + 4,
               note='configs[2] as worded: the center loss is the focal extension (loss/focal.py, '
                    'not in the reference); its divisor is the number of heat-map peaks, which no '
                    '1 B/px count gives: same single call, but the center gradient is computed in '
                    'backward (the other three are confirmed)'),
           'ce_fwd_bwd': _leg(ms_ce, n_px, (2 * C + 1) + 2 * C,
                              note='algorithmic: logits + labels read once, gradient written')}
    out['backward_launches'] = {k: spec_mse[k] - spec0[k] for k in spec_mse}
    return out


def secondary_ce(dev, B=16, C=150, H=768, W=1024):
    """BASELINE configs[4] shape: the weighted cross entropy at 150 classes, bf16 — the class
    column does not fit the registers, forward + gradient come from one two-walk launch"""
    from nicr_mt_scene_analysis_amd.loss import (CrossEntropyLossSemantic, reset_speculation_state,
                                                 speculation_stats)
    reset_speculation_state()
    g = torch.Generator(device=dev).manual_seed(13)
    logits = (torch.randn((B, C, H, W), device=dev, generator=g) * 3).to(torch.bfloat16)
    logits.requires_grad_(True)
    labels = torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8)
    ce = CrossEntropyLossSemantic(weights=torch.rand(C, device=dev, generator=g) + 0.5)

    def fwd():
        (lc, n), = ce([logits], [labels])
        return lc / n

    def fwd_bwd():
        logits.grad = None
        fwd().backward()
    n_px = B * H * W
    spec0 = speculation_stats()
    with torch.no_grad():
        ms_f = hip_timed(fwd, reps=8, warm=2)
    ms_fb = hip_timed(fwd_bwd, reps=8, warm=2)
    spec1 = speculation_stats()
    return {'shape': f'B={B} C={C} {W}x{H}', 'pred_dtype': 'bfloat16',
            'fwd': _leg(ms_f, n_px, 2 * C + 1),
            'fwd_bwd': _leg(ms_fb, n_px, (2 * C + 1) + 2 * C,
                            note='algorithmic: logits + labels read once, gradient written; the '
                                 'second walk over each tile is served by the caches'),
            'backward_launches': {k: spec1[k] - spec0[k] for k in spec1}}


def cos_oracle_check(cos, emb, idx, lut, n_cols=4096):
    """after the timing: image 0 of the SAME batch (same launch geometry: the other images' indices
    are zeroed) through the loss + backward, against the C oracle — loss sum 1e-5, count exact,
    gradient on sampled pixel columns to bf16 rounding, images without targets exactly zero"""
    from oracle import oracle as orc
    B, D, H, W = emb.shape
    idx0 = idx.clone()
    idx0[1:] = 0
    emb.grad = None
    l0, n0 = cos.lut_sum(emb, idx0, lut)
    (l0 / n0).backward()
    torch.cuda.synchronize()
    want, want_n, want_grad = orc.loss_cosine_embedding(
        emb[0:1].detach().float().cpu().numpy(), idx0[0:1].cpu().numpy(), lut[0:1].cpu().numpy(), want_grad=True)
    rs = np.random.default_rng(5)
    cols = np.unique(np.concatenate([rs.integers(0, H * W, n_cols), [0, 1, 255, 256, H * W - 1]]))
    got = emb.grad[0].reshape(D, H * W)[:, torch.from_numpy(cols).to(emb.device)].float().cpu().numpy()
    ref = want_grad.reshape(D, H * W)[:, cols] / max(want_n, 1)
    scale = float(np.abs(ref).max())
    err = np.abs(got - ref) / (np.abs(ref) + 0.05 * scale)
    loss_rel = abs(float(l0) - want) / max(abs(want), 1e-30)
    others_zero = bool(B == 1 or not emb.grad[B - 1].any())
    ok = bool(loss_rel <= 1e-5 and int(n0) == want_n and float(err.max()) <= 2 ** -7 and others_zero)
    emb.grad = None
    return {'matches_oracle': ok, 'loss_rel_err': float(loss_rel), 'count_exact': bool(int(n0) == want_n),
            'grad_cols_checked': int(cols.size), 'grad_max_rel_err': float(err.max()),
            'what': 'image 0 of the timed batch at the timed launch geometry vs the C oracle '
                    '(loss 1e-5, gradient to bf16 rounding 2^-7)'}


def secondary_cos_emb(dev, B=8, D=512, H=768, W=1024, L=64):
    """BASELINE configs[4]: dense visual-embedding cosine loss at the DVEFormer shape
    (2D+4 B/px forward, +2D gradient write backward, bf16 predictions)"""
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss
    g = torch.Generator(device=dev).manual_seed(11)
    pred = torch.empty((B, D, H, W), device=dev, dtype=torch.bfloat16)
    for b in range(B):                                  # per image: bounds the f32 temporary
        pred[b] = torch.randn((D, H, W), device=dev, generator=g).to(torch.bfloat16)
    pred.requires_grad_(True)
    idx = torch.randint(0, L + 1, (B, H // 16, W // 16), device=dev, generator=g,
                        dtype=torch.int32)
    idx = idx.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()    # segments
    lut = torch.nn.functional.normalize(torch.randn((B, L, D), device=dev, generator=g), dim=-1)
    cos = CosineEmbeddingLoss()

    def fwd():
        return cos.lut_sum(pred, idx, lut)

    def fwd_bwd():
        pred.grad = None
        l, n = cos.lut_sum(pred, idx, lut)
        (l / n).backward()
    n_px = B * H * W
    from nicr_mt_scene_analysis_amd.loss import _multi as _m
    one_pass = bool(_m.cos_supported(pred, lut))
    with torch.no_grad():
        ms_f = hip_timed(fwd, reps=5, warm=2)
    ms_fb = hip_timed(fwd_bwd, reps=5, warm=2)
    return {'shape': f'B={B} D={D} {W}x{H} L={L}', 'pred_dtype': 'bfloat16', 'one_pass_kernel': one_pass,
            'oracle_check': cos_oracle_check(cos, pred, idx, lut),
            'fwd': _leg(ms_f, n_px, 2 * D + 4),
            'fwd_bwd': _leg(ms_fb, n_px, 2 * D + 4 + 2 * D,
                            moved_bytes_per_px=(2 * 2 * D + 8) if one_pass else (3 * 2 * D + 8),
                            note=('forward + gradient in ONE pass over the prediction (k_cos_split: the column '
                                  'stays in the registers of D / 64 waves): 2x2D+8 B/px moved') if one_pass else
                                 'the backward re-reads the prediction: 3x2D+8 B/px moved')}


def secondary_cfg5_full(ops, syn, dev, B=16, C=150, H=768, W=1024, K=48, D=512, L=64):
    """BASELINE configs[4] as ONE figure ("full pipeline incl. losses"): per step of a
    B=16 1024x768 batch with 150 classes, bf16 predictions — the panoptic pipeline (center NMS,
    grouping, merge) + the mIoU / PQ accumulator updates + forward and backward of the multi-task
    losses (CE at 150 classes, center MSE, offset L1, von Mises orientation: one multi-loss call)
    + forward and backward of the dense visual-embedding cosine loss (D=512, 64-row LUTs).
    Mpix/s = B H W / wall time of the step on one stream; the algorithmic bytes are the sums of
    the legs' SURVEY §8d figures."""
    from tools import bench_support
    from nicr_mt_scene_analysis_amd.loss import CosineEmbeddingLoss, _multi, reset_speculation_state
    reset_speculation_state()
    dt = torch.bfloat16
    inp = syn.make_panoptic_inputs_torch(B, C, H, W, n_centers=K, seed=4321, device=dev, logits_dtype=dt)
    a = (inp['semantic_logits'], inp['instance_center'], inp['instance_offset'],
         inp['semantic_classes_is_thing'])
    metrics = bench_support.MetricAccumulators(C + 1, dev, inp, 0, side_stream=False)
    g = torch.Generator(device=dev).manual_seed(17)

    def rnd(*shape):
        return torch.randn(shape, device=dev, generator=g)
    logits = a[0].detach().clone().requires_grad_(True)          # the loss sees the same logits
    labels = torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8)
    w = torch.rand(C, device=dev, generator=g) + 0.5
    center = torch.rand((B, H, W), device=dev, generator=g).to(dt).requires_grad_(True)
    center_t = torch.rand((B, H, W), device=dev, generator=g)
    offset = rnd(B, 2, H, W).to(dt).requires_grad_(True)
    offset_t = rnd(B, 2, H, W)
    ori = rnd(B, 2, H, W).to(dt).requires_grad_(True)
    ori_t = torch.nn.functional.normalize(rnd(B, 2, H, W), dim=1)
    m1 = torch.rand((B, H, W), device=dev, generator=g) < 0.7
    m2 = torch.rand((B, H, W), device=dev, generator=g) < 0.5
    m3 = torch.rand((B, H, W), device=dev, generator=g) < 0.3
    emb = torch.empty((B, D, H, W), device=dev, dtype=dt)
    for b in range(B):                                  # per image: bounds the f32 temporary
        emb[b] = torch.randn((D, H, W), device=dev, generator=g).to(dt)
    emb.requires_grad_(True)
    idx = torch.randint(0, L + 1, (B, H // 16, W // 16), device=dev, generator=g, dtype=torch.int32)
    idx = idx.repeat_interleave(16, 1).repeat_interleave(16, 2).contiguous()
    lut = torch.nn.functional.normalize(torch.randn((B, L, D), device=dev, generator=g), dim=-1)
    cos = CosineEmbeddingLoss()
    spec = _multi.SpecState(4)
    items = [{'kind': 'ce', 'pred': logits, 'mask': labels, 'weights': w, 'total': 0},
             {'kind': 'mse', 'pred': center, 'target': center_t, 'mask': m1, 'total': 1},
             {'kind': 'l1', 'pred': offset, 'target': offset_t, 'mask': m2, 'total': 2},
             {'kind': 'vonmises', 'pred': ori, 'target': ori_t, 'mask': m3, 'param': 1.0, 'total': 3,
              'clamp': True}]
    leaves = (logits, center, offset, ori, emb)

    def step():
        r = ops.panoptic_pipeline(*a, want_foreground=False)
        metrics.update_and_reduce(r)
        for t in leaves:
            t.grad = None
        total = _multi.multi_loss(items, 4, spec).total_losses.sum()
        l, n = cos.lut_sum(emb, idx, lut)
        (total + l / n).backward()
    ms = hip_timed(step, reps=6, warm=3)
    metrics.pq._check_status()
    metrics.miou._check_status()
    n_px = B * H * W
    parts = {'pipeline': 2 * C + 21, 'metrics': 17, 'multitask_losses_fwd_bwd': (2 * C + 34) + (2 * C + 10),
             'cos_emb_fwd_bwd': 2 * D + 4 + 2 * D}
    out = _leg(ms, n_px, sum(parts.values()), algorithmic_bytes_per_px_by_part=parts,
               shape=f'B={B} C={C} {W}x{H} D={D} L={L}', pred_dtype='bfloat16',
               what='panoptic pipeline + mIoU/PQ updates + multi-task losses fwd+bwd + cosine-embedding '
                    'loss fwd+bwd, one stream, HIP events around the whole step')
    out['cos_oracle_check'] = cos_oracle_check(cos, emb, idx, lut)
    del inp, a, emb, logits
    torch.cuda.empty_cache()
    return out


def secondary_next_rows(ops, syn, dev, B=32, C=40, H=480, W=640):
    """SURVEY §8(f) rows that sit either side of the headline path, each as one HIP-event-timed
    call on B=32 640x480 inputs: f2 the full-resolution step (crop + bilinear resize + softmax
    + max fused, 640x480 -> 730x530 and -> 1024x768; the nearest resize of the panoptic map),
    f3 `compute_scores`, f4 on-device target generation.  Algorithmic bytes per OUTPUT pixel
    for f2 (source logits read once + index / score written), per pixel otherwise."""
    out = {}
    inp = syn.make_panoptic_inputs_torch(B, C, H, W, n_centers=24, seed=99, device=dev)
    x = inp['semantic_logits']
    p = ops.panoptic_pipeline(x, inp['instance_center'], inp['instance_offset'],
                              inp['semantic_classes_is_thing'], want_score=True)
    n_px = B * H * W
    in_bytes = x.numel() * x.element_size()
    for size in ((530, 730), (768, 1024)):
        n_out = B * size[0] * size[1]
        ms = hip_timed(lambda: ops.semantic_argmax_resized(x, size, None, want_score=False),
                       reps=10, warm=3)
        out[f'f2_argmax_resized_{size[1]}x{size[0]}'] = _leg(
            ms, n_out, round((in_bytes + n_out * 8) / n_out, 2),
            what='fused crop + bilinear + softmax + max, int64 index out (semantic.py:61-80); '
                 'VALU / LDS-issue bound, not HBM-bound')
        ms = hip_timed(lambda: ops.resize_nearest(p['panoptic'], size, None), reps=10, warm=3)
        out[f'f2_nearest_i64_{size[1]}x{size[0]}'] = _leg(
            ms, n_out, round((n_px * 8 + n_out * 8) / n_out, 2), what='dense_base.py:15-58')
    tab = torch.zeros((B, 256), dtype=torch.float32, device=dev)
    tab[:, 1:] = p['center_scores'][:, :255]
    ms = hip_timed(lambda: ops.panoptic_scores(x, p['semantic_idx_u8'], p['semantic_score'],
                                               p['instance'], p['panoptic'], p['pan_of_inst'],
                                               tab, 1 << 16), reps=10, warm=3)
    out['f3_compute_scores'] = _leg(ms, n_px, 4 + 1 + 1 + 8 + 4 + 4 + 1 + 8 + 8,
                                    what='panoptic.py:171-239: two kernels')
    m = syn.make_label_maps(2, C + 1, H, W, 30, seed=2)
    reps = (B + 1) // 2
    sem = torch.from_numpy(np.tile(m['semantic'], (reps, 1, 1))[:B]).to(dev)
    ins = torch.from_numpy(np.tile(m['instance'], (reps, 1, 1))[:B]).to(dev)
    th = torch.from_numpy(m['semantic_classes_is_thing'].astype(np.uint8)).to(dev)
    st = torch.from_numpy((~m['semantic_classes_is_thing']).astype(np.uint8)).to(dev)
    ops.instance_clear_stuff(sem, ins, st)
    ms = hip_timed(lambda: ops.instance_targets(sem, ins, C + 1, th, st, 8, True), reps=10, warm=3)
    out['f4_instance_targets'] = _leg(ms, n_px, 1 + 4 + 4 + 8 + 1 + 1,
                                      what='data/preprocessing/instance.py:97-286 per batch: labels in, center / '
                                           'offset / masks out; memset + one scan launch (presence, statistics, '
                                           'rank, decide) + the paint launch')
    ms = hip_timed(lambda: ops.panoptic_targets(sem, ins, C + 1, th, 1 << 16), reps=10, warm=3)
    out['f4_panoptic_targets'] = _leg(ms, n_px, 1 + 4 + 8,
                                      what='data/preprocessing/panoptic.py:16-85 per batch: memset + one scan '
                                           'launch (presence, class histograms, rank, naive ranks) + the paint launch')
    return out


def secondary_api(syn, dev):
    """the reference-shaped API on the driver's clock: wall time per call incl.
    the host, and the host's own share (`host_issue_ms`: time until the last launch of a call has
    been queued).  postprocess: PanopticPostprocessing.postprocess B=32 (eager = one device->host
    copy per call; deferred = none); validation_step: postprocess + PanopticTaskHelper +
    SemanticTaskHelper validation steps; training_step: SemanticTaskHelper + InstanceTaskHelper
    training_step + backward, B=64 bf16, main + two side outputs (1/2, 1/4)."""
    from nicr_mt_scene_analysis_amd.data.preprocessing import APPLIED_PREPROCESSING_KEY
    from nicr_mt_scene_analysis_amd.model.postprocessing import get_postprocessing_class
    from nicr_mt_scene_analysis_amd.task_helper import (InstanceTaskHelper, PanopticTaskHelper,
                                                        SemanticTaskHelper)
    from nicr_mt_scene_analysis_amd.task_helper.base import get_total_loss_key
    from nicr_mt_scene_analysis_amd.loss import reset_speculation_state, speculation_stats
    out = {}

    def timed(fn, n, warm):
        for i in range(warm):
            fn(i)
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(n):
                fn(i)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if best is None or t2 - t0 < best[0]:
                best = (t2 - t0, t1 - t0)
        return best[0] / n * 1e3, best[1] / n * 1e3

    B, C, H, W = 32, 40, 480, 640
    inp = syn.make_panoptic_inputs_torch(B, C, H, W, device=dev, seed=1)
    is_thing = tuple(bool(x) for x in inp['semantic_classes_is_thing'].cpu().tolist())
    g = torch.Generator(device=dev).manual_seed(3)
    batch = {'rgb_fullres': torch.zeros((B, 3, H, W)),
             'semantic': torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8),
             'semantic_fullres': torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8),
             'panoptic_ids_to_instance_dict': [{} for _ in range(B)],
             APPLIED_PREPROCESSING_KEY: [[{'type': 'Resize', 'valid_region_slice_y': slice(0, H),
                                           'valid_region_slice_x': slice(0, W)}]] * B}
    data = ((inp['semantic_logits'], (inp['instance_center'], inp['instance_offset'])), (None, None))

    def make_post(defer):
        return get_postprocessing_class('panoptic')(
            semantic_postprocessing=get_postprocessing_class('semantic')(),
            instance_postprocessing=get_postprocessing_class('instance')(),
            semantic_classes_is_thing=is_thing, semantic_class_has_orientation=is_thing,
            defer_host_sync=defer)
    pan = make_post(False).postprocess(data, batch, is_training=False)['panoptic_segmentation_deeplab_fullres']
    tgt = torch.roll(pan, shifts=(3, 3), dims=(1, 2)).contiguous()
    tgt[:, :3] = 0
    batch['panoptic_fullres'] = tgt
    n_px = B * H * W
    for defer in (False, True):
        post = make_post(defer)
        ms, host = timed(lambda i: post.postprocess(data, batch, is_training=False), 40, 5)
        out['postprocess_deferred' if defer else 'postprocess'] = _leg(
            ms, n_px, 4 * C + 21, host_issue_ms=round(host, 4),
            what='PanopticPostprocessing.postprocess, B=32 640x480 C=40 f32'
                 + (' (defer_host_sync=True)' if defer else ''))
        ph = PanopticTaskHelper(C + 1, (False,) + is_thing)
        sh = SemanticTaskHelper(n_classes=C, disable_multiscale_supervision=True)
        ph.initialize(dev)
        sh.initialize(dev)

        def vstep(i):
            r = post.postprocess(data, batch, is_training=False)
            ph.validation_step(batch, i, r)
            sh.validation_step(batch, i, r)
        ms, host = timed(vstep, 30, 5)
        out['validation_step_deferred' if defer else 'validation_step'] = _leg(
            ms, n_px, 4 * C + 21 + 17 + 4 * C + 1 + 2, host_issue_ms=round(host, 4),
            what='postprocess + PanopticTaskHelper.validation_step (PQ + mIoU) + '
                 'SemanticTaskHelper.validation_step (CE + mIoU)')
        ph.validation_epoch_end()
        sh.validation_epoch_end()
    del inp, data, batch, pan, tgt
    torch.cuda.empty_cache()

    # ---- training step through the task helpers, B=64 bf16, three supervision scales -------------
    B = 64
    dt = torch.bfloat16

    def rnd(*shape, dtype=torch.float32):
        return torch.randn(shape, device=dev, generator=g).to(dtype)
    batch = {
        'semantic': torch.randint(0, C + 1, (B, H, W), device=dev, generator=g).to(torch.uint8),
        'instance_center': torch.rand((B, H, W), device=dev, generator=g),
        'instance_center_mask': torch.rand((B, H, W), device=dev, generator=g) < 0.7,
        'instance_offset': rnd(B, 2, H, W) * 0.1,
        'instance_foreground': torch.rand((B, H, W), device=dev, generator=g) < 0.5,
        'orientation': torch.nn.functional.normalize(rnd(B, 2, H, W), dim=1),
        'orientation_foreground': torch.rand((B, H, W), device=dev, generator=g) < 0.3,
    }
    for sc in (2, 4):
        batch[f'_down_{sc}'] = {k: v[..., ::sc, ::sc].contiguous() for k, v in batch.items()
                               if isinstance(v, torch.Tensor)}
    sem_main = rnd(B, C, H, W, dtype=dt).requires_grad_(True)
    sem_side = tuple(rnd(B, C, H // sc, W // sc, dtype=dt).requires_grad_(True) for sc in (2, 4))

    def inst(sc):
        return tuple(t.requires_grad_(True) for t in (
            rnd(B, 1, H // sc, W // sc, dtype=dt), rnd(B, 2, H // sc, W // sc, dtype=dt),
            torch.nn.functional.normalize(rnd(B, 2, H // sc, W // sc), dim=1).to(dt)))
    preds = {'semantic_output': sem_main, 'semantic_side_outputs': sem_side,
             'instance_output': inst(1), 'instance_side_outputs': (inst(2), inst(4))}
    sem = SemanticTaskHelper(n_classes=C, class_weights=torch.rand(C) + 0.5)
    ins = InstanceTaskHelper(semantic_n_classes=C + 1, semantic_classes_is_thing=(False,) + is_thing)
    sem.initialize(dev)
    ins.initialize(dev)
    leaves = [sem_main, *sem_side, *preds['instance_output'],
              *(t for side in preds['instance_side_outputs'] for t in side)]
    loss_weights = {'semantic': 1.0, 'instance_center': 2.0, 'instance_offset': 0.5,
                    'instance_orientation': 1.0}                  # FixedLossWeighting-style constants

    def tstep(i):
        for t in leaves:
            t.grad = None
        ls, _ = sem.training_step(batch, i, preds)
        li, _ = ins.training_step(batch, i, preds)
        total = sum(wt * (ls if k == 'semantic' else li)[get_total_loss_key(k)]
                    for k, wt in loss_weights.items())
        total.backward()
    reset_speculation_state()
    s0 = speculation_stats()
    ms, host = timed(tstep, 20, 4)
    s1 = speculation_stats()
    if os.environ.get('NMSA_BENCH_API_PROFILE'):                   # where the host's share goes (stderr)
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(10):
            tstep(i)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr, stream=sys.stderr).sort_stats('cumulative').print_stats(45)
    scale_px = int(B * H * W * (1 + 1 / 4 + 1 / 16))
    out['training_step'] = _leg(
        ms, scale_px, 2 * C + 34 + 2 * C + 2 + 4 + 4, host_issue_ms=round(host, 4),
        what='SemanticTaskHelper + InstanceTaskHelper training_step + backward, B=64 bf16, main + two '
             'side outputs, loss weights (1, 2, 0.5, 1) never announced (no backward_scale)',
        backward_totals={k: s1[k] - s0[k] for k in s1})
    return out


def secondary(ops, syn, dev):
    out = {}
    legs = (
        ('cfg2_bf16', lambda: secondary_pipeline(ops, syn, dev, 32, 40, 480, 640, 24,
                                                 torch.bfloat16)),
        ('cfg3_losses', lambda: secondary_losses(dev)),
        ('cfg5_bf16', lambda: secondary_pipeline(ops, syn, dev, 16, 150, 768, 1024, 48,
                                                 torch.bfloat16)),
        ('cfg5_ce_C150', lambda: secondary_ce(dev)),
        ('cfg5_cos_emb_D512', lambda: secondary_cos_emb(dev, B=16, D=512)),
        ('cfg5_cos_emb_D768', lambda: secondary_cos_emb(dev, B=16, D=768)),
        ('cfg5_full', lambda: secondary_cfg5_full(ops, syn, dev)),
        ('next_rows', lambda: secondary_next_rows(ops, syn, dev)),
        ('api', lambda: secondary_api(syn, dev)),
    )
    for name, fn in legs:
        try:
            out[name] = fn()
        except Exception as e:                      # a failed leg must not lose the headline ...
            out[name] = {'error': f'{type(e).__name__}: {e}'[:300]}
            out['failed_legs'] = out.get('failed_legs', []) + [name]      # ... but it must show
        torch.cuda.empty_cache()
    return out


def main():
    args = parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args))
    launched = 'RANK' in os.environ
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    from nicr_mt_scene_analysis_amd import ops
    from nicr_mt_scene_analysis_amd.testing import synthetic as syn

    host_threads = host_threads_per_rank(world)
    torch.set_num_threads(host_threads)          # intra-op pool of this rank (see launch_ranks)
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)       # rehearsal: several ranks may share one GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device('cuda', dev_index)
    dist = None
    backend = None
    rccl_ranks = None
    rccl_error = None
    if launched:                                 # under a launcher also at N=1 (same code path)
        import torch.distributed as dist_
        dist = dist_
        # 'nccl' == RCCL on ROCm.  NMSA_BENCH_BACKEND=gloo is only for rehearsing the
        # multi-rank path on a box with fewer GPUs than ranks (RCCL refuses shared devices).
        backend = os.environ.get('NMSA_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            try:
                if os.environ.get('NMSA_BENCH_FORCE_RCCL_FAIL'):            # test hook
                    raise RuntimeError('NMSA_BENCH_FORCE_RCCL_FAIL: RCCL declared unavailable on request')
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
                one = torch.ones((1,), dtype=torch.float64, device=dev)
                dist.all_reduce(one, op=dist.ReduceOp.SUM)     # an actual RCCL all-reduce
                rccl_ranks = int(one.item())
            except Exception as e:               # noqa: BLE001 — whatever the runtime raises
                # A node whose RCCL cannot start (IPC, topology) still gets its scaling figures:
                # the ONE ~15 KB accumulator all-reduce of the run goes through gloo (staged
                # through the host), said so in the line (`collective.backend`, `rccl_error`).
                # The data path never had a collective.
                rccl_error = f'{type(e).__name__}: {e}'.replace('\n', ' ')[:400]
                print(f'[bench] rank {rank}: RCCL unavailable ({rccl_error}); the accumulator '
                      f'all-reduce falls back to gloo', file=sys.stderr, flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = 'gloo'
                import datetime
                dist.init_process_group('gloo', rank=rank, world_size=world,
                                        timeout=datetime.timedelta(seconds=180))
            if rccl_ranks is not None and rccl_ranks != world:
                raise SystemExit(f'RCCL all-reduce saw {rccl_ranks} ranks, expected {world}')
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if os.environ.get('NMSA_BENCH_FAIL_RANK') == str(rank):    # test hook (tests/test_bench_launch.py)
            raise RuntimeError(f'NMSA_BENCH_FAIL_RANK: rank {rank} fails on request')

    B, C, H, W = args.batch_per_gpu, args.classes, args.height, args.width
    logits_dtype = {'f32': None, 'bf16': torch.bfloat16, 'f16': torch.float16}[args.dtype]
    inp = syn.make_panoptic_inputs_torch(B, C, H, W, n_centers=args.centers,
                                         seed=1234 + rank, device=dev, logits_dtype=logits_dtype)
    logits, center, offset = inp['semantic_logits'], inp['instance_center'], inp['instance_offset']
    is_thing = inp['semantic_classes_is_thing']
    torch.cuda.synchronize()

    metrics = None
    if not args.no_metrics:
        from tools import bench_support
        metrics = bench_support.MetricAccumulators(C + 1, dev, inp, rank, world_size=world,
                                                   side_stream=not args.no_side_stream,
                                                   sync_every_step=args.metric_sync == 'step')

    events = []
    # consecutive batches are independent: they alternate over `--streams` HIP streams so that
    # the latency-bound per-image kernels of one batch (top-k select, class/rank assignment,
    # PQ matching) overlap with the streaming kernels of the next one
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(args.streams, 1))]

    def step(i, record):
        with torch.cuda.stream(streams[i % len(streams)]):
            # (the foreground mask — a lookup of the class map, 1 B/px more to store — is not part of
            # the step: neither the merge nor the metrics read it; the API builds it when read)
            r = ops.panoptic_pipeline(logits, center, offset, is_thing, want_foreground=False,
                                      fused_kernel_events=events if record else None)
            if metrics is not None:
                metrics.update_and_reduce(r, dist)
        return r

    for i in range(args.warmup):
        r = step(i, False)
    # hipGraphs of one step per batch stream (captured behind the warm-up, on the very streams the
    # eager steps use: same persistent vote table / PQ workspace per stream, so eager and replayed
    # steps may alternate): [pipeline graph on the batch stream] -> event -> [metric-chain graph on
    # the metric stream].  The metric chains of consecutive steps stay on ONE stream (the fp64
    # accumulation is ordered), so they are graphs of their own.  Per-step reduction over the ranks
    # (--metric-sync step) and runs without the side stream stay eager.
    graphs = None
    use_graph = (not args.no_graph and len(streams) >= 1
                 and (metrics is None or (metrics.stream is not None and not metrics.sync_every_step)))
    graph_error = None
    if use_graph:
        # (thread-local capture mode: the RCCL watchdog thread of a launched run polls events
        # while we capture; a failed capture falls back to the eager loop and says so in the line)
        try:
            torch.cuda.synchronize()
            graphs = []
            # TWO graph slots per batch stream (slot k runs on stream k % n): a slot's outputs sit at
            # fixed addresses and the next replay of that slot rewrites them, so it has to wait for
            # the metric chain that still reads them (`metric_done`, graph_step) — with two slots per
            # stream that chain is four steps old and long finished; with one it was the chain of
            # the step before last, still running next to the other stream's batch
            for k in range(2 * len(streams)):
                st = streams[k % len(streams)]
                gp = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gp, stream=st, capture_error_mode='thread_local'):
                    out = ops.panoptic_pipeline(logits, center, offset, is_thing, want_foreground=False)
                gm = None
                if metrics is not None:
                    torch.cuda.synchronize()
                    gm = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gm, stream=metrics.stream, capture_error_mode='thread_local'):
                        metrics.enqueue(out)
                graphs.append((gp, gm, out, torch.cuda.Event(), torch.cuda.Event()))
            torch.cuda.synchronize()
        except Exception as e:                  # noqa: BLE001 — any capture problem: measure eagerly
            graphs = None
            graph_error = f'{type(e).__name__}: {e}'[:200]
            print(f'bench.py: hipGraph capture failed, eager loop instead: {graph_error}', file=sys.stderr, flush=True)
            torch.cuda.synchronize()

    replayed = [False] * (2 * len(streams))

    def graph_step(i):
        slot = i % (2 * len(streams))
        gp, gm, out, ready, metric_done = graphs[slot]
        st = streams[slot % len(streams)]              # (= streams[i % len(streams)])
        with torch.cuda.stream(st):
            # the pipeline graph rewrites `out` in place (fixed addresses in the graph's pool): the
            # metric chain of this slot's PREVIOUS step, which reads `out` on the metric stream, must
            # be through with it (a real pipeline with changing inputs needs this dependency; with
            # identical inputs every step its absence went unnoticed)
            if gm is not None and replayed[slot]:
                st.wait_event(metric_done)
            gp.replay()
            ready.record(st)
        if gm is not None:
            metrics.stream.wait_event(ready)
            with torch.cuda.stream(metrics.stream):
                gm.replay()
                metric_done.record(metrics.stream)
            replayed[slot] = True
        return out

    if graphs is not None:                      # one untimed replay per slot: part of the warm-up
        for i in range(2 * len(streams)):
            r = graph_step(i)
    if metrics is not None:
        metrics.warm_collective(dist)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()                               # no collector pause inside the (few-ms) timed region
    # the live duration of the dominant kernel is sampled on every 8th step: each pair
    # of event records costs the pipeline ~4 us, 1.2 % of a K = 20 run when every step carries one
    # (neither the first nor the last step: a replayed step reaches the GPU at once, an eager one
    # over the ~0.17 ms its launches take from Python — at the head and in the tail that is idle time)
    sampled = [i % 8 == 2 and i < args.steps - 1 for i in range(args.steps)]
    if not any(sampled):
        sampled[args.steps // 2] = True
    done = [torch.cuda.Event() for _ in range(len(streams) + 1)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        r = step(i, True) if (sampled[i] or graphs is None) else graph_step(i)
    host_issue = time.perf_counter() - t0      # the host is done queueing; the GPU is not
    if metrics is not None:
        metrics.finalize(dist)                 # --metric-sync end: the one all-reduce, timed
    # spin on an event behind the last kernel of every stream (hipEventSynchronize of a
    # non-blocking event polls; the device-wide wait below sleeps on an interrupt and wakes up
    # 30-60 us late), then the synchronisation the contract asks for
    for ev, st in zip(done, streams + ([metrics.stream] if metrics is not None and metrics.stream else [])):
        ev.record(st)
    for ev, st in zip(done, streams + ([metrics.stream] if metrics is not None and metrics.stream else [])):
        ev.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device=dev if dist.get_backend() == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    host_issue_ranks = None
    if dist is not None:
        hi = torch.tensor([host_issue / args.steps * 1e3], dtype=torch.float64,
                          device=dev if dist.get_backend() == 'nccl' else 'cpu')
        got = [torch.zeros_like(hi) for _ in range(world)]
        dist.all_gather(got, hi)
        host_issue_ranks = [round(float(g.item()), 4) for g in got]

    # after the timed region: every rank must hold the same reduced accumulators (checksum of
    # the confusion matrix and the PQ vectors, gathered over the ranks)
    totals_identical = None
    if dist is not None and metrics is not None:
        metrics.wait()
        torch.cuda.synchronize()
        chk = torch.stack([metrics.total_confmat.sum().double(),
                           metrics.total_pq.double().sum()]).to(dev)
        if dist.get_backend() != 'nccl':
            chk = chk.cpu()
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        totals_identical = bool(all(torch.equal(g, gathered[0]) for g in gathered))

    # every step (warm-up, graph warm-up replays, timed) must have reached the accumulators exactly
    # once: the confusion matrix counts one entry per pixel and update
    if metrics is not None and dist is None:
        metrics.wait()
        torch.cuda.synchronize()
        n_updates = args.warmup + (2 * len(streams) if graphs is not None else 0) + args.steps
        counted = int(metrics.total_confmat.sum().item())
        if counted != n_updates * B * H * W:
            raise SystemExit(f'metric accumulators saw {counted / (B * H * W):.3f} updates, expected {n_updates}')

    n_px_step = B * H * W * world
    ms_per_step = elapsed / args.steps * 1e3
    value = n_px_step / (elapsed / args.steps) / 1e6

    # ---- roofline of the dominant kernel (fused argmax + grouping + votes) ----------
    # algorithmic bytes per launch: logits 4C + offsets 8 read, instance u8 1 written,
    # per pixel (DESIGN.md "Kernels"); the intermediate sem u8 (1 B/px) is NOT counted
    fused_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))
    esize = logits.element_size()
    fused_bytes_px = esize * C + 8 + 1
    fused_bytes = fused_bytes_px * B * H * W
    achieved = fused_bytes / (fused_ms * 1e-3) / 1e9
    pipeline_bytes_px = esize * C + 4 + 8 + 8 + 1           # 181 B/px at f32, C=40
    if metrics is not None:
        pipeline_bytes_px += 8 + 8 + 1                      # + pred pan, target pan, target sem
    # the same kernel without the metric kernels overlapping on the side stream (untimed)
    # (200 launches: in a `rocprofv3 --kernel-trace --stats` run of this very command they outnumber
    # the ~45 launches of warm-up and timed region, so the summary's average duration of the kernel
    # — profiles/<round>_kernel_stats.csv — is this number to within a few percent)
    n_iso = 200
    iso_events = []
    for _ in range(n_iso + 2):
        ops.panoptic_pipeline(logits, center, offset, is_thing, want_foreground=False,
                              fused_kernel_events=iso_events)
    torch.cuda.synchronize()
    iso_ms = float(np.mean([a.elapsed_time(b) for a, b in iso_events[2:]]))
    # HBM bytes per launch from the committed PMC passes of this command (separate
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, gfx950 wide-stream correction applied by
    # tools/summarize_profile.py); null when the workload differs from the profiled one
    traffic = None
    traffic_sha = None
    tpath = os.path.join(ROOT, 'profiles', 'latest_traffic.json')
    if os.path.exists(tpath) and (B, C, H, W) == (32, 40, 480, 640) and esize == 4:
        import hashlib
        with open(tpath, 'rb') as f:
            raw = f.read()
        traffic_sha = hashlib.sha256(raw).hexdigest()[:16]
        traffic = next((v.get('hbm_bytes_per_launch') for k, v in json.loads(raw).items()
                        if k.startswith('k_panoptic_fused')), None)
    # `frac`: the kernel alone on the chip (the isolated pass above: what a rocprofv3 kernel trace of
    # this command shows for it); `frac_eager`: HIP events around the launches sampled INSIDE the
    # timed region — these launches are issued kernel by kernel from Python and queue behind the
    # other batch in flight, so the figure includes waiting for wave slots and reads lower
    roofline = {
        'bound': 'hbm', 'kernel': 'k_panoptic_fused',
        'achieved': round(fused_bytes / (iso_ms * 1e-3) / 1e9, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
        'frac': round(fused_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        'kernel_ms': round(iso_ms, 4), 'kernel_launches_isolated': n_iso,
        'frac_eager': round(achieved / HBM_PEAK_GBS, 4), 'achieved_eager': round(achieved, 1),
        'kernel_ms_eager': round(fused_ms, 4), 'kernel_launches_timed': len(events),
        # HBM bytes per launch are NOT measured by this run: a constant from the committed PMC passes
        # of this command (profiles/latest_traffic.json; its content hash says which)
        'traffic': traffic, 'traffic_profiled': traffic,
        'traffic_source': ('profiles/latest_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                           'this command, per launch, gfx950 wide-stream correction)') if traffic else None,
        'traffic_profile_sha256': traffic_sha,
        'algorithmic_bytes_per_launch': fused_bytes,
        'algorithmic_bytes_per_px': fused_bytes_px,
        'pipeline_algorithmic_bytes_per_px': pipeline_bytes_px,
        'pipeline_frac': round(pipeline_bytes_px * B * H * W / (ms_per_step * 1e-3) / 1e9
                               / HBM_PEAK_GBS, 4),
    }

    out = {
        'metric': 'Mpix/s panoptic merge+metrics, 640x480xB',
        'value': round(value, 1), 'unit': 'Mpix/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4),
        # host time to queue one step (this rank): below ms_per_step = the GPU sets the pace
        'host_issue_ms_per_step': round(host_issue / args.steps * 1e3, 4),
        'host_threads_per_rank': host_threads,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {torch.float32: 'f32', torch.bfloat16: 'bf16', torch.float16: 'f16'}[logits.dtype],
        'data': 'synthetic',
        'config': {'workload': 'configs[1]: center-NMS + offset grouping + panoptic merge'
                               + (' + mIoU/PQ accumulators (configs[3])' if metrics else ''),
                   'batch_per_gpu': B, 'global_batch': B * world, 'classes': C,
                   'height': H, 'width': W, 'centers_per_image': args.centers,
                   'batches_in_flight': len(streams),
                   'launch': ('hipGraph replay (pipeline graph + metric-chain graph per step); the steps '
                              'that carry event records are launched kernel by kernel') if graphs is not None
                   else 'kernel by kernel from Python' + (f' (graph capture failed: {graph_error})' if graph_error else ''),
                   'parallelism': f'dp{world} (images sharded, accumulators all-reduced '
                                  f'{"every step" if args.metric_sync == "step" else "once per run, timed"})'},
        'collective': {'backend': ('rccl' if backend == 'nccl' else backend), 'rccl_ranks': rccl_ranks,
                       'rccl_error': rccl_error,
                       'payload_bytes': metrics.payload_bytes if metrics is not None else 0,
                       'calls': 'Metric.sync(): one all-reduce per state dtype (int64, float64)',
                       'totals_identical_on_all_ranks': totals_identical,
                       'host_issue_ms_per_step_ranks': host_issue_ranks}
        if dist is not None else None,
        'roofline': roofline,
    }

    solo = rank == 0 and world == 1
    if solo and not args.no_cpu_baseline:                          # contract: rank 0 at N=1 only
        cb, (idx, inst, pan) = cpu_baseline(inp, args.cpu_sample_images, C, H, W, metrics)
        n = idx.shape[0]
        ok = bool((r['semantic_idx_u8'][:n].cpu().numpy() == idx).all()
                  and (r['instance'][:n].cpu().numpy() == inst).all()
                  and (r['panoptic'][:n].cpu().numpy() == pan).all())
        cb['gpu_matches_oracle_bit_exact'] = ok
        out['cpu_baseline'] = cb
    elif rank == 0:
        out['cpu_baseline'] = None
    if solo and not args.no_secondary:
        del inp, logits, center, offset, r, metrics
        torch.cuda.empty_cache()
        out['secondary'] = secondary(ops, syn, dev)

    failed = bool(out.get('secondary', {}).get('failed_legs'))
    if failed:
        out['secondary_failed'] = out['secondary']['failed_legs']
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        # the line above is complete and says so at top level (`secondary_failed`); the headline
        # measurement stands, so the exit code stays 0 unless NMSA_BENCH_STRICT=1 asks otherwise
        print(f'bench.py: secondary legs raised: {out["secondary_failed"]}', file=sys.stderr, flush=True)
        if os.environ.get('NMSA_BENCH_STRICT') == '1':
            sys.exit(3)


if __name__ == '__main__':
    main()
