"""CPU tier: code-generation guard for the hot kernels (hipcc cross-compiles without a GPU).

Round 4 found three kinds of silent regressions in the generated code, none of which changes a
result: register spills to scratch memory (100 B per lane were 8 % of the wide cross entropy's HBM
writes), hundreds of loop-invariant scalar plane offsets spilled to lane registers, and hash-probe
loops unrolled into 37 900 instructions.  This test compiles the three files that hold those
kernels with LLVM's kernel-resource-usage remarks and checks the hot instantiations.
(`tools/isa_survey.sh` prints the same survey for every kernel of the library.)"""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'nicr_mt_scene_analysis_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=off', '-Wno-unused-function',
         '--offload-device-only', '-Rpass-analysis=kernel-resource-usage', '-c', '-o', os.devnull]


def _usage(src, extra=()):
    """{mangled kernel name: {'vgpr', 'scratch', 'sgpr_spill', 'occupancy'}} of one source file"""
    out = subprocess.run([HIPCC, *FLAGS, *extra, src], cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r'remark:\s+Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            kernels[name] = {}
            continue
        for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                         ('sgpr_spill', r'SGPRs Spill: (\d+)'), ('occupancy', r'Occupancy \[waves/SIMD\]: (\d+)')):
            m = re.search(pat, line)
            if m and name:
                kernels[name][key] = int(m.group(1))
    return kernels


@pytest.fixture(scope='module')
def usage():
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not available')
    jobs = {'losses_split.hip': (), 'losses_cos.hip': (), 'metrics.hip': ()}
    with ThreadPoolExecutor(max_workers=3) as pool:
        res = dict(zip(jobs, pool.map(lambda kv: _usage(kv[0], kv[1]), jobs.items())))
    return res


def test_wide_cross_entropy_has_no_scratch(usage):
    """every instantiation of k_ce_split (dtype x plane groups x smoothing x mode x row form)"""
    ks = {k: v for k, v in usage['losses_split.hip'].items() if 'k_ce_split' in k}
    assert len(ks) >= 48
    bad = {k: v for k, v in ks.items() if v['scratch'] != 0}
    assert not bad, bad
    # configs[4]: bf16, 150 classes (5 groups of 8 planes per wave), loss + gradient, 8-byte rows
    hot = [v for k, v in ks.items() if 'ILi1ELi5ELb0ELi0ELb1E' in k]
    assert len(hot) == 1 and hot[0]['occupancy'] >= 3 and hot[0]['sgpr_spill'] <= 16, hot


def test_cosine_kernels_keep_their_plane_offsets_out_of_lane_registers(usage):
    ks = usage['losses_cos.hip']
    split = {k: v for k, v in ks.items() if 'k_cos_split' in k}
    parts = {k: v for k, v in ks.items() if 'k_cos_parts' in k}
    assert split and parts
    for k, v in {**split, **parts}.items():
        # (the gradient-only instantiation — the recompute after a missed expectation — may keep a
        # few bytes: it sits at the 256-register limit)
        recompute = re.search(r'k_cos_(parts|split)ILi\dELi2E', k) is not None
        assert v['scratch'] <= (16 if recompute else 0), (k, v)
    for k, v in split.items():
        assert v['sgpr_spill'] <= 32, (k, v)                    # were 202-211: one base per plane
    # the non-ragged column kernels (the D % 64 == 0 instantiations: third template argument false)
    nonragged = {k: v for k, v in ks.items() if re.search(r'k_cos_partsILi\dELi\dELb0E', k)}
    assert len(nonragged) >= 6
    for k, v in nonragged.items():
        assert v['sgpr_spill'] <= 48, (k, v)                    # were 245-268


def test_pq_count_is_not_unrolled_into_its_probe_loops(usage):
    ks = {k: v for k, v in usage['metrics.hip'].items() if 'k_pq_count' in k}
    assert len(ks) == 5            # (with / without confusion matrix) x (pow2 / generic) on the map + k_pq_count_parts
    for k, v in ks.items():
        assert v['scratch'] == 0 and v['sgpr_spill'] <= 64, (k, v)   # the unrolled form: 4994-5467
