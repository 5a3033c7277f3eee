"""CPU tier: the C-ABI library builds for gfx950, loads, and exports every
symbol that include/nmsa.h declares (no compute calls without a GPU)."""
import ctypes

import pytest
import torch

from nicr_mt_scene_analysis_amd import _lib as L


def test_library_builds_and_exports_declared_symbols():
    path = L.build()
    handle = ctypes.CDLL(path)
    declared = L.declared_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(handle, name), f'{name} declared in nmsa.h but not exported'
    # every declared symbol has a ctypes signature in the binding (and vice versa)
    assert sorted(L._SIGNATURES.keys()) == declared


def test_version_and_error_strings():
    lib = L.lib()
    assert lib.nmsa_version() >= 100
    assert lib.nmsa_strerror(0) == b'ok'
    assert b'argument' in lib.nmsa_strerror(-1)


def test_device_geometry_query_and_override(monkeypatch):
    """grids are sized from what the HIP runtime reports for the device (csrc/api.hip); without a
    device the MI355X's numbers; NMSA_ASSUME_CUS / NMSA_ASSUME_XCDS override per call — and the
    workspace of the cooperating-workgroup cosine kernel follows (fewer CUs: fewer groups)"""
    monkeypatch.delenv('NMSA_ASSUME_CUS', raising=False)
    monkeypatch.delenv('NMSA_ASSUME_XCDS', raising=False)
    cus, xcds, lds = L.device_geometry()
    assert cus >= 1 and 1 <= xcds <= cus and lds >= 64 * 1024
    if not torch.cuda.is_available():
        assert (cus, xcds, lds) == (256, 8, 160 * 1024)
    full = L.lib().nmsa_loss_cos_emb_fwd_grad_workspace_bytes(2, 768, 256, 512, 64)
    monkeypatch.setenv('NMSA_ASSUME_CUS', '64')
    monkeypatch.setenv('NMSA_ASSUME_XCDS', '2')
    assert L.device_geometry()[:2] == (64, 2)
    quarter = L.lib().nmsa_loss_cos_emb_fwd_grad_workspace_bytes(2, 768, 256, 512, 64)
    assert 0 < quarter < full
    monkeypatch.setenv('NMSA_ASSUME_XCDS', '4096')             # never more XCDs than CUs
    assert L.device_geometry()[:2] == (64, 64)


def test_cpu_tensors_are_rejected_loudly():
    from nicr_mt_scene_analysis_amd import ops
    with pytest.raises(L.NmsaError):
        ops.semantic_argmax(torch.zeros((1, 3, 4, 4)))


def test_raw_pointer_helper_refuses_pageable_host_memory():
    """`_lib.ptr` is the one place a tensor becomes a kernel argument: a pageable host tensor
    there would be a GPU memory fault (a test of round 5 did exactly that), so it raises"""
    with pytest.raises(ValueError, match='pageable'):
        L.ptr(torch.zeros(8))
    assert L.ptr(None) is None


def test_header_is_plain_c99():
    """include/nmsa.h is the FFI contract: it must compile as C (no C++ / HIP types)."""
    import os
    import shutil
    import subprocess
    gcc = shutil.which('gcc')
    if gcc is None:
        import pytest
        pytest.skip('no gcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([gcc, '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror',
                           '-fsyntax-only', '-x', 'c', os.path.join(root, 'include', 'nmsa.h')])
