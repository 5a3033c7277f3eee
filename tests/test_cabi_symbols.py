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


def test_cpu_tensors_are_rejected_loudly():
    from nicr_mt_scene_analysis_amd import ops
    with pytest.raises(L.NmsaError):
        ops.semantic_argmax(torch.zeros((1, 3, 4, 4)))


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
