"""
ctypes binding of the C ABI in include/nmsa.h (library csrc/libnmsa_hip.so).

The product path has NO fallback: if the library is missing or a call fails,
an exception is raised.  `import torch` must happen before the library is
loaded so that both share one HIP runtime (libamdhip64.so.7 is resolved by
soname against the copy torch already mapped).
"""
import ctypes as C
import os
import re
import subprocess
from typing import Optional

import torch   # noqa: F401  (must be loaded first: shares libamdhip64 with us)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC_DIR = os.path.join(_HERE, 'csrc')
# NMSA_LIB_PATH: another build of the library (same-box A/B measurements against an older build)
LIB_PATH = os.environ.get('NMSA_LIB_PATH') or os.path.join(CSRC_DIR, 'libnmsa_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'nmsa.h')

NMSA_F32, NMSA_BF16, NMSA_F16 = 0, 1, 2
NMSA_U8, NMSA_I16, NMSA_I32, NMSA_I64 = 0, 1, 2, 3


class NmsaError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ['make', '-C', CSRC_DIR, '-j4']
    if force:
        args.append('-B')
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


def declared_symbols():
    """Entry points declared in include/nmsa.h."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nmsa_[a-z0-9_]+)\s*\(', text)))


_vp, _i, _f, _i64, _sz = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t

_SIGNATURES = {
    'nmsa_version': (C.c_int, []),
    'nmsa_strerror': (C.c_char_p, [_i]),
    'nmsa_last_hip_error': (C.c_int, []),
    'nmsa_device_geometry': (_i, [_vp, _vp, _vp]),
    'nmsa_center_nms_workspace_bytes': (_sz, [_i, _i, _i]),
    'nmsa_center_nms_topk': (_i, [_vp, _vp, _i, _i, _i, _f, _i, _i, _i, _i,
                                  _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_group_offsets': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _i, _f,
                                _vp, _vp, _vp]),
    'nmsa_semantic_argmax': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'nmsa_semantic_softmax': (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    'nmsa_panoptic_fused': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                 _f, _f, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'nmsa_panoptic_assign': (_i, [_vp, _i, _i, _i, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    'nmsa_panoptic_paint': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64,
                                 _vp, _vp, _vp]),
    'nmsa_panoptic_merge': (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i64, _i64,
                                 _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'nmsa_panoptic_merge_wide_workspace_bytes': (_sz, [_i, _i, _i]),
    'nmsa_panoptic_merge_wide': (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_resize_nearest': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'nmsa_resize_bilinear': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    'nmsa_semantic_argmax_resized': (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                          _vp, _vp, _vp, _vp]),
    'nmsa_panoptic_scores_workspace_bytes': (_sz, [_i]),
    'nmsa_panoptic_scores': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64,
                                  _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_targets_workspace_bytes': (_sz, [_i, _i, _i]),
    'nmsa_instance_clear_stuff': (_i, [_vp, _i, _vp, _i, _vp, _i, _i64, _vp]),
    'nmsa_instance_targets': (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    'nmsa_panoptic_targets': (_i, [_vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i64, _i64, _i, _i,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _vp]),
    'nmsa_dve_targets': (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'nmsa_instance_orientation_wide': (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                                            _vp, _vp, _sz, _vp]),
    'nmsa_instance_orientation': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    'nmsa_confmat_workspace_bytes': (_sz, [_i]),
    'nmsa_confmat_update': (_i, [_vp, _i, _i64, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_pq_confmat_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'nmsa_pq_update_with_confmat': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64,
                                         _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _i,
                                         _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_pq_update_with_confmat_parts': (_i, [_vp, _vp, _vp, _vp, _i, _i64, _vp, _vp, _i, _i, _i,
                                               _i, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _vp,
                                               _vp, _vp, _sz, _i, _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_pack_tables': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    'nmsa_loss_workspace_bytes': (_sz, [_i, _i, _i]),
    'nmsa_loss_ce_fwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                              _vp]),
    'nmsa_loss_ce_bwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    'nmsa_loss_ce_fwd_i16': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                  _vp]),
    'nmsa_loss_ce_bwd_i16': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    'nmsa_loss_masked_fwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_loss_masked_bwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'nmsa_loss_vonmises_fwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_loss_vonmises_bwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp]),
    'nmsa_loss_ce_fwd_grad_supported': (_i, [_i, _i]),
    'nmsa_loss_ce_fwd_grad': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _sz, _vp]),
    'nmsa_loss_ce_bwd_unless': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    'nmsa_count_workspace_bytes': (_sz, []),
    'nmsa_count_u8': (_i, [_vp, _i64, _i, _i, _vp, _vp, _f, _vp, _sz, _vp]),
    'nmsa_loss_cos_emb_fwd_grad_supported': (_i, [_i, _i, _i, _i, _i]),
    'nmsa_loss_cos_emb_fwd_grad_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'nmsa_loss_cos_emb_fwd_grad': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_loss_cos_emb_bwd_unless': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_multitask_loss_workspace_bytes': (_sz, [_vp, _i]),
    'nmsa_multitask_loss_fwd_grad': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_multitask_loss_bwd_unless': (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_loss_masked_fwd_grad': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp,
                                       _sz, _vp]),
    'nmsa_loss_masked_bwd_unless': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp,
                                         _vp]),
    'nmsa_loss_vonmises_fwd_grad': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp,
                                         _sz, _vp]),
    'nmsa_loss_vonmises_bwd_unless': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp]),
    'nmsa_loss_cos_emb_fwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'nmsa_loss_cos_emb_bwd': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    'nmsa_loss_cos_emb_can_keep_dots': (_i, [_i, _i, _i, _i]),
    'nmsa_loss_elementwise_none_fwd': (_i, [_vp, _i, _vp, _i64, _i, _i, _vp, _vp]),
    'nmsa_loss_elementwise_none_bwd': (_i, [_vp, _i, _vp, _i64, _i, _i, _vp, _vp, _vp]),
    'nmsa_loss_cos_rows_fwd': (_i, [_vp, _i, _vp, _vp, _i64, _i, _f, _vp, _vp]),
    'nmsa_loss_cos_rows_bwd': (_i, [_vp, _i, _vp, _vp, _i64, _i, _f, _vp, _i, _vp, _vp]),
    'nmsa_loss_vonmises_rows_fwd': (_i, [_vp, _i, _vp, _i64, _f, _vp, _vp]),
    'nmsa_loss_vonmises_rows_bwd': (_i, [_vp, _i, _vp, _i64, _f, _vp, _vp, _vp]),
    'nmsa_pq_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'nmsa_pq_update': (_i, [_vp, _vp, _i, _i, _i, _i, _i64, _i64, _i64, _i64,
                            _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _i, _vp]),
}

_lib = None


def lib():
    """The loaded library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NmsaError(
                f'{LIB_PATH} is missing: the HIP extension was not built. Run '
                '`python -c "import __graft_entry__ as g; g.build()"` '
                '(or `make -C nicr_mt_scene_analysis_amd/csrc`). There is no '
                'CPU/PyTorch fallback for this path.')
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        l_ = lib()
        msg = l_.nmsa_strerror(rc).decode()
        hip = l_.nmsa_last_hip_error()
        raise NmsaError(f'{what}: {msg} (code {rc}, hipError {hip})')


def device_geometry():
    """(compute units, XCDs, LDS bytes per CU) the library sizes its grids from on the current device"""
    cus, xcds, lds = C.c_int(0), C.c_int(0), C.c_size_t(0)
    check(lib().nmsa_device_geometry(C.byref(cus), C.byref(xcds), C.byref(lds)), 'nmsa_device_geometry')
    return cus.value, xcds.value, lds.value


def ptr(t: Optional[torch.Tensor]):
    """Address of a tensor the kernels may touch: device memory, or pinned host memory (mapped
    into the GPU's address space).  A pageable host tensor raises here — handed to a kernel it
    would be a GPU memory fault, not a Python error."""
    if t is None:
        return None
    if not t.is_cuda and not t.is_pinned():
        raise ValueError(f'the C-ABI takes device (or pinned host) memory; got a pageable '
                         f'{t.device} tensor of shape {tuple(t.shape)}')
    return C.c_void_p(t.data_ptr())


def stream_ptr(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def float_dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return NMSA_F32
    if t.dtype == torch.bfloat16:
        return NMSA_BF16
    if t.dtype == torch.float16:
        return NMSA_F16
    raise TypeError(f'unsupported floating dtype {t.dtype}')


def int_dtype_code(t: torch.Tensor) -> int:
    if t.dtype in (torch.uint8, torch.bool):
        return NMSA_U8
    if t.dtype == torch.int16:
        return NMSA_I16
    if t.dtype == torch.int32:
        return NMSA_I32
    if t.dtype == torch.int64:
        return NMSA_I64
    raise TypeError(f'unsupported integer dtype {t.dtype}')


def require_device_tensor(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise NmsaError(
            f'{name} is on {t.device}: the HIP path needs tensors on the MI355X '
            '(there is no CPU fallback).')
    return t.contiguous()
