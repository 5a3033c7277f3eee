"""Postprocessing factory (reference model/postprocessing/__init__.py:24-44).

On the hot path: 'semantic', 'instance', 'panoptic'.  'normal', 'scene' and
'dense-visual-embedding' are other tasks' postprocessors (resize / softmax / a
conv2d GEMM) and out of scope of this package."""
from typing import Any

from ...utils import partial_class
from .base import PostprocessingBase
from .dense_base import DensePostprocessingBase
from .instance import InstancePostprocessing
from .panoptic import PanopticPostprocessing
from .semantic import SemanticPostprocessing

_OUT_OF_SCOPE = ('normal', 'scene', 'dense-visual-embedding')


def get_postprocessing_class(name: str, **kwargs: Any):
    if name == 'semantic':
        cls = SemanticPostprocessing
    elif name == 'instance':
        cls = InstancePostprocessing
    elif name == 'panoptic':
        cls = PanopticPostprocessing
    elif name in _OUT_OF_SCOPE:
        raise NotImplementedError(
            f"postprocessing '{name}' is not part of the MI355X hot path; use the "
            "reference implementation for it")
    else:
        raise ValueError(f"Unknown postprocessing: '{name}'")
    return partial_class(cls, **kwargs)
