"""`DensePostprocessingBase` (reference model/postprocessing/dense_base.py:14-58).

Crop to the valid region and resize to the dataset resolution.  This is the
"full-resolution step either side of the merge" that SURVEY.md §8(f) ranks as
next after the hot path: it stays a torch (ATen, on-device) call here —
integer maps take the same float32 round-trip as the reference so that the
nearest-neighbour source pixel is chosen identically.
"""
from typing import Tuple

import torch
import torch.nn.functional as F

from .base import PostprocessingBase


class DensePostprocessingBase(PostprocessingBase):
    def _crop_to_valid_region_and_resize_prediction(
        self,
        prediction: torch.Tensor,
        valid_region_slices: Tuple[slice, slice],
        shape: Tuple[int, int],     # (h, w)
        mode: str = 'nearest'
    ) -> torch.Tensor:
        sl_h, sl_w = valid_region_slices
        out = prediction[..., sl_h, sl_w]
        h, w = shape
        if tuple(out.shape[-2:]) == (h, w):
            return out

        squeeze = out.ndim == 3
        if squeeze:
            out = out.unsqueeze(1)          # interpolate wants BCHW
        orig_dtype = out.dtype
        if not out.is_floating_point():
            out = out.to(torch.float32)
        extra = {} if mode == 'nearest' else {'align_corners': False}
        out = F.interpolate(out, size=(h, w), mode=mode, **extra).to(orig_dtype)
        if squeeze:
            out = out.squeeze(1)
        return out
