"""`DensePostprocessingBase` (reference model/postprocessing/dense_base.py:14-58).

Crop to the valid region and resize to the dataset resolution — the
"full-resolution step either side of the merge" (SURVEY.md §8 f2).  The crop is
folded into the HIP resize kernels (csrc/resize.hip: `nmsa_resize_nearest`,
`nmsa_resize_bilinear`), whose index / weight arithmetic reproduces the ATen
CPU kernels the reference's evaluation path runs, including the float32 round
trip the reference takes for integer maps.
"""
from typing import Tuple

import torch

from ... import ops
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
        cropped = prediction[..., sl_h, sl_w]
        h, w = shape
        if tuple(cropped.shape[-2:]) == (h, w):
            return cropped              # nothing to resize (dense_base.py:28-31)

        if mode == 'nearest':
            return ops.resize_nearest(prediction, (h, w), valid_region_slices)
        if mode == 'bilinear':
            return ops.resize_bilinear(prediction, (h, w), valid_region_slices)
        raise NotImplementedError(
            f"resize mode '{mode}' (the reference only uses 'nearest' and 'bilinear')")
