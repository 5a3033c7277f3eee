"""Full-resolution bookkeeping helpers (reference data/preprocessing/resize.py:19-78).
The cv2 `Resize` transform itself is CPU dataloader work and out of scope; the
postprocessing only needs to know the valid region and the original shape."""
from typing import Any, Dict, Tuple

from .base import get_applied_preprocessing_meta

FULLRES_SUFFIX = '_fullres'
_RESIZE_TYPE_NAME = 'Resize'


def get_fullres_key(key: str) -> str:
    return key + FULLRES_SUFFIX


def get_fullres(sample: Dict[str, Any], key: str) -> Any:
    return sample.get(get_fullres_key(key), None)


def get_fullres_shape(sample: Dict[str, Any], key: str) -> Tuple[int, int]:
    # the key itself first, then the input modalities (at least one is present)
    for k in (key, 'rgb', 'depth'):
        img = get_fullres(sample, k)
        if img is not None:
            return tuple(img.shape[-2:])
    raise ValueError(f"Unable to get fullres shape for `{key}`.")


def get_valid_region_slices(sample: Dict[str, Any]) -> Tuple[slice, slice]:
    meta = get_applied_preprocessing_meta(sample)
    # all samples of a batch share the original resolution: first element
    if len(meta):
        for entry in meta[0]:
            if entry['type'] == _RESIZE_TYPE_NAME:
                return entry['valid_region_slice_y'], entry['valid_region_slice_x']
    raise ValueError("Unable to get get valid region slices.")


def get_valid_region_slices_and_fullres_shape(
    sample: Dict[str, Any], key: str
) -> Tuple[Tuple[slice, slice], Tuple[int, int]]:
    return get_valid_region_slices(sample), get_fullres_shape(sample, key)
