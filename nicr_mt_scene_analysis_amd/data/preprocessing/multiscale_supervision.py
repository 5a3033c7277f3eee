"""`get_downscale` (reference data/preprocessing/multiscale_supervision.py):
side-output targets live in batch['_down_<k>']."""
from typing import Any, Dict, Optional

from .base import MULTI_DOWNSCALE_KEY_FMT


def get_downscale(sample: Dict[str, Any], downscale: int) -> Optional[Dict[str, Any]]:
    return sample.get(MULTI_DOWNSCALE_KEY_FMT.format(downscale), None)
