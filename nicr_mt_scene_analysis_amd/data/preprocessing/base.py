"""Batch-meta keys written by the reference's preprocessing pipeline
(reference data/preprocessing/base.py:17-30); the hot path only READS them."""
from typing import Any, Dict, List

MULTI_DOWNSCALE_KEY_FMT = '_down_{}'
APPLIED_PREPROCESSING_KEY = '_applied_preprocessing'


def get_applied_preprocessing_meta(sample: Dict[str, Any]) -> List[Any]:
    return sample.setdefault(APPLIED_PREPROCESSING_KEY, [])
