"""Batch-level, on-device twin of `DenseVisualEmbeddingTargetGenerator`
(reference data/preprocessing/dense_visual_embedding.py:14-93).

Batch format: `panoptic` i64 [B,H,W]; `panoptic_embedding_keys` i64 [B,K] with
`panoptic_embedding_n` i32 [B] valid entries per image, `panoptic_embedding` f32 [B,K,D]
(the reference's per-sample dict, padded), `image_embedding` f32 [B,D]."""
from typing import Any, Dict

from ... import ops


class DenseVisualEmbeddingTargetGenerator:
    def __init__(self, diff_factor: float = 0.65, **kwargs) -> None:
        self.diff_factor = diff_factor

    def __call__(self, batch: Dict[str, Any]) -> Dict[str, Any]:
        if 'image_embedding' not in batch or 'panoptic_embedding' not in batch:
            return batch                                      # inference call
        r = ops.dve_targets(batch['panoptic'], batch['panoptic_embedding_keys'],
                            batch['panoptic_embedding_n'], batch['panoptic_embedding'],
                            batch['image_embedding'], self.diff_factor)
        batch['dense_visual_embedding_lut'] = r['lut']
        batch['dense_visual_embedding_indices'] = r['indices']
        return batch
