"""`SemanticPostprocessing` on the MI355X
(reference model/postprocessing/semantic.py:17-82).

argmax + score come from ONE pass of `nmsa_semantic_argmax` over the logits
(the reference materialises the full softmax first); the softmax map itself is
a lazy entry produced by `nmsa_semantic_softmax` when somebody reads it.
"""
import torch

from ... import ops
from ...data.preprocessing.resize import get_fullres_key
from ...data.preprocessing.resize import get_valid_region_slices_and_fullres_shape
from ...types import BatchType
from ...types import DecoderRawOutputType
from ...types import PostprocessingOutputType
from ._lazy import LazyDict
from .dense_base import DensePostprocessingBase


class SemanticPostprocessing(DensePostprocessingBase):
    def __init__(self, **kwargs):
        super().__init__()

    def _postprocess_training(
        self, data: DecoderRawOutputType, batch: BatchType
    ) -> PostprocessingOutputType:
        output, side_outputs = data
        return {'semantic_output': output, 'semantic_side_outputs': side_outputs}

    @staticmethod
    def _argmax_entries(r: LazyDict, logits: torch.Tensor, suffix: str = '') -> None:
        """idx now; the softmax tensor and the score of the winning class when they are read
        (semantic.py:52-59 / :71-80): the argmax alone needs no exponentials, and validation
        loops only consume the class map."""
        narrow = logits.shape[1] <= 256
        am = ops.semantic_argmax(logits, want_u8=narrow, want_i64=not narrow, want_score=False)
        r.set_lazy('semantic_softmax_scores' + suffix, lambda: ops.semantic_softmax(logits))
        r.set_lazy('semantic_segmentation_score' + suffix,
                   lambda: ops.semantic_argmax(logits, want_u8=False, want_i64=False,
                                               want_score=True)['score'])
        if narrow:      # uint8 class map for the metrics, the reference's int64 map when read
            idx_u8 = am['idx_u8']
            r.aux['semantic_segmentation_idx' + suffix] = idx_u8
            r.set_lazy('semantic_segmentation_idx' + suffix, lambda: idx_u8.long())
        else:
            r['semantic_segmentation_idx' + suffix] = am['idx']

    def _fullres_entries(self, r: LazyDict, output: torch.Tensor, batch: BatchType) -> None:
        """semantic.py:61-80.  With a real resize, idx / score come from ONE fused pass over the
        network-resolution logits (`nmsa_semantic_argmax_resized`); the full-resolution logits
        and their softmax are produced only when somebody reads them."""
        crop, shape = get_valid_region_slices_and_fullres_shape(batch, 'semantic')
        cropped = output[..., crop[0], crop[1]]
        k_out, k_sm, k_score, k_idx = (get_fullres_key(k) for k in (
            'semantic_output', 'semantic_softmax_scores', 'semantic_segmentation_score',
            'semantic_segmentation_idx'))
        if tuple(cropped.shape[-2:]) != tuple(shape):
            r.set_lazy(k_out, lambda: ops.resize_bilinear(output, shape, crop))
            r.set_derived(k_sm, lambda d: ops.semantic_softmax(d[k_out]))
            narrow = output.shape[1] <= 256
            am = ops.semantic_argmax_resized(output, shape, crop, want_u8=narrow,
                                             want_i64=not narrow, want_score=False)
            r.set_lazy(k_score, lambda: ops.semantic_argmax_resized(
                output, shape, crop, want_u8=False, want_i64=False, want_score=True)['score'])
            if narrow:
                idx_u8 = am['idx_u8']
                r.aux[k_idx] = idx_u8
                r.set_lazy(k_idx, lambda: idx_u8.long())
            else:
                r[k_idx] = am['idx']
            return
        r[k_out] = cropped
        if cropped.shape == output.shape:
            # nothing was cropped or resized: the fullres entries are the same
            # functions of the same logits -> share them (bit-identical)
            for k in ('semantic_softmax_scores', 'semantic_segmentation_score',
                      'semantic_segmentation_idx'):
                if r.is_pending(k):
                    r.set_derived(get_fullres_key(k), (lambda kk: (lambda d: d[kk]))(k))
                else:
                    r[get_fullres_key(k)] = r[k]
                if k in r.aux:
                    r.aux[get_fullres_key(k)] = r.aux[k]
        else:
            self._argmax_entries(r, cropped.contiguous(), suffix='_fullres')

    def _postprocess_inference(
        self, data: DecoderRawOutputType, batch: BatchType
    ) -> PostprocessingOutputType:
        output, side_outputs = data
        r = LazyDict(semantic_output=output, semantic_side_outputs=side_outputs)
        self._argmax_entries(r, output)
        self._fullres_entries(r, output, batch)
        return r
