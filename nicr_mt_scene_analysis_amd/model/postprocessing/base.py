"""`PostprocessingBase` — the interface decoders call
(reference model/postprocessing/base.py:13-41)."""
import abc

from ...types import BatchType
from ...types import DecoderRawOutputType
from ...types import PostprocessingOutputType


class PostprocessingBase(abc.ABC):
    def postprocess(
        self,
        data: DecoderRawOutputType,
        batch: BatchType,
        is_training: bool = True
    ) -> PostprocessingOutputType:
        fn = self._postprocess_training if is_training else self._postprocess_inference
        return fn(data, batch)

    @abc.abstractmethod
    def _postprocess_training(
        self, data: DecoderRawOutputType, batch: BatchType
    ) -> PostprocessingOutputType:
        ...

    def _postprocess_inference(
        self, data: DecoderRawOutputType, batch: BatchType
    ) -> PostprocessingOutputType:
        # default: inference == training
        return self._postprocess_training(data, batch)
