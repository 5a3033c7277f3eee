"""Type aliases used in the signatures of the hot-path API
(names as in the reference's types.py:33-47)."""
from typing import Any, Dict, Tuple, Union

from torch import Tensor

BatchType = Dict[str, Any]
DecoderRawOutputType = Tuple[Any, Any]          # (outputs, side_outputs)
DecoderPostprocessedOutputType = Dict[str, Any]
PostprocessingOutputType = DecoderPostprocessedOutputType
TensorOrTuple = Union[Tensor, Tuple[Tensor, ...]]
