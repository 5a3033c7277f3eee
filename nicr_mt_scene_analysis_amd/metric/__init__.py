"""Metric accumulators of the hot path (reference metric/__init__.py).
`RootMeanSquaredError` belongs to the normals task and is out of scope."""
from .base import Metric
from .mae import MeanAbsoluteAngularError
from .mae import PanopticQualityWithOrientationMAE
from .miou import MeanIntersectionOverUnion
from .pq import PanopticQuality
