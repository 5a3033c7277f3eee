"""Losses of the hot path (reference loss/__init__.py)."""
from .base import LossBase
from .ce import CrossEntropyLossSemantic
from .cos_emb import CosineEmbeddingLoss
from .focal import CenterFocalLoss
from .l1 import L1Loss
from .mse import MSELoss
from .vonmises import VonMisesLossBiternion
from ._functional import check_loss_status
from ._functional import reset_speculation_state
from ._functional import speculation_stats
