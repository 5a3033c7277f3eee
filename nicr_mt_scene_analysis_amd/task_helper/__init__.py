"""Task helpers on the hot path (reference task_helper/__init__.py).
`SceneTaskHelper` / `NormalTaskHelper` belong to other tasks and are out of scope."""
from .base import TaskHelperBase
from .base import get_total_loss_key
from .dense_visual_embedding import DenseVisualEmbeddingTaskHelper
from .instance import InstanceTaskHelper
from .panoptic import PanopticTaskHelper
from .semantic import SemanticTaskHelper
