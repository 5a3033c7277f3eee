"""The batch helpers the hot-path callers use, plus the on-device target generators of
SURVEY.md §8 f4 (the numpy/cv2 preprocessing pipeline itself is out of scope)."""
from .base import APPLIED_PREPROCESSING_KEY
from .base import get_applied_preprocessing_meta
from .multiscale_supervision import get_downscale
from .resize import FULLRES_SUFFIX
from .resize import get_fullres
from .resize import get_fullres_key
from .resize import get_fullres_shape
from .resize import get_valid_region_slices
from .resize import get_valid_region_slices_and_fullres_shape
from .dense_visual_embedding import DenseVisualEmbeddingTargetGenerator
from .instance import InstanceClearStuffIDs
from .instance import InstanceTargetGenerator
from .panoptic import PanopticTargetGenerator
