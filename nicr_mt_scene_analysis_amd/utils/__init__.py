from ._misc import partial_class
from ._orientation import biternion2deg
from ._orientation import biternion2rad
from . import panoptic_merge
