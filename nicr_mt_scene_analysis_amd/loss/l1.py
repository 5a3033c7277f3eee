"""`L1Loss` (reference loss/l1.py:13-41)."""
from ._elementwise import _ElementwiseLoss


class L1Loss(_ElementwiseLoss):
    _kind = 'l1'
