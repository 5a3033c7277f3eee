"""`MSELoss` (reference loss/mse.py:13-41)."""
from ._elementwise import _ElementwiseLoss


class MSELoss(_ElementwiseLoss):
    _kind = 'mse'
