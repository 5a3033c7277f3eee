"""Synthetic, seed-deterministic inputs for the hot path.

Counterpart of the reference's `testing/` helpers (which need datasets that are
not available offline).  Used by tests/, bench.py and oracle/gen_golden.py.
"""
from .synthetic import make_panoptic_inputs
from .synthetic import make_metric_inputs
from .synthetic import make_loss_inputs
from .synthetic import input_digest
from .synthetic import make_panoptic_inputs_torch
