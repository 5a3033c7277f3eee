"""Biternion helpers used by the hot path (reference utils/_orientation.py:39-47)."""
import torch


def biternion2rad(biternion: torch.Tensor) -> torch.Tensor:
    # channel 0 = cos, channel 1 = sin
    return torch.atan2(biternion[:, 1], biternion[:, 0])


def biternion2deg(biternion: torch.Tensor) -> torch.Tensor:
    return torch.rad2deg(biternion2rad(biternion)) % 360
