"""Stage-1 -> stage-2 contract; same field names as the reference (splat/schema.py:13-25)."""
from typing import NamedTuple

import torch


class PreprocessedScene(NamedTuple):
    points: torch.Tensor                 # (Nv,2) pixel means (== points_xy)
    colors: torch.Tensor                 # (Nv,3)
    covariance_2d: torch.Tensor          # (Nv,2,2)
    depths: torch.Tensor                 # (Nv,)
    inverse_covariance_2d: torch.Tensor  # (Nv,2,2)
    radius: torch.Tensor                 # (Nv,)
    points_xy: torch.Tensor              # (Nv,2)
    min_x: torch.Tensor
    min_y: torch.Tensor
    max_x: torch.Tensor
    max_y: torch.Tensor
    sigmoid_opacity: torch.Tensor        # (Nv,1)
