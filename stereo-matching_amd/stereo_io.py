"""Host-side helpers shared with the pipeline -- subset of the reference's stereo_io.py."""
from __future__ import annotations

from typing import Tuple

import numpy as np


def ensure_same_shape(left: np.ndarray, right: np.ndarray) -> Tuple[int, int]:
    """Reference stereo_io.py:79-95: shapes must match and be 2-D; returns (height, width)."""
    if left.shape != right.shape:
        raise ValueError("left/right image sizes differ")
    if left.ndim != 2:
        raise ValueError("gray images must be 2-D")
    return int(left.shape[0]), int(left.shape[1])
