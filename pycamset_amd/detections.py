"""Input format of the path: the detection table (pyCamSet calibration_targets/target_detections.py).

Only what the cost/Jacobian path consumes is mirrored: the ``(N, 4+k)`` float64 table
``| cam | im_num | key ... | u | v |`` (td:51-55), ``get_data()``, ``max_ims`` (td:74-78) and
``return_flattened_keys`` (td:333-351).  Detection of targets in images is out of scope.
"""
from __future__ import annotations

from copy import copy

import numpy as np


class TargetDetection:
    def __init__(self, cam_names: list, data: np.ndarray | None = None, max_ims: int = 0):
        if len(set(cam_names)) != len(cam_names):
            raise ValueError("input camera names must be unique")  # td:63-64
        self.cam_names = list(cam_names)
        self._data = None if data is None else np.array(copy(data), dtype=np.float64)
        self._max_ims = max_ims

    def get_data(self) -> np.ndarray | None:
        return self._data

    @property
    def max_ims(self) -> int:  # td:74-78
        temp = int(np.max(self._data[:, 1])) + 1 if self._data is not None and self._data.shape[0] else 0   # an empty shard keeps its stated max_ims
        self._max_ims = max(temp, self._max_ims)
        return self._max_ims

    @max_ims.setter
    def max_ims(self, val: int):
        self._max_ims = val

    def return_flattened_keys(self, keydims) -> "TargetDetection":
        """Row-major flattening of multi-dimensional keys, matching ``point_data.reshape(-1, 3)``
        (td:333-351)."""
        if self._data.shape[1] == 5:  # already flat (td:342-343)
            return self
        # row-major strides of the key grid: the last key dimension is contiguous, as in point_data.reshape(-1, 3)
        dims = [int(d) for d in keydims]
        strides = np.ones(len(dims))
        for axis in range(len(dims) - 2, -1, -1):
            strides[axis] = strides[axis + 1] * dims[axis + 1]
        table = self._data
        flat = np.empty((table.shape[0], 5))
        flat[:, :2] = table[:, :2]
        flat[:, 2] = table[:, 2:-2] @ strides
        flat[:, 3:] = table[:, -2:]
        return TargetDetection(self.cam_names, flat, self.max_ims)

    def delete_row(self, im_num=None) -> "TargetDetection":
        """Drop all rows of the given image numbers (used by get_detection_data, th:399-403)."""
        mask = ~np.isin(self._data[:, 1], np.atleast_1d(im_num))
        return TargetDetection(self.cam_names, self._data[mask], self._max_ims)
