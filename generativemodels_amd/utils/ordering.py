"""Projection of a 2-D / 3-D latent grid onto a 1-D token sequence (host-side index bookkeeping; contract of the reference's
generative/utils/ordering.py:20-205): raster scan, boustrophedon ("s_curve") or random order of an index template that may first
be transposed, rotated by 90 degrees and reflected, in a configurable order."""
from __future__ import annotations

import numpy as np
import torch


class OrderingType:
    RASTER_SCAN = "raster_scan"
    S_CURVE = "s_curve"
    RANDOM = "random"
    _ALL = (RASTER_SCAN, S_CURVE, RANDOM)


class OrderingTransformations:
    ROTATE_90 = "rotate_90"
    TRANSPOSE = "transpose"
    REFLECT = "reflect"
    _ALL = (ROTATE_90, TRANSPOSE, REFLECT)


class Ordering:
    def __init__(self, ordering_type: str, spatial_dims: int, dimensions, reflected_spatial_dims=(), transpositions_axes=(),
                 rot90_axes=(), transformation_order=(OrderingTransformations.TRANSPOSE, OrderingTransformations.ROTATE_90,
                                                      OrderingTransformations.REFLECT)) -> None:
        self.ordering_type = ordering_type
        if ordering_type not in OrderingType._ALL:
            raise ValueError(f"ordering_type must be one of the following {list(OrderingType._ALL)}, but got {ordering_type}.")
        self.spatial_dims = spatial_dims
        self.dimensions = dimensions
        if len(dimensions) != spatial_dims + 1:
            raise ValueError(f"dimensions must be of length {spatial_dims + 1}, but got {len(dimensions)}.")
        self.reflected_spatial_dims = reflected_spatial_dims
        self.transpositions_axes = transpositions_axes
        self.rot90_axes = rot90_axes
        if len(set(transformation_order)) != len(transformation_order):
            raise ValueError(f"No duplicates are allowed. Received {transformation_order}.")
        for tr in transformation_order:
            if tr not in OrderingTransformations._ALL:
                raise ValueError(f"Valid transformations are {list(OrderingTransformations._ALL)} but received {tr}.")
        self.transformation_order = transformation_order
        self.template = self._transformed_template()
        self._sequence_ordering = self._walk(self.template)
        self._revert_sequence_ordering = np.argsort(self._sequence_ordering)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return x[self._sequence_ordering]

    def get_sequence_ordering(self) -> np.ndarray:
        return self._sequence_ordering

    def get_revert_sequence_ordering(self) -> np.ndarray:
        return self._revert_sequence_ordering

    def _transformed_template(self) -> np.ndarray:
        spatial = self.dimensions[1:]
        t = np.arange(int(np.prod(spatial))).reshape(*spatial)
        for tr in self.transformation_order:
            if tr == OrderingTransformations.TRANSPOSE:
                for axes in self.transpositions_axes:
                    t = np.transpose(t, axes=axes)
            elif tr == OrderingTransformations.ROTATE_90:
                for axes in self.rot90_axes:
                    t = np.rot90(t, axes=axes)
            else:
                for axis, flag in enumerate(self.reflected_spatial_dims):
                    if flag:
                        t = np.flip(t, axis=axis)
        return t

    def _walk(self, t: np.ndarray) -> np.ndarray:
        snake = self.ordering_type == OrderingType.S_CURVE
        coords = []
        for r in range(t.shape[0]):
            cols = range(t.shape[1] - 1, -1, -1) if (snake and r % 2) else range(t.shape[1])
            for c in cols:
                if self.spatial_dims == 3:
                    deps = range(t.shape[2] - 1, -1, -1) if (snake and c % 2) else range(t.shape[2])
                    coords.extend((r, c, d) for d in deps)
                else:
                    coords.append((r, c))
        idx = np.array(coords)
        if self.ordering_type == OrderingType.RANDOM:
            np.random.shuffle(idx)  # the reference shuffles with numpy's global generator (ordering.py:193-205)
        return np.array([t[tuple(e)] for e in idx])
