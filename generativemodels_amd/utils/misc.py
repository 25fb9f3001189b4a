"""Shape helpers (reference generative/utils/misc.py:19-26)."""
from __future__ import annotations


def unsqueeze_right(arr, ndim: int):
    """Append size-1 dims until `arr` has `ndim` dims."""
    return arr[(...,) + (None,) * (ndim - arr.ndim)]


def unsqueeze_left(arr, ndim: int):
    """Prepend size-1 dims until `arr` has `ndim` dims."""
    return arr[(None,) * (ndim - arr.ndim)]
