"""Axis-label handling and small helpers (reference: ``tomobar/supp/funcs.py:84-206``)."""

from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from ..projector import vec_geom_init3D as _vec_geom_init3D  # noqa: F401  (supp/funcs.py:45-65)

_VALID = ("angles", "detX", "detY")


def _first_mismatch(current: List[str], required: List[str]) -> Optional[Tuple[int, int]]:
    """(position in `required`, position in `current`) of the first label that sits at the wrong place."""
    for want_pos, label in enumerate(required):
        have_pos = current.index(label)
        if have_pos != want_pos:
            return (want_pos, have_pos)
    return None


def _swap_data_axes_to_accepted(data_axes_labels: list, required_labels_order: list) -> list:
    """At most two axis swaps that bring ``data_axes_labels`` into ``required_labels_order``; entries are
    ``None`` when no swap is needed (same tuples as the reference, cf. its tests/test_tools.py:36-68)."""
    if len(data_axes_labels) != len(required_labels_order):
        raise ValueError("Warning: The mismatch in length between provided labels and data dimensions.")
    for label in data_axes_labels:
        if label not in required_labels_order:
            raise ValueError(
                f'Axis title "{label}" is not valid, please use one of these: "angles", "detX", or "detY"')
    labels = list(data_axes_labels)
    swaps = []
    for _ in range(2):
        swap = _first_mismatch(labels, required_labels_order) if (not swaps or swaps[-1] is not None) else None
        if swap is not None:
            i, j = swap
            labels[i], labels[j] = labels[j], labels[i]
        swaps.append(swap)
    return swaps


def swap_tuple_elements(tup: tuple, idx1: int, idx2: int) -> tuple:
    items = list(tup)
    items[idx1], items[idx2] = items[idx2], items[idx1]
    return tuple(items)


def _data_swap(data, data_swap_list: list):
    """Apply the swaps to an array (as views) or to a shape tuple."""
    for swap in data_swap_list:
        if swap is None:
            continue
        if isinstance(data, tuple):
            data = swap_tuple_elements(data, swap[0], swap[1])
        elif isinstance(data, np.ndarray):
            data = np.swapaxes(data, swap[0], swap[1])
        else:
            data = data.transpose(swap[0], swap[1])  # torch view
    return data


def _data_dims_swapper(data, data_axes_labels_order: list, required_labels_order: list):
    return _data_swap(data, _swap_data_axes_to_accepted(data_axes_labels_order, required_labels_order))


def _parse_device_argument(device_int_or_string) -> Tuple[str, int]:
    """'cpu' / 'gpu' / GPU index -> (architecture, index)   (supp/funcs.py:174-187)."""
    if isinstance(device_int_or_string, int):
        return "gpu", device_int_or_string
    if device_int_or_string == "gpu":
        return "gpu", 0
    if device_int_or_string == "cpu":
        return "cpu", -1
    raise ValueError('Unknown device {0}. Expecting either "cpu" or "gpu" strings OR the gpu device integer'.format(
        device_int_or_string))
