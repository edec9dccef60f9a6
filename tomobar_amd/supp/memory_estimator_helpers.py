"""Dry-run accounting of device memory for the direct methods.

Interface of the reference's ``tomobar/supp/memory_estimator_helpers.py`` (``DeviceMemStack`` with ``malloc`` / ``free`` /
``highwater`` / ``instance()`` and context-manager activation) so that callers written against it keep working:

    with DeviceMemStack() as stack:
        out_shape = rectools.FOURIER_INV((nz, nproj, ndet), data_dtype=np.float32)
    peak_bytes = stack.highwater

While a stack is active, a method that is handed a SHAPE instead of an array books the device buffers it would create and
returns the shape of its result.  Sizes are booked in whole 512-byte granules, the granularity of the device allocators.
"""
from __future__ import annotations

from collections import Counter
from typing import ClassVar, List, Optional

ALLOCATION_UNIT_SIZE = 512


def _granules(nbytes: int) -> int:
    return -(-int(nbytes) // ALLOCATION_UNIT_SIZE) * ALLOCATION_UNIT_SIZE


class DeviceMemStack:
    """Live-bytes counter with a high-water mark.  The outermost active ``with`` block is the one methods report to."""

    _active: ClassVar[List["DeviceMemStack"]] = []

    def __init__(self) -> None:
        self._live: Counter = Counter()   # requested size -> number of live blocks of that size
        self.current = 0                  # booked bytes (granule-rounded)
        self.highwater = 0                # maximum of ``current`` so far

    # ---- activation -------------------------------------------------------------------------------
    def __enter__(self) -> "DeviceMemStack":
        DeviceMemStack._active.append(self)
        return self

    def __exit__(self, exc_type, exc_value, traceback) -> None:
        DeviceMemStack._active.pop()

    @classmethod
    def instance(cls) -> Optional["DeviceMemStack"]:
        return cls._active[0] if cls._active else None

    # ---- bookkeeping ------------------------------------------------------------------------------
    @property
    def allocations(self) -> List[int]:
        return sorted(self._live.elements())

    def malloc(self, byte_count) -> None:
        size = int(byte_count)
        self._live[size] += 1
        self.current += _granules(size)
        if self.current > self.highwater:
            self.highwater = self.current

    def free(self, byte_count) -> None:
        size = int(byte_count)
        if self._live[size] <= 0:
            raise ValueError(f"free({size}) without a matching malloc")
        self._live[size] -= 1
        if not self._live[size]:
            del self._live[size]
        self.current -= _granules(size)
