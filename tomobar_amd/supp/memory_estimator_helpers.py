"""Device-memory bookkeeping for "dry runs" of the direct methods (reference: tomobar/supp/memory_estimator_helpers.py).

Inside ``with DeviceMemStack() as stack:`` a method called with a SHAPE instead of an array records the device
allocations it would make (rounded up to 512-byte units) and returns its output shape; ``stack.highwater`` is the peak."""

ALLOCATION_UNIT_SIZE = 512


class DeviceMemStack:
    _instance = None
    _depth = 0

    def __init__(self) -> None:
        self.allocations = []
        self.current = 0
        self.highwater = 0

    def __enter__(self):
        if DeviceMemStack._depth == 0:
            DeviceMemStack._instance = self
        DeviceMemStack._depth += 1
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        DeviceMemStack._depth -= 1
        if DeviceMemStack._depth == 0:
            DeviceMemStack._instance = None

    @classmethod
    def instance(cls):
        return cls._instance

    @staticmethod
    def _round_up(size: int) -> int:
        return (int(size) + ALLOCATION_UNIT_SIZE - 1) // ALLOCATION_UNIT_SIZE * ALLOCATION_UNIT_SIZE

    def malloc(self, byte_count) -> None:
        self.allocations.append(int(byte_count))
        self.current += self._round_up(byte_count)
        self.highwater = max(self.highwater, self.current)

    def free(self, byte_count) -> None:
        self.allocations.remove(int(byte_count))
        self.current -= self._round_up(byte_count)
