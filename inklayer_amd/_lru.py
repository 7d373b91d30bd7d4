"""Bounded per-input-size caches (plans, HIP graphs): a directory of sketches with many aspect ratios must not grow
HBM without bound."""
from collections import OrderedDict
from typing import Callable, Hashable, TypeVar

V = TypeVar("V")


class LRU:
    def __init__(self, capacity: int):
        assert capacity >= 1
        self.capacity = capacity
        self._d: "OrderedDict[Hashable, object]" = OrderedDict()

    def __len__(self) -> int:
        return len(self._d)

    def __contains__(self, key) -> bool:
        return key in self._d

    def get(self, key, default=None):
        if key in self._d:
            self._d.move_to_end(key)
            return self._d[key]
        return default

    def put(self, key, value) -> None:
        self._d[key] = value
        self._d.move_to_end(key)
        while len(self._d) > self.capacity:
            self._d.popitem(last=False)          # dropping the last reference frees the entry's device buffers / graph pool

    def get_or_make(self, key, make: Callable[[], V]) -> V:
        v = self.get(key)
        if v is None:
            v = make()
            self.put(key, v)
        return v

    def clear(self) -> None:
        self._d.clear()
