"""A dict whose expensive entries are computed on first access.

The reference materialises every derived tensor eagerly (e.g. the full
B x C x H x W softmax, 160 B/px of extra HBM writes at C=40).  Consumers of the
postprocessing dict usually read a handful of keys, so entries that are pure
functions of other entries are registered as thunks here: the key is listed by
keys()/in/len() from the start and the value is produced (by a HIP kernel) the
first time it is read.
"""
from typing import Any, Callable, Dict


class _Derived:
    __slots__ = ('fn',)

    def __init__(self, fn):
        self.fn = fn


class LazyDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._thunks: Dict[str, Callable[[], Any]] = {}
        # compact twins of entries for in-package consumers (e.g. the uint8 class map behind the
        # int64 'semantic_segmentation_idx'): NOT keys of the dict, so the key set stays the
        # reference's
        self.aux: Dict[str, Any] = {}

    def set_lazy(self, key: str, thunk: Callable[[], Any]) -> None:
        self._thunks[key] = thunk
        super().__setitem__(key, None)        # reserve the slot / ordering

    def set_derived(self, key: str, fn: Callable[['LazyDict'], Any]) -> None:
        """lazy entry computed from OTHER entries: `fn` receives the dict it is read from
        (a thunk that captured the dict itself would form a reference cycle and keep every
        tensor of the result alive until the cycle collector runs)"""
        self._thunks[key] = _Derived(fn)
        super().__setitem__(key, None)

    def _force(self, key):
        thunk = self._thunks.pop(key, None)
        if thunk is not None:
            value = thunk.fn(self) if isinstance(thunk, _Derived) else thunk()
            super().__setitem__(key, value)

    def __getitem__(self, key):
        self._force(key)
        return super().__getitem__(key)

    def __setitem__(self, key, value):
        self._thunks.pop(key, None)
        super().__setitem__(key, value)

    def get(self, key, default=None):
        if key in self:
            return self[key]
        return default

    def pop(self, key, *default):
        self._force(key)
        return super().pop(key, *default)

    def items(self):
        for k in list(self._thunks):
            self._force(k)
        return super().items()

    def values(self):
        for k in list(self._thunks):
            self._force(k)
        return super().values()

    def copy(self):
        new = LazyDict()
        new.merge(self)
        return new

    def merge(self, other: dict) -> 'LazyDict':
        """update() that keeps the other dict's pending thunks pending."""
        if isinstance(other, LazyDict):
            self.aux.update(other.aux)
            for k in other.keys():
                if k in other._thunks:
                    self._thunks[k] = other._thunks[k]
                    dict.__setitem__(self, k, None)
                else:
                    self[k] = dict.__getitem__(other, k)
        else:
            for k, v in other.items():
                self[k] = v
        return self

    def is_pending(self, key: str) -> bool:
        return key in self._thunks
