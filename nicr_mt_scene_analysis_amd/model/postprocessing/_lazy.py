"""A dict whose expensive entries are computed on first access.

The reference materialises every derived tensor eagerly (e.g. the full
B x C x H x W softmax, 160 B/px of extra HBM writes at C=40).  Consumers of the
postprocessing dict usually read a handful of keys, so entries that are pure
functions of other entries are registered as thunks here: the key is listed by
keys()/in/len() from the start and the value is produced (by a HIP kernel) the
first time it is read.

The result must still behave like the plain dict the reference returns
(reference model/postprocessing/panoptic.py:75,94 merges the per-task dicts
with `{**a, **b}`).  Every way of reading a value therefore forces the thunk:

* `d[k]`, `get`, `pop`, `setdefault`, `items()`, `values()`, `==`, `repr`;
* `{**d}`, `dict(d)`, `x.update(d)`, `f(**d)`: CPython copies the hash table of
  a dict subclass directly ONLY while `tp_iter` is `dict.__iter__`; with the
  `__iter__` override below it goes through `keys()` + `__getitem__`;
* `pickle` / `copy.deepcopy` see a plain `dict` of the forced values
  (`__reduce__`); `copy.copy` / `.copy()` keep pending entries pending.

Writes (`d[k] = v`, `update`, `del`, `clear`, `|=`) drop the pending thunk so a
stale thunk can never overwrite a value the caller has set.
"""
from typing import Any, Callable, Dict

_MISSING = object()


class _Derived:
    __slots__ = ('fn',)

    def __init__(self, fn):
        self.fn = fn


class LazyDict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self._thunks: Dict[str, Callable[[], Any]] = {}
        # compact twins of entries for in-package consumers (e.g. the uint8 class map behind the
        # int64 'semantic_segmentation_idx'): NOT keys of the dict, so the key set stays the
        # reference's
        self.aux: Dict[str, Any] = {}
        if args or kwargs:
            self.update(*args, **kwargs)

    # ---- registration ---------------------------------------------------------------
    def set_lazy(self, key: str, thunk: Callable[[], Any]) -> None:
        self._thunks[key] = thunk
        super().__setitem__(key, None)        # reserve the slot / ordering

    def set_derived(self, key: str, fn: Callable[['LazyDict'], Any]) -> None:
        """lazy entry computed from OTHER entries: `fn` receives the dict it is read from
        (a thunk that captured the dict itself would form a reference cycle and keep every
        tensor of the result alive until the cycle collector runs)"""
        self._thunks[key] = _Derived(fn)
        super().__setitem__(key, None)

    def is_pending(self, key: str) -> bool:
        return key in self._thunks

    def _force(self, key):
        thunk = self._thunks.pop(key, None)
        if thunk is not None:
            value = thunk.fn(self) if isinstance(thunk, _Derived) else thunk()
            super().__setitem__(key, value)

    def force_all(self) -> 'LazyDict':
        for k in list(self._thunks):
            self._force(k)
        return self

    # ---- reads ----------------------------------------------------------------------
    def __iter__(self):
        # overriding tp_iter is what keeps `{**d}` / `dict(d)` / `x.update(d)` off CPython's
        # raw hash-table copy (which would hand out the None placeholders)
        return iter(super().keys())

    def __getitem__(self, key):
        self._force(key)
        return super().__getitem__(key)

    def get(self, key, default=None):
        if key in self:
            return self[key]
        return default

    def pop(self, key, *default):
        self._force(key)
        return super().pop(key, *default)

    def popitem(self):
        if not self:
            raise KeyError('popitem(): dictionary is empty')
        key = next(reversed(super().keys()))
        return key, self.pop(key)

    def setdefault(self, key, default=None):
        if key in self:
            return self[key]
        self[key] = default
        return default

    def items(self):
        self.force_all()
        return super().items()

    def values(self):
        self.force_all()
        return super().values()

    def __eq__(self, other):
        self.force_all()
        if isinstance(other, LazyDict):
            other.force_all()
        return super().__eq__(other)

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = None

    def __repr__(self):
        self.force_all()
        return super().__repr__()

    # ---- writes ---------------------------------------------------------------------
    def __setitem__(self, key, value):
        self._thunks.pop(key, None)
        super().__setitem__(key, value)

    def __delitem__(self, key):
        self._thunks.pop(key, None)
        super().__delitem__(key)

    def clear(self):
        self._thunks.clear()
        self.aux.clear()
        super().clear()

    def update(self, *args, **kwargs):
        if len(args) > 1:
            raise TypeError(f'update expected at most 1 argument, got {len(args)}')
        if args:
            other = args[0]
            if hasattr(other, 'keys'):
                for k in other.keys():
                    self[k] = other[k]
            else:
                for k, v in other:
                    self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    def __or__(self, other):
        if not isinstance(other, dict):
            return NotImplemented
        new = self.copy()
        new.merge(other)
        return new

    def __ror__(self, other):
        if not isinstance(other, dict):
            return NotImplemented
        new = LazyDict()
        new.merge(other)
        new.merge(self)
        return new

    def __ior__(self, other):
        self.update(other)
        return self

    # ---- copies ---------------------------------------------------------------------
    def copy(self):
        new = LazyDict()
        new.merge(self)
        return new

    __copy__ = copy

    def __reduce__(self):
        # pickle / deepcopy: a plain dict of the forced values (thunks hold kernels' inputs and
        # closures, nothing a consumer on the other side could run)
        return (dict, (dict(self.items()),))

    def __reduce_ex__(self, protocol):
        return self.__reduce__()

    def merge(self, other: dict) -> 'LazyDict':
        """update() that keeps the other dict's pending thunks pending."""
        if isinstance(other, LazyDict):
            self.aux.update(other.aux)
            for k in dict.keys(other):
                if k in other._thunks:
                    self._thunks[k] = other._thunks[k]
                    dict.__setitem__(self, k, None)
                else:
                    self[k] = dict.__getitem__(other, k)
        else:
            for k, v in other.items():
                self[k] = v
        return self
