"""`partial_class` (reference utils/_misc.py:11-21): a subclass whose __init__
has some arguments bound — used by the get_*_class factories."""
import functools


@functools.lru_cache(maxsize=None)
def _make(cls, args, kwargs_items):
    kwargs = dict(kwargs_items)

    class _Bound(cls):
        __init__ = functools.partialmethod(cls.__init__, *args, **kwargs)

    _Bound.__name__ = cls.__name__
    _Bound.__qualname__ = cls.__qualname__
    return _Bound


def partial_class(cls, *args, **kwargs):
    if not args and not kwargs:
        return cls
    try:
        return _make(cls, args, tuple(sorted(kwargs.items())))
    except TypeError:        # unhashable argument: build uncached
        return _make.__wrapped__(cls, args, tuple(kwargs.items()))
