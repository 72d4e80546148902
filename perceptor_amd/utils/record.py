"""Frozen record with ``replace`` — the role lantern.FunctionalBase plays in the reference
(perceptor/models/guided_diffusion/predictions.py:9, velocity_diffusion/predictions.py:9)."""
from __future__ import annotations


class FrozenRecord:
    _fields: tuple = ()

    def __init__(self, **kw):
        missing = [f for f in self._fields if f not in kw]
        extra = [k for k in kw if k not in self._fields]
        if missing or extra:
            raise TypeError(f"{type(self).__name__}: missing fields {missing}, unexpected {extra}")
        for k, v in kw.items():
            object.__setattr__(self, k, v)

    def __setattr__(self, k, v):
        raise AttributeError(f"{type(self).__name__} is immutable; use .replace(**changes)")

    def replace(self, **kw):
        d = {f: getattr(self, f) for f in self._fields}
        d.update(kw)
        return type(self)(**d)

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{f}=<{tuple(getattr(self, f).shape)}>' for f in self._fields)})"
