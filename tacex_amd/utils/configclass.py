"""Minimal stand-in for `isaaclab.utils.configclass` (IsaacLab is not part of this build).

The reference declares every cfg with IsaacLab's @configclass (e.g. gelsight_sensor_cfg.py:12,
gpu_taxim/taxim_sim_cfg.py:11): a dataclass that tolerates mutable defaults, un-annotated fields and
nested cfg instances as defaults, plus copy()/replace()/to_dict()/validate().  This keeps the same
field names and construction syntax so task code written against the reference cfgs reads the same.
"""
from __future__ import annotations

import copy
import dataclasses
from dataclasses import MISSING, dataclass, field, fields, is_dataclass
from typing import Any

__all__ = ["configclass", "MISSING"]

_IMMUTABLE = (int, float, str, bool, bytes, type(None), tuple, frozenset, type)


def _default_factory(value):
    return lambda: copy.deepcopy(value)


def configclass(cls=None, **kwargs):
    def wrap(c):
        ann = dict(c.__dict__.get("__annotations__", {}))
        # give un-annotated, non-callable class attributes an annotation so they become fields
        for name, value in list(c.__dict__.items()):
            if name.startswith("_") or name in ann:
                continue
            if isinstance(value, (staticmethod, classmethod, property)) or callable(value) and not isinstance(value, type):
                continue
            if isinstance(value, type) and is_dataclass(value) and value.__qualname__.startswith(c.__qualname__ + "."):
                continue  # nested cfg class definition, not a field
            ann[name] = type(value)
        c.__annotations__ = ann
        # mutable defaults -> default_factory(deepcopy)
        for name in ann:
            if name in c.__dict__:
                value = c.__dict__[name]
                if isinstance(value, dataclasses.Field):
                    continue
                if value is MISSING:
                    continue
                if not isinstance(value, _IMMUTABLE):
                    setattr(c, name, field(default_factory=_default_factory(value)))
        # MISSING class defaults stay as dataclasses.MISSING sentinels *values* (IsaacLab semantics:
        # "must be set by the user"); dataclass() would treat them as "no default", which forbids
        # following defaulted fields -> store the sentinel through a factory.
        for name in ann:
            if c.__dict__.get(name, None) is MISSING:
                setattr(c, name, field(default_factory=lambda: MISSING))
        c = dataclass(c, **kwargs)

        def _copy(self):
            return copy.deepcopy(self)

        def _replace(self, **changes):
            new = copy.deepcopy(self)
            for k, v in changes.items():
                if not hasattr(new, k):
                    raise AttributeError(f"{type(self).__name__} has no field '{k}'")
                setattr(new, k, v)
            return new

        def _to_dict(self) -> dict[str, Any]:
            out = {}
            for f in fields(self):
                v = getattr(self, f.name)
                out[f.name] = v.to_dict() if hasattr(v, "to_dict") and is_dataclass(v) else v
            return out

        def _validate(self, prefix: str = ""):
            missing = []
            for f in fields(self):
                v = getattr(self, f.name)
                if v is MISSING:
                    missing.append(prefix + f.name)
                elif is_dataclass(v) and hasattr(v, "validate"):
                    try:
                        v.validate(prefix + f.name + ".")
                    except TypeError as e:
                        missing.append(str(e))
            if missing:
                raise TypeError(f"Missing values detected in object {type(self).__name__} for the following fields: {missing}")

        c.copy = _copy
        c.replace = _replace
        c.to_dict = _to_dict
        c.validate = _validate
        return c

    return wrap if cls is None else wrap(cls)
