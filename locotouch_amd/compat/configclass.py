"""`configclass` stand-in for the `isaaclab.utils.configclass` decorator the reference's cfg files use.

Behaviour the reference relies on (SURVEY.md §5 "Config / flags"): nested class defaults, mutable
defaults (deep-copied per instance), un-annotated class attributes becoming fields, `MISSING`,
`.replace()`, `.copy()`, `.to_dict()`, `__post_init__` chains through inheritance, attribute
injection after construction (`self.scene.object = ...`,
reference locotouch/config/locotouch/object_transport_teacher_env_cfg.py:63) and setting a term to
None to delete it (`:108`).  IsaacLab itself is absent from this image, so this is a restatement
of its documented behaviour, not a copy.
"""
from __future__ import annotations

import copy
import dataclasses
import types
from dataclasses import MISSING, field
from typing import Any, Callable

_IMMUTABLE = (int, float, str, bool, type(None), tuple, frozenset, bytes, type, types.FunctionType,
              types.BuiltinFunctionType, types.MethodType)


def _is_immutable(v) -> bool:
    if isinstance(v, tuple):
        return all(_is_immutable(x) for x in v)
    return isinstance(v, _IMMUTABLE) or v is MISSING


def _to_dict(obj) -> Any:
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        out = {}
        for k, v in obj.__dict__.items():
            if k.startswith("__"):
                continue
            out[k] = _to_dict(v)
        return out
    if isinstance(obj, dict):
        return {k: _to_dict(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_dict(v) for v in obj) if not hasattr(obj, "_fields") else obj
    if not isinstance(obj, type) and hasattr(obj, "__dict__") and callable(getattr(type(obj), "to_dict", None)):
        # permissive records (isaaclab_shim._AnyCfg) and any other cfg-like object: plain dict of their attributes
        return {k: _to_dict(v) for k, v in vars(obj).items() if not k.startswith("__")}
    if isinstance(obj, type) or callable(obj):
        mod = getattr(obj, "__module__", None)
        name = getattr(obj, "__qualname__", getattr(obj, "__name__", None))
        if mod and name:
            return f"{mod}:{name}"
    return obj


def _update_from_dict(obj, data: dict, _ns: str = "") -> None:
    for key, value in data.items():
        if not hasattr(obj, key):
            raise KeyError(f"[configclass] key not found under namespace: {_ns}/{key}")
        member = getattr(obj, key)
        if isinstance(value, dict) and dataclasses.is_dataclass(member):
            _update_from_dict(member, value, f"{_ns}/{key}")
        elif isinstance(value, dict) and isinstance(member, dict):
            member.update(value)
        else:
            if isinstance(member, tuple) and isinstance(value, list):
                value = tuple(value)
            setattr(obj, key, value)


def _replace(obj, **kwargs):
    new = copy.deepcopy(obj)
    for k, v in kwargs.items():
        setattr(new, k, v)
    return new


def _validate(obj, prefix: str = "") -> list[str]:
    missing = []
    if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
        for k, v in obj.__dict__.items():
            missing += _validate(v, f"{prefix}.{k}" if prefix else k)
    elif obj is MISSING:
        missing.append(prefix)
    elif isinstance(obj, dict):
        for k, v in obj.items():
            missing += _validate(v, f"{prefix}.{k}")
    return missing


def _missing():
    return MISSING


def configclass(cls=None, **kwargs) -> Callable:
    """Turn a plain class with (possibly un-annotated, possibly mutable) attributes into a dataclass."""

    def wrap(c):
        ann = dict(c.__dict__.get("__annotations__", {}))
        # un-annotated public class attributes become fields (isaaclab allows `terrain = TerrainImporterCfg(...)`)
        for name, value in list(c.__dict__.items()):
            if name.startswith("_") or name in ann:
                continue
            if isinstance(value, (types.FunctionType, classmethod, staticmethod, property)):
                continue
            if isinstance(value, type) and value.__qualname__.startswith(c.__qualname__ + "."):
                continue  # nested class definition, not a field
            ann[name] = type(value) if value is not None else Any
        # inherited fields whose default is overridden without annotation are handled above; keep order:
        # base-class fields first (dataclass does that itself).
        c.__annotations__ = ann
        for name in ann:
            if name in c.__dict__:
                value = c.__dict__[name]
                if isinstance(value, dataclasses.Field):
                    continue
                if value is MISSING:
                    setattr(c, name, field(default_factory=_missing))
                elif not _is_immutable(value):
                    setattr(c, name, field(default_factory=(lambda v=value: copy.deepcopy(v))))
            else:
                # annotated without default: inherit a default if a base has it, else MISSING sentinel
                inherited = MISSING
                for b in c.__mro__[1:]:
                    if name in getattr(b, "__dataclass_fields__", {}):
                        inherited = None  # dataclass inheritance will take care of it
                        break
                if inherited is MISSING:
                    setattr(c, name, field(default_factory=_missing))
        if not any("__post_init__" in b.__dict__ for b in c.__mro__):
            c.__post_init__ = lambda self: None  # so that `super().__post_init__()` chains terminate (isaaclab provides one)
        c = dataclasses.dataclass(c, eq=False, **kwargs)
        # kw-only style reordering is not needed: every field has a default by construction.
        c.to_dict = _to_dict
        c.from_dict = _update_from_dict
        c.replace = _replace
        c.copy = lambda self: copy.deepcopy(self)
        c.validate = lambda self: _raise_missing(_validate(self), type(self).__name__)
        return c

    if cls is None:
        return wrap
    return wrap(cls)


def _raise_missing(missing: list[str], name: str) -> None:
    if missing:
        raise TypeError(f"Missing values detected in object {name} for the following fields:\n" +
                        "\n".join(f"  - {m}" for m in missing))
