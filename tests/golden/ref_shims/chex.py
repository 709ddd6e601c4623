"""chex stand-in: `chex.Array` annotation and `chex.dataclass` (frozen dataclass, kw-only)."""
import dataclasses
from typing import Any

Array = Any
ArrayTree = Any
PRNGKey = Any
Numeric = Any
Shape = Any


def dataclass(cls=None, *, frozen=False, **kw):
  def wrap(c):
    return dataclasses.dataclass(c, frozen=frozen, eq=True)
  return wrap if cls is None else wrap(cls)
