"""jax.numpy stand-in = numpy (+ the keyword differences the executed code uses)."""
import numpy as _np
from numpy import *  # noqa: F401,F403

ndarray = _np.ndarray
bfloat16 = getattr(_np, "bfloat16", _np.float32)   # only used as a dtype tag by casting.py


def repeat(a, repeats, axis=None, total_repeat_length=None):
  out = _np.repeat(a, repeats, axis=axis)
  if total_repeat_length is not None:
    n = out.shape[axis if axis is not None else 0]
    if n != total_repeat_length:
      raise ValueError(f"repeat: total_repeat_length {total_repeat_length} != {n}")
  return out
