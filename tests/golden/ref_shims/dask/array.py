import numpy as _np


def zeros_like(a, *args, **kw):
  return _np.zeros_like(a)
