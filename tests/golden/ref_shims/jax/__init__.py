"""jax stand-in on numpy: tree utilities, jax.nn activations, no-op jit/device_get."""
import numpy as _np

from . import lax  # noqa: F401
from . import numpy  # noqa: F401
from . import tree_util  # noqa: F401
from . import tree_util as tree  # noqa: F401  (jax.tree.map / flatten / unflatten)

Array = _np.ndarray


class nn:  # noqa: N801
  @staticmethod
  def swish(x):
    """jax.nn.swish / silu: x * sigmoid(x)."""
    return x / (1.0 + _np.exp(-x))

  silu = swish

  @staticmethod
  def relu(x):
    return _np.maximum(x, 0)

  @staticmethod
  def sigmoid(x):
    return 1.0 / (1.0 + _np.exp(-x))


def jit(f, *a, **k):
  return f


def device_get(x):
  return x


def device_put(x, *a, **k):
  return x


def local_devices():
  return [None]


def devices():
  return [None]


class random:  # noqa: N801
  @staticmethod
  def PRNGKey(seed):  # noqa: N802
    return _np.array([0, seed], dtype=_np.uint32)

  @staticmethod
  def split(key, num=2):
    rng = _np.random.default_rng(int(key[-1]) + 7919 * int(key[0]))
    return rng.integers(0, 2**31, size=(num, 2)).astype(_np.uint32)
