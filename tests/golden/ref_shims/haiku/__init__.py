"""dm-haiku stand-in on numpy: Module naming, parameters, Linear, nets.MLP, LayerNorm,
Sequential -- the entry points called at
/root/reference/weathernext/utils/legacy/deep_typed_graph_net.py:53,206-208,231-233,246.

Published behaviour restated here:
  * module naming: a module constructed while method `m` of module P is running is
    named "P/~m/<name>" ("P/<name>" when m is __call__, "P/~/<name>" when m is
    __init__); repeated names in one scope get _1, _2 ... suffixes;
  * hk.Linear: y = x @ w + b, w [in, out] ~ TruncatedNormal(1/sqrt(in)), b = 0;
  * hk.nets.MLP: Linear layers named linear_<i> created in __init__, activation
    between layers, activate_final=False;
  * hk.LayerNorm(axis, create_scale, create_offset, eps=1e-5): biased variance,
    (x - mean) * rsqrt(var + eps) * scale + offset;
  * hk.name_like / hk.transparent (naming scopes of utils/dense.py and utils/deep_gnn.py), hk.Bias.
Parameters live in a {module_name: {param_name: array}} dict installed with
`haiku.running(params, init_rng=None)`; with `init_rng` missing leaves are created
(that is how make_golden.py learns the reference's parameter tree).
"""
import contextlib
import functools
import types

import numpy as np

from . import nets  # noqa: F401,E402  (defined below via late import)

_frames = []          # (module, method_name) of running module methods
_state = {"params": None, "rng": None, "dtype": np.float64, "created": None, "counts": {}}


@contextlib.contextmanager
def running(params, init_rng=None, dtype=np.float64):
  old = dict(_state)
  _state.update(params=params, rng=init_rng, dtype=dtype, created=[], counts={})
  try:
    yield _state
  finally:
    _state.clear()
    _state.update(old)


def _wrap_method(name, fn):
  if getattr(fn, "_hk_transparent", False):      # hk.transparent: no scope of its own
    return fn
  scope_name = getattr(fn, "_hk_name_like", name)     # hk.name_like: scoped as if it were that method

  @functools.wraps(fn)
  def wrapped(self, *a, **k):
    _frames.append((self, scope_name))
    try:
      return fn(self, *a, **k)
    finally:
      _frames.pop()
  return wrapped


class _ModuleMeta(type):
  def __new__(mcs, clsname, bases, ns):
    for k, v in list(ns.items()):
      if isinstance(v, types.FunctionType) and (not k.startswith("__") or k in ("__call__", "__init__")):
        ns[k] = _wrap_method(k, v)
    return super().__new__(mcs, clsname, bases, ns)


def _scope_prefix():
  if not _frames:
    return ""
  mod, meth = _frames[-1]
  base = mod.module_name
  if meth == "__call__":
    return base + "/"
  if meth == "__init__":
    return base + "/~/"
  return f"{base}/~{meth}/"


class Module(metaclass=_ModuleMeta):
  def __init__(self, name=None):
    # NOTE: runs inside the subclass's wrapped __init__ frame; the creating scope is
    # the frame BELOW every __init__ frame of this very object.
    if name is None:
      name = _camel_to_snake(type(self).__name__)
    k = len(_frames)
    while k > 0 and _frames[k - 1][0] is self:
      k -= 1
    saved = _frames[k:]
    del _frames[k:]
    prefix = _scope_prefix()
    _frames.extend(saved)
    counts = _state["counts"]
    n = counts.get(prefix + name, 0)
    counts[prefix + name] = n + 1
    self.name = name if n == 0 else f"{name}_{n}"
    self.module_name = prefix + self.name


def _camel_to_snake(s):
  out = []
  for i, c in enumerate(s):
    if c.isupper() and i and not s[i - 1].isupper():
      out.append("_")
    out.append(c.lower())
  return "".join(out)


def get_parameter(name, shape, dtype=None, init=None):
  mod = _frames[-1][0]
  store = _state["params"]
  if store is None:
    raise RuntimeError("haiku stand-in: no parameters installed (use haiku.running)")
  leafs = store.setdefault(mod.module_name, {}) if _state["rng"] is not None else store.get(mod.module_name)
  if leafs is None or name not in leafs:
    if _state["rng"] is None:
      raise KeyError(f"missing parameter {mod.module_name!r}/{name!r}")
    leafs[name] = np.asarray(init(tuple(shape), _state["rng"]), dtype=np.float32)
    _state["created"].append((mod.module_name, name, tuple(shape)))
  p = np.asarray(leafs[name])
  if tuple(p.shape) != tuple(shape):
    raise ValueError(f"{mod.module_name}/{name}: stored shape {p.shape}, module wants {tuple(shape)}")
  return p.astype(_state["dtype"])


def _truncated_normal(stddev):
  def init(shape, rng):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2
    while bad.any():
      x[bad] = rng.standard_normal(int(bad.sum()))
      bad = np.abs(x) > 2
    return x * stddev
  return init


def _constant(v):
  return lambda shape, rng: np.full(shape, v, dtype=np.float64)


class Linear(Module):
  def __init__(self, output_size, with_bias=True, w_init=None, b_init=None, name=None):
    super().__init__(name=name)
    self.output_size, self.with_bias = output_size, with_bias

  def __call__(self, x):
    k = x.shape[-1]
    w = get_parameter("w", (k, self.output_size), init=_truncated_normal(1.0 / np.sqrt(k)))
    y = x @ w
    if self.with_bias:
      y = y + get_parameter("b", (self.output_size,), init=_constant(0.0))
    return y


class LayerNorm(Module):
  def __init__(self, axis, create_scale, create_offset, eps=1e-5, name=None):
    super().__init__(name=name)
    if axis != -1:
      raise NotImplementedError("stand-in: axis=-1 only")
    self.create_scale, self.create_offset, self.eps = create_scale, create_offset, eps

  def __call__(self, x):
    mean = x.mean(axis=-1, keepdims=True)
    var = np.square(x - mean).mean(axis=-1, keepdims=True)
    y = (x - mean) / np.sqrt(var + self.eps)
    if self.create_scale:
      y = y * get_parameter("scale", (x.shape[-1],), init=_constant(1.0))
    if self.create_offset:
      y = y + get_parameter("offset", (x.shape[-1],), init=_constant(0.0))
    return y


class Sequential(Module):
  def __init__(self, layers, name=None):
    super().__init__(name=name)
    self.layers = tuple(layers)

  def __call__(self, x, *args, **kwargs):
    for i, layer in enumerate(self.layers):
      x = layer(x, *args, **kwargs) if i == 0 else layer(x)
    return x


class initializers:  # noqa: N801
  """hk.initializers names used in annotations / defaults of imported code."""
  Initializer = object

  @staticmethod
  def TruncatedNormal(stddev=1.0, mean=0.0):  # noqa: N802
    return _truncated_normal(stddev)

  @staticmethod
  def Constant(v):  # noqa: N802
    return _constant(v)


def name_like(method_name):
  """hk.name_like(m): modules / parameters created in the decorated method are named as if they were
  created in method `m` (utils/dense.py decorates every __init__ with name_like("__call__"): no "~"
  path segment for sub-modules built in constructors)."""
  def deco(f):
    f._hk_name_like = method_name
    return f
  return deco


def transparent(f):
  """hk.transparent: the decorated method opens no naming scope -- what it creates is named in the scope
  it is called from (utils/deep_gnn.py:185 `_networks_builder`)."""
  f._hk_transparent = True
  return f


class Bias(Module):
  """hk.Bias(bias_dims=[-1]): y = x + b, b [x.shape[-1]] (zeros by default)."""

  def __init__(self, bias_dims=None, b_init=None, name=None):
    super().__init__(name=name)
    if bias_dims not in (None, [-1], (-1,)):
      raise NotImplementedError("stand-in: bias over the last axis only")

  def __call__(self, x):
    return x + get_parameter("b", (x.shape[-1],), init=_constant(0.0))


def remat(f, *a, **k):
  return f


def scan(f, init, xs, length=None):
  raise NotImplementedError("stand-in: hk.scan is not on the executed path")
