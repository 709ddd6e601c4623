"""A minimal labelled-array container: the subset of xarray the GraphCast path touches.

xarray is not installable in the target image, and the reference's Predictor /
rollout API takes and returns ``xarray.Dataset`` (``predictor_base.py:43-84``,
``rollout.py:326-364``).  This module implements exactly the surface listed in
SURVEY.md appendix A.6 with xarray's semantics, so that

  * ``graphcast_amd.model_utils`` / ``rollout`` read like the reference's code, and
  * the reference's own ``model_utils.py`` / ``graphcast.py`` can be executed on
    top of it when golden vectors are generated (``tests/golden/make_golden.py``).

Surface (reference call sites: ``graphcast.py:689-699,711-723``,
``model_utils.py:161,171-177,668-674,699-710,738-776``, ``rollout.py:417-435,
453-463,499-559,587-604``, ``normalization.py:29-132``):

  Variable(dims, data): .dims .data .shape .sizes .dtype .stack(channels=[...])
      .set_dims({...}) .unstack({"channels": {...}}) .transpose(*dims|...)
      .isel({...}) .astype() Variable.concat(list, dim)
  DataArray(data, coords=None, dims=None, name=None): .variable .coords .name
      + the Variable surface, arithmetic with by-name broadcasting
  Dataset(data_vars, coords=None): .data_vars .variables .coords .sizes .dims
      .keys() [name] [[names]] .copy() .isel(time=...) .compute()
      .assign_coords(...) .assign(other) .tail(time=n) .astype() .lat/.lon
  concat([...], dim, data_vars="all"|"different", compat=...)

``data`` may be a numpy array or a torch tensor (device-resident rollouts keep
the state on the GPU); every operation used here works on both.
"""
import collections.abc
import math
from typing import Any, Dict, Mapping, Sequence

import numpy as np


# ----------------------------------------------------------------------------- backends
def _is_torch(a):
  return type(a).__module__.split(".")[0] == "torch"


def _asdata(a):
  if _is_torch(a) or isinstance(a, np.ndarray):
    return a
  if isinstance(a, (Variable, DataArray)):
    return a.data
  return np.asarray(a)


def _transpose(a, axes):
  return a.permute(*axes) if _is_torch(a) else np.transpose(a, axes)


def _reshape(a, shape):
  return a.reshape(tuple(int(s) for s in shape))


def _broadcast_to(a, shape):
  return a.expand(*shape) if _is_torch(a) else np.broadcast_to(a, shape)


def _concatenate(arrays, axis):
  if any(_is_torch(a) for a in arrays):
    import torch
    ref = next(a for a in arrays if _is_torch(a))
    arrays = [a if _is_torch(a) else torch.as_tensor(a, device=ref.device) for a in arrays]
    return torch.cat(arrays, dim=axis)
  return np.concatenate(arrays, axis=axis)


def _astype(a, dtype):
  if _is_torch(a):
    import torch
    if not isinstance(dtype, torch.dtype):
      dtype = getattr(torch, np.dtype(dtype).name)
    return a.to(dtype)
  if type(dtype).__module__.split(".")[0] == "torch":      # numpy data, torch dtype requested
    dtype = np.dtype(str(dtype).split(".")[-1])
  return a.astype(dtype)


def _equal(a, b):
  a, b = _asdata(a), _asdata(b)
  if tuple(a.shape) != tuple(b.shape):
    return False
  if _is_torch(a) or _is_torch(b):
    import torch
    return bool(torch.equal(torch.as_tensor(a), torch.as_tensor(b).to(torch.as_tensor(a).device)))
  return bool(np.array_equal(a, b))


# ----------------------------------------------------------------------------- Variable
class Variable:
  """N-d array with named dimensions (xarray.Variable subset)."""

  __slots__ = ("_dims", "_data")

  def __init__(self, dims, data):
    if isinstance(dims, str):
      dims = (dims,)
    data = _asdata(data)
    if len(dims) != data.ndim:
      raise ValueError(f"dimensions {tuple(dims)} must have the same length as the number of "
                       f"data dimensions, ndim={data.ndim}")
    if len(set(dims)) != len(dims):
      raise ValueError(f"repeated dimension names in {tuple(dims)}")
    self._dims, self._data = tuple(dims), data

  dims = property(lambda self: self._dims)
  data = property(lambda self: self._data)
  shape = property(lambda self: tuple(self._data.shape))
  ndim = property(lambda self: self._data.ndim)
  dtype = property(lambda self: self._data.dtype)
  variable = property(lambda self: self)

  @property
  def values(self):
    return self._data.cpu().numpy() if _is_torch(self._data) else np.asarray(self._data)

  @property
  def sizes(self) -> Dict[str, int]:
    return dict(zip(self._dims, self.shape))

  def __array__(self, dtype=None, copy=None):
    return np.asarray(self.values, dtype=dtype)

  def __repr__(self):
    return f"<xarray_lite.Variable {self.sizes} {self.dtype}>"

  def _axes(self, dims):
    return [self._dims.index(d) for d in dims]

  def transpose(self, *dims):
    if not dims:
      dims = self._dims[::-1]
    if any(d is Ellipsis for d in dims):
      named = [d for d in dims if d is not Ellipsis]
      rest = [d for d in self._dims if d not in named]
      out = []
      for d in dims:
        out.extend(rest if d is Ellipsis else [d])
      dims = out
    if set(dims) != set(self._dims) or len(dims) != len(self._dims):
      raise ValueError(f"{tuple(dims)} must be a permuted list of {self._dims}, unless `...` is included")
    if tuple(dims) == self._dims:
      return Variable(self._dims, self._data)
    return Variable(dims, _transpose(self._data, self._axes(dims)))

  def stack(self, dimensions=None, **dimensions_kwargs):
    """Stacks dims into a new trailing dim (C order over the listed dims)."""
    dimensions = dict(dimensions or {}, **dimensions_kwargs)
    out = self
    for new_dim, dims in dimensions.items():
      dims = list(dims)
      if not set(dims) <= set(out._dims):
        raise ValueError(f"invalid existing dimensions: {dims}")
      if new_dim in out._dims:
        raise ValueError("cannot create a new dimension with the same name as an existing dimension")
      other = [d for d in out._dims if d not in dims]
      t = out.transpose(*(other + dims))
      new_shape = t.shape[:len(other)] + (math.prod(t.shape[len(other):]),)
      out = Variable(other + [new_dim], _reshape(t._data, new_shape))
    return out

  def unstack(self, dimensions=None, **dimensions_kwargs):
    """{"old": {"new_a": size, ...}}: splits `old` into new trailing dims."""
    dimensions = dict(dimensions or {}, **dimensions_kwargs)
    out = self
    for old_dim, new in dimensions.items():
      names, sizes = list(new.keys()), [int(s) for s in new.values()]
      if old_dim not in out._dims:
        raise ValueError(f"invalid existing dimension: {old_dim}")
      if math.prod(sizes) != out.sizes[old_dim]:
        raise ValueError("the product of the new dimension sizes must equal the size of the old dimension")
      other = [d for d in out._dims if d != old_dim]
      t = out.transpose(*(other + [old_dim]))
      out = Variable(other + names, _reshape(t._data, t.shape[:-1] + tuple(sizes)))
    return out

  def set_dims(self, dims, shape=None):
    """Result has exactly `dims` (in that order); missing ones are broadcast."""
    if isinstance(dims, str):
      dims = [dims]
    if shape is None and isinstance(dims, Mapping):
      shape = list(dims.values())
    dims = list(dims)
    missing = set(self._dims) - set(dims)
    if missing:
      raise ValueError(f"new dimensions {dims} must be a superset of existing dimensions {self._dims}")
    new = [d for d in dims if d not in self._dims]
    data = self._data
    for _ in new:
      data = data[None]
    expanded = Variable(new + list(self._dims), data)
    if shape is not None:
      want = dict(zip(dims, shape))
      target = tuple(int(want[d]) for d in expanded._dims)
      for d, have, w in zip(expanded._dims, expanded.shape, target):
        if have != w and d in self._dims:
          raise ValueError(f"dimension {d!r} has size {have}, set_dims asked for {w}")
      expanded = Variable(expanded._dims, _broadcast_to(expanded._data, target))
    return expanded.transpose(*dims)

  def isel(self, indexers=None, **indexers_kwargs):
    indexers = dict(indexers or {}, **indexers_kwargs)
    bad = set(indexers) - set(self._dims)
    if bad:
      raise ValueError(f"Dimensions {bad} do not exist. Expected one or more of {self._dims}")
    # ORTHOGONAL (outer) indexing like xarray, one dimension at a time: handing the whole tuple to
    # numpy would pair list indexers pointwise and, when an integer and a list are separated by a
    # slice, move the advanced axes to the front while `dims` keeps the original order
    data = self._data
    drop = []
    for axis, d in enumerate(self._dims):
      k = indexers.get(d, slice(None))
      if isinstance(k, (int, np.integer)):
        drop.append(axis)
        k = slice(int(k), int(k) + 1) if k != -1 else slice(-1, None)
      if isinstance(k, slice):
        if k != slice(None):
          data = data[(slice(None),) * axis + (k,)]
      else:
        idx = np.asarray(k)
        if idx.dtype == bool:
          idx = np.nonzero(idx)[0]
        if idx.ndim != 1:
          raise IndexError(f"indexer for dimension {d!r} must be an int, a slice or 1-d")
        if _is_torch(data):
          import torch
          data = torch.index_select(data, axis, torch.as_tensor(idx.astype(np.int64), device=data.device))
        else:
          data = np.take(data, idx, axis=axis)
    dims = [d for axis, d in enumerate(self._dims) if axis not in drop]
    if drop:
      data = _reshape(data, tuple(n for axis, n in enumerate(data.shape) if axis not in drop))
    return Variable(dims, data)

  def astype(self, dtype):
    return Variable(self._dims, _astype(self._data, dtype))

  def copy(self, deep=True):
    data = self._data
    if deep:
      data = data.clone() if _is_torch(data) else data.copy()
    return Variable(self._dims, data)

  def equals(self, other):
    other = getattr(other, "variable", other)
    return self._dims == other._dims and _equal(self._data, other._data)

  @classmethod
  def concat(cls, variables, dim="concat_dim"):
    variables = [getattr(v, "variable", v) for v in variables]
    first = variables[0]
    if dim in first._dims:
      axis = first._dims.index(dim)
      arrays = [v.transpose(*first._dims)._data for v in variables]
      return cls(first._dims, _concatenate(arrays, axis))
    arrays = [v.transpose(*first._dims)._data[None] for v in variables]
    return cls((dim,) + first._dims, _concatenate(arrays, 0))


# ----------------------------------------------------------------------------- label indexing
def _label(x, dtype):
  """A user label (str / pandas.Timedelta / number ...) in the coordinate's own dtype."""
  if np.issubdtype(dtype, np.timedelta64):
    if isinstance(x, np.timedelta64):
      return x.astype(dtype)
    import pandas as pd
    return np.timedelta64(pd.Timedelta(x).value, "ns").astype(dtype)
  if np.issubdtype(dtype, np.datetime64):
    return np.datetime64(x).astype(dtype)
  return x


def _positions(coord: "Variable", dim: str, key):
  """Label-based indexer along `dim` -> positional indexer (xarray `.sel`, exact matches;
  slices include both end points, as label slices do)."""
  if coord.ndim != 1:
    raise ValueError(f"cannot select along {dim!r}: its coordinate is not 1-d")
  values = coord.values
  if isinstance(key, slice):
    if key.step is not None:
      raise NotImplementedError("label slices with a step")
    ok = np.ones(len(values), dtype=bool)
    if key.start is not None:
      ok &= values >= _label(key.start, values.dtype)
    if key.stop is not None:
      ok &= values <= _label(key.stop, values.dtype)
    idx = np.flatnonzero(ok)
    if len(idx) and not np.array_equal(idx, np.arange(idx[0], idx[-1] + 1)):
      raise ValueError(f"label slice along unsorted coordinate {dim!r}")
    return slice(int(idx[0]), int(idx[-1]) + 1) if len(idx) else slice(0, 0)
  scalar = np.ndim(key) == 0 and not isinstance(key, (list, tuple))
  out = []
  for lab in ([key] if scalar else list(key)):
    hit = np.flatnonzero(values == _label(lab, values.dtype))
    if not len(hit):
      raise KeyError(f"{lab!r} not found in coordinate {dim!r}")
    out.append(int(hit[0]))
  return out[0] if scalar else out


# ----------------------------------------------------------------------------- coords
class _Coords(collections.abc.MutableMapping):
  """name -> DataArray; stored as name -> Variable on the owner."""

  def __init__(self, store: Dict[str, Variable]):
    self._store = store

  def __getitem__(self, k):
    v = self._store[k]
    own = {k: v} if v.dims == (k,) else {}
    return DataArray(v, coords=own, name=k)

  def __setitem__(self, k, v):
    self._store[k] = _as_variable(v, default_dim=k)

  def __delitem__(self, k):
    del self._store[k]

  def __iter__(self):
    return iter(self._store)

  def __len__(self):
    return len(self._store)

  def __repr__(self):
    return f"Coordinates({ {k: v.sizes for k, v in self._store.items()} })"


def _as_variable(v, default_dim=None) -> Variable:
  if isinstance(v, Variable):
    return v
  if isinstance(v, DataArray):
    return v.variable
  if isinstance(v, tuple) and len(v) == 2 and not isinstance(v[0], (int, float, np.number)):
    return Variable(v[0], v[1])
  data = _asdata(v)
  if data.ndim == 0:
    return Variable((), data)
  if data.ndim == 1 and default_dim is not None:
    return Variable((default_dim,), data)
  raise ValueError(f"cannot infer dimensions for {default_dim!r} from an array of shape {data.shape}")


def _coords_for(dims, coords) -> Dict[str, Variable]:
  """Keeps the coordinates whose dims are all present (xarray drops the others)."""
  out = {}
  for k, v in (coords or {}).items():
    var = _as_variable(v, default_dim=k)
    if set(var.dims) <= set(dims):
      out[k] = var
  return out


# ----------------------------------------------------------------------------- DataArray
class DataArray:
  """A Variable + name + coordinates (xarray.DataArray subset)."""

  def __init__(self, data=None, coords=None, dims=None, name=None):
    if isinstance(data, DataArray):
      coords = dict(data._coords) if coords is None else coords
      name = data.name if name is None else name
      data = data.variable
    if isinstance(data, Variable):
      if dims is not None and tuple(dims) != data.dims:
        raise ValueError("dims conflict with the Variable's dims")
      var = data
    else:
      data = _asdata(data)
      if dims is None:
        if coords is not None and not isinstance(coords, Mapping):
          dims = [c[0] for c in coords]
          coords = {c[0]: c[1] for c in coords}
        elif coords is not None and data.ndim == len(coords):
          dims = list(coords.keys())
        elif data.ndim == 0:
          dims = ()
        else:
          raise ValueError("DataArray needs dims (or one coordinate per axis)")
      var = Variable(dims, data)
    self._variable = var
    self._coords = _coords_for(var.dims, coords)
    for k, c in self._coords.items():
      for d, n in c.sizes.items():
        if var.sizes[d] != n:
          raise ValueError(f"coordinate {k!r} has size {n} along {d!r}, data has {var.sizes[d]}")
    self.name = name

  variable = property(lambda self: self._variable)
  dims = property(lambda self: self._variable.dims)
  data = property(lambda self: self._variable.data)
  values = property(lambda self: self._variable.values)
  shape = property(lambda self: self._variable.shape)
  sizes = property(lambda self: self._variable.sizes)
  dtype = property(lambda self: self._variable.dtype)
  ndim = property(lambda self: self._variable.ndim)

  @property
  def coords(self):
    return _Coords(self._coords)

  def __array__(self, dtype=None, copy=None):
    return np.asarray(self.values, dtype=dtype)

  def __repr__(self):
    return f"<xarray_lite.DataArray {self.name!r} {self.sizes} {self.dtype}>"

  def __getattr__(self, name):
    coords = self.__dict__.get("_coords", {})
    if name in coords:
      return self.coords[name]
    raise AttributeError(name)

  def __len__(self):
    return self.shape[0]

  def _new(self, var, coords=None):
    return DataArray(var, coords=self._coords if coords is None else coords, name=self.name)

  def transpose(self, *dims):
    return self._new(self._variable.transpose(*dims))

  def astype(self, dtype):
    return self._new(self._variable.astype(dtype))

  def isel(self, indexers=None, **kw):
    indexers = dict(indexers or {}, **kw)
    coords = {k: c.isel({d: i for d, i in indexers.items() if d in c.dims})
              for k, c in self._coords.items()}
    return self._new(self._variable.isel(indexers), coords)

  def sel(self, indexers=None, **kw):
    indexers = dict(indexers or {}, **kw)
    return self.isel({d: _positions(self._coords[d], d, k) for d, k in indexers.items()})

  def __getitem__(self, key):
    """Positional indexing along the leading dimension(s) (`time[-1]`, `x[:, 0]`)."""
    key = key if isinstance(key, tuple) else (key,)
    return self.isel(dict(zip(self.dims, key)))

  def item(self):
    return self.values.item()

  def squeeze(self, dim=None):
    dims = [dim] if isinstance(dim, str) else list(dim if dim is not None else
                                                    [d for d, n in self.sizes.items() if n == 1])
    for d in dims:
      if self.sizes[d] != 1:
        raise ValueError("cannot select a dimension to squeeze out which has length greater than one")
    return self.isel({d: 0 for d in dims})

  def expand_dims(self, dim, axis=0):
    """A new leading (axis=0) dimension of size 1; coordinates keep their own dims."""
    if axis != 0:
      raise NotImplementedError("expand_dims: only axis=0")
    var = Variable((dim,) + self.dims, self.data[None])
    return DataArray(var, coords=dict(self._coords), name=self.name)

  def copy(self, deep=True):
    return DataArray(self._variable.copy(deep), coords=dict(self._coords), name=self.name)

  def rename(self, new_name):
    return DataArray(self._variable, coords=dict(self._coords), name=new_name)

  def compute(self):
    return self

  def equals(self, other):
    return self._variable.equals(getattr(other, "variable", other))

  def assign_coords(self, coords=None, **kw):
    new = dict(self._coords)
    for k, v in dict(coords or {}, **kw).items():
      new[k] = _as_variable(v, default_dim=k)
    return DataArray(self._variable, coords=new, name=self.name)

  # by-name broadcasting arithmetic (normalization.py:29-48,113-132)
  def _binary(self, other, op):
    if isinstance(other, (DataArray, Variable)):
      # xarray aligns on coordinate LABELS before the arithmetic (inner join): the reference's
      # normalisation relies on it -- 37-level statistics applied to 13-level inputs
      # (normalization.py:29-48).  Shared dims whose 1-d dimension coordinates differ are
      # reindexed to the common labels, in this operand's order.
      me = self
      if isinstance(other, DataArray):
        for d in self.dims:
          if d in other.dims and d in self._coords and d in other._coords:
            mine, theirs = np.asarray(self._coords[d].values), np.asarray(other._coords[d].values)
            if mine.ndim == 1 and theirs.ndim == 1 and not (mine.shape == theirs.shape and (mine == theirs).all()):
              where = {l: i for i, l in enumerate(theirs.tolist())}
              keep = [i for i, l in enumerate(mine.tolist()) if l in where]
              if not keep:
                raise ValueError(f"no overlapping labels along {d!r}")
              if len(keep) != len(mine):
                me = me.isel({d: keep})
              other = other.isel({d: [where[l] for l in np.asarray(me._coords[d].values).tolist()]})
      if me is not self:
        return me._binary(other, op)
      ov = other.variable
      dims = list(self.dims) + [d for d in ov.dims if d not in self.dims]
      sizes = dict(ov.sizes, **self.sizes)
      a = self._variable.set_dims(dims, [self.sizes.get(d, 1) for d in dims]) if dims != list(self.dims) else self._variable
      b = ov.set_dims(dims, [ov.sizes.get(d, 1) for d in dims])
      for d in dims:
        if d in self.sizes and d in ov.sizes and self.sizes[d] != ov.sizes[d]:
          raise ValueError(f"size mismatch along {d!r}: {self.sizes[d]} vs {ov.sizes[d]}")
      bd = b.data
      if _is_torch(a.data) and not _is_torch(bd):
        import torch
        # (np.array: a fresh, writable buffer -- broadcast views / read-only statistics would
        #  otherwise reach torch as non-writable memory)
        bd = torch.as_tensor(np.array(bd, order="C"), device=a.data.device)
      coords = dict(getattr(other, "_coords", {}))
      coords.update(self._coords)
      del sizes
      return DataArray(Variable(dims, op(a.data, bd)), coords=coords, name=self.name)
    if type(other).__module__.startswith("pandas") and hasattr(other, "to_numpy"):
      other = other.to_numpy()          # pd.Timedelta / pd.Timestamp -> numpy scalar
    return self._new(Variable(self.dims, op(self.data, other)))

  __add__ = lambda s, o: s._binary(o, lambda a, b: a + b)
  __sub__ = lambda s, o: s._binary(o, lambda a, b: a - b)
  __mul__ = lambda s, o: s._binary(o, lambda a, b: a * b)
  __truediv__ = lambda s, o: s._binary(o, lambda a, b: a / b)
  __radd__ = __add__
  __rmul__ = __mul__


# ----------------------------------------------------------------------------- Dataset
class Dataset(collections.abc.Mapping):
  """name -> DataArray with shared coordinates (xarray.Dataset subset)."""

  def __init__(self, data_vars=None, coords=None):
    self._vars: Dict[str, Variable] = {}
    self._names: Dict[str, Any] = {}
    self._coords: Dict[str, Variable] = {}
    for k, v in (coords or {}).items():
      self._coords[k] = _as_variable(v, default_dim=k)
    for k, v in (data_vars or {}).items():
      if isinstance(v, DataArray):
        for ck, cv in v._coords.items():
          if ck in self._coords and not self._coords[ck].equals(cv):
            raise ValueError(f"conflicting values for coordinate {ck!r}")
          self._coords.setdefault(ck, cv)
        self._names[k] = v.name
      self._vars[k] = _as_variable(v)
    self._check_sizes()

  def _check_sizes(self):
    sizes = {}
    for k, v in list(self._coords.items()) + list(self._vars.items()):
      for d, n in v.sizes.items():
        if sizes.setdefault(d, n) != n:
          raise ValueError(f"conflicting sizes for dimension {d!r}: {sizes[d]} vs {n} (on {k!r})")
    return sizes

  @classmethod
  def _construct(cls, variables, coords):
    ds = cls.__new__(cls)
    ds._vars, ds._coords, ds._names = dict(variables), dict(coords), {}
    ds._check_sizes()
    return ds

  # mapping protocol over data variables
  def __getitem__(self, key):
    if isinstance(key, str):
      if key in self._vars:
        v = self._vars[key]
        return DataArray(v, coords=_coords_for(v.dims, self._coords), name=key)
      if key in self._coords:
        return self.coords[key]
      raise KeyError(key)
    keys = list(key)
    missing = [k for k in keys if k not in self._vars]
    if missing:
      raise KeyError(missing[0])
    dims = {d for k in keys for d in self._vars[k].dims}
    coords = {k: c for k, c in self._coords.items() if set(c.dims) <= dims or not keys}
    return Dataset._construct({k: self._vars[k] for k in keys}, coords)

  def __iter__(self):
    return iter(self._vars)

  def __len__(self):
    return len(self._vars)

  def __contains__(self, k):
    return k in self._vars or k in self._coords

  def __getattr__(self, name):
    d = self.__dict__
    if name in d.get("_vars", {}) or name in d.get("_coords", {}):
      return self[name]
    raise AttributeError(name)

  def __repr__(self):
    return f"<xarray_lite.Dataset sizes={self.sizes} vars={list(self._vars)}>"

  @property
  def data_vars(self):
    return {k: self[k] for k in self._vars}

  @property
  def variables(self) -> Dict[str, Variable]:
    out = dict(self._coords)
    out.update(self._vars)
    return out

  @property
  def coords(self):
    return _Coords(self._coords)

  @property
  def sizes(self) -> Dict[str, int]:
    return self._check_sizes()

  dims = sizes

  def copy(self, deep=False):
    return Dataset._construct({k: v.copy(deep) for k, v in self._vars.items()} if deep else self._vars,
                              self._coords)

  def compute(self):
    return self

  def isel(self, indexers=None, drop=False, **kw):
    """Positional selection; `drop=True` removes the coordinates an integer index turns into
    scalars (xarray's meaning), instead of keeping them as 0-d coordinates."""
    indexers = dict(indexers or {}, **kw)
    bad = set(indexers) - set(self.sizes)
    if bad:
      raise ValueError(f"Dimensions {bad} do not exist. Expected one or more of {tuple(self.sizes)}")
    pick = lambda v: v.isel({d: i for d, i in indexers.items() if d in v.dims})
    scalar_dims = {d for d, i in indexers.items() if isinstance(i, (int, np.integer))}
    coords = {k: pick(v) for k, v in self._coords.items()
              if not (drop and k in scalar_dims and v.dims == (k,))}
    return Dataset._construct({k: pick(v) for k, v in self._vars.items()}, coords)

  def sel(self, indexers=None, **kw):
    """Label-based selection along dimension coordinates (lists, inclusive slices, scalars)."""
    indexers = dict(indexers or {}, **kw)
    for d in indexers:
      if d not in self._coords:
        raise KeyError(f"no coordinate for dimension {d!r}")
    return self.isel({d: _positions(self._coords[d], d, k) for d, k in indexers.items()})

  def squeeze(self, dim=None):
    dims = [dim] if isinstance(dim, str) else list(dim if dim is not None else
                                                    [d for d, n in self.sizes.items() if n == 1])
    for d in dims:
      if self.sizes[d] != 1:
        raise ValueError("cannot select a dimension to squeeze out which has length greater than one")
    return self.isel({d: 0 for d in dims})

  def update(self, other):
    """In place: adds / replaces variables (name -> Variable | DataArray | (dims, data))."""
    new = self.assign(other)
    self._vars, self._coords = new._vars, new._coords
    return self

  def tail(self, indexers=None, **kw):
    indexers = dict(indexers or {}, **kw)
    return self.isel({d: slice(max(self.sizes[d] - int(n), 0), None) for d, n in indexers.items()})

  def head(self, indexers=None, **kw):
    indexers = dict(indexers or {}, **kw)
    return self.isel({d: slice(0, int(n)) for d, n in indexers.items()})

  def assign_coords(self, coords=None, **kw):
    new = dict(self._coords)
    for k, v in dict(coords or {}, **kw).items():
      new[k] = _as_variable(v, default_dim=k)
    return Dataset._construct(self._vars, new)

  def assign(self, variables=None, **kw):
    new_vars, new_coords = dict(self._vars), dict(self._coords)
    items = dict(variables or {}, **kw)
    if isinstance(variables, Dataset):
      for ck, cv in variables._coords.items():
        new_coords.setdefault(ck, cv)
    for k, v in items.items():
      if isinstance(v, DataArray):
        for ck, cv in v._coords.items():
          new_coords.setdefault(ck, cv)
      new_vars[k] = _as_variable(v)
    return Dataset._construct(new_vars, new_coords)

  def drop_vars(self, names):
    names = [names] if isinstance(names, str) else list(names)
    return Dataset._construct({k: v for k, v in self._vars.items() if k not in names},
                              {k: v for k, v in self._coords.items() if k not in names})

  def astype(self, dtype):
    return Dataset._construct({k: v.astype(dtype) for k, v in self._vars.items()}, self._coords)

  def map(self, fn):
    return Dataset({k: fn(self[k]) for k in self._vars}, coords=self._coords)


# ----------------------------------------------------------------------------- concat
def concat(objs: Sequence, dim: str, data_vars: str = "all", compat: str = "equals", **_):
  """xarray.concat along an existing (or new) dimension for DataArrays / Datasets."""
  objs = list(objs)
  if not objs:
    raise ValueError("must supply at least one object to concatenate")
  if all(isinstance(o, DataArray) for o in objs):
    var = Variable.concat([o.variable for o in objs], dim)
    coords = {}
    for k, c in objs[0]._coords.items():
      if dim in c.dims:
        if all(k in o._coords for o in objs):
          coords[k] = Variable.concat([o._coords[k] for o in objs], dim)
      else:
        coords[k] = c
    return DataArray(var, coords=coords, name=objs[0].name)
  if not all(isinstance(o, Dataset) for o in objs):
    raise TypeError("concat: objects must all be DataArrays or all be Datasets")
  if data_vars not in ("all", "different", "minimal"):
    raise ValueError(f"unexpected value for data_vars: {data_vars}")
  names = []
  for o in objs:
    names += [k for k in o._vars if k not in names]
  new_vars = {}
  for k in names:
    have = [o._vars[k] for o in objs if k in o._vars]
    with_dim = [dim in v.dims for v in have]
    differs = (data_vars == "different" and len(have) == len(objs)
               and any(not have[0].equals(v) for v in have[1:]))
    if any(with_dim) or data_vars == "all" or differs:
      if len(have) != len(objs):
        raise ValueError(f"{k!r} is not present in all datasets.")
      new_vars[k] = Variable.concat(have, dim)
    else:
      if compat == "equals" and any(not have[0].equals(v) for v in have[1:]):
        raise ValueError(f"variable {k!r} is not equal across datasets")
      new_vars[k] = have[0]
  new_coords = {}
  cnames = []
  for o in objs:
    cnames += [k for k in o._coords if k not in cnames]
  for k in cnames:
    have = [o._coords[k] for o in objs if k in o._coords]
    if dim in have[0].dims:
      if len(have) == len(objs):
        new_coords[k] = Variable.concat(have, dim)
    else:
      new_coords[k] = have[0]
  return Dataset._construct(new_vars, new_coords)


def merge(objects: Sequence, compat: str = "no_conflicts", join: str = "outer", **_):
  """xarray.merge for named DataArrays / Datasets whose shared coordinates are identical
  (``join="exact"``: anything else raises, like xarray)."""
  data_vars, coords = {}, {}
  for o in objects:
    items = ({o.name: o} if isinstance(o, DataArray) else {k: o[k] for k in o.keys()})
    if isinstance(o, DataArray) and o.name is None:
      raise ValueError("cannot merge an unnamed DataArray")
    for k, da in items.items():
      for ck, cv in da._coords.items():
        is_index = cv.dims == (ck,)          # a dimension coordinate: what `join` aligns on
        if ck in coords and is_index and not coords[ck].equals(cv):
          raise ValueError(f"cannot align objects with join='exact' where index/labels/sizes are "
                           f"not equal along dimension {ck!r}")
        if ck in coords and not is_index and compat != "override" and not coords[ck].equals(cv):
          raise ValueError(f"conflicting values for coordinate {ck!r}")
        if ck in coords and coords[ck].shape != cv.shape:
          raise ValueError(f"cannot align objects with join='exact' along {ck!r}")
        coords.setdefault(ck, cv)
      data_vars[k] = da.variable
  return Dataset._construct(data_vars, coords)


def zeros_like(obj, dtype=None):
  if isinstance(obj, Dataset):
    return obj.map(lambda v: zeros_like(v, dtype))
  data = obj.data
  if _is_torch(data):
    import torch
    z = torch.zeros_like(data)
  else:
    z = np.zeros_like(data, dtype=dtype)
  return DataArray(Variable(obj.dims, z), coords=getattr(obj, "_coords", None), name=getattr(obj, "name", None))


# ----------------------------------------------------------------------------- real-xarray boundary
def is_host(dataset) -> bool:
  """True for a Dataset none of whose data variables is a torch tensor (numpy-backed: host memory)."""
  return isinstance(dataset, Dataset) and not any(_is_torch(v.data) for v in dataset._vars.values())


def to_device(dataset, device):
  """Dataset with every data variable on ``device`` as a torch tensor (coordinates stay numpy): one H2D copy per
  variable straight from the caller's arrays -- the HIP runtime moves pageable memory at the link's rate on the
  MI355X host (50-56 GB/s, scripts/probes/pcie_probe.py; staging through pinned pages first measured slower)."""
  import torch
  def put(v):
    data = v.data if _is_torch(v.data) else torch.from_numpy(np.ascontiguousarray(v.data))
    # (non_blocking only from PINNED pages: a pageable source may be a temporary -- `ascontiguousarray` of a view --
    #  that is dropped as soon as this returns; the measured gain is the per-variable copy, not asynchrony)
    return Variable(v.dims, data.to(device, non_blocking=bool(data.device.type == "cpu" and data.is_pinned())))
  return Dataset._construct({k: put(v) for k, v in dataset._vars.items()}, dict(dataset._coords))


def to_host(dataset):
  """``jax.device_get`` of the reference (rollout.py:362): torch-backed variables and coordinates -> numpy."""
  def get(v):
    data = v.data
    return Variable(v.dims, data.detach().cpu().numpy() if _is_torch(data) else data)
  return Dataset._construct({k: get(v) for k, v in dataset._vars.items()},
                            {k: get(v) for k, v in dataset._coords.items()})


def is_lite(obj) -> bool:
  return isinstance(obj, (Dataset, DataArray, Variable))


def from_xarray(obj):
  """A real ``xarray.Dataset`` / ``xarray.DataArray`` (or anything that quacks like one: ``.data_vars`` /
  ``.coords`` mappings whose values expose ``.dims`` and ``.values`` / ``.data``) -> the xarray_lite
  equivalent; xarray_lite objects and None pass through.  The Predictor boundary (``GraphCast.__call__``,
  ``rollout.chunked_prediction*``, ``normalization.InputsAndResiduals``) calls this on what it is handed, so a
  host that HAS xarray can pass its own Datasets (reference ``utils/predictor_base.py:43-84``); xarray itself is
  never imported here (it is not installable in this image -- the test uses a duck-typed stand-in).
  Data stay where they are: numpy arrays are not copied, torch tensors stay on their device; lazily loaded
  (dask) variables are materialised through ``.values``."""
  if obj is None or is_lite(obj):
    return obj

  def var(v):
    data = getattr(v, "data", None)
    if data is None or not (isinstance(data, np.ndarray) or _is_torch(data)):
      data = np.asarray(v.values)
    return Variable(tuple(v.dims), data)

  if hasattr(obj, "data_vars"):
    return Dataset._construct({str(k): var(v) for k, v in obj.data_vars.items()},
                              {str(k): var(v) for k, v in obj.coords.items()})
  if hasattr(obj, "dims") and hasattr(obj, "coords"):
    arr = DataArray(var(obj).data, dims=tuple(obj.dims), name=getattr(obj, "name", None))
    return arr.assign_coords({str(k): var(v) for k, v in obj.coords.items()})
  raise TypeError(f"cannot adapt {type(obj).__name__}: expected an xarray / xarray_lite Dataset or DataArray")


def to_xarray(ds, xarray_module=None):
  """The way back for a host that has xarray: ``to_xarray(predictions, xarray)`` builds ``xarray.Dataset`` (or
  ``DataArray``) objects from xarray_lite ones through the public constructors ``Dataset(data_vars, coords)`` /
  ``DataArray(data, coords, dims, name)`` with ``{name: (dims, array)}`` entries -- torch-backed variables are
  copied to host numpy first.  Without a module, returns those constructor arguments."""
  def arr(v):
    data = v.data
    return data.detach().cpu().numpy() if _is_torch(data) else np.asarray(data)

  if isinstance(ds, Dataset):
    data_vars = {k: (tuple(v.dims), arr(v)) for k, v in ds._vars.items()}
    coords = {k: (tuple(v.dims), arr(v)) for k, v in ds._coords.items()}
    return xarray_module.Dataset(data_vars, coords=coords) if xarray_module is not None else (data_vars, coords)
  if isinstance(ds, DataArray):
    coords = {k: (tuple(v.variable.dims), arr(v.variable)) for k, v in ds.coords.items()}
    args = dict(data=arr(ds.variable), coords=coords, dims=tuple(ds.dims), name=ds.name)
    return xarray_module.DataArray(**args) if xarray_module is not None else args
  raise TypeError(f"expected an xarray_lite Dataset / DataArray, got {type(ds).__name__}")
