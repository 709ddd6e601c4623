"""`map_structure` over containers of DataArrays that keeps Datasets Datasets.

Mirror of the reference's ``weathernext/utils/xarray_tree.py`` (used by its normalisation and
loss code): a Dataset is traversed as a mapping name -> DataArray, and the mapped results are
merged back into a Dataset when that is possible -- every result a DataArray (``None`` results
are dropped) with exactly matching dimension coordinates -- and returned as a plain dict
otherwise.  dicts / lists / tuples / sets are traversed recursively; anything else is a leaf.
"""
from typing import Any, Callable

from graphcast_amd import xarray_lite as xarray


def map_structure(func: Callable[..., Any], *structures: Any) -> Any:
  """Applies `func` leaf-wise through parallel structures (reference xarray_tree.py:46-69)."""
  if not callable(func):
    raise TypeError(f"func must be callable, got: {func}")
  if not structures:
    raise ValueError("Must provide at least one structure")
  head = structures[0]
  if isinstance(head, xarray.Dataset):
    mapped = {name: func(*[s[name] for s in structures]) for name in head.keys()}
    if all(v is None or isinstance(v, xarray.DataArray) for v in mapped.values()):
      named = [v.rename(name) for name, v in mapped.items() if v is not None]
      try:
        return xarray.merge(named, join="exact", compat="override")
      except ValueError:            # dimension coordinates differ: not one Dataset any more
        pass
    return mapped
  if isinstance(head, dict):
    return {k: map_structure(func, *[s[k] for s in structures]) for k in head.keys()}
  if isinstance(head, (list, tuple, set)):
    return type(head)(map_structure(func, *group) for group in zip(*structures))
  return func(*structures)
