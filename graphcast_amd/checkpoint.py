"""Checkpoint (de)serialisation in the reference's ``.npz`` layout
(``weathernext/utils/checkpoint.py:25-170``): a tree of dicts / dataclasses /
lists / tuples is flattened to ``"outer:inner:leaf"`` keys of one ``np.savez``
archive; ``load`` rebuilds it against a type used as schema.  A published
GraphCast checkpoint (``graphcast.CheckPoint``: ``params`` keyed by haiku module
path then ``w|b|scale|offset``, ``model_config``, ``task_config``,
``description``, ``license``) therefore loads with

    with open(path, "rb") as f:
      ckpt = checkpoint.load(f, graphcast.CheckPoint)
    model = graphcast.GraphCast(ckpt.model_config, ckpt.task_config, params=ckpt.params)

Written as two explicit tree walks (encode / decode against a schema).
"""
import dataclasses
import io
import types
import typing
from typing import Any, BinaryIO, Dict

import numpy as np

SEPARATOR = ":"


# ----------------------------------------------------------------------------- encode
def _children(node):
  """Container -> ordered (key, child) pairs; None fields of dataclasses are dropped."""
  if dataclasses.is_dataclass(node) and not isinstance(node, type):
    return [(f.name, getattr(node, f.name)) for f in dataclasses.fields(node)
            if getattr(node, f.name) is not None]
  if isinstance(node, dict):
    return list(node.items())
  if isinstance(node, (list, tuple)):
    return list(enumerate(node))
  return None


def flatten(tree) -> Dict[str, Any]:
  """{"a:b:c": leaf} for every leaf of the tree."""
  if _children(tree) is None:
    raise TypeError("the root of a checkpoint must be a dict, dataclass, list or tuple")
  flat = {}
  stack = [((), tree)]
  while stack:
    path, node = stack.pop()
    kids = _children(node)
    if kids is None:
      if node is None:
        raise ValueError(f"None leaf at {SEPARATOR.join(path)!r} (only dataclass fields may be None)")
      flat[SEPARATOR.join(path)] = node
      continue
    for k, child in kids:
      k = str(k)
      if SEPARATOR in k:
        raise ValueError(f"key {k!r} contains the separator {SEPARATOR!r}")
      stack.append((path + (k,), child))
  return flat


def dump(dest: BinaryIO, value: Any) -> None:
  """Writes ``value`` to a binary file object (seek not required)."""
  buffer = io.BytesIO()
  flat = flatten(value)
  np.savez(buffer, **{k: flat[k] for k in sorted(flat)})
  dest.write(buffer.getvalue())


# ----------------------------------------------------------------------------- decode
def unflatten(flat) -> Dict[str, Any]:
  tree: Dict[str, Any] = {}
  for key in flat:
    *parents, leaf = key.split(SEPARATOR)
    node = tree
    for p in parents:
      node = node.setdefault(p, {})
    node[leaf] = flat[key]
  return tree


def _by_index(d):
  return [v for _, v in sorted(d.items(), key=lambda kv: int(kv[0]))]


def _optional_inner(tp):
  """T for Optional[T] / T | None, else None."""
  if typing.get_origin(tp) in (typing.Union, types.UnionType):
    inner = [a for a in typing.get_args(tp) if a is not type(None)]
    if len(inner) != 1:
      raise TypeError("Optional works, Union with anything except None doesn't")
    return inner[0]
  return None


def convert(tp, value):
  """Coerces the plain tree ``value`` into the schema type ``tp``."""
  if tp is Any or tp is Ellipsis:
    return value
  if tp in (int, float, str, bool):
    return tp(value)
  if tp is np.ndarray:
    if not isinstance(value, np.ndarray):
      raise TypeError(f"expected an array, found {type(value)}")
    return value
  if dataclasses.is_dataclass(tp):
    hints = typing.get_type_hints(tp)
    kwargs = {}
    for f in dataclasses.fields(tp):
      ftype = hints.get(f.name, f.type)
      inner = _optional_inner(ftype)
      if inner is not None:
        kwargs[f.name] = convert(inner, value[f.name]) if f.name in value else None
      elif f.name in value:
        kwargs[f.name] = convert(ftype, value[f.name])
      else:
        raise ValueError(f"Missing value: {f.name}")
    return tp(**kwargs)
  origin, args = typing.get_origin(tp), typing.get_args(tp)
  if origin is dict:
    kt, vt = args
    return {convert(kt, k): convert(vt, v) for k, v in value.items()}
  if origin is list:
    return [convert(args[0], v) for v in _by_index(value)]
  if origin is tuple:
    items = _by_index(value)
    if len(args) == 2 and args[1] is Ellipsis:
      return tuple(convert(args[0], v) for v in items)
    if len(args) != len(items):
      raise ValueError(f"tuple schema has {len(args)} entries, data has {len(items)}")
    return tuple(convert(t, v) for t, v in zip(args, items))
  try:
    return tp(value)
  except TypeError as e:
    raise TypeError(f"cannot build {tp} from a checkpoint leaf; schema types must be dataclasses "
                    "or constructors taking one numpy value") from e


def load(source: BinaryIO, typ):
  """Reads an archive written by ``dump`` (or by the reference) as ``typ``."""
  with np.load(source) as archive:
    flat = {k: archive[k] for k in archive.files}
  return convert(typ, unflatten(flat))
