"""jraph stand-in: the two functions the GraphCast path calls + type aliases.

Published algorithms (jraph/_src/utils.py, jraph/_src/models.py):
  segment_sum(data, segment_ids, num_segments): out[i] = sum_{e: ids[e]==i} data[e]
  concatenated_args(update)(*args, **kwargs) =
      update(concatenate(tree_leaves(args) + tree_leaves(kwargs), axis=-1))
"""
from typing import Any, Callable

import numpy as np

import jax.tree_util as _tree

ArrayTree = Any
GraphsTuple = Any
NodeFeatures = EdgeFeatures = Globals = SenderFeatures = ReceiverFeatures = Any
AggregateEdgesToNodesFn = AggregateNodesToGlobalsFn = AggregateEdgesToGlobalsFn = Callable
GNUpdateEdgeFn = GNUpdateNodeFn = GNUpdateGlobalFn = InteractionUpdateEdgeFn = Callable
InteractionUpdateNodeFn = EmbedEdgeFn = EmbedNodeFn = EmbedGlobalFn = Callable


def segment_sum(data, segment_ids, num_segments=None, indices_are_sorted=False,
                unique_indices=False):
  segment_ids = np.asarray(segment_ids)
  if num_segments is None:
    num_segments = int(segment_ids.max()) + 1
  out = np.zeros((num_segments,) + data.shape[1:], dtype=data.dtype)
  np.add.at(out, segment_ids, data)
  return out


def concatenated_args(update=None, *, axis=-1):
  def _curry(update):
    def wrapper(*args, **kwargs):
      combined = _tree.tree_flatten(args)[0] + _tree.tree_flatten(kwargs)[0]
      return update(np.concatenate(combined, axis=axis))
    return wrapper
  return _curry if update is None else _curry(update)
