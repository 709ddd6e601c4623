"""Typed-graph containers with the names, field order and lookup behaviour of the reference's
``weathernext/utils/typed_graph.py:45-97``: plain tuples (``_replace`` / unpacking / equality as
there), built here with ``collections.namedtuple``."""
import collections

NodeSet = collections.namedtuple("NodeSet", "n_node features")
EdgesIndices = collections.namedtuple("EdgesIndices", "senders receivers")
EdgeSet = collections.namedtuple("EdgeSet", "n_edge indices features")             # indices: EdgesIndices
Context = collections.namedtuple("Context", "n_graph features")
EdgeSetKey = collections.namedtuple("EdgeSetKey", "name node_sets")                # node_sets: (sender set, receiver set)


class TypedGraph(collections.namedtuple("TypedGraph", "context nodes edges")):
  """context: Context; nodes: {name: NodeSet}; edges: {EdgeSetKey: EdgeSet}."""
  __slots__ = ()

  def edge_key_by_name(self, name):
    keys = list(self.edges)
    hits = [key for key in keys if key.name == name]
    if len(hits) == 1:
      return hits[0]
    # (same text as the reference's KeyError, typed_graph.py:87-90)
    raise KeyError("invalid edge key '{}'. Available edges: [{}]".format(name, ", ".join(key.name for key in keys)))

  def edge_by_name(self, name):
    return self.edges[self.edge_key_by_name(name)]
