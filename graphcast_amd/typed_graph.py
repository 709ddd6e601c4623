"""Typed-graph containers, same names/fields as the reference's
``weathernext/utils/typed_graph.py:45-97`` (NodeSet, EdgesIndices, EdgeSet,
Context, EdgeSetKey, TypedGraph incl. ``edge_key_by_name`` / ``edge_by_name``
and their KeyError behaviour)."""
from typing import Any, Mapping, NamedTuple, Tuple


class NodeSet(NamedTuple):
  n_node: Any
  features: Any


class EdgesIndices(NamedTuple):
  senders: Any
  receivers: Any


class EdgeSet(NamedTuple):
  n_edge: Any
  indices: EdgesIndices
  features: Any


class Context(NamedTuple):
  n_graph: Any
  features: Any


class EdgeSetKey(NamedTuple):
  name: str
  node_sets: Tuple[str, str]   # (sender node set, receiver node set)


class TypedGraph(NamedTuple):
  context: Context
  nodes: Mapping[str, NodeSet]
  edges: Mapping[EdgeSetKey, EdgeSet]

  def edge_key_by_name(self, name: str) -> EdgeSetKey:
    found = [k for k in self.edges.keys() if k.name == name]
    if len(found) != 1:
      raise KeyError("invalid edge key '{}'. Available edges: [{}]".format(
          name, ", ".join(k.name for k in self.edges.keys())))
    return found[0]

  def edge_by_name(self, name: str) -> EdgeSet:
    return self.edges[self.edge_key_by_name(name)]
