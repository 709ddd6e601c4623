"""jax.tree_util stand-in: pytrees of dict (sorted keys) / list / tuple / NamedTuple / None."""


def _is_namedtuple(x):
  return isinstance(x, tuple) and hasattr(x, "_fields")


def tree_flatten(tree):
  leaves = []

  def go(x):
    if x is None:
      return ("none",)
    if isinstance(x, dict):
      keys = sorted(x.keys())
      return ("dict", keys, [go(x[k]) for k in keys])
    if _is_namedtuple(x):
      return ("nt", type(x), [go(v) for v in x])
    if isinstance(x, (list, tuple)):
      return ("seq", type(x), [go(v) for v in x])
    leaves.append(x)
    return ("leaf",)

  return leaves, go(tree)


def tree_unflatten(treedef, leaves):
  it = iter(leaves)

  def go(d):
    kind = d[0]
    if kind == "none":
      return None
    if kind == "leaf":
      return next(it)
    if kind == "dict":
      return {k: go(c) for k, c in zip(d[1], d[2])}
    if kind == "nt":
      return d[1](*[go(c) for c in d[2]])
    return d[1](go(c) for c in d[2])

  return go(treedef)


def tree_leaves(tree):
  return tree_flatten(tree)[0]


def tree_structure(tree):
  return tree_flatten(tree)[1]


def tree_map(f, tree, *rest):
  leaves, treedef = tree_flatten(tree)
  others = [tree_flatten(r)[0] for r in rest]
  for o in others:
    if len(o) != len(leaves):
      raise ValueError("tree_map: trees do not match")
  return tree_unflatten(treedef, [f(*xs) for xs in zip(leaves, *others)])


flatten, unflatten, leaves, map, structure = (tree_flatten, tree_unflatten, tree_leaves, tree_map,
                                              tree_structure)
