"""Oracle: typed-graph network arithmetic (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates, in numpy and for any float dtype:
  * third-party primitives the reference calls (source NOT under /root/reference;
    dm-haiku / jraph / jax are un-pinned, ``setup.py:37,42,43`` -- published
    semantics restated, **parity unpinned**):
      ``linear``      hk.Linear: ``x @ w + b``, ``w`` stored [in, out]
      ``mlp``         hk.nets.MLP(activate_final=False) (call site
                      ``deep_typed_graph_net.py:205-209``)
      ``layer_norm``  hk.LayerNorm(axis=-1, create_scale, create_offset), eps=1e-5,
                      biased variance (call site ``:231-233``)
      ``swish``       jax.nn.swish = x * sigmoid(x) (``:444-449``)
      ``segment_sum`` jraph.segment_sum (``:455-458``)
      concatenated_args: concat of the positional args' leaves on the last axis
  * ``DeepTypedGraphNet`` forward (``deep_typed_graph_net.py:180-401``) on a plain
    dict graph: embed (GraphMapFeatures, ``typed_graph_net.py:657-696``), N
    InteractionNetwork steps (``typed_graph_net.py:272-350,369-546,590-654``:
    all edge sets first, then all node sets using the UPDATED edges, sent
    messages dropped) with residuals on every node and edge set
    (``deep_typed_graph_net.py:372-393``), optional decoder MLPs without
    LayerNorm (``:313-322``).

Graph container (all features are [rows, batch, channels] like the reference):
  graph = {"nodes": {name: features},
           "edges": {name: dict(senders_set, receivers_set, senders, receivers,
                                features)}}
"""
import numpy as np
import scipy.sparse

LN_EPS = 1e-5   # haiku default; no in-tree override (dense.py:182-188)


def swish(x):
  with np.errstate(over="ignore"):      # exp(-x) -> inf gives x / inf = -0.0, the right limit
    if ACTIVATIONS == "bf16":           # jax.nn.swish = x * sigmoid(x): lax.logistic, then lax.mul
      return _bf16(x * _bf16(1.0 / (1.0 + np.exp(-x))))
    return x / (1.0 + np.exp(-x))


# None, or "bf16": round BOTH operands of every Linear product to bfloat16 (nearest even) before
# multiplying, accumulate in the working dtype -- the arithmetic of the GC_PREC_BF16 tier (the
# GEMM-operand part of the reference's casting.Bfloat16Cast, utils/casting.py:31-65,155-205).
GEMM_OPERANDS = None


class gemm_operands:
  """Context manager: ``with gnn.gemm_operands("bf16"): ...``"""

  def __init__(self, mode):
    self.mode = mode

  def __enter__(self):
    global GEMM_OPERANDS
    self.prev, GEMM_OPERANDS = GEMM_OPERANDS, self.mode

  def __exit__(self, *exc):
    global GEMM_OPERANDS
    GEMM_OPERANDS = self.prev


def _bf16(a):
  u = np.ascontiguousarray(np.asarray(a, dtype=np.float32)).view(np.uint32)
  r = ((u + (((u >> 16) & 1) + np.uint32(0x7FFF))) >> 16) << 16
  return r.view(np.float32).astype(np.asarray(a).dtype)


# None, or "bf16": the reference's ``casting.Bfloat16Cast`` run (utils/casting.py:31-65,155-205) --
# inputs, parameters (fp32-stored, read through ``bfloat16_variable_view``) and EVERY array the
# traced program materialises are bfloat16.  Restated op by op: the result of every jnp / lax
# operation the reference's Python emits (dot, bias add, logistic, multiply, the six element-wise
# ops of hk.LayerNorm, residual add, segment_sum) is rounded to bfloat16 (nearest even); dots and
# the reductions of jnp.mean / jnp.var accumulate in float32 and round once (jax's ``_upcast_f16``
# computation dtype).  **Parity unpinned**: XLA may fuse element-wise chains and keep float32
# intermediates inside a fusion, which this op-by-op restatement cannot know; segment_sum in
# bfloat16 is order dependent in XLA -- here it is a float32 sum rounded once.  Containers stay
# float32 (numpy has no bfloat16).
ACTIVATIONS = None


class activations:
  """Context manager: ``with gnn.activations("bf16"): ...`` (run the oracle with dtype=np.float32)."""

  def __init__(self, mode):
    self.mode = mode

  def __enter__(self):
    global ACTIVATIONS
    self.prev, ACTIVATIONS = ACTIVATIONS, self.mode

  def __exit__(self, *exc):
    global ACTIVATIONS
    ACTIVATIONS = self.prev


def act(x):
  """Rounds a freshly produced array to the activation dtype (identity unless ACTIVATIONS == "bf16")."""
  return _bf16(x) if ACTIVATIONS == "bf16" else x


def linear(x, w, b):
  if GEMM_OPERANDS == "bf16" or ACTIVATIONS == "bf16":
    x, w = _bf16(x), _bf16(w)
  # 2-D GEMM: numpy would treat [rows, batch, k] @ [k, n] as `rows` tiny [batch, k] products
  y = (x.reshape(-1, x.shape[-1]) @ w).reshape(x.shape[:-1] + (w.shape[1],))
  if ACTIVATIONS == "bf16":
    return _bf16(_bf16(y) + _bf16(b))
  return y + b


def mlp(x, layers):
  """layers = [(w0, b0), (w1, b1), ...]; swish between, none after the last."""
  for i, (w, b) in enumerate(layers):
    x = linear(x, w, b)
    if i < len(layers) - 1:
      x = swish(x)
  return x


def layer_norm(x, scale, offset, eps=LN_EPS):
  mean = x.mean(axis=-1, keepdims=True)
  var = np.square(x - mean).mean(axis=-1, keepdims=True)
  if ACTIVATIONS == "bf16":
    # hk.LayerNorm.__call__ as its Python emits it: mean = jnp.mean(x), variance = jnp.var(x) (both
    # reduced in float32 from the up-cast input, each rounded once), eps cast to the variance's dtype,
    # inv = scale * lax.rsqrt(variance + eps); return inv * (x - mean) + offset
    r = _bf16
    inv = r(r(scale) * r(1.0 / np.sqrt(r(r(var) + r(np.float32(eps))))))
    return r(r(inv * r(x - r(mean))) + r(offset))
  return (x - mean) / np.sqrt(var + eps) * scale + offset


def segment_sum(data, segment_ids, num_segments):
  """out[i] = sum_{e: ids[e] == i} data[e]; zeros for empty segments."""
  flat = data.reshape(data.shape[0], -1)
  sel = scipy.sparse.csr_matrix(
      (np.ones(len(segment_ids), dtype=data.dtype),
       (np.asarray(segment_ids), np.arange(len(segment_ids)))),
      shape=(num_segments, len(segment_ids)))
  return np.asarray(sel @ flat).reshape((num_segments,) + data.shape[1:])


class Net:
  """MLP(+LayerNorm) addressed by its reference module name.

  ``norm_conditioning`` [batch, C_cond] switches every LayerNorm to the reference's conditional
  form (deep_typed_graph_net.py:210-246 + dense.py:360-393, the GenCast encoder / decoder):
  LayerNorm WITHOUT learned scale / offset, then ``x * (1 + s) + o`` with
  ``[s | o] = norm_conditioning @ w + b`` of the module ``<name>_norm_conditioning/linear``,
  broadcast over the node / edge axis (``global_norm_conditioning[None]``)."""

  def __init__(self, params, gnn_name, dtype, norm_conditioning=None):
    self._p, self._g, self._dt = params, gnn_name, dtype
    self._cond = None if norm_conditioning is None else np.asarray(norm_conditioning, dtype=dtype)

  def _get(self, module, leaf):
    key = f"{self._g}/~_networks_builder/{module}"
    return act(np.asarray(self._p[key][leaf], dtype=self._dt))      # (bf16 run: the bfloat16 view of the fp32 parameter)

  def has(self, name):
    return f"{self._g}/~_networks_builder/{name}_mlp/~/linear_0" in self._p

  def apply(self, name, *args, use_layer_norm=True):
    x = np.concatenate(args, axis=-1) if len(args) > 1 else args[0]
    layers, k = [], 0
    while f"{self._g}/~_networks_builder/{name}_mlp/~/linear_{k}" in self._p:
      layers.append((self._get(f"{name}_mlp/~/linear_{k}", "w"),
                     self._get(f"{name}_mlp/~/linear_{k}", "b")))
      k += 1
    y = mlp(x, layers)
    if use_layer_norm and self._cond is not None:
      y = layer_norm(y, 1.0, 0.0)
      so = self._cond @ self._get(f"{name}_norm_conditioning/linear", "w") \
          + self._get(f"{name}_norm_conditioning/linear", "b")                  # [batch, 2 C]
      c = y.shape[-1]
      y = y * (so[..., :c] + 1.0)[None] + so[..., c:][None]
    elif use_layer_norm:
      y = layer_norm(y, self._get(f"{name}_layer_norm", "scale"),
                     self._get(f"{name}_layer_norm", "offset"))
    return y


def _edge_update(net, name, edge, nodes, chunk):
  """e' = f([e | h_send[senders] | h_recv[receivers]]), row-chunked."""
  e = edge["features"]
  hs, hr = nodes[edge["senders_set"]], nodes[edge["receivers_set"]]
  pieces = []
  for lo in range(0, e.shape[0], chunk):
    hi = min(lo + chunk, e.shape[0])
    pieces.append(net.apply(name, e[lo:hi], hs[edge["senders"][lo:hi]],
                            hr[edge["receivers"][lo:hi]]))
  return np.concatenate(pieces, axis=0)


def deep_typed_graph_net(params, gnn_name, graph, *, num_steps, embed_nodes,
                         embed_edges, node_output=(), dtype=np.float64,
                         chunk=1 << 16, live_nodes=None, live_edges=None,
                         f32_aggregation=False, norm_conditioning=None):
  """Returns {"nodes": {...}, "edges": {...}} of output features.

  ``live_nodes`` / ``live_edges`` optionally restrict which outputs are computed
  on the LAST step (the reference computes everything; GraphCast only reads
  some of it -- ``graphcast.py:602-603,639,676``).  Results that are computed
  are identical either way.

  ``f32_aggregation`` restates ``deep_typed_graph_net.py:273-281``: the edge messages are
  cast to float32 around the segment-sum and the result cast back (an up-cast for the
  reference's bf16 activations, a no-op in fp32, a DOWN-cast when the oracle runs in float64).

  ``norm_conditioning`` [batch, C_cond]: the reference's ``use_norm_conditioning=True`` /
  ``global_norm_conditioning`` (see ``Net``).
  """
  if ACTIVATIONS is not None and (norm_conditioning is not None or np.dtype(dtype) != np.float32):
    raise ValueError("the bf16-activation restatement runs unconditioned nets with dtype=np.float32")
  net = Net(params, gnn_name, dtype, norm_conditioning=norm_conditioning)
  nodes = {k: act(np.asarray(v, dtype=dtype)) for k, v in graph["nodes"].items()}
  edges = {k: dict(v, features=act(np.asarray(v["features"], dtype=dtype)))
           for k, v in graph["edges"].items()}

  # _embed (deep_typed_graph_net.py:325-353)
  if embed_edges:
    for k, e in edges.items():
      e["features"] = net.apply(f"encoder_edges_{k}", e["features"])
  if embed_nodes:
    for k in nodes:
      nodes[k] = net.apply(f"encoder_nodes_{k}", nodes[k])

  # _process (:355-393)
  for step in range(num_steps):
    last = step == num_steps - 1
    new_edges = {k: _edge_update(net, f"processor_edges_{step}_{k}", e, nodes, chunk)
                 for k, e in edges.items()}
    new_nodes = {}
    for k, h in nodes.items():
      if last and live_nodes is not None and k not in live_nodes:
        continue
      agg_in = (lambda a: a.astype(np.float32)) if f32_aggregation else (lambda a: a)
      received = [act(segment_sum(agg_in(new_edges[ek]), e["receivers"], h.shape[0]).astype(dtype))
                  for ek, e in sorted(edges.items()) if e["receivers_set"] == k]
      new_nodes[k] = net.apply(f"processor_nodes_{step}_{k}", h, *received)
    for k in list(nodes):
      if k in new_nodes:
        nodes[k] = act(nodes[k] + new_nodes[k])
      else:
        del nodes[k]
    for k, e in edges.items():
      if last and live_edges is not None and k not in live_edges:
        e["features"] = None
      else:
        e["features"] = act(e["features"] + new_edges[k])

  # _output (:395-401)
  for k in node_output:
    nodes[k] = net.apply(f"decoder_nodes_{k}", nodes[k], use_layer_norm=False)
  return {"nodes": nodes, "edges": {k: e["features"] for k, e in edges.items()}}
