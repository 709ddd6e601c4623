"""WN2's ``DeepGNN`` processor on the MI355X row-MLP kernels (second half).

Host-side mirror of ``weathernext/utils/deep_gnn.py:45-400``: same constructor arguments, same
``__call__(input_graph, global_norm_conditioning=None, is_training=None) -> TypedGraph``, same parameter tree

    "<name>/processor_edges_<i>_<edge set>/mlp/linear_{0,1}"          {"w", "b"}
    "<name>/processor_edges_<i>_<edge set>/normalization/layer_norm"  {"scale", "offset"}
    "<name>/processor_nodes_<i>_<node set>/..."                       likewise
    pre_gather_matmul=True:  linear_0 of an edge MLP holds the bias only and its matrix lives in
    "<name>/processor_edges_<i>_{edge,sender,receiver}_<edge set>" {"w"}   (deep_gnn.py:224-262)

(haiku names under ``hk.transparent`` / ``hk.name_like("__call__")``; pinned by the reference source executed
on the stand-ins, tests/golden/make_golden_deepgnn.py).  Parameters are passed to the constructor instead of
through ``hk.transform`` like everywhere in this package; features are float32 torch tensors on the device,
``[rows, batch, 512]``.

One message-passing step is the launches of the GraphCast processor (engine.py):
  (h_send . W_s), (h_recv . W_r) per node            GC_MODE_LINEAR             (``pre_gather_matmul`` is what the
                                                      kernels always do: x[idx].W == (x.W)[idx]; the concat form's
                                                      first matrix is cut into its [e | s | r] row blocks)
  e' = LN(MLP(e . W_e + gathered addends)) + e        GC_MODE_MLP_LN + receiver segment-sum (+ fix-up launches)
  h' = LN(MLP([h | agg])) + h                         GC_MODE_MLP_LN, K = 1024
Edges are re-packed receiver-sorted into 64-row tiles once per graph structure (cached on the index arrays).
The reference hands DeepGNN receiver-sorted padded edge sets already (``utils/padding_utils.py``); any order is
accepted here, padding edges with receiver index >= n_nodes are not.

What is built: ``dense.DenseLayer`` with ONE hidden layer, hidden = output = 512, "swish", "layer_norm",
biases, no norm conditioning (the configuration the kernels are specialised for); one edge set; the node set
that receives it and, optionally, a separate sender node set; ``use_edge_residuals`` on or off;
``num_processor_repetitions``; ``aggregate_normalization`` (folded into the node MLP's aggregate block).
Anything else raises NotImplementedError loudly; rematerialisation / sharding arguments are accepted and
ignored (inference, one device)."""
from typing import Any, Mapping, Optional

import numpy as np
import torch

from graphcast_amd import _native as nat
from graphcast_amd import launch
from graphcast_amd import packing
from graphcast_amd import typed_graph

D = packing.LATENT


def _check_dense(kw, what):
  want = dict(hidden_size=D, output_size=D, num_hidden_layers=1, activation="swish",
              activation_normalization="layer_norm")
  for k, v in want.items():
    if kw.get(k) != v:
      raise NotImplementedError(f"DeepGNN on MI355X: {what}[{k!r}] must be {v!r} (got {kw.get(k)!r}): the row-MLP "
                                "kernels fuse exactly one hidden layer of 512 + LayerNorm")
  if kw.get("activate_final") or not kw.get("with_bias", True) or kw.get("stack") or \
     (kw.get("activation_normalization_kwargs") or {}).get("use_norm_conditioning"):
    raise NotImplementedError(f"DeepGNN on MI355X: {what}: activate_final / no bias / stacked layers / norm "
                              "conditioning are not built")


class DeepGNN(launch.LaunchBase):
  """See the module docstring."""

  def __init__(self, *, dense_kwargs: Mapping[str, Any], num_message_passing_steps: int,
               num_processor_repetitions: int = 1, edge_dense_kwargs: Optional[Mapping[str, Any]] = None,
               use_edge_residuals: bool = True, gather_from_receivers: bool = True,
               edge_update_remat_block_size: Optional[int] = None, remat_block_size: Optional[int] = None,
               f32_aggregation: bool = False, gather_scatter_kwargs=None,
               aggregate_normalization: Optional[float] = None, pre_gather_matmul: bool = False,
               name: str = "DeepGNN", params: Optional[Mapping[str, Mapping[str, Any]]] = None,
               device="cuda:0", precision: Optional[str] = None):
    del edge_update_remat_block_size, remat_block_size, f32_aggregation, gather_scatter_kwargs   # (see docstring)
    _check_dense(dense_kwargs, "dense_kwargs")
    _check_dense(edge_dense_kwargs if edge_dense_kwargs is not None else dense_kwargs, "edge_dense_kwargs")
    if not gather_from_receivers:
      raise NotImplementedError("DeepGNN on MI355X: gather_from_receivers=False is not built")
    if params is None:
      raise ValueError("DeepGNN has no parameters: pass params= (haiku tree, see the module docstring)")
    self._name, self._params = name, params
    self._steps, self._reps = int(num_message_passing_steps), int(num_processor_repetitions)
    self._edge_res, self._pre, self._agg_norm = bool(use_edge_residuals), bool(pre_gather_matmul), aggregate_normalization
    # ---- the parts of StepEngine this class uses (it does not run the GraphCast step program)
    self.dev = torch.device(device)
    self.lib = nat.lib()
    precision = precision or launch.DEFAULT_PRECISION
    if precision not in ("f16x3", "f32"):
      raise NotImplementedError("DeepGNN on MI355X runs in the fp32-grade precisions (f16x3 | f32)")
    self.precision, self.prec = precision, nat.PRECISIONS[precision]
    self.half = self.prec == nat.PREC_F16X3
    self.onepass, self.scratch, self._keep = False, None, []
    # latents come from the caller: every launch that reads rows carries the f16x3 range flag (launch.LaunchBase.check_range)
    self.check_all_rows = True
    self.range_flag = (torch.zeros((1,), dtype=torch.int32, device=self.dev)
                       if self.half and self.prec == nat.PREC_F16X3 else None)
    self._mlps = None
    self._graphs = {}

  # ---------------------------------------------------------------- parameters
  def _legacy_view(self, i, edge_name, node_names):
    """The step's modules under the names launch._Mlp reads (``<stem>_mlp/~/linear_k``, ``<stem>_layer_norm``);
    the pre-gather form's three matrices stacked back into [W_e; W_s; W_r]."""
    p, n = self._params, self._name
    view = {}

    def dense(stem, src, w0=None):
      l0 = dict(p[f"{src}/mlp/linear_0"])
      if w0 is not None:
        l0["w"] = w0
      view[f"{stem}_mlp/~/linear_0"] = l0
      view[f"{stem}_mlp/~/linear_1"] = p[f"{src}/mlp/linear_1"]
      view[f"{stem}_layer_norm"] = p[f"{src}/normalization/layer_norm"]

    w0 = None
    if self._pre:
      w0 = np.concatenate([np.asarray(p[f"{n}/processor_edges_{i}_{part}_{edge_name}"]["w"], np.float32)
                           for part in ("edge", "sender", "receiver")])
    dense("E", f"{n}/processor_edges_{i}_{edge_name}", w0)
    for k, node in enumerate(node_names):
      src = f"{n}/processor_nodes_{i}_{node}"
      w = np.asarray(p[f"{src}/mlp/linear_0"]["w"], np.float32)
      if k == 0 and self._agg_norm:                  # agg / c == agg . (W_a / c): folded into the aggregate's rows
        w = w.copy()
        w[D:] /= np.float32(self._agg_norm)
      dense(f"N{k}", src, w)
    return view

  def _build_mlps(self, edge_name, recv_set, send_set):
    key = (edge_name, recv_set, send_set)
    if self._mlps is not None and self._mlps[0] == key:
      return self._mlps[1]
    nodes = [recv_set] + ([send_set] if send_set != recv_set else [])
    out = []
    for i in range(self._steps):
      view = self._legacy_view(i, edge_name, nodes)
      out.append(dict(
          edge=launch._Mlp(view, "E", self.dev, split=("e", "s", "r"), prec=self.prec),
          recv=launch._Mlp(view, "N0", self.dev, prec=self.prec),
          send=launch._Mlp(view, "N1", self.dev, prec=self.prec) if len(nodes) > 1 else None))
    self._keep.append(out)
    self._mlps = (key, out)          # (a graph with other set names reads other parameter modules: rebuilt)
    return out

  def _edges_of(self, senders, receivers, n_recv, n_send):
    s, r = np.asarray(senders).astype(np.int64), np.asarray(receivers).astype(np.int64)
    key = (n_recv, n_send, s.tobytes(), r.tobytes())       # (the index bytes themselves: no hash collisions)
    if key not in self._graphs:
      if len(r) and (r.min() < 0 or r.max() >= n_recv):
        raise NotImplementedError("DeepGNN on MI355X: padding edges (receiver index outside the node set) are not built")
      if len(s) and (s.min() < 0 or s.max() >= n_send):
        raise ValueError(f"DeepGNN: sender index outside the sender node set (0 <= s < {n_send})")
      e = launch._Edges(packing.pack_edges(s, r, n_recv), self.dev)
      ok = e.pk.perm >= 0
      e.src = torch.from_numpy(np.where(ok, e.pk.perm, 0).astype(np.int64)).to(self.dev)       # packed row -> edge
      e.ok = torch.from_numpy(ok).to(self.dev)
      e.rows_of_edge = torch.from_numpy(np.argsort(np.where(ok, e.pk.perm, len(s) + np.arange(len(ok))))[:len(s)]
                                        .astype(np.int64)).to(self.dev)                          # edge -> packed row
      self._graphs[key] = e
      self._keep.append(e)
    return self._graphs[key]

  # ---------------------------------------------------------------- forward
  def __call__(self, input_graph: typed_graph.TypedGraph, global_norm_conditioning=None, is_training=None):
    del is_training
    if global_norm_conditioning is not None:
      raise NotImplementedError("DeepGNN on MI355X: norm conditioning is not built (use conditioned.py's encoder / decoder)")
    if len(input_graph.edges) != 1:
      raise NotImplementedError("DeepGNN on MI355X: exactly one edge set")
    (ekey, eset), = input_graph.edges.items()
    send_set, recv_set = ekey.node_sets
    extra = set(input_graph.nodes) - {send_set, recv_set}
    if extra:
      raise NotImplementedError(f"DeepGNN on MI355X: node sets without edges are not built: {sorted(extra)}")
    h_recv, h_send = input_graph.nodes[recv_set].features, input_graph.nodes[send_set].features
    e_in = eset.features
    for t, what in ((h_recv, "node"), (h_send, "node"), (e_in, "edge")):
      if (not isinstance(t, torch.Tensor) or t.dtype != torch.float32 or t.dim() != 3 or t.shape[2] != D
          or t.device != self.dev):
        raise ValueError(f"{what} features must be float32 [rows, batch, {D}] tensors on {self.dev}")
    batch = h_recv.shape[1]
    n_recv, n_send = h_recv.shape[0], h_send.shape[0]
    edges = self._edges_of(eset.indices.senders, eset.indices.receivers, n_recv, n_send)
    mlps = self._build_mlps(ekey.name, recv_set, send_set)
    same = send_set == recv_set
    new = lambda rows: torch.empty((rows, D), dtype=torch.float32, device=self.dev)
    out_recv, out_send = torch.empty_like(h_recv), (None if same else torch.empty_like(h_send))
    out_e = torch.empty_like(e_in)
    pre_s, pre_r, agg = new(n_send), new(n_recv), new(n_recv)
    for b in range(batch):
      # private copies: the node launches below write in place (res = out = hr / hs), and with batch == 1
      # `h_recv[:, b].contiguous()` is a VIEW of the caller's tensor -- the reference never modifies its input
      hr = h_recv[:, b].clone()
      hs = hr if same else h_send[:, b].clone()
      e = e_in[:, b].index_select(0, edges.src)                        # packed rows (padding rows: edge 0, masked by seg = -1)
      ops = []
      for _ in range(self._reps):
        for m in mlps:
          me = m["edge"]
          ops.append(self._op_mlp("proc_pre", self._desc(nat.MODE_LINEAR, n_send, a0=hs, k0=D, w1p=me.w1["s"], out=pre_s)))
          ops.append(self._op_mlp("proc_pre", self._desc(nat.MODE_LINEAR, n_recv, a0=hr, k0=D, w1p=me.w1["r"], out=pre_r)))
          ops.append(self._op_mlp("proc_edge", self._mlp_ln(
              edges.n_rows, me, a0=e, k0=D, w1p=me.w1["e"], b1=me.b1, g0=pre_s, idx0=edges.snd, g1=pre_r,
              idx1=edges.rcv, res=e if self._edge_res else None, out=e, edges=edges, agg=agg)))
          ops += self._ops_after_segsum(edges, agg)
          if not same:                               # the sender set receives nothing: h' = h + LN(MLP(h))
            ms = m["send"]
            ops.append(self._op_mlp("proc_node", self._mlp_ln(n_send, ms, a0=hs, k0=D, w1p=ms.w1, b1=ms.b1, res=hs, out=hs)))
          mn = m["recv"]
          ops.append(self._op_mlp("proc_node", self._mlp_ln(n_recv, mn, a0=hr, k0=D, a1=agg, k1=D, w1p=mn.w1, b1=mn.b1,
                                                            res=hr, out=hr)))
      self._run(ops)
      out_recv[:, b] = hr
      if not same:
        out_send[:, b] = hs
      out_e[:, b] = e.index_select(0, edges.rows_of_edge)
    self.check_range()           # (f16x3: latents beyond +-65504 raise instead of coming back wrong)
    nodes = {recv_set: input_graph.nodes[recv_set]._replace(features=out_recv)}
    if not same:
      nodes[send_set] = input_graph.nodes[send_set]._replace(features=out_send)
    return input_graph._replace(nodes=nodes, edges={ekey: eset._replace(features=out_e)})

  # the GraphCast step API of the base class does not apply here
  def forward(self, *a, **k):
    raise TypeError("DeepGNN is called on a TypedGraph")
