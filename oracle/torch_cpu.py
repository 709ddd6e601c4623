"""The oracle's step on torch CPU tensors (fp32): the same restatement as oracle/graphcast.py +
oracle/gnn.py -- reference weathernext1_graph/graphcast.py:550-678, utils/legacy/
deep_typed_graph_net.py:325-401, utils/typed_graph_net.py:369-546 -- with torch's multi-threaded
GEMM / elementwise kernels instead of numpy's (whose elementwise ops run on one core).

TEST INFRASTRUCTURE like the rest of oracle/: it exists so that bench.py's `cpu_baseline` times a
CPU implementation that actually uses the host's cores ("the fp32 restatement
executed with torch-CPU GEMMs"); tests/test_oracle_torch_cpu.py pins it to the numpy oracle.
"""
import os

import numpy as np
import torch

LN_EPS = 1e-5


def _t(a):
  return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float32)))


class _Net:
  def __init__(self, params, gnn_name):
    self._p, self._g = params, gnn_name
    self._cache = {}

  def _get(self, module, leaf):
    key = (module, leaf)
    if key not in self._cache:
      self._cache[key] = _t(self._p[f"{self._g}/~_networks_builder/{module}"][leaf])
    return self._cache[key]

  def apply(self, name, *args, use_layer_norm=True):
    x = torch.cat(args, dim=-1) if len(args) > 1 else args[0]
    lead = x.shape[:-1]
    y = torch.addmm(self._get(f"{name}_mlp/~/linear_0", "b"), x.reshape(-1, x.shape[-1]),
                    self._get(f"{name}_mlp/~/linear_0", "w"))
    y = torch.nn.functional.silu(y)
    y = torch.addmm(self._get(f"{name}_mlp/~/linear_1", "b"), y, self._get(f"{name}_mlp/~/linear_1", "w"))
    if use_layer_norm:
      y = torch.nn.functional.layer_norm(y, (y.shape[-1],), self._get(f"{name}_layer_norm", "scale"),
                                         self._get(f"{name}_layer_norm", "offset"), eps=LN_EPS)
    return y.reshape(lead + (y.shape[-1],))


def _segment_sum(data, ids, n):
  out = torch.zeros((n,) + tuple(data.shape[1:]), dtype=data.dtype)
  return out.index_add_(0, ids, data)


def _gnn(params, gnn_name, nodes, edges, *, num_steps, embed_nodes, embed_edges, node_output=(),
         live_nodes=None, chunk=1 << 17, taps=None):
  """edges: name -> dict(features, senders, receivers (LongTensors), senders_set, receivers_set).
  `taps` (dict or None) receives the last step's received aggregates as "agg:<edge set>"."""
  net = _Net(params, gnn_name)
  if embed_edges:
    for k, e in edges.items():
      e["features"] = net.apply(f"encoder_edges_{k}", e["features"])
  if embed_nodes:
    for k in nodes:
      nodes[k] = net.apply(f"encoder_nodes_{k}", nodes[k])
  for step in range(num_steps):
    last = step == num_steps - 1
    new_edges = {}
    for k, e in edges.items():                 # e' = f([e | h_send[senders] | h_recv[receivers]]), row-chunked
      hs, hr = nodes[e["senders_set"]], nodes[e["receivers_set"]]
      pieces = []
      for lo in range(0, e["features"].shape[0], chunk):
        hi = min(lo + chunk, e["features"].shape[0])
        pieces.append(net.apply(f"processor_edges_{step}_{k}", e["features"][lo:hi],
                                hs[e["senders"][lo:hi]], hr[e["receivers"][lo:hi]]))
      new_edges[k] = torch.cat(pieces, dim=0)
    new_nodes = {}
    for k, h in nodes.items():
      if last and live_nodes is not None and k not in live_nodes:
        continue
      received = [_segment_sum(new_edges[ek], e["receivers"], h.shape[0])
                  for ek, e in sorted(edges.items()) if e["receivers_set"] == k]
      if taps is not None and last:
        for (ek, e), r in zip([(ek, e) for ek, e in sorted(edges.items()) if e["receivers_set"] == k],
                              received):
          taps[f"agg:{ek}"] = r
      new_nodes[k] = net.apply(f"processor_nodes_{step}_{k}", h, *received)
    for k in list(nodes):
      if k in new_nodes:
        nodes[k] = nodes[k] + new_nodes[k]
      else:
        del nodes[k]
    if not last:
      for k, e in edges.items():
        e["features"] = e["features"] + new_edges[k]
  for k in node_output:
    nodes[k] = net.apply(f"decoder_nodes_{k}", nodes[k], use_layer_norm=False)
  return nodes


def set_threads(n=None):
  """Threads for the CPU kernels: the host's PHYSICAL cores by default (SMT siblings only add
  contention to fp32 GEMMs).  Returns the count in effect."""
  if n is None:
    n = os.cpu_count() or 1
    try:
      import psutil
      n = psutil.cpu_count(logical=False) or n
    except Exception:                                            # pragma: no cover
      pass
  torch.set_num_threads(max(1, int(n)))
  return torch.get_num_threads()


def forward(params, graphs, x_grid, steps, taps=None):
  """x_grid [N_grid, B, C_in] (numpy) -> [N_grid, B, C_out] (numpy float32).

  `taps` (a dict) additionally receives the stage boundaries of graphcast.py:311-319 as numpy
  arrays: "enc_agg_mesh" (the grid2mesh segment-sum the encoder's mesh-node update consumes,
  typed_graph_net.py:532-538), "latent_mesh" / "latent_grid" (after _run_grid2mesh_gnn),
  "updated_mesh" (after _run_mesh_gnn)."""
  x = _t(x_grid)
  b = x.shape[1]
  batch = lambda a: _t(a)[:, None, :].expand(-1, b, -1)
  ids = lambda a: torch.as_tensor(np.asarray(a, dtype=np.int64))
  edge = lambda g, s, r: dict(features=batch(g["feat"]).contiguous(), senders=ids(g["senders"]),
                              receivers=ids(g["receivers"]), senders_set=s, receivers_set=r)
  n_mesh = graphs["n_mesh"]
  with torch.no_grad():
    enc = _gnn(params, "grid2mesh_gnn",
               {"grid_nodes": torch.cat([x, batch(graphs["grid_node_feat"])], -1),
                "mesh_nodes": torch.cat([torch.zeros((n_mesh,) + tuple(x.shape[1:])), batch(graphs["mesh_node_feat"])], -1)},
               {"grid2mesh": edge(graphs["g2m"], "grid_nodes", "mesh_nodes")},
               num_steps=1, embed_nodes=True, embed_edges=True, taps=taps)
    if taps is not None:
      taps["enc_agg_mesh"] = taps.pop("agg:grid2mesh").numpy()
      taps["latent_mesh"] = enc["mesh_nodes"].numpy()
      taps["latent_grid"] = enc["grid_nodes"].numpy()
    proc = _gnn(params, "mesh_gnn", {"mesh_nodes": enc["mesh_nodes"]},
                {"mesh": edge(graphs["mesh"], "mesh_nodes", "mesh_nodes")},
                num_steps=steps, embed_nodes=False, embed_edges=True)
    dec = _gnn(params, "mesh2grid_gnn", {"mesh_nodes": proc["mesh_nodes"], "grid_nodes": enc["grid_nodes"]},
               {"mesh2grid": edge(graphs["m2g"], "mesh_nodes", "grid_nodes")},
               num_steps=1, embed_nodes=False, embed_edges=True, node_output=("grid_nodes",),
               live_nodes=("grid_nodes",))
    if taps is not None:
      taps["updated_mesh"] = proc["mesh_nodes"].numpy()
  return dec["grid_nodes"].numpy()
