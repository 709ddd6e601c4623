"""Parameter tree of the GraphCast step in the reference's haiku layout.

Keys are the module paths a reference ``CheckPoint.params`` carries
(``weathernext1_graph/graphcast.py:145-151``): modules are created inside
``DeepTypedGraphNet._networks_builder`` (``utils/legacy/deep_typed_graph_net.py:198-323``)
of hk.Modules named ``grid2mesh_gnn`` / ``mesh_gnn`` / ``mesh2grid_gnn``
(``graphcast.py:217,233,261``):

  "<gnn>/~_networks_builder/<prefix><set>_mlp/~/linear_<k>" -> {"w" [in, out], "b" [out]}
  "<gnn>/~_networks_builder/<prefix><set>_layer_norm"       -> {"scale", "offset"}
"""
from typing import Dict, List, Tuple

import numpy as np


def mlp_table(c_in: int, c_out: int, latent: int, steps: int, n_node_struct: int = 3,
              n_edge_struct: int = 4) -> List[Tuple[str, int, int, bool]]:
  """(module stem, input width, output width, followed by LayerNorm) for every MLP."""
  d = latent
  rows = []
  enc, proc, dec = "grid2mesh_gnn", "mesh_gnn", "mesh2grid_gnn"
  put = lambda gnn, stem, k, n, ln=True: rows.append((f"{gnn}/~_networks_builder/{stem}", k, n, ln))
  put(enc, "encoder_edges_grid2mesh", n_edge_struct, d)
  put(enc, "encoder_nodes_grid_nodes", c_in + n_node_struct, d)
  put(enc, "encoder_nodes_mesh_nodes", c_in + n_node_struct, d)
  put(enc, "processor_edges_0_grid2mesh", 3 * d, d)
  put(enc, "processor_nodes_0_grid_nodes", d, d)
  put(enc, "processor_nodes_0_mesh_nodes", 2 * d, d)
  put(proc, "encoder_edges_mesh", n_edge_struct, d)
  for i in range(steps):
    put(proc, f"processor_edges_{i}_mesh", 3 * d, d)
    put(proc, f"processor_nodes_{i}_mesh_nodes", 2 * d, d)
  put(dec, "encoder_edges_mesh2grid", n_edge_struct, d)
  put(dec, "processor_edges_0_mesh2grid", 3 * d, d)
  put(dec, "processor_nodes_0_grid_nodes", 2 * d, d)
  put(dec, "processor_nodes_0_mesh_nodes", d, d)      # present in checkpoints, never read (graphcast.py:676)
  put(dec, "decoder_nodes_grid_nodes", d, c_out, False)
  return rows


def random_params(c_in: int, c_out: int, latent: int, steps: int, seed: int = 1) -> Dict[str, Dict[str, np.ndarray]]:
  """Random-init weights of the architecture (synthetic benchmarks; no checkpoint offline).

  w ~ N(0, 1/fan_in) clipped at 2 sigma, small random b / scale / offset so that every
  term of every fused kernel is exercised."""
  rng = np.random.default_rng(seed)
  out = {}
  for stem, k, n, ln in mlp_table(c_in, c_out, latent, steps):
    for layer, (fan_in, fan_out) in enumerate(((k, latent), (latent, n))):
      w = np.clip(rng.standard_normal((fan_in, fan_out), dtype=np.float32), -2, 2)
      out[f"{stem}_mlp/~/linear_{layer}"] = {
          "w": (w / np.sqrt(fan_in)).astype(np.float32),
          "b": (0.1 * rng.standard_normal(fan_out)).astype(np.float32)}
    if ln:
      out[f"{stem}_layer_norm"] = {
          "scale": (1 + 0.1 * rng.standard_normal(n)).astype(np.float32),
          "offset": (0.1 * rng.standard_normal(n)).astype(np.float32)}
  return out


def check_params(params, c_in: int, c_out: int, latent: int, steps: int) -> None:
  """Raises ValueError naming the first missing / mis-shaped leaf."""
  for stem, k, n, ln in mlp_table(c_in, c_out, latent, steps):
    for layer, shape in enumerate(((k, latent), (latent, n))):
      key = f"{stem}_mlp/~/linear_{layer}"
      if key not in params:
        raise ValueError(f"missing parameters for module {key!r}")
      if tuple(np.shape(params[key]["w"])) != shape or tuple(np.shape(params[key]["b"])) != shape[1:]:
        raise ValueError(f"{key}: expected w {shape}, got {np.shape(params[key]['w'])}")
    if ln and f"{stem}_layer_norm" not in params:
      raise ValueError(f"missing parameters for module {stem + '_layer_norm'!r}")
