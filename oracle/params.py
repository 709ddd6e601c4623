"""Oracle: parameter tree of the three GNNs (TEST INFRASTRUCTURE, see oracle/__init__.py).

haiku's module-naming rules applied to the names built at
``/root/reference/weathernext/utils/legacy/deep_typed_graph_net.py:205-323`` inside
modules named ``grid2mesh_gnn`` / ``mesh_gnn`` / ``mesh2grid_gnn``
(``weathernext1_graph/graphcast.py:217,233,261``):

  "<gnn>/~_networks_builder/<prefix><set>_mlp/~/linear_<k>"  -> {"w" [in,out], "b" [out]}
  "<gnn>/~_networks_builder/<prefix><set>_layer_norm"        -> {"scale", "offset"}

Default initialisation restates hk.Linear's defaults
(w ~ TruncatedNormal(stddev=1/sqrt(fan_in), +-2 sigma), b = 0) and
hk.LayerNorm's (scale = 1, offset = 0); corroborated in-tree by
``utils/dense.py:450-517``.
"""
import numpy as np
from scipy import stats


def module_specs(c_in, c_out, latent, steps, n_struct_node=3, n_struct_edge=4):
  """[(module stem, [layer sizes incl. input], has_layer_norm)], haiku order-free."""
  d = latent
  specs = []
  def add(gnn, stem, fan_in, out, ln=True):
    specs.append((f"{gnn}/~_networks_builder/{stem}", [fan_in, d, out], ln))
  g = "grid2mesh_gnn"
  add(g, "encoder_edges_grid2mesh", n_struct_edge, d)
  add(g, "encoder_nodes_grid_nodes", c_in + n_struct_node, d)
  add(g, "encoder_nodes_mesh_nodes", c_in + n_struct_node, d)
  add(g, "processor_edges_0_grid2mesh", 3 * d, d)
  add(g, "processor_nodes_0_grid_nodes", d, d)
  add(g, "processor_nodes_0_mesh_nodes", 2 * d, d)
  g = "mesh_gnn"
  add(g, "encoder_edges_mesh", n_struct_edge, d)
  for i in range(steps):
    add(g, f"processor_edges_{i}_mesh", 3 * d, d)
    add(g, f"processor_nodes_{i}_mesh_nodes", 2 * d, d)
  g = "mesh2grid_gnn"
  add(g, "encoder_edges_mesh2grid", n_struct_edge, d)
  add(g, "processor_edges_0_mesh2grid", 3 * d, d)
  add(g, "processor_nodes_0_grid_nodes", 2 * d, d)
  add(g, "processor_nodes_0_mesh_nodes", d, d)
  add(g, "decoder_nodes_grid_nodes", d, c_out, ln=False)
  return specs


def init_params(c_in, c_out, latent, steps, seed=1, nontrivial=False):
  """float32 haiku-layout params.  ``nontrivial`` also randomises b / scale / offset."""
  rng = np.random.default_rng(seed)
  params = {}
  for stem, sizes, ln in module_specs(c_in, c_out, latent, steps):
    for k in range(len(sizes) - 1):
      fan_in, fan_out = sizes[k], sizes[k + 1]
      u = rng.random((fan_in, fan_out))
      w = stats.truncnorm.ppf(u, -2.0, 2.0) / np.sqrt(fan_in)
      b = (0.1 * rng.standard_normal(fan_out) if nontrivial else np.zeros(fan_out))
      params[f"{stem}_mlp/~/linear_{k}"] = {
          "w": w.astype(np.float32), "b": b.astype(np.float32)}
    if ln:
      scale = 1.0 + (0.1 * rng.standard_normal(sizes[-1]) if nontrivial else 0.0)
      offset = 0.1 * rng.standard_normal(sizes[-1]) if nontrivial else np.zeros(sizes[-1])
      params[f"{stem}_layer_norm"] = {
          "scale": (np.zeros(sizes[-1]) + scale).astype(np.float32),
          "offset": np.asarray(offset).astype(np.float32)}
  return params


def conditioned_module_specs(c_grid, c_mesh, c_edge, c_out, latent):
  """The GenCast encoder / decoder pair (weathernext1_gen/denoiser.py:303-363): the same
  DeepTypedGraphNet modules as GraphCast's grid2mesh / mesh2grid GNNs, built with
  ``use_norm_conditioning=True`` -- no LayerNorm parameters, one ``<stem>_norm_conditioning/linear``
  per normalised MLP instead (deep_typed_graph_net.py:210-246, dense.py:360-393)."""
  d = latent
  specs = []
  def add(gnn, stem, fan_in, out, cond=True):
    specs.append((f"{gnn}/~_networks_builder/{stem}", [fan_in, d, out], cond))
  g = "grid2mesh_gnn"
  add(g, "encoder_edges_grid2mesh", c_edge, d)
  add(g, "encoder_nodes_grid_nodes", c_grid, d)
  add(g, "encoder_nodes_mesh_nodes", c_mesh, d)
  add(g, "processor_edges_0_grid2mesh", 3 * d, d)
  add(g, "processor_nodes_0_grid_nodes", d, d)
  add(g, "processor_nodes_0_mesh_nodes", 2 * d, d)
  g = "mesh2grid_gnn"
  add(g, "encoder_edges_mesh2grid", c_edge, d)
  add(g, "processor_edges_0_mesh2grid", 3 * d, d)
  add(g, "processor_nodes_0_grid_nodes", 2 * d, d)
  add(g, "processor_nodes_0_mesh_nodes", d, d)
  add(g, "decoder_nodes_grid_nodes", d, c_out, cond=False)
  return specs


def init_conditioned_params(c_grid, c_mesh, c_edge, c_cond, c_out, latent, seed=1):
  """Seeded float32 parameters of the norm-conditioned encoder / decoder.  The conditioning
  layers are drawn at O(0.3) (the reference initialises them at ~1e-8, dense.py:381,385, which
  would make the conditioning invisible to a test); biases are non-zero."""
  rng = np.random.default_rng(seed)
  params = {}
  for stem, sizes, cond in conditioned_module_specs(c_grid, c_mesh, c_edge, c_out, latent):
    for k in range(len(sizes) - 1):
      fan_in, fan_out = sizes[k], sizes[k + 1]
      w = stats.truncnorm.ppf(rng.random((fan_in, fan_out)), -2.0, 2.0) / np.sqrt(fan_in)
      params[f"{stem}_mlp/~/linear_{k}"] = {
          "w": w.astype(np.float32), "b": (0.1 * rng.standard_normal(fan_out)).astype(np.float32)}
    if cond:
      params[f"{stem}_norm_conditioning/linear"] = {
          "w": (0.3 * rng.standard_normal((c_cond, 2 * sizes[-1]))).astype(np.float32),
          "b": (0.3 * rng.standard_normal(2 * sizes[-1])).astype(np.float32)}
  return params


def digest(params):
  """sha256 over the sorted float32 leaves (detects drift of a seed-regenerated tree)."""
  import hashlib
  h = hashlib.sha256()
  for mod in sorted(params):
    for leaf in sorted(params[mod]):
      h.update(f"{mod}:{leaf}".encode())
      h.update(np.ascontiguousarray(params[mod][leaf], dtype=np.float32).tobytes())
  return h.hexdigest()


def count(params):
  return sum(int(np.prod(a.shape)) for m in params.values() for a in m.values())


def init_deep_gnn_params(latent, steps, node_sets, edge_sets, pre_gather_matmul=False, name="DeepGNN", seed=1):
  """Seeded float32 parameters of the WN2 ``DeepGNN`` processor (reference utils/deep_gnn.py:45-400 with
  ``dense.DenseLayer`` MLPs of one hidden layer + "layer_norm"): haiku names as the reference creates
  them (``hk.transparent`` builder, ``hk.name_like("__call__")`` constructors -- executed on the
  stand-ins by tests/golden/make_golden_deepgnn.py, which asserts that this tree is exactly what the
  reference asks for):
    "<name>/processor_edges_<i>_<edge set>/mlp/linear_0|1", ".../normalization/layer_norm",
    "<name>/processor_nodes_<i>_<node set>/...";  with ``pre_gather_matmul``: linear_0 of an edge MLP is
    a bias only and its matrix lives in "<name>/processor_edges_<i>_{edge,sender,receiver}_<edge set>" {"w"}.
  node_sets: {name: number of edge sets it receives from}; edge_sets: names.  Non-trivial biases / LayerNorm."""
  rng = np.random.default_rng(seed)
  d = latent
  tn = lambda fan_in, shape: (stats.truncnorm.ppf(rng.random(shape), -2.0, 2.0) / np.sqrt(fan_in)).astype(np.float32)
  vec = lambda scale, mean=0.0: (mean + scale * rng.standard_normal(d)).astype(np.float32)
  params = {}

  def dense(stem, fan_in, first_matrix=True):
    if first_matrix:
      params[f"{stem}/mlp/linear_0"] = {"w": tn(fan_in, (fan_in, d)), "b": vec(0.1)}
    else:
      params[f"{stem}/mlp/linear_0"] = {"b": vec(0.1)}
    params[f"{stem}/mlp/linear_1"] = {"w": tn(d, (d, d)), "b": vec(0.1)}
    params[f"{stem}/normalization/layer_norm"] = {"scale": vec(0.1, 1.0), "offset": vec(0.1)}

  for i in range(steps):
    for e in edge_sets:
      if pre_gather_matmul:
        for part in ("sender", "receiver", "edge"):
          params[f"{name}/processor_edges_{i}_{part}_{e}"] = {"w": tn(3 * d, (d, d))}
      dense(f"{name}/processor_edges_{i}_{e}", 3 * d, first_matrix=not pre_gather_matmul)
    for n, n_recv in node_sets.items():
      dense(f"{name}/processor_nodes_{i}_{n}", (1 + n_recv) * d)
  return params
