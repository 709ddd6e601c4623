"""Pins the oracle's GNN wiring against tests/golden/gnn_tiny.npz, which was produced by
executing the reference's own graphcast.py / deep_typed_graph_net.py / typed_graph_net.py on
numpy stand-ins for haiku / jraph / jax (tests/golden/make_golden.py, part 2)."""
import os

import numpy as np
import pytest

from oracle import gnn as ognn
from oracle import graphcast as ogc
from oracle import params as oparams


@pytest.fixture(scope="module")
def fx(golden_dir):
  z = np.load(os.path.join(golden_dir, "gnn_tiny.npz"))
  params = {}
  for k in z.files:
    if k.startswith("params:"):
      _, mod, leaf = k.split(":")
      params.setdefault(mod, {})[leaf] = z[k]
  res, mesh_size, latent, steps, batch, c_in = z["config"]
  return dict(z=z, params=params, mesh_size=int(mesh_size), latent=int(latent), steps=int(steps))


def test_parameter_tree_names_and_shapes_match_reference(fx):
  """The oracle's haiku naming (oracle/params.py) == what the reference's module code creates."""
  z, c_in = fx["z"], int(fx["z"]["config"][5])
  mine = oparams.init_params(c_in, fx["z"]["out"].shape[-1], fx["latent"], fx["steps"], seed=0)
  assert sorted(mine) == sorted(fx["params"])
  for mod, leaves in mine.items():
    assert sorted(leaves) == sorted(fx["params"][mod]), mod
    for leaf, v in leaves.items():
      assert v.shape == fx["params"][mod][leaf].shape, (mod, leaf)


def test_graph_indices_match_reference_run(fx):
  z = fx["z"]
  g = ogc.build_graphs(z["lat"], z["lon"], fx["mesh_size"])
  np.testing.assert_array_equal(g["g2m"]["senders"], z["g2m_senders"])
  np.testing.assert_array_equal(g["g2m"]["receivers"], z["g2m_receivers"])
  np.testing.assert_array_equal(g["m2g"]["senders"], z["m2g_senders"])
  np.testing.assert_array_equal(g["m2g"]["receivers"], z["m2g_receivers"])


def test_forward_reproduces_reference_stage_outputs(fx):
  z = fx["z"]
  g = ogc.build_graphs(z["lat"], z["lon"], fx["mesh_size"])
  out, lat = ogc.forward(fx["params"], g, z["x"], steps=fx["steps"], dtype=np.float64,
                         return_latents=True, f32_aggregation=True)   # graphcast.py:215
  for got, name in ((lat["latent_mesh"], "latent_mesh"), (lat["latent_grid"], "latent_grid"),
                    (lat["updated_mesh"], "updated_mesh"), (out, "out")):
    want = z[name]
    assert got.shape == want.shape, name
    err = np.linalg.norm(got - want) / np.linalg.norm(want)
    assert err < 1e-12, (name, err)


def test_primitives_against_torch():
  """hk.Linear / LayerNorm / swish / segment_sum restatements vs torch's own implementations
  (independent code; the third-party originals are not installable -> 'parity unpinned')."""
  torch = pytest.importorskip("torch")
  rng = np.random.default_rng(0)
  x = rng.standard_normal((37, 2, 24))
  w, b = rng.standard_normal((24, 16)), rng.standard_normal(16)
  sc, of = rng.standard_normal(16), rng.standard_normal(16)
  tx = torch.from_numpy(x)
  np.testing.assert_allclose(ognn.linear(x, w, b),
                             torch.nn.functional.linear(tx, torch.from_numpy(w.T.copy()),
                                                        torch.from_numpy(b)).numpy(), atol=1e-12)
  np.testing.assert_allclose(ognn.swish(x), torch.nn.functional.silu(tx).numpy(), atol=1e-14)
  y = ognn.linear(x, w, b)
  np.testing.assert_allclose(
      ognn.layer_norm(y, sc, of),
      torch.nn.functional.layer_norm(torch.from_numpy(y), (16,), torch.from_numpy(sc),
                                     torch.from_numpy(of), eps=1e-5).numpy(), atol=1e-12)
  ids = rng.integers(0, 9, size=37)
  want = torch.zeros((11, 2, 24), dtype=torch.float64).index_add_(0, torch.from_numpy(ids), tx)
  np.testing.assert_allclose(ognn.segment_sum(x, ids, 11), want.numpy(), atol=1e-13)


def load_latent512(golden_dir):
  """(fixture dict, params regenerated from the stored seed) -- digest-checked."""
  import hashlib
  z = np.load(os.path.join(golden_dir, "gnn_latent512.npz"))
  c_in, c_out, latent, steps, seed = (int(v) for v in z["params_seed"])
  params = oparams.init_params(c_in, c_out, latent, steps, seed=seed, nontrivial=True)
  h = hashlib.sha256()
  for mod in sorted(params):
    for leaf in sorted(params[mod]):
      h.update(f"{mod}:{leaf}".encode())
      h.update(np.ascontiguousarray(params[mod][leaf], dtype=np.float32).tobytes())
  assert h.hexdigest() == str(z["params_sha256"]), "seed-regenerated parameters drifted"
  return z, params, steps


def test_oracle_reproduces_latent512_reference_run(golden_dir):
  z, params, steps = load_latent512(golden_dir)
  g = ogc.build_graphs(z["lat"], z["lon"], int(z["config"][1]))
  out, lat = ogc.forward(params, g, z["x"].astype(np.float64), steps=steps, dtype=np.float64,
                         return_latents=True, f32_aggregation=True)
  assert np.linalg.norm(out - z["out"]) / np.linalg.norm(z["out"]) < 1e-12
  np.testing.assert_allclose(
      [lat["latent_mesh"].sum(), np.abs(lat["updated_mesh"]).sum()], z["latent_mesh_checksum"],
      rtol=1e-10)


def test_norm_conditioned_nets_reproduce_reference_execution(golden_dir):
  """the GenCast encoder / decoder are the same DeepTypedGraphNet with
  `use_norm_conditioning=True` (weathernext1_gen/denoiser.py:303-363): LayerNorm without learned
  scale / offset followed by dense.LinearNormConditioning.  tests/golden/gnn_conditioned.npz is
  the reference's own code (deep_typed_graph_net.py + dense.py) executed on the stand-ins; the
  oracle must reproduce both stages, including the module names of the conditioning layers."""
  z = np.load(os.path.join(golden_dir, "gnn_conditioned.npz"))
  params = {}
  for k in z.files:
    if k.startswith("params:"):
      _, mod, leaf = k.split(":")
      params.setdefault(mod, {})[leaf] = z[k]
  assert not any(m.endswith("_layer_norm") for m in params)       # no learned scale / offset in this mode
  assert "grid2mesh_gnn/~_networks_builder/processor_nodes_0_mesh_nodes_norm_conditioning/linear" in params
  cond = z["cond"]
  enc_graph = {"nodes": {"grid_nodes": z["grid_x"], "mesh_nodes": z["mesh_x"]},
               "edges": {"grid2mesh": dict(features=z["g2m_e"], senders=z["g2m_senders"], receivers=z["g2m_receivers"],
                                           senders_set="grid_nodes", receivers_set="mesh_nodes")}}
  enc = ognn.deep_typed_graph_net(params, "grid2mesh_gnn", enc_graph, num_steps=1, embed_nodes=True,
                                  embed_edges=True, dtype=np.float64, f32_aggregation=True,
                                  norm_conditioning=cond)
  for name, want in (("grid_nodes", z["enc_grid"]), ("mesh_nodes", z["enc_mesh"])):
    got = enc["nodes"][name]
    # (f32_aggregation casts the messages to float32 around the segment-sum, as the reference does)
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-12, name
  dec_graph = {"nodes": {"grid_nodes": enc["nodes"]["grid_nodes"], "mesh_nodes": enc["nodes"]["mesh_nodes"]},
               "edges": {"mesh2grid": dict(features=z["m2g_e"], senders=z["m2g_senders"], receivers=z["m2g_receivers"],
                                           senders_set="mesh_nodes", receivers_set="grid_nodes")}}
  dec = ognn.deep_typed_graph_net(params, "mesh2grid_gnn", dec_graph, num_steps=1, embed_nodes=False,
                                  embed_edges=True, node_output=("grid_nodes",), dtype=np.float64,
                                  norm_conditioning=cond)
  got, want = dec["nodes"]["grid_nodes"], z["dec_grid"]
  assert got.shape == want.shape == (40, 2, 7)
  assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-12
  # and the conditioning really matters in this fixture: without it the result is different
  with pytest.raises(KeyError):
    ognn.deep_typed_graph_net(params, "mesh2grid_gnn", dec_graph, num_steps=1, embed_nodes=False,
                              embed_edges=True, node_output=("grid_nodes",), dtype=np.float64)


def load_conditioned512(golden_dir):
  """(fixture, params regenerated from the stored seed -- digest-checked)."""
  z = np.load(os.path.join(golden_dir, "gnn_conditioned512.npz"))
  c_grid, c_mesh, c_edge, c_cond, c_out, latent, seed = (int(v) for v in z["config"])
  params = oparams.init_conditioned_params(c_grid, c_mesh, c_edge, c_cond, c_out, latent, seed=seed)
  assert oparams.digest(params) == str(z["params_sha256"]), "seed-regenerated parameters drifted"
  return z, params


def conditioned_oracle(z, params, dtype=np.float64):
  b = z["cond"].shape[0]
  rep = lambda e: np.repeat(np.asarray(e, dtype)[:, None, :], b, axis=1)
  enc = ognn.deep_typed_graph_net(
      params, "grid2mesh_gnn",
      {"nodes": {"grid_nodes": z["grid_x"], "mesh_nodes": z["mesh_x"]},
       "edges": {"grid2mesh": dict(features=rep(z["g2m_e"]), senders=z["g2m_senders"], receivers=z["g2m_receivers"],
                                   senders_set="grid_nodes", receivers_set="mesh_nodes")}},
      num_steps=1, embed_nodes=True, embed_edges=True, dtype=dtype, f32_aggregation=True,
      norm_conditioning=z["cond"])
  dec = ognn.deep_typed_graph_net(
      params, "mesh2grid_gnn",
      {"nodes": {"grid_nodes": enc["nodes"]["grid_nodes"], "mesh_nodes": enc["nodes"]["mesh_nodes"]},
       "edges": {"mesh2grid": dict(features=rep(z["m2g_e"]), senders=z["m2g_senders"], receivers=z["m2g_receivers"],
                                   senders_set="mesh_nodes", receivers_set="grid_nodes")}},
      num_steps=1, embed_nodes=False, embed_edges=True, node_output=("grid_nodes",), dtype=dtype,
      norm_conditioning=z["cond"])
  return enc["nodes"]["grid_nodes"], enc["nodes"]["mesh_nodes"], dec["nodes"]["grid_nodes"]


def test_oracle_reproduces_conditioned_latent512_reference_run(golden_dir):
  """gnn_conditioned512.npz: the reference's norm-conditioned encoder / decoder executed at the
  width the HIP kernels are built for; the oracle reproduces it (it is what the device path is
  compared with on larger graphs, tests/test_conditioned_gpu.py)."""
  z, params = load_conditioned512(golden_dir)
  g, m, d = conditioned_oracle(z, params)
  for got, want in ((g, z["enc_grid"]), (m, z["enc_mesh"]), (d, z["dec_grid"])):
    assert got.shape == want.shape
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-9       # (f32_aggregation casts messages)
