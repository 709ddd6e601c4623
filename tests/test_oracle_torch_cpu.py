"""oracle/torch_cpu.py (what bench.py's cpu_baseline times) against the numpy oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import graphcast as ogc          # noqa: E402
from oracle import params as oparams         # noqa: E402
from oracle import torch_cpu                 # noqa: E402


@pytest.mark.parametrize("batch", [1, 2])
def test_torch_cpu_step_equals_numpy_oracle(batch):
  res, mesh_size, steps = 10.0, 2, 3
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  graphs = ogc.build_graphs(lat, lon, mesh_size)
  c_in, c_out = 20, 9
  params = oparams.init_params(c_in, c_out, 64, steps, seed=2, nontrivial=True)
  x = np.random.default_rng(batch).standard_normal((graphs["n_grid"], batch, c_in)).astype(np.float32)
  want = ogc.forward(params, graphs, x, steps=steps, dtype=np.float64)
  got = torch_cpu.forward(params, graphs, x, steps)
  assert got.shape == want.shape and got.dtype == np.float32
  err = np.linalg.norm(got - want) / np.linalg.norm(want)
  assert err < 5e-6, err


def test_torch_cpu_stage_taps_equal_numpy_oracle():
  """The stage boundaries the full-size GPU parity test compares (tests/test_fullsize_gpu.py)."""
  from oracle import gnn
  res, mesh_size, steps = 10.0, 2, 2
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  graphs = ogc.build_graphs(lat, lon, mesh_size)
  c_in, c_out = 20, 9
  params = oparams.init_params(c_in, c_out, 64, steps, seed=3, nontrivial=True)
  x = np.random.default_rng(5).standard_normal((graphs["n_grid"], 1, c_in)).astype(np.float32)
  _, lat64 = ogc.forward(params, graphs, x, steps=steps, dtype=np.float64, return_latents=True)
  taps = {}
  torch_cpu.forward(params, graphs, x, steps, taps=taps)
  rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
  assert rel(taps["latent_mesh"], lat64["latent_mesh"]) < 5e-6
  assert rel(taps["latent_grid"], lat64["latent_grid"]) < 5e-6
  assert rel(taps["updated_mesh"], lat64["updated_mesh"]) < 5e-6
  # the encoder aggregate, recomputed from the float64 pieces: e' of the grid2mesh edges summed by receiver
  net = gnn.Net(params, "grid2mesh_gnn", np.float64)
  b = lambda a: np.repeat(np.asarray(a, np.float64)[:, None, :], 1, axis=1)
  hg = net.apply("encoder_nodes_grid_nodes", np.concatenate([x.astype(np.float64), b(graphs["grid_node_feat"])], -1))
  hm = net.apply("encoder_nodes_mesh_nodes", np.concatenate(
      [np.zeros((graphs["n_mesh"], 1, c_in)), b(graphs["mesh_node_feat"])], -1))
  e = net.apply("encoder_edges_grid2mesh", b(graphs["g2m"]["feat"]))
  e2 = net.apply("processor_edges_0_grid2mesh", e, hg[graphs["g2m"]["senders"]], hm[graphs["g2m"]["receivers"]])
  agg = gnn.segment_sum(e2, graphs["g2m"]["receivers"], graphs["n_mesh"])
  assert rel(taps["enc_agg_mesh"], agg) < 5e-6
  assert torch_cpu.set_threads(2) == 2
