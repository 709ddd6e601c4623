"""GPU, BASELINE.json's full size (0.25 deg / 37 levels / M6: 1,038,240 grid nodes, 40,962 mesh
nodes, 5.06 M edges): properties that need no oracle -- the float64 oracle would take minutes
per step here; it pins the same kernels at sizes it finishes in seconds (test_step_gpu.py).

  * determinism: two runs give identical bits (no float atomics anywhere on the path);
  * the two fp32-grade arithmetic modes (exact fp32 MFMA / 3 x f16-split MFMA) agree on the whole
    output far inside the 1e-4 budget;
  * batch independence: element b of a batched call equals the single-element call bit for bit;
  * graph structure at this size matches the fingerprints of the reference's own builders
    (tests/golden/structure_hashes.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from graphcast_amd import graphcast as gc          # noqa: E402
from graphcast_amd import params as gparams        # noqa: E402


@pytest.fixture(scope="module")
def full():
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  res, mesh_size, steps = 0.25, 6, 16
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_out = gc.num_output_channels(gc.TASK)
  c_in = 2 * (5 + 6 * 37) + 2 * 5 + 2 + 5
  params = gparams.random_params(c_in, c_out, 512, steps)
  model = gc.GraphCast(cfg, gc.TASK, params=params).init_from_coordinates(lat, lon)
  x = torch.from_numpy(np.random.default_rng(0).standard_normal(
      (len(lat) * len(lon), 1, c_in), dtype=np.float32)).to("cuda:0")
  return dict(model=model, x=x, c_in=c_in, c_out=c_out)


def test_full_size_shapes_and_determinism(full):
  m, x = full["model"], full["x"]
  y1 = m.forward_grid_node_features(x).clone()
  y2 = m.forward_grid_node_features(x).clone()
  assert y1.shape == (1038240, 1, 227)
  assert torch.isfinite(y1).all()
  assert torch.equal(y1, y2)
  full["y"] = y1


def test_full_size_modes_agree(full):
  m, x = full["model"], full["x"]
  y = full.get("y")
  if y is None:
    y = m.forward_grid_node_features(x).clone()
  prev = m.set_precision("f32")
  try:
    y32 = m.forward_grid_node_features(x).clone()
  finally:
    m.set_precision(prev)
  rel = float(torch.linalg.vector_norm((y - y32).double()) / torch.linalg.vector_norm(y32.double()))
  print(f"0.25 deg: f16x3 vs exact-fp32 MFMA rel-RMSE {rel:.2e}")
  assert rel < 5e-6


def test_full_size_batch_independence(full):
  m, x = full["model"], full["x"]
  xb = torch.cat([x, x.flip(0)], dim=1).contiguous()           # two different batch elements
  yb = m.forward_grid_node_features(xb)
  y0 = m.forward_grid_node_features(x)
  assert torch.equal(yb[:, 0], y0[:, 0])


def test_full_size_graph_fingerprints(full, golden_dir):
  """Index arrays the PRODUCT built for this size vs the SHA-256 fingerprints of the arrays the
  reference's own builders produce (tests/golden/make_golden.py): bit-exact."""
  want = json.load(open(os.path.join(golden_dir, "structure_hashes.json")))["0p25deg_M6"]
  g = full["model"].graph_arrays()
  h16 = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
  got = {"mesh_senders": np.asarray(g["mesh"]["senders"], np.int32),
         "mesh_receivers": np.asarray(g["mesh"]["receivers"], np.int32),
         "g2m_grid_idx": np.asarray(g["g2m"]["senders"], np.int64),
         "g2m_mesh_idx": np.asarray(g["g2m"]["receivers"], np.int64)}
  for name, arr in got.items():
    assert list(arr.shape) == want[name]["shape"], name
    assert h16(arr) == want[name]["sha256_16"], name
  assert repr(float(g["radius"])) == want["radius_repr"]
  for name, key in (("grid_node_feat", "grid_node_feat"), ("mesh_node_feat", "mesh_node_feat")):
    a = np.asarray(g[key], np.float64)
    assert abs(np.abs(a).sum() - want[name]["abs_sum_f64"]) <= 1e-6 * want[name]["abs_sum_f64"], name


def test_config0_1deg_13level_step_matches_oracle():
  """BASELINE.json configs[0] -- GraphCast_small-like 1 deg / 13 levels / M5, 16 processor steps
  (65,160 grid nodes, 10,242 mesh nodes): the largest case the float64 oracle finishes in about a
  minute on the GPU box's host cores.  rel-RMSE <= 2e-5 (budget 1e-4)."""
  from oracle import graphcast as ogc
  res, mesh_size, steps = 1.0, 5, 16
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_in, c_out = 183, gc.num_output_channels(gc.TASK_13)
  params = gparams.random_params(c_in, c_out, 512, steps)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params).init_from_coordinates(lat, lon)
  x = np.random.default_rng(2).standard_normal((len(lat) * len(lon), 1, c_in)).astype(np.float32)
  y = model.forward_grid_node_features(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
  want = ogc.forward(params, ogc.build_graphs(lat, lon, mesh_size), x, steps=steps, dtype=np.float64)
  err = float(np.linalg.norm(y - want) / np.linalg.norm(want))
  print(f"1 deg / 13 levels / M5 / 16 steps: rel-RMSE vs float64 oracle {err:.2e}")
  assert err <= 2e-5
