"""GPU parity of the norm-conditioned encoder / decoder (GenCast's reuse of the
two bipartite DeepTypedGraphNets, weathernext1_gen/denoiser.py:303-363, utils/dense.py:360-393).

  * vs tests/golden/gnn_conditioned512.npz -- the reference's own deep_typed_graph_net.py +
    dense.py executed (float64, numpy stand-ins for haiku / jraph / jax) at latent 512 on a graph
    with empty receivers and receivers spanning tile borders: rel-RMSE <= 2e-5 per stage;
  * vs the float64 oracle on a 6 deg / M2 GraphCast-shaped graph pair;
  * the conditioning is per BATCH ELEMENT: swapping the conditioning rows swaps nothing else."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from graphcast_amd import conditioned              # noqa: E402
from oracle import graphcast as ogc                # noqa: E402
from oracle import params as oparams               # noqa: E402
from tests.test_oracle_gnn_golden import conditioned_oracle, load_conditioned512   # noqa: E402

TOL = 2e-5


def rel(got, want):
  return float(np.linalg.norm(np.asarray(got, np.float64) - want) / np.linalg.norm(want))


def _device_run(z, params, precision):
  c_grid, c_mesh, c_edge, c_cond, c_out, latent, _ = (int(v) for v in z["config"])
  graphs = dict(n_grid=z["grid_x"].shape[0], n_mesh=z["mesh_x"].shape[0],
                g2m=dict(senders=z["g2m_senders"], receivers=z["g2m_receivers"], feat=z["g2m_e"]),
                m2g=dict(senders=z["m2g_senders"], receivers=z["m2g_receivers"], feat=z["m2g_e"]))
  net = conditioned.ConditionedEncoderDecoder(graphs, params, c_grid=c_grid, c_mesh=c_mesh, c_cond=c_cond,
                                              c_out=c_out, precision=precision)
  dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda:0")
  cond = dev(z["cond"])
  lat_mesh, lat_grid = net.encode(dev(z["grid_x"]), dev(z["mesh_x"]), cond)
  y = net.decode(lat_mesh, lat_grid, cond)
  torch.cuda.synchronize()
  return net, lat_grid.cpu().numpy(), lat_mesh.cpu().numpy(), y.cpu().numpy()


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_conditioned_nets_match_reference_golden_vectors(golden_dir, precision):
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  z, params = load_conditioned512(golden_dir)
  _, g, m, y = _device_run(z, params, precision)
  errs = dict(enc_grid=rel(g, z["enc_grid"]), enc_mesh=rel(m, z["enc_mesh"]), dec_grid=rel(y, z["dec_grid"]))
  print(f"norm-conditioned encoder/decoder vs reference-executed golden ({precision}): {errs}")
  assert max(errs.values()) <= TOL
  for b in range(y.shape[1]):
    assert rel(y[:, b], z["dec_grid"][:, b]) <= TOL


def test_conditioning_is_per_batch_element(golden_dir):
  z, params = load_conditioned512(golden_dir)
  net, _, _, y = _device_run(z, params, "f16x3")
  zs = {k: z[k] for k in z.files}
  for k in ("grid_x", "mesh_x"):
    zs[k] = z[k][:, ::-1].copy()
  zs["cond"] = z["cond"][::-1].copy()
  _, _, _, y_swapped = _device_run(zs, params, "f16x3")
  np.testing.assert_array_equal(y_swapped[:, ::-1], y)            # bitwise: batch elements are independent
  zs["cond"] = z["cond"].copy()                                   # inputs swapped, conditioning not
  _, _, _, y_mixed = _device_run(zs, params, "f16x3")
  assert rel(y_mixed[:, ::-1], z["dec_grid"]) > 1e-2              # the conditioning matters
  with pytest.raises(ValueError):
    net.encode(torch.zeros((3, 1, 5), device="cuda:0"), torch.zeros((3, 1, 5), device="cuda:0"),
               torch.zeros((1, 16), device="cuda:0"))
  with pytest.raises(TypeError):
    net(torch.zeros(1))


def test_conditioned_nets_match_oracle_on_graphcast_graphs():
  """GraphCast-shaped bipartite graphs (6 deg grid, M2 mesh: 1,860 grid nodes, 162 mesh nodes,
  radius-query encoder edges, containing-triangle decoder edges), batch 3."""
  res, mesh_size = 6.0, 2
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  g = ogc.build_graphs(lat, lon, mesh_size)
  c_grid, c_mesh, c_edge, c_cond, c_out = 86, 3, 4, 16, 83
  params = oparams.init_conditioned_params(c_grid, c_mesh, c_edge, c_cond, c_out, 512, seed=9)
  rng = np.random.default_rng(2)
  batch = 3
  z = dict(config=np.array([c_grid, c_mesh, c_edge, c_cond, c_out, 512, 9]),
           grid_x=rng.standard_normal((g["n_grid"], batch, c_grid)).astype(np.float32),
           mesh_x=np.repeat(np.asarray(g["mesh_node_feat"], np.float32)[:, None, :], batch, axis=1),
           cond=rng.standard_normal((batch, c_cond)).astype(np.float32),
           g2m_e=np.asarray(g["g2m"]["feat"], np.float32), m2g_e=np.asarray(g["m2g"]["feat"], np.float32),
           g2m_senders=g["g2m"]["senders"], g2m_receivers=g["g2m"]["receivers"],
           m2g_senders=g["m2g"]["senders"], m2g_receivers=g["m2g"]["receivers"])
  want_g, want_m, want_y = conditioned_oracle(z, params)

  class Z(dict):
    files = property(lambda self: list(self))
  _, got_g, got_m, got_y = _device_run(Z(z), params, "f16x3")
  errs = dict(enc_grid=rel(got_g, want_g), enc_mesh=rel(got_m, want_m), dec_grid=rel(got_y, want_y))
  print(f"norm-conditioned encoder/decoder vs float64 oracle (6 deg / M2, batch 3): {errs}")
  assert max(errs.values()) <= TOL
