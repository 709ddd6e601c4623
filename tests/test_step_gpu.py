"""GPU parity of the whole encode-process-decode step (StepEngine through the C-ABI)
against the float64 oracle, on graphs small enough for the oracle to finish in
seconds.  Tolerance (BASELINE.json): rel-RMSE <= 1e-4 vs the reference in fp32;
we assert <= 2e-5 against the float64 oracle and report the measured value."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from graphcast_amd import _native as nat            # noqa: E402
from graphcast_amd import graphcast as gc          # noqa: E402
from oracle import graphcast as ogc                # noqa: E402
from oracle import params as oparams               # noqa: E402

REL_RMSE_TOL = 2e-5


def rel_rmse(got, want):
  return float(np.linalg.norm(np.asarray(got, np.float64) - want) / np.linalg.norm(want))


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def small(request):
  # "f16x3": the shipped default (every launch in the half-N formulation, two persistent workgroups per CU);
  # "f32": the exact-fp32 chunked kernel.  (Round 5 retired the chunked f16x3 kernels and the "bf16gemm" tier.)
  precision = request.param
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  res, mesh_size, steps = 4.0, 3, 3
  lat = np.arange(-90, 90 + res / 2, res)
  lon = np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_in, c_out = 183, gc.num_output_channels(gc.TASK_13)
  params = oparams.init_params(c_in, c_out, 512, steps, seed=1, nontrivial=True)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params, precision=precision).init_from_coordinates(lat, lon)
  graphs = ogc.build_graphs(lat, lon, mesh_size)
  return dict(model=model, precision=precision, graphs=graphs, params=params, steps=steps, c_in=c_in, c_out=c_out)


def test_product_graphs_equal_oracle_graphs(small):
  got, want = small["model"].graph_arrays(), small["graphs"]
  for k in ("g2m", "mesh", "m2g"):
    np.testing.assert_array_equal(got[k]["senders"], want[k]["senders"])
    np.testing.assert_array_equal(got[k]["receivers"], want[k]["receivers"])
    np.testing.assert_allclose(got[k]["feat"], want[k]["feat"], atol=1e-12)


@pytest.mark.parametrize("batch", [1, 2])
def test_step_matches_oracle(small, batch):
  rng = np.random.default_rng(batch)
  x = rng.standard_normal((small["graphs"]["n_grid"], batch, small["c_in"])).astype(np.float32)
  want = ogc.forward(small["params"], small["graphs"], x, steps=small["steps"], dtype=np.float64)
  y = small["model"].forward_grid_node_features(torch.from_numpy(x).to("cuda:0"))
  torch.cuda.synchronize()
  got = y.cpu().numpy()
  assert got.shape == want.shape == (small["graphs"]["n_grid"], batch, small["c_out"])
  err = rel_rmse(got, want)
  print(f"step rel-RMSE vs float64 oracle (batch={batch}, {small['precision']}): {err:.3e}")
  assert np.isfinite(got).all()
  tol = REL_RMSE_TOL
  assert err <= tol
  for b in range(batch):                        # per batch element too
    assert rel_rmse(got[:, b], want[:, b]) <= tol


def test_step_is_deterministic_and_batch_independent(small):
  rng = np.random.default_rng(9)
  n = small["graphs"]["n_grid"]
  x = torch.from_numpy(rng.standard_normal((n, 2, small["c_in"])).astype(np.float32)).to("cuda:0")
  m = small["model"]
  y1 = m.forward_grid_node_features(x).clone()
  y2 = m.forward_grid_node_features(x).clone()
  assert torch.equal(y1, y2)                    # bitwise: no float atomics
  y_single = m.forward_grid_node_features(x[:, 1:2].contiguous())
  assert torch.equal(y_single[:, 0], y1[:, 1])  # batch is a pure broadcast axis (graphcast.py:726-730)


def test_step_gives_the_same_bits_in_the_one_workgroup_per_cu_forms(small):
  """The big node-side launches of the headline size (grid embedder, grid-node update, decoder node update + output
  MLP: no gather, no segment-sum) run as ONE eight-wave workgroup per CU -- round 5: in the WIDE form (eight multiplying
  waves, 128 rows against one weight ring; csrc/rowmlp_half.inc: rowmlp16w_kernel), round 4: the helper form (four
  multiplying + four staging waves).  Here, where no launch reaches the row threshold, the engine knobs put EVERY such
  launch into either form: the whole step's output is bitwise the four-wave form's, batch 1 and 2."""
  if small["precision"] != "f16x3":
    pytest.skip("the eight-wave forms are GC_PREC_F16X3 kernels")
  rng = np.random.default_rng(21)
  n = small["graphs"]["n_grid"]
  x = torch.from_numpy(rng.standard_normal((n, 2, small["c_in"])).astype(np.float32)).to("cuda:0")
  want = small["model"].forward_grid_node_features(x).clone()
  cfg = small["model"]._model_config
  lat = np.arange(-90, 90 + cfg.resolution / 2, cfg.resolution)
  lon = np.arange(0, 360, cfg.resolution)
  for knob in ("wide_min_rows", "helpers_min_rows"):
    m = gc.GraphCast(cfg, gc.TASK_13, params=small["params"], precision="f16x3").init_from_coordinates(lat, lon)
    setattr(m._get_engine(small["c_in"]), knob, 1)            # (read when the launch program is recorded: the first step)
    y = m.forward_grid_node_features(x)
    torch.cuda.synchronize()
    ops, _ = m._engine.bind(x)
    marked = [ops[k].mlp.flags & (nat.WG_WIDE | nat.WG_HELPERS) for k in range(len(ops)) if ops[k].kind == nat.OP_ROWMLP]
    assert sum(f == (nat.WG_WIDE if knob == "wide_min_rows" else nat.WG_HELPERS) for f in marked) >= 3, (knob, marked)
    assert torch.equal(y, want), knob
  # round 6: ... and with EVERY edge update -- the one-pass ones (encoder, decoder, processor step 0) and the two-pass
  # ones (the other processor steps) -- in the wide form too: segment-sums per 64-row sub-tile, ten of sixteen parked
  # n-blocks in LDS; together with the node-side launches above the whole step then runs one workgroup per CU
  m = gc.GraphCast(cfg, gc.TASK_13, params=small["params"], precision="f16x3").init_from_coordinates(lat, lon)
  e = m._get_engine(small["c_in"])
  e.wide_edges, e.wide_min_rows = 3, 1
  y = m.forward_grid_node_features(x)
  torch.cuda.synchronize()
  ops, _ = e.bind(x)
  edge_ops = [ops[k].mlp for k in range(len(ops)) if ops[k].kind == nat.OP_ROWMLP and ops[k].mlp.seg]
  batch = x.shape[1]                           # (the program runs the batch elements one after another)
  assert len(edge_ops) == batch * (2 + cfg.gnn_msg_steps) and all(o.flags & nat.WG_WIDE for o in edge_ops)
  assert sum(bool(o.flags & nat.W2_NATURAL) for o in edge_ops) == batch * 3
  assert torch.equal(y, want)


def test_missing_params_and_bad_shapes_raise(small):
  cfg = gc.ModelConfig(resolution=4.0, mesh_size=3, latent_size=512, gnn_msg_steps=3,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  empty = gc.GraphCast(cfg, gc.TASK_13)
  with pytest.raises(ValueError):
    empty.forward_grid_node_features(torch.zeros((4, 1, 183), device="cuda:0"))
  with pytest.raises(ValueError):
    small["model"].forward_grid_node_features(torch.zeros((5, 1, 183), device="cuda:0"))


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_step_matches_reference_golden_vectors(golden_dir, precision):
  """HIP path vs tests/golden/gnn_latent512.npz = the reference's own graphcast.py /
  deep_typed_graph_net.py / typed_graph_net.py executed (float64) on numpy stand-ins for
  haiku / jraph / jax (tests/golden/make_golden.py).  fp32 tolerance: rel-RMSE <= 2e-5."""
  from tests.test_oracle_gnn_golden import load_latent512
  z, params, steps = load_latent512(golden_dir)
  res, mesh_size = float(z["config"][0]), int(z["config"][1])
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params,
                       precision=precision).init_from_coordinates(z["lat"], z["lon"])
  y = model.forward_grid_node_features(torch.from_numpy(z["x"]).to("cuda:0"))
  torch.cuda.synchronize()
  err = rel_rmse(y.cpu().numpy(), z["out"])
  print(f"step rel-RMSE vs reference-executed golden vectors ({precision}): {err:.3e}")
  assert err <= REL_RMSE_TOL


def test_step_with_injected_mesh2grid_faces_matches_oracle():
  """SURVEY 8 a6 escape hatch: mesh2grid face indices supplied from outside (what a host with
  trimesh would compute, reference grid_mesh_connectivity.py:114-119) drive the product's and the
  oracle's decoder graph alike; the step on them matches the float64 oracle."""
  from graphcast_amd import grid_mesh_connectivity as gmc
  from graphcast_amd import icosahedral_mesh as im
  res, mesh_size, steps = 6.0, 2, 2
  lat = np.arange(-90, 90 + res / 2, res).astype(np.float32)
  lon = np.arange(0, 360, res).astype(np.float32)
  mesh = im.get_hierarchy_of_triangular_meshes_for_sphere(splits=mesh_size)[-1]
  faces = gmc._nearest_face_on_surface(gmc._grid_lat_lon_to_coordinates(lat, lon).reshape([-1, 3]), mesh)
  rng = np.random.default_rng(3)
  pick = rng.choice(len(faces), size=40, replace=False)        # a different (valid) face for 40 points
  faces = faces.copy()
  faces[pick] = (faces[pick] + 1 + rng.integers(0, 5, size=40)) % len(mesh.faces)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_in, c_out = 183, gc.num_output_channels(gc.TASK_13)
  params = oparams.init_params(c_in, c_out, 512, steps, seed=4, nontrivial=True)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params,
                       mesh2grid_face_indices=faces).init_from_coordinates(lat, lon)
  graphs = ogc.build_graphs(lat, lon, mesh_size, m2g_face_indices=faces)
  np.testing.assert_array_equal(model.graph_arrays()["m2g"]["senders"], graphs["m2g"]["senders"])
  x = rng.standard_normal((graphs["n_grid"], 1, c_in)).astype(np.float32)
  want = ogc.forward(params, graphs, x, steps=steps, dtype=np.float64)
  got = model.forward_grid_node_features(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
  base = gc.GraphCast(cfg, gc.TASK_13, params=params).init_from_coordinates(lat, lon)
  differs = base.forward_grid_node_features(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
  err = rel_rmse(got, want)
  print(f"injected mesh2grid faces: rel-RMSE vs float64 oracle {err:.2e}; "
        f"vs the restated-rule graph {rel_rmse(differs, want):.2e}")
  assert err <= REL_RMSE_TOL
  assert rel_rmse(differs, want) > 10 * REL_RMSE_TOL           # the injection really changed the graph


def test_f16x3_range_guard_raises_instead_of_wrong_numbers(small):
  """the f16x3 split is exact only for |x| <= 65504; un-normalised fields (geopotential ~5e5)
  must raise, never come back silently wrong.  Inside the range (values up to 6e4) the step still meets the budget;
  f32 (the reference's arithmetic) takes anything."""
  from graphcast_amd import _native as nat
  rng = np.random.default_rng(5)
  n_grid, c_in = small["graphs"]["n_grid"], small["c_in"]
  x = rng.standard_normal((n_grid, 1, c_in)).astype(np.float32)
  big = x.copy()
  big[:, 0, 7] = np.abs(big[:, 0, 7]) * 1e4 + 5e5                 # one geopotential-scale channel
  big[3, 0, c_in - 1] = -7e4                                       # ... and one value in the 32-column tail
  model = small["model"]
  engine = model._get_engine(c_in)
  if small["precision"] == "f16x3" and engine.half:
    model.forward_grid_node_features(torch.from_numpy(big).to("cuda:0"))
    with pytest.raises(nat.GcastRangeError, match="65504"):
      engine.check_range()
    engine.check_range()                                           # the flag is cleared by the raise
    only_tail = x.copy()
    only_tail[3, 0, c_in - 1] = -7e4
    model.forward_grid_node_features(torch.from_numpy(only_tail).to("cuda:0"))
    with pytest.raises(nat.GcastRangeError):
      engine.check_range()
  elif small["precision"] == "f32":
    y = model.forward_grid_node_features(torch.from_numpy(big).to("cuda:0")).cpu().numpy()
    want = ogc.forward(small["params"], small["graphs"], big, steps=small["steps"], dtype=np.float64)
    assert rel_rmse(y, want) <= REL_RMSE_TOL
  if small["precision"] in ("f16x3", "f32"):
    ok = x.copy()
    ok[:, 0, 7] = np.abs(ok[:, 0, 7]) * 1e3 + 5e4                  # large but inside the exact range (< 65504)
    ok[:, 0, 7] = np.minimum(ok[:, 0, 7], 6.5e4)
    y = model.forward_grid_node_features(torch.from_numpy(ok).to("cuda:0")).cpu().numpy()
    engine.check_range()
    want = ogc.forward(small["params"], small["graphs"], ok, steps=small["steps"], dtype=np.float64)
    err = rel_rmse(y, want)
    print(f"in-range large inputs ({small['precision']}): rel-RMSE {err:.2e}")
    assert err <= 1e-4


def _params_with_large_g2m_messages(c_in, c_out, steps, seed=6):
  """Every grid2mesh message e' = LN(...) stays inside the f16x3 range (|e'| ~ 2e4 in four columns), but a mesh
  node SUMS 4 .. 40 of them: the aggregate -- the layer-1 operand of the encoder's mesh-node update -- does not."""
  params = {m: {k: np.array(v) for k, v in leaves.items()}
            for m, leaves in oparams.init_params(c_in, c_out, 512, steps, seed=seed, nontrivial=True).items()}
  ln = params["grid2mesh_gnn/~_networks_builder/processor_edges_0_grid2mesh_layer_norm"]
  ln["offset"][[3, 100, 257, 511]] = [2.0e4, -2.1e4, 1.9e4, 2.2e4]
  return params


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_aggregate_beyond_the_f16x3_range_raises_instead_of_saturating(precision):
  """the launches whose layer-1 operand is an AGGREGATE (encoder mesh-node update, processor
  node updates, decoder grid-node update) carry the range flag too -- a sum over up to 3,753 edges of LayerNorm
  outputs is not a LayerNorm output (the reference up-casts this very sum to fp32 because it is large,
  graphcast.py:215).  With in-range inputs and in-range messages whose SUM exceeds 65504 the f16x3 step raises;
  f32 computes it."""
  from graphcast_amd import _native as nat
  res, mesh_size, steps = 4.0, 3, 2
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_in, c_out = 183, gc.num_output_channels(gc.TASK_13)
  params = _params_with_large_g2m_messages(c_in, c_out, steps)
  graphs = ogc.build_graphs(lat, lon, mesh_size)
  x = np.random.default_rng(2).standard_normal((graphs["n_grid"], 1, c_in)).astype(np.float32)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params, precision=precision).init_from_coordinates(lat, lon)
  engine = model._get_engine(c_in)
  xt = torch.from_numpy(x).to("cuda:0")
  engine.run_until(xt, "enc_node_mesh")
  torch.cuda.synchronize()
  top = float(engine.agg_mesh.abs().max())
  assert top > 65504.0, f"the test's aggregate must leave the range (max |agg| = {top:.3g})"
  y = model.forward_grid_node_features(xt)
  if precision == "f16x3":
    assert engine.half
    with pytest.raises(nat.GcastRangeError, match="65504"):
      engine.check_range()
    engine.check_range()                       # cleared by the raise
  else:
    engine.check_range()
    want = ogc.forward(params, graphs, x, steps=steps, dtype=np.float64)
    err = rel_rmse(y.cpu().numpy(), want)
    print(f"aggregate up to {top:.3g} (f32): rel-RMSE {err:.2e}")
    assert err <= 1e-4


def test_checkpoint_file_to_hip_step(tmp_path):
  """a CheckPoint in the reference's .npz layout (utils/checkpoint.py:26-54) written to a
  FILE, read back with checkpoint.load, handed to GraphCast -> the HIP step equals the step on the in-memory
  parameters bit for bit (and the configs survive the round trip)."""
  import dataclasses
  from graphcast_amd import checkpoint
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  res, mesh_size, steps = 6.0, 2, 2
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_in, c_out = 183, gc.num_output_channels(gc.TASK_13)
  params = {m: {k: np.asarray(v, np.float32) for k, v in leaves.items()}
            for m, leaves in oparams.init_params(c_in, c_out, 512, steps, seed=11, nontrivial=True).items()}
  ckpt = gc.CheckPoint(params=params, model_config=cfg, task_config=gc.TASK_13, description="round-4 test", license="none")
  path = tmp_path / "graphcast_toy.npz"
  with open(path, "wb") as f:
    checkpoint.dump(f, ckpt)
  with open(path, "rb") as f:
    back = checkpoint.load(f, gc.CheckPoint)
  assert back.model_config == cfg and back.task_config == gc.TASK_13 and back.description == "round-4 test"
  assert sorted(back.params) == sorted(params)
  x = torch.from_numpy(np.random.default_rng(2).standard_normal((len(lat) * len(lon), 1, c_in)).astype(np.float32)).to("cuda:0")
  y_mem = gc.GraphCast(cfg, gc.TASK_13, params=params).init_from_coordinates(lat, lon).forward_grid_node_features(x)
  loaded = gc.GraphCast(back.model_config, back.task_config)
  loaded.load_params(back.params)
  y_file = loaded.init_from_coordinates(lat, lon).forward_grid_node_features(x)
  torch.cuda.synchronize()
  assert torch.isfinite(y_file).all()
  assert torch.equal(y_file, y_mem)
