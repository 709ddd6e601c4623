"""GPU: the spatially partitioned step (BASELINE.json config 5: receiver-owned edge partition +
18 halo exchanges per step) against the unpartitioned step on the same device, same kernels.
P ranks are emulated in one process (partition.EmulatedPartitionedStep): local graphs, local
engines, halo rows copied between the ranks' tables at the exchange points.  Edges keep their
relative order inside every receiver's segment, so results agree to fp32 rounding of the
tile-boundary partial sums (asserted 2e-6 rel-RMSE; typically bit-identical rows)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from graphcast_amd import graphcast as gc          # noqa: E402
from graphcast_amd import partition                # noqa: E402
from oracle import graphcast as ogc                # noqa: E402
from oracle import params as oparams               # noqa: E402


@pytest.fixture(scope="module")
def setup():
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  res, mesh_size, steps = 4.0, 3, 3
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_in, c_out = 183, gc.num_output_channels(gc.TASK_13)
  params = oparams.init_params(c_in, c_out, 512, steps, seed=1, nontrivial=True)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params).init_from_coordinates(lat, lon)
  x = torch.from_numpy(np.random.default_rng(0).standard_normal(
      (len(lat) * len(lon), 2, c_in)).astype(np.float32)).to("cuda:0")
  y = model.forward_grid_node_features(x).clone()
  return dict(model=model, params=params, steps=steps, c_in=c_in, c_out=c_out, x=x, y=y,
              lat=lat, lon=lon, mesh_size=mesh_size)


@pytest.mark.parametrize("n_parts", [2, 3, 8])
def test_partitioned_step_equals_full_step(setup, n_parts):
  # (round 5: each rank's engine is a C++ plan of its LOCAL graphs -- sender tables with a halo suffix,
  #  gc_model_desc.n_*_senders -- and the exchange points are read off the plan's program)
  m = setup["model"]
  step = partition.EmulatedPartitionedStep(
      m.graph_arrays(), setup["params"], m._grid_nodes_lon, m._mesh_nodes_lon, n_parts,
      num_steps=setup["steps"], c_in=setup["c_in"], c_out=setup["c_out"],
      grid_lat=m._grid_nodes_lat, mesh_lat=m._mesh_nodes_lat)          # 2 / 8 parts: hemispheres / octants; 3: bands
  y = step(setup["x"])
  torch.cuda.synchronize()
  assert step.exchanges_per_call == 2 * (2 + setup["steps"])        # batch 2 x (enc + steps + dec)
  diff = (y - setup["y"]).double()
  rel = float(torch.linalg.vector_norm(diff) / torch.linalg.vector_norm(setup["y"].double()))
  print(f"{n_parts} parts: rel diff vs unpartitioned {rel:.2e}, "
        f"halo rows per rank (g2m/mesh/m2g): "
        f"{[ (len(r.halo_g2m.halo_global), len(r.halo_mesh.halo_global), len(r.halo_m2g.halo_global)) for r in step.ranks][:3]}")
  assert torch.isfinite(y).all()
  assert rel < 2e-6


@pytest.mark.parametrize("n_parts", [2, 8])
def test_partitioned_step_in_the_bfloat16_tier(setup, n_parts, monkeypatch):
  """The same partition with every rank's engine in the "bf16" arithmetic (GC_PREC_BF16: bfloat16 row tables, so
  the halo rows exchanged are bfloat16 too), behind blocking exchanges.  Partitioning moves tile boundaries, i.e.
  the order of fp32 partial sums in front of ONE bfloat16 rounding: the two bf16 runs decorrelate at bfloat16
  resolution, and must sit at the same distance from the fp32-grade step."""
  from graphcast_amd import engine
  m = setup["model"]
  full = engine.StepEngine(m.graph_arrays(), setup["params"], num_steps=setup["steps"], c_in=setup["c_in"],
                           c_out=setup["c_out"], device="cuda:0", precision="bf16")
  y_full = full.forward(setup["x"]).clone()
  step = partition.EmulatedPartitionedStep(
      m.graph_arrays(), setup["params"], m._grid_nodes_lon, m._mesh_nodes_lon, n_parts,
      num_steps=setup["steps"], c_in=setup["c_in"], c_out=setup["c_out"], precision="bf16",
      grid_lat=m._grid_nodes_lat, mesh_lat=m._mesh_nodes_lat)
  y = step(setup["x"])
  torch.cuda.synchronize()
  assert torch.isfinite(y).all()
  rel = lambda a, b: float(torch.linalg.vector_norm((a - b).double()) / torch.linalg.vector_norm(b.double()))
  between, d_part, d_full = rel(y, y_full), rel(y, setup["y"]), rel(y_full, setup["y"])
  print(f"bf16 tier, {n_parts} parts: vs the unpartitioned bf16 step {between:.2e}; distance to the fp32-grade step: "
        f"partitioned {d_part:.2e}, unpartitioned {d_full:.2e}")
  assert between < 2e-2
  assert d_part < 1.25 * d_full + 1e-3


def test_partitioned_step_against_oracle(setup):
  """And against the float64 oracle directly (the partition must not hide behind the engine)."""
  graphs = ogc.build_graphs(setup["lat"], setup["lon"], setup["mesh_size"])
  want = ogc.forward(setup["params"], graphs, setup["x"].cpu().numpy(), steps=setup["steps"])
  m = setup["model"]
  step = partition.EmulatedPartitionedStep(
      m.graph_arrays(), setup["params"], m._grid_nodes_lon, m._mesh_nodes_lon, 4,
      num_steps=setup["steps"], c_in=setup["c_in"], c_out=setup["c_out"],
      grid_lat=m._grid_nodes_lat, mesh_lat=m._mesh_nodes_lat)          # quadrants
  got = step(setup["x"]).cpu().numpy().astype(np.float64)
  assert np.linalg.norm(got - want) / np.linalg.norm(want) < 2e-5


def test_distributed_partitioned_step_two_ranks_share_one_gpu(setup, tmp_path):
  """partition.DistributedPartitionedStep itself -- one process per rank, ONE all_to_all_single per
  halo exchange -- executed with world_size 2: both ranks on this GPU, gloo group, host-staged
  exchanger (RCCL refuses two ranks on one device).  Its assembled output equals the unpartitioned
  step and the float64 oracle.  Reference design: utils/gather_scatter_ops.py:102-144,267-283."""
  import os
  import socket
  import subprocess
  import sys
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_partition_gpu_worker.py")
  procs = []
  for rank in range(2):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path)], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
  outs = [p.communicate(timeout=600)[0] for p in procs]
  for p, o in zip(procs, outs):
    assert p.returncode == 0, o[-3000:]
  n_grid = setup["x"].shape[0]
  y = np.full((n_grid, 2, setup["c_out"]), np.nan, dtype=np.float32)
  seen = np.zeros(n_grid, dtype=int)
  for rank in range(2):
    rows = np.load(tmp_path / f"rows_rank{rank}.npy")
    y[rows] = np.load(tmp_path / f"y_rank{rank}.npy")
    seen[rows] += 1
  assert (seen == 1).all()                                         # every grid row owned exactly once
  full = setup["y"].cpu().numpy()
  rel = np.linalg.norm((y - full).astype(np.float64)) / np.linalg.norm(full.astype(np.float64))
  graphs = ogc.build_graphs(setup["lat"], setup["lon"], setup["mesh_size"])
  want = ogc.forward(setup["params"], graphs, setup["x"].cpu().numpy(), steps=setup["steps"])
  rel_oracle = np.linalg.norm(y - want) / np.linalg.norm(want)
  print(f"DistributedPartitionedStep, 2 ranks on one GPU (gloo, host-staged): rel diff vs unpartitioned "
        f"{rel:.2e}, vs float64 oracle {rel_oracle:.2e}")
  assert rel < 2e-6
  assert rel_oracle < 2e-5


def test_rccl_path_executes_on_the_gpu():
  """the `nccl` (RCCL) backend initialised with world_size 1 on the MI355X; DistExchanger.exchange
  through all_to_all_single on DEVICE tensors (no host staging), the overlap-stream pattern, and one
  DistributedPartitionedStep -- in a process of its own (tests/_nccl_ws1_worker.py)."""
  import os
  import socket
  import subprocess
  import sys
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_nccl_ws1_worker.py")
  env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
             HSA_ENABLE_IPC_MODE_LEGACY="0")
  p = subprocess.run([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                     timeout=600)
  assert p.returncode == 0 and "NCCL_WS1_OK" in p.stdout, p.stdout[-3000:]
  print(p.stdout.strip().splitlines()[-1])
