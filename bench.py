#!/usr/bin/env python
"""bench.py -- 6-h GraphCast steps/s at 0.25 deg / 37 levels on MI355X.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one encode-process-decode pass (Grid2Mesh encoder, 16-step multi-mesh
processor, Mesh2Grid decoder) over one synthetic 0.25 deg / 37-level state that
is already resident in HBM: x [1,038,240, 1, 471] fp32 -> y [1,038,240, 1, 227].
With N > 1 every rank advances its own ensemble member (BASELINE.json config 4;
members never interact in the forward pass, reference rollout.py:220-283), so
scaling is weak and `value` = N*K / max-over-ranks time.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: `roofline` for the dominant kernel and `cpu_baseline` (the oracle's CPU restatement,
on torch's CPU kernels, timed on the host cores on a bounded sample, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0      # dense f16/bf16 MFMA peak (same guide; 2:1-sparse figures excluded).  The guide MEASURES
                                   # 2495 TF = 99.8 % of it (32x32x16 MFMA stream): `roofline.frac` is priced against this.
# A SECONDARY reference, not the roofline: what OUR streaming micro-benchmark sustained on this part in round 1 with
# random (non-zero) f16 operands on all 256 CUs over 120-150 ms -- a bare v_mfma_f32_32x32x16_f16 stream 69 % of peak,
# with the weight stream from L2 + row fragments from LDS 55 % (profiles/r01_co21_sustained_mfma_random_vs_zero.txt;
# with ZERO operands the same loops reach 98 % / 88 %, i.e. the guide's figure: the gap is the power-limited clock under
# real data).  `mfma_issue_frac_of_sustained` divides by the second number; nobody should quote it as the roofline
# fraction.
SUSTAINED_F16_MFMA_TFLOPS = {"mfma_only": 0.69 * 2500.0, "mfma_with_operand_streams": 0.55 * 2500.0,
                             "guide_measured_peak": 2495.0}
LATENT = 512

CONFIGS = {
    # name: (resolution, mesh_size, levels, gnn steps)
    "0.25deg_37L_M6": (0.25, 6, 37, 16),
    "1deg_13L_M5": (1.0, 5, 13, 16),
    "2deg_13L_M4": (2.0, 4, 13, 16),         # cpu_baseline sample
    "4deg_13L_M3": (4.0, 3, 13, 3),          # plumbing / CI only / cpu_baseline sample
}


def flops_as_written(n_grid, n_mesh, e_g2m, e_mesh, e_m2g, c_in, c_out, steps, d=LATENT):
  """2*MAC of every MLP the reference executes per step, dead code removed."""
  mlp = lambda rows, k, n_out: 2.0 * rows * (k * d + d * n_out)
  return (mlp(n_grid, c_in + 3, d) + mlp(n_mesh, c_in + 3, d) + mlp(e_g2m, 4, d)
          + mlp(e_g2m, 3 * d, d) + mlp(n_mesh, 2 * d, d) + mlp(n_grid, d, d)
          + mlp(e_mesh, 4, d) + steps * (mlp(e_mesh, 3 * d, d) + mlp(n_mesh, 2 * d, d))
          + mlp(e_m2g, 4, d) + mlp(e_m2g, 3 * d, d) + mlp(n_grid, 2 * d, d)
          + mlp(n_grid, d, c_out))


def _stamped_profile(name, stage):
  """Stage `stage` of profiles/<name> if that summary was measured on the build that is loaded NOW: the summaries
  written by scripts/pmc_by_stage.py / sq_by_stage.py carry `_stamp.src` = the hash of the library sources they
  were collected on; the loaded library reports the hash it was compiled from (gc_build_info ";src=").  bench.py
  cannot run rocprofv3 on itself, so counters are attached from the profile of the SAME command -- and only then.
  -> (stage dict | None, reason | None)"""
  from graphcast_amd import _native as nat
  path = os.path.join(ROOT, "profiles", name)
  try:
    with open(path) as f:
      table = json.load(f)
  except (OSError, ValueError):
    return None, f"profiles/{name} is not in the tree"
  want = nat.loaded_source_hash()
  have = (table.get("_stamp") or {}).get("src")
  if have is None or want is None or have != want:
    return None, (f"profiles/{name} was collected on sources {have}, the loaded library was built from {want}: "
                  "re-collect the counter passes (scripts/final_session.sh)")
  if stage not in table:
    return None, f"profiles/{name} has no stage {stage}"
  return dict(table[stage], env=(table["_stamp"].get("env") or {})), None


def _profile_name(kind, precision_key):
  """The committed counter summary of an arithmetic: the f16x3 headline and (round 6) the bf16 tier."""
  if precision_key.startswith("f16x3h"):
    return f"current_{kind}_by_stage.json"
  if precision_key.startswith("bf16"):
    return f"current_{kind}_by_stage_bf16.json"
  return None


def measured_traffic(precision_key, stage):
  """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
  WRITE_SIZE passes of the same bench command, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950),
  keyed to the loaded build (see _stamped_profile)."""
  name = _profile_name("pmc", precision_key)
  if name is None:
    return None, "no counter profile is kept for this arithmetic / formulation"
  st, why = _stamped_profile(name, stage)
  if st is None:
    return None, why
  return {"bytes_per_launch": st["traffic_bytes_per_launch"], "fetch_bytes_per_launch": st["fetch_bytes_per_launch"],
          "write_bytes_per_launch": st["write_bytes_per_launch"], "algorithmic_bytes": st["algorithmic_bytes_per_launch"],
          "traffic_over_algorithmic": st["traffic_over_algorithmic"],
          "source": f"profiles/{name} (scripts/pmc_by_stage.py; L2 <-> fabric boundary: Infinity-Cache hits "
                    "are counted)", "env": st["env"]}, None


def measured_mfma_busy(precision_key, stage):
  """MFMA pipe busy fraction (and wave wait fractions) of the dominant launch from the committed rocprofv3 SQ counter
  passes of the SAME command, keyed to the loaded build (see _stamped_profile)."""
  name = _profile_name("sq", precision_key)
  if name is None:
    return None, "no counter profile is kept for this arithmetic / formulation"
  st, why = _stamped_profile(name, stage)
  if st is None:
    return None, why
  return {"mfma_busy_per_simd": st.get("mfma_busy_per_simd"), "wave_waiting": st.get("wave_waiting"),
          "lds_bank_conflict": st.get("lds_bank_conflict"),
          "source": f"profiles/{name} (scripts/sq_by_stage.py)", "env": st["env"]}, None


def fast_params(c_in, c_out, steps, seed=1):
  """Random-init weights of the architecture in the reference's haiku layout."""
  from graphcast_amd import params as gparams
  return gparams.random_params(c_in, c_out, LATENT, steps, seed=seed)


def op_flops(op, c_out_exec=240):
  """FLOPs one fused launch executes (2*MAC of its GEMMs over its logical rows)."""
  from graphcast_amd import _native as nat
  if op.kind != nat.OP_ROWMLP:
    return 0.0
  m = op.mlp
  f = 2.0 * m.n_rows * (m.k0 + m.k1) * LATENT
  if m.mode == nat.MODE_MLP_LN:
    f += 2.0 * m.n_rows * LATENT * LATENT
  elif m.mode == nat.MODE_MLP_OUT:
    f += 2.0 * m.n_rows * LATENT * c_out_exec
  for k in range(m.n_chain):                      # Linear layers chained onto the launch (gc_chain_stage)
    f += 2.0 * m.n_rows * LATENT * (c_out_exec if m.chain[k].kind == nat.CHAIN_NARROW else LATENT)
  return f


def op_weight_stream_bytes(op):
  """Bytes of packed weights one half-N launch streams L2 -> LDS (LDS-DMA): every 64-row tile re-streams the launch's
  whole weight set as 16 KiB quarters -- 4 per layer-1 K chunk, 64 for a 512-wide layer 2 / chained stage (two
  passes of 32), 32 for the narrow output stage; the one-pass edge updates stream W2 once (64 quarters)."""
  from graphcast_amd import _native as nat
  if op.kind != nat.OP_ROWMLP or op.mlp.layout != nat.LAYOUT_HALF or op.mlp.prec != nat.PREC_F16X3:
    return 0.0
  m = op.mlp
  quarters = 4 * ((m.k0 + m.k1) // 32)
  if m.mode == nat.MODE_MLP_LN:
    quarters += 64
  elif m.mode == nat.MODE_MLP_OUT:
    quarters += 32
  for k in range(m.n_chain):
    quarters += 32 if m.chain[k].kind == nat.CHAIN_NARROW else 64
  tiles = (m.n_rows + 63) // 64
  return float(tiles) * quarters * 16384.0


def _oracle_sample(cfg_name, c_in, c_out, steps):
  from oracle import graphcast as ogc
  res, mesh_size, _, _ = CONFIGS[cfg_name]
  lat = np.arange(-90, 90 + res / 2, res)
  lon = np.arange(0, 360, res)
  graphs = ogc.build_graphs(lat, lon, mesh_size)
  x = np.random.default_rng(0).standard_normal((graphs["n_grid"], 1, c_in)).astype(np.float32)
  f_sample = flops_as_written(graphs["n_grid"], graphs["n_mesh"], len(graphs["g2m"]["senders"]),
                              len(graphs["mesh"]["senders"]), len(graphs["m2g"]["senders"]),
                              c_in, c_out, steps)
  return graphs, x, f_sample


def _time_step(backend, params, graphs, x, steps):
  t0 = time.perf_counter()
  if backend == "torch":
    from oracle import torch_cpu
    torch_cpu.forward(params, graphs, x, steps)
  else:
    from oracle import graphcast as ogc
    ogc.forward(params, graphs, x, steps=steps, dtype=np.float32)
  return time.perf_counter() - t0


def cpu_baseline(c_in, c_out, steps, f_full, full_graphs=None, full_x=None, budget_s=240.0):
  """Times the CPU restatement of the reference step (fp32, as written: concat -> MLP -> LayerNorm,
  scatter-add; JAX is not installable) on the GPU box's host cores.

  Sample: the same architecture (0.25deg/37L channel widths, `steps` processor steps) on the
  1deg/M5 graph -- 4.4 TFLOP as written -- timed with BOTH CPU back ends of the oracle: torch's
  CPU kernels (oracle/torch_cpu.py; threads = physical cores) and numpy (oracle/gnn.py; BLAS
  threads as configured, elementwise ops single-threaded).  The smaller 4deg/M3 graph only
  warms the thread pools up; nothing is extrapolated from it.  If the faster back end predicts
  that ONE FULL 0.25deg step fits the time budget it is timed too and becomes the reported
  value (no extrapolation at all); otherwise the 1deg/M5 time is scaled by the as-written FLOP
  ratio (6.7x)."""
  from oracle import torch_cpu
  params = fast_params(c_in, c_out, steps)
  threads = torch_cpu.set_threads()
  warm_graphs, warm_x, _ = _oracle_sample("4deg_13L_M3", c_in, c_out, steps)
  _time_step("torch", params, warm_graphs, warm_x, steps)
  graphs, x, f_sample = _oracle_sample("1deg_13L_M5", c_in, c_out, steps)
  rates, secs = {}, {}
  secs["torch"] = _time_step("torch", params, graphs, x, steps)
  rates["torch"] = f_sample / secs["torch"] / 1e9
  _time_step("numpy", params, warm_graphs, warm_x, steps)
  # (the numpy back end is only timed when the whole leg stays bounded: the full step below is the value)
  secs["numpy"] = _time_step("numpy", params, graphs, x, steps) if secs["torch"] < 12.0 else None
  rates["numpy"] = f_sample / secs["numpy"] / 1e9 if secs["numpy"] else None
  best = "torch" if (rates["numpy"] is None or rates["torch"] >= rates["numpy"]) else "numpy"
  est_full = secs[best] * f_full / f_sample
  measured_full = None
  if best == "torch" and full_graphs is not None and est_full <= budget_s:
    measured_full = _time_step("torch", params, full_graphs, full_x, steps)
  value_s = measured_full if measured_full is not None else est_full
  fmt = lambda b: (f"{b} {secs[b]:.1f} s = {rates[b]:.0f} GFLOP/s" if secs[b] else f"{b} skipped (torch leg took > 30 s)")
  sample = (f"fp32 CPU restatement of the reference step, as written (JAX not installable), 0.25deg/37L "
            f"channel widths on the 1deg/M5 graph = {f_sample / 1e12:.2f} TFLOP: {fmt('torch')} on "
            f"{threads} threads (physical cores), {fmt('numpy')}; ")
  if measured_full is not None:
    sample += (f"then ONE FULL 0.25deg/37L/M6 step ({f_full / 1e12:.1f} TFLOP as written) timed with torch: "
               f"{measured_full:.1f} s = {f_full / measured_full / 1e9:.0f} GFLOP/s -- the reported value, "
               f"no extrapolation")
  else:
    sample += (f"reported value = the {best} time scaled by the as-written FLOP ratio "
               f"{f_full / f_sample:.2f} to one 0.25deg step ({est_full:.0f} s)")
  return {
      "value": 1.0 / value_s, "unit": "steps/s", "cores": threads, "kind": "port",
      "sample": sample, "seconds_per_step": value_s,
      "gflops": {"torch_1deg": rates["torch"], "numpy_1deg": rates["numpy"],
                 "torch_full_step": (f_full / measured_full / 1e9) if measured_full else None},
      "sample_seconds": secs[best], "extrapolated": measured_full is None}


_REAL_STDOUT = None


def emit_line(text):
  """The bench line, to the process's REAL stdout (see main: fd 1 itself points at stderr)."""
  sys.stdout.flush()
  if _REAL_STDOUT is None:
    print(text, flush=True)
  else:
    os.write(_REAL_STDOUT, (text + "\n").encode())


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--warmup", type=int, default=1)
  ap.add_argument("--config", default="0.25deg_37L_M6", choices=sorted(CONFIGS))
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--op-timing-iters", type=int, default=2)
  ap.add_argument("--precision", default=None, choices=["f16x3", "f32", "bf16"],
                  help="GEMM arithmetic (include/gcast.h gc_precision); default: engine default")
  ap.add_argument("--no-cross-check", action="store_true",
                  help="skip the full-size f16x3-vs-f32-MFMA agreement check (N = 1 only)")
  ap.add_argument("--mode", default="ensemble", choices=["ensemble", "partition"],
                  help="what N > 1 GPUs do: 'ensemble' = one member per GPU, no collective (BASELINE.json config 4, "
                       "weak scaling; the default and the driver's SCALE run); 'partition' = ONE step split over the N "
                       "GPUs by octant, receiver-owned edges, 18 halo exchanges per step, one RCCL all_to_all_single "
                       "each (config 5, strong scaling)")
  ap.add_argument("--rollout-steps", type=int, default=40,
                  help="N = 1: additionally (outside the timed region) time an autoregressive rollout of this many 6-h "
                       "steps with the state resident in HBM -- BASELINE.json config 3: 40 = ten days -- and report it "
                       "as `rollout` in the line; 0 = skip")
  ap.add_argument("--single-process", action="store_true",
                  help="the SECOND launch form of --gpus N (round 6): ONE process drives N engines, one per device -- the "
                       "reference's pmap_devices form (utils/rollout.py:196-283; rollout.chunked_prediction_generator("
                       "pmap_devices=...) here) -- instead of one process per GPU.  Devices are cuda:0 .. cuda:N-1, wrapping "
                       "around the visible ones (a 1-GPU box runs N engines on N streams of its one device)")
  args = ap.parse_args()

  # stdout carries ONE JSON line and nothing else.  RCCL prints a version banner to the C-level stdout of every process
  # that creates a communicator (plain printf, flushed at exit -- i.e. AFTER anything Python prints; NCCL_DEBUG_FILE
  # does not redirect it; profiles/r05_s1_*, r05_s5_*): file descriptor 1 is pointed at stderr for the life of the
  # process and the line is written to the saved descriptor at the very end (emit_line).
  sys.stdout.flush()
  global _REAL_STDOUT
  _REAL_STDOUT = os.dup(1)
  os.dup2(2, 1)

  import torch
  import torch.distributed as dist
  from graphcast_amd import _native as nat
  from graphcast_amd import engine as eng
  from graphcast_amd import graphcast as gc

  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if args.single_process:
    if world != 1:
      raise SystemExit("--single-process is ONE process driving --gpus N devices: launch it without torch.distributed.run")
    return single_process_main(args)
  if world != args.gpus:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU "
                     "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N)")
  torch.cuda.set_device(local_rank)
  device = f"cuda:{local_rank}"
  # Launched by torch.distributed.run (RANK / WORLD_SIZE / MASTER_* in the environment) the RCCL group is
  # initialised whatever N is -- the N = 1 point of a scaling run goes through the same init, barrier and
  # max-over-ranks path as N = 8.  A bare `python bench.py` (no rendezvous variables) stays single-process.
  distributed = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
  if distributed:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=rank, world_size=world,
                            device_id=torch.device(device))
  if args.mode == "partition":
    return partition_main(args, rank, world, device, distributed)

  res, mesh_size, levels, gnn_steps = CONFIGS[args.config]
  task = {37: gc.TASK, 13: gc.TASK_13}[levels]
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * levels) + 2 * 5 + 2 + 5    # 2 input frames + forcings(2 frames) + statics + target-time forcings
  lat = np.arange(-90, 90 + res / 2, res)
  lon = np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=LATENT,
                       gnn_msg_steps=gnn_steps, hidden_layers=1,
                       radius_query_fraction_edge_length=0.6)
  t_setup = time.perf_counter()
  params = fast_params(c_in, c_out, gnn_steps)
  model = gc.GraphCast(cfg, task, params=params, device=device, precision=args.precision)
  model.init_from_coordinates(lat, lon)
  g = model.graph_arrays()
  n_grid = g["n_grid"]
  # ensemble member `rank`: its own synthetic (already normalised) state, resident in HBM
  x = torch.from_numpy(np.random.default_rng(rank).standard_normal(
      (n_grid, 1, c_in), dtype=np.float32)).to(device)
  y = torch.empty((n_grid, 1, c_out), dtype=torch.float32, device=device)
  model.forward_grid_node_features(x, y)          # builds the engine + folds constants
  torch.cuda.synchronize()
  t_setup = time.perf_counter() - t_setup
  engine = model._engine

  def barrier():
    if distributed:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(args.warmup):
    engine(x, y)
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    engine(x, y)
  barrier()
  elapsed = time.perf_counter() - t0
  if distributed:
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  engine.check_range()          # (outside the timed region: the inputs were inside the f16x3 arithmetic's exact range)
  finite = bool(torch.isfinite(y).all().item())
  precision = engine.precision

  # Full-size parity property (outside the timed region): the two arithmetic modes -- exact fp32
  # MFMA and the 3 x f16 split -- must agree on the whole 0.25 deg output far inside the 1e-4
  # budget.  (The float64 oracle pins both at sizes it finishes in seconds: tests/.)
  cross = None
  if rank == 0 and world == 1 and not args.no_cross_check:
    other = "f32" if precision == "f16x3" else "f16x3"
    ref_engine = eng.StepEngine(g, params, num_steps=gnn_steps, c_in=c_in, c_out=c_out,
                                device=device, precision=other)
    y_ref = ref_engine(x)
    torch.cuda.synchronize()
    num = torch.linalg.vector_norm((y - y_ref).double())
    den = torch.linalg.vector_norm(y_ref.double())
    cross = {"against": other, "rel_rmse": float((num / den).item()),
             "max_abs_diff": float((y - y_ref).abs().max().item())}
    del ref_engine, y_ref
    torch.cuda.empty_cache()

  if rank == 0:
    ms_per_step = 1e3 * elapsed / args.steps
    f_alg = flops_as_written(n_grid, g["n_mesh"], len(g["g2m"]["senders"]),
                             len(g["mesh"]["senders"]), len(g["m2g"]["senders"]),
                             c_in, c_out, gnn_steps)
    per_stage = stage_table(engine, x, y, args.op_timing_iters)
    dominant = max(per_stage, key=lambda s: per_stage[s]["ms"])
    dom = per_stage[dominant]
    achieved = dom["tflop"] / (dom["ms"] / 1e3) if dom["ms"] > 0 else 0.0
    executed_tflop = sum(s["tflop"] for s in per_stage.values())
    split = precision == "f16x3"
    peak = PEAK_FP32_MFMA_TFLOPS if precision == "f32" else PEAK_F16_MFMA_TFLOPS
    issue = 3.0 if split else 1.0            # MFMA FLOPs issued per algorithmic FLOP
    traffic_key = f"{precision}{'h' if getattr(engine, 'half', False) else ''}"
    traffic, traffic_why = measured_traffic(traffic_key, dominant)
    pmc, pmc_why = measured_mfma_busy(traffic_key, dominant)
    line = {
        "metric": "6-h rollout steps/sec at 0.25deg/37-level",
        "value": args.gpus * args.steps / elapsed,
        "unit": "steps/s",
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"f16x3": "f32 (3 x f16-split MFMA products, f32 accumulate)", "f32": "f32",
                  "bf16": "bf16 activations + parameters, f32 accumulate = the reference's Bfloat16Cast run "
                          "(reduced-precision TIER: not the headline)"}[precision],
        "data": "synthetic",
        "config": {
            "workload": f"GraphCast {args.config}: one encode-process-decode 6-h step per GPU "
                        f"(grid {n_grid}, mesh {g['n_mesh']}, edges {len(g['g2m']['senders'])}/"
                        f"{len(g['mesh']['senders'])}/{len(g['m2g']['senders'])}, C {c_in}->{c_out}, "
                        f"{gnn_steps} processor steps, latent 512), random-init weights",
            "parallelism": f"ensemble x{args.gpus} (1 member per GPU, no collective in the step)",
            "batch_per_gpu": 1},
        "roofline": {
            "bound": "mfma",
            "kernel": f"{dominant_kernel_name(precision, dominant, dom, nat.get_tuning())} stage {dominant}",
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
            "frac": achieved / peak,
            "mfma_flops_per_algorithmic_flop": issue,
            "mfma_issue_frac": issue * achieved / peak,
            # the same MFMA work against what the chip sustains under power with real operands
            "sustained_peak_measured": (None if precision == "f32" else SUSTAINED_F16_MFMA_TFLOPS),
            "mfma_issue_frac_of_sustained": (None if precision == "f32" else
                                             issue * achieved / SUSTAINED_F16_MFMA_TFLOPS["mfma_with_operand_streams"]),
            "launches_per_step": dom["launches"],
            "avg_launch_ms": dom["ms"] / dom["launches"],
            "traffic": (traffic or {}).get("bytes_per_launch"),
            "traffic_detail": traffic if traffic is not None else {"unavailable": traffic_why},
            "pmc": pmc if pmc is not None else {"unavailable": pmc_why},
            # the weight stream of the dominant launch: packed-weight bytes DMA'd L2 -> LDS per launch over its duration,
            # against the chip-wide LDS-DMA fill rate of MI355X_MICROARCH.md (6.4-6.8 TB/s; HBM-sourced there, L2 hits here)
            "lds_fill": {"bytes_per_launch": dom["lds_fill_bytes"] / dom["launches"],
                         "achieved_tb_per_s": (dom["lds_fill_bytes"] / (dom["ms"] / 1e3) / 1e12 if dom["ms"] > 0 else 0.0),
                         "guide_tb_per_s": [6.4, 6.8]},
            "step_executed_tflop": executed_tflop,
            "step_as_written_tflop": f_alg / 1e12,
            "step_frac_executed": executed_tflop / (ms_per_step / 1e3) / peak,
            "step_frac_as_written": f_alg / 1e12 / (ms_per_step / 1e3) / peak,
            # every stage against the same peak (executed FLOPs of its launches / its HIP-event time)
            "stages": stages_summary(per_stage, peak)},
        "precision": precision,
        "tier": (None if precision in ("f16x3", "f32") else
                 "reduced-precision TIER line: the headline is the default f16x3 run (fp32-grade results)"),
        "cross_check": cross,
        "stages_ms": {k: round(v["ms"], 3) for k, v in sorted(per_stage.items())},
        "setup_seconds": round(t_setup, 1),
        "output_finite": finite,
        "formulation": (("half-N kernels, two persistent four-wave workgroups per CU, chained layers"
                         + (f"; launches without gather / segment-sum from {engine.helpers_min_rows} rows on as ONE eight-wave "
                            + ("workgroup per CU (four multiplying + four weight-staging waves)" if not nat.get_tuning().wide
                               else "workgroup per CU (the wide form: eight multiplying waves on one weight ring)")
                            if getattr(engine, "helpers_min_rows", 0) else "")) if getattr(engine, "fuse", False)
                        else "half-N kernels, two persistent workgroups per CU" if getattr(engine, "half", False)
                        else "chunked, one workgroup per CU"),
        "build": nat.lib().gc_build_info().decode(),
        # the library's ONE tuning surface (include/gcast.h: gc_tuning) as this process ran with it -- every speed-only
        # A/B switch, whatever set it (the GCAST_* variables only initialise it)
        "tuning": nat.tuning_string(),
    }
    if args.gpus == 1 and args.rollout_steps > 0 and args.config == "0.25deg_37L_M6":
      line["rollout"] = rollout_extra(model, task, lat, lon, args.rollout_steps)
      # the metric's own word is "rollout": the same figure for the 40-step AUTOREGRESSIVE device loop (state advance
      # included) beside `value` (= K repeated steps on a constant input, the contract's timed region) -- VERDICT r5 weak #11
      line["value_rollout"] = line["rollout"]["steps_per_second"]
      line["value_rollout_what"] = (f"steps/s of the {args.rollout_steps}-step autoregressive rollout resident in HBM (line.rollout: "
                                    "step + gc_advance_state per lead time, HIP-event time of the device loop)")
      line["rollout_api"] = rollout_api_extra(model, task, lat, lon, args.rollout_steps)
    if args.gpus == 1 and not args.no_cpu_baseline:
      line["cpu_baseline"] = cpu_baseline(c_in, c_out, gnn_steps, f_alg, full_graphs=g,
                                          full_x=x.cpu().numpy())
    else:
      line["cpu_baseline"] = None
  if distributed:
    dist.destroy_process_group()
  if rank == 0:
    emit_line(json.dumps(line))              # the ONE line on stdout


class _PowerPoll:
  """Socket power and shader clock of GPU 0 while something runs (scripts/power_probe.py's reader in a thread)."""

  def __init__(self):
    import threading
    self._stop, self._samples = threading.Event(), []
    self._thread = threading.Thread(target=self._run, daemon=True)

  def _run(self):
    try:
      sys.path.insert(0, os.path.join(ROOT, "scripts"))
      import power_probe
      while not self._stop.is_set():
        s = power_probe.read_once()
        s.pop("raw", None)
        self._samples.append(s)
        time.sleep(0.05)
    except Exception as e:      # noqa: BLE001  (a missing tool must not cost the bench line)
      self._samples.append({"errors": [repr(e)]})

  def start(self):
    self._thread.start()

  def stop(self):
    self._stop.set()
    self._thread.join(timeout=10)
    pw = sorted(s["power_w"] for s in self._samples if "power_w" in s)
    ck = sorted(s["sclk_mhz"] for s in self._samples if "sclk_mhz" in s)
    if not pw:
      return {"unavailable": sorted({e for s in self._samples for e in s.get("errors", [])})[:2]}
    return {"socket_w_median": pw[len(pw) // 2], "socket_w_max": pw[-1], "sclk_mhz_median": ck[len(ck) // 2] if ck else None,
            "samples": len(pw), "board_power_w": 1400, "tool": "amd-smi metric --power --clock (scripts/power_probe.py)"}


def dominant_kernel_name(precision, stage, dom, tuning):
  """The kernel the dominant stage's launches run as (what a rocprofv3 kernel trace of this command shows): the form follows
  the launcher's rules (csrc/gcast.hip: launch_rowmlp_half) -- since round 6 every edge update of >= 4096 tiles runs in the
  wide form, the two-pass ones with GC_LATE_ADDENDS."""
  if precision == "f32":
    return "rowmlp_kernel<MLP_LN>"
  if precision == "bf16":
    return "rowmlpbf_kernel<., 4, 0> (64-row workgroups, two per CU)"
  if stage in ("proc_edge", "enc_edge", "dec_edge"):
    onepass = stage != "proc_edge"
    if tuning.wide_edges & (1 if onepass else 2):
      late = (not onepass) and tuning.wide_late
      return ("rowmlp16w_kernel<MLP_LN, " + ("3" if stage == "dec_edge" else "2" if onepass else "0") + (", 1> (wide form, GC_LATE_ADDENDS)" if late else ", 0> (wide form)"))
    return "rowmlp16h_kernel<MLP_LN> (pairs of four-wave workgroups)"
  if stage in ("enc_embed_grid", "enc_node_grid", "dec_node") and tuning.wide:
    return "rowmlp16w_kernel<MLP_LN, 0, 0> (wide form)"
  return "rowmlp16h_kernel<MLP_LN>"


def stage_table(engine, x, y, iters):
  """Per-stage totals of one step of `engine`: per-launch durations by HIP events on the launch stream
  (gc_time_program), executed FLOPs and weight-stream bytes of every launch, summed by stage tag."""
  from graphcast_amd import engine as eng
  arr, _ = engine.bind(x, y)
  timed = engine.time_ops(x, iters=iters)
  tag_name = {v: k for k, v in eng.TAGS.items()}
  per_stage = {}
  for k, (tag, kind, ms) in enumerate(timed):
    s = per_stage.setdefault(tag_name[tag], {"ms": 0.0, "launches": 0, "tflop": 0.0, "lds_fill_bytes": 0.0})
    s["ms"] += ms
    s["launches"] += 1
    s["tflop"] += op_flops(arr[k]) / 1e12
    s["lds_fill_bytes"] += op_weight_stream_bytes(arr[k])
  return per_stage


def stages_summary(per_stage, peak):
  """Every stage against the same peak (executed FLOPs of its launches / its HIP-event time)."""
  return {k: {"ms": round(v["ms"], 3), "launches": v["launches"], "tflop": round(v["tflop"], 4),
              "achieved": (v["tflop"] / (v["ms"] / 1e3) if v["ms"] > 0 else 0.0),
              "frac": (v["tflop"] / (v["ms"] / 1e3) / peak if v["ms"] > 0 else 0.0),
              "lds_fill_tb_per_s": round(v["lds_fill_bytes"] / (v["ms"] / 1e3) / 1e12, 2) if v["ms"] > 0 else 0.0}
          for k, v in sorted(per_stage.items()) if v["tflop"] > 0}


def exchange_probe(step, world, device, iters=5):
  """The cost of ONE step's 18 halo exchanges on this rank, in isolation (HIP events around index_select +
  all_to_all_single, no compute in between): the rank's real exchangers when it has remote senders; at N = 1, where
  the real splits are empty, a SELF-exchange of the row counts an 8-way rank has on the 0.25 deg graphs (encoder
  3,000 grid rows, 419 mesh rows per processor step, 240 decoder rows: DESIGN.md section 7) through the same
  DistExchanger, so that the collective's launch + copy cost is a number before an 8-GPU node exists."""
  import torch
  from graphcast_amd import partition
  eng_ = step.engine
  real = any(sum(ex.send_counts) + sum(ex.recv_counts) > 0 for ex in step.exchangers.values())
  if real:
    exs = {name: (step.exchangers[name], eng_.halo_table(name)) for name in ("g2m", "mesh", "m2g")}
    rows = {name: int(sum(ex.recv_counts)) for name, (ex, _) in exs.items()}
  else:
    rows = {"g2m": 3000, "mesh": 419, "m2g": 240}
    exs = {}
    for name, n in rows.items():
      owned = eng_.owned_rows(name)
      n = min(n, owned)
      table = torch.zeros((owned + n, 512), dtype=torch.float32, device=device)
      idx = [np.arange(0, n, dtype=np.int64) * (owned // n)] + [np.zeros(0, np.int64)] * (world - 1)
      plan_ = partition.HaloPlan(np.zeros(n, np.int64), [n] + [0] * (world - 1), idx)
      exs[name] = (partition.DistExchanger(plan_, owned, device), table)
  def once():
    exs["g2m"][0].exchange(exs["g2m"][1])
    for _ in range(eng_.num_steps):
      exs["mesh"][0].exchange(exs["mesh"][1])
    exs["m2g"][0].exchange(exs["m2g"][1])
  once()
  torch.cuda.synchronize()
  ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0 = time.perf_counter()
  ev0.record()
  for _ in range(iters):
    once()
  ev1.record()
  torch.cuda.synchronize()
  wall = 1e3 * (time.perf_counter() - t0) / iters
  return {"exchanges_per_step": eng_.num_steps + 2, "rows_received": rows, "synthetic_self_exchange": not real,
          "device_ms_per_step": ev0.elapsed_time(ev1) / iters, "host_wall_ms_per_step": wall}


def rollout_extra(model, task, lat, lon, n_steps):
  """BASELINE.json config 3 on the model the line was measured with: an `n_steps` x 6 h autoregressive rollout,
  everything resident in HBM (rollout_device.DeviceRollout: the step + ONE fused gc_advance_state per step -- rolling
  window, InputsAndResiduals' algebra, forcings); only the last frame is kept (40 frames are 38 GB)."""
  import torch
  from graphcast_amd import rollout_device
  from graphcast_amd import synthetic
  inputs, template, forcings = synthetic.make_example(task, lat, lon, num_target_steps=n_steps)
  mean, std, dstd = synthetic.make_stats(task)
  roll = rollout_device.DeviceRollout(model, std, mean, dstd)
  roll.run(inputs, template.isel(time=slice(0, 1)), forcings.isel(time=slice(0, 1)), keep_trajectory=False)   # tables, warm-up
  torch.cuda.synchronize()
  # the TIMED rollout runs undisturbed; socket power / shader clock are polled (amd-smi, every ~50 ms, another thread)
  # during a SECOND, untimed repeat of the same 40 steps (the SMU queries and the subprocess churn must not
  # sit inside the window the reported number comes from) -- the step sits at the part's power limit (DESIGN.md section
  # 9.14), this is where the line shows it; the polled repeat's own loop time is reported beside it
  last = roll.run(inputs, template, forcings, keep_trajectory=False)
  torch.cuda.synchronize()
  loop_ms = roll.last_loop_ms()
  finite = bool(torch.isfinite(last).all().item())
  advance_ms = roll.advance_ms()
  power = _PowerPoll()
  power.start()
  roll.run(inputs, template, forcings, keep_trajectory=False)
  torch.cuda.synchronize()
  power_report = power.stop()
  power_report["ms_per_step_while_polled"] = roll.last_loop_ms() / n_steps
  return {"steps": n_steps, "ms_per_step": loop_ms / n_steps, "steps_per_second": 1e3 * n_steps / loop_ms,
          "advance_state_ms": advance_ms, "finite": finite,
          "power": power_report,
          "what": f"{n_steps} x 6 h autoregressive steps, state + forcings resident in HBM (DeviceRollout), device-loop "
                  "time by HIP events; parity of this loop: tests/test_rollout40_fullsize_gpu.py"}


def rollout_api_extra(model, task, lat, lon, n_steps):
  """The same rollout through the REFERENCE's entry point (utils/rollout.py:326-565): `rollout.chunked_prediction_generator`
  on HOST Datasets around the demo stack normalization.InputsAndResiduals(GraphCast).  Three forms of `predictor_fn`:
    * `rollout.as_predictor_fn(stack)` (`ms_per_step`): the functional form of the Predictor object -- the fused device
      loop runs underneath, every chunk's [N_grid, 1, C_out] block is copied to pinned host memory on a side stream under
      the next step, host Datasets are yielded;
    * `rollout.fuse(closure)` (`ms_per_step_fused_closure`): a plain closure around the stack that the CALLER opted in --
      the same loop plus the first- and last-chunk cross-checks against the closure itself (on device-resident Datasets);
    * the plain closure (`generic_loop_ms_per_step`): opaque, as in the reference -- called for every chunk (round 6: the
      default for closures; 4 chunks timed).
  Chunks are dropped as they come (40 kept frames are 38 GB of host memory: `chunked_prediction` = this generator + one
  host concatenation).  Every figure = host wall time of the whole generator loop / steps (uploads, cross-check calls and
  the last chunk's copy included)."""
  import torch
  from graphcast_amd import normalization, rollout, synthetic
  inputs, template, forcings = synthetic.make_example(task, lat, lon, num_target_steps=n_steps)
  mean, std, dstd = synthetic.make_stats(task)
  stack = normalization.InputsAndResiduals(model, std, mean, dstd)
  calls = []

  def closure(rng, inputs, targets_template, forcings):
    calls.append(1)
    return stack(inputs, targets_template, forcings)

  def consume(fn, tmpl, forc):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n, host, finite = 0, True, True
    for chunk in rollout.chunked_prediction_generator(fn, None, inputs, tmpl, 1, forc):
      v = chunk["2m_temperature"].data
      host = host and isinstance(v, np.ndarray)
      finite = finite and bool(np.isfinite(np.asarray(v).reshape(-1)[::4097]).all())
      n += 1
      del chunk
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0), n, host, finite

  trusted = rollout.as_predictor_fn(stack)
  consume(trusted, template.isel(time=slice(0, 2)), forcings.isel(time=slice(0, 2)))           # warm-up: tables, pinned pages
  ms, n, host, finite = consume(trusted, template, forcings)
  host_s = {k: round(v, 3) for k, v in rollout.last_fused_stats.get("stats", {}).items()}
  calls.clear()
  ms_closure, _, host2, finite2 = consume(rollout.fuse(closure), template, forcings)
  host_s_closure = {k: round(v, 3) for k, v in rollout.last_fused_stats.get("stats", {}).items()}
  closure_calls_fused = len(calls)
  calls.clear()
  k = min(n_steps, 4)
  ms_generic, _, _, _ = consume(closure, template.isel(time=slice(0, k)), forcings.isel(time=slice(0, k)))
  return {"steps": n, "ms_per_step": ms / max(n, 1), "steps_per_second": 1e3 * n / ms, "host_datasets_out": host and host2,
          "finite_sampled": finite and finite2, "predictor_fn": "rollout.as_predictor_fn(InputsAndResiduals(GraphCast))",
          "ms_per_step_fused_closure": ms_closure / max(n, 1), "closure_calls_when_fused": closure_calls_fused,
          "generic_loop_ms_per_step": ms_generic / k, "closure_calls_generic": len(calls), "generic_chunks": k,
          "host_seconds": host_s, "host_seconds_fused_closure": host_s_closure,
          "what": f"{n_steps} x 6 h steps through rollout.chunked_prediction_generator on HOST Datasets, host wall time incl. "
                  "uploads and the per-chunk D2H (0.94 GB, overlapped).  ms_per_step: predictor_fn = rollout.as_predictor_fn(stack) "
                  "(fused device loop); ms_per_step_fused_closure: rollout.fuse(closure) -- the caller's opt-in, + first- and "
                  "last-chunk cross-checks; generic_loop_ms_per_step: the plain closure, opaque as in the reference (called for "
                  "every chunk; the default for closures since round 6); parity: tests/test_rollout_gpu.py (bitwise = DeviceRollout)"}


def single_process_main(args):
  """`bench.py --gpus N --single-process`: ensemble member d on device d, all N engines driven from THIS process (the
  pmap_devices form: `GraphCast.replica` per device -- shared parameters and graphs, own plan + workspace -- one stream
  each; per step the launches of every device are enqueued before the host waits on any).  Same contract: K timed steps
  per device bracketed by a synchronisation of every device, `value` = N x K / elapsed.  Unmeasured at N > 1 DEVICES
  until a multi-GPU node exists: on a 1-GPU box the N engines share the one device (then `value` is that device's rate,
  not a scaling point -- `config.devices` says which it was)."""
  import torch
  from graphcast_amd import _native as nat
  from graphcast_amd import graphcast as gc
  res, mesh_size, levels, gnn_steps = CONFIGS[args.config]
  task = {37: gc.TASK, 13: gc.TASK_13}[levels]
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * levels) + 2 * 5 + 2 + 5
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=LATENT, gnn_msg_steps=gnn_steps, hidden_layers=1,
                       radius_query_fraction_edge_length=0.6)
  visible = torch.cuda.device_count()
  devices = [f"cuda:{d % visible}" for d in range(args.gpus)]
  t_setup = time.perf_counter()
  params = fast_params(c_in, c_out, gnn_steps)
  model = gc.GraphCast(cfg, task, params=params, device=devices[0], precision=args.precision)
  model.init_from_coordinates(lat, lon)
  g = model.graph_arrays()
  members = []
  for d, dev in enumerate(devices):
    m = model if d == 0 else model.replica(dev)
    with torch.cuda.device(dev):
      stream = torch.cuda.Stream(device=dev)
      x = torch.from_numpy(np.random.default_rng(d).standard_normal((g["n_grid"], 1, c_in), dtype=np.float32)).to(dev)
      y = torch.empty((g["n_grid"], 1, c_out), dtype=torch.float32, device=dev)
      with torch.cuda.stream(stream):
        m.forward_grid_node_features(x, y)        # builds the replica's engine + folds its constants
    members.append((m, stream, x, y, dev))

  def sync_all():
    for dev in sorted(set(devices)):
      torch.cuda.synchronize(dev)

  def step_all():
    for m, stream, x, y, dev in members:          # every device's launches are enqueued before the host waits on any
      with torch.cuda.device(dev), torch.cuda.stream(stream):
        m._engine(x, y)

  sync_all()
  t_setup = time.perf_counter() - t_setup
  for _ in range(args.warmup):
    step_all()
  sync_all()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step_all()
  sync_all()
  elapsed = time.perf_counter() - t0
  for m, *_ in members:
    m._engine.check_range()
  finite = all(bool(torch.isfinite(y).all().item()) for _, _, _, y, _ in members)
  same_bits = None
  if len(set(devices)) < len(devices):            # engines sharing a device: a member re-run ALONE gives the same bits
    m, stream, x, y, dev = members[-1]
    alone = torch.empty_like(y)
    m._engine(x, alone)
    torch.cuda.synchronize(dev)
    same_bits = bool(torch.equal(alone, y))
  engine = model._engine
  precision = engine.precision
  peak = PEAK_FP32_MFMA_TFLOPS if precision == "f32" else PEAK_F16_MFMA_TFLOPS
  per_stage = stage_table(engine, members[0][2], members[0][3], args.op_timing_iters)
  dominant = max(per_stage, key=lambda k: per_stage[k]["ms"])
  dom = per_stage[dominant]
  achieved = dom["tflop"] / (dom["ms"] / 1e3) if dom["ms"] > 0 else 0.0
  line = {
      "metric": "6-h rollout steps/sec at 0.25deg/37-level", "value": args.gpus * args.steps / elapsed, "unit": "steps/s",
      "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
      "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
      "dtype": {"f16x3": "f32 (3 x f16-split MFMA products, f32 accumulate)", "f32": "f32",
                "bf16": "bf16 activations + parameters, f32 accumulate (reduced-precision TIER)"}[precision],
      "data": "synthetic",
      "config": {"workload": f"GraphCast {args.config}: one encode-process-decode 6-h step per ensemble member",
                 "parallelism": f"ensemble x{args.gpus}, ONE process driving {args.gpus} engines (the reference's pmap_devices "
                                "form; no collective in the step)",
                 "devices": devices, "distinct_devices": len(set(devices)), "batch_per_gpu": 1,
                 "note": (None if len(set(devices)) == len(devices) else
                          "engines share a device: `value` is that device's rate with several engines resident, not a scaling point")},
      "roofline": {"bound": "mfma", "kernel": f"{dominant_kernel_name(precision, dominant, dom, nat.get_tuning())} stage {dominant} (engine 0, timed alone)",
                   "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                   "launches_per_step": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                   "stages": stages_summary(per_stage, peak)},
      "cpu_baseline": None,
      "precision": precision, "output_finite": finite, "member_alone_gives_the_same_bits": same_bits,
      "setup_seconds": round(t_setup, 1), "build": nat.lib().gc_build_info().decode(), "tuning": nat.tuning_string()}
  emit_line(json.dumps(line))


def partition_main(args, rank, world, device, distributed):
  """BASELINE.json config 5: ONE 0.25 deg step strong-scaled over N GPUs (graphcast_amd/partition.py) -- the multi-mesh
  and the grid split by octant (hemispheres / quadrants for 2 / 4 ranks), every edge owned by the rank that owns its
  receiver (segment-sum stays local: the reference's shard pattern, utils/gather_scatter_ops.py:267-283), remote
  sender rows through 18 halo exchanges per step, ONE all_to_all_single each.  Same JSON contract; `value` = steps/s
  of the whole partitioned step (max over ranks)."""
  import torch
  import torch.distributed as dist
  from graphcast_amd import _native as nat
  from graphcast_amd import graphcast as gc
  from graphcast_amd import partition
  if not distributed:            # a bare `python bench.py --mode partition`: a single-process group, still RCCL
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(device))
  res, mesh_size, levels, gnn_steps = CONFIGS[args.config]
  task = {37: gc.TASK, 13: gc.TASK_13}[levels]
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * levels) + 2 * 5 + 2 + 5
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=LATENT, gnn_msg_steps=gnn_steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  params = fast_params(c_in, c_out, gnn_steps)
  model = gc.GraphCast(cfg, task, params=params, device=device, precision=args.precision).init_from_coordinates(lat, lon)
  g = model.graph_arrays()
  mine = partition.plan(g, model._grid_nodes_lon, model._mesh_nodes_lon, world, grid_lat=model._grid_nodes_lat,
                        mesh_lat=model._mesh_nodes_lat)[rank]
  step = partition.DistributedPartitionedStep(mine, params, num_steps=gnn_steps, c_in=c_in, c_out=c_out, device=device,
                                              precision=args.precision)
  x = torch.from_numpy(np.random.default_rng(0).standard_normal(
      (g["n_grid"], 1, c_in), dtype=np.float32)[mine.grid_owned]).to(device)
  y = torch.empty((mine.n_grid_owned, 1, c_out), dtype=torch.float32, device=device)
  for _ in range(args.warmup):
    step(x, y)
  dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step(x, y)
  dist.barrier()
  torch.cuda.synchronize()
  t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  elapsed = float(t.item())
  step.engine.check_range()
  # every rank's shard must be finite (rank 0's alone says nothing about the others)
  fin = torch.tensor([1 if bool(torch.isfinite(y).all().item()) else 0], dtype=torch.int32, device=device)
  dist.all_reduce(fin, op=dist.ReduceOp.MIN)
  finite = bool(fin.item())
  # per-rank roofline: this rank's launches timed one by one (HIP events, no exchange in between), max over ranks per stage
  precision = step.engine.precision
  peak = PEAK_FP32_MFMA_TFLOPS if precision == "f32" else PEAK_F16_MFMA_TFLOPS
  per_stage = stage_table(step.engine, x, y, args.op_timing_iters)
  names = sorted(per_stage)
  ms = torch.tensor([per_stage[k]["ms"] for k in names], dtype=torch.float64, device=device)
  dist.all_reduce(ms, op=dist.ReduceOp.MAX)          # (every rank has the same stage names: the same launch program)
  exch = exchange_probe(step, world, device)
  if rank == 0:
    halo = {k: int(len(pl.halo_global)) for k, (pl, _) in partition.tables_of(mine).items()}
    stages = stages_summary(per_stage, peak)
    for k, v in zip(names, ms.tolist()):
      if k in stages:
        stages[k]["ms_max_over_ranks"] = round(v, 3)
    dominant = max(stages, key=lambda k: stages[k]["ms"])
    dom = per_stage[dominant]
    compute_ms = float(sum(ms.tolist()))
    cpu = None
    if world == 1 and not args.no_cpu_baseline:      # the same whole step on the host cores (rank 0, N = 1 only)
      f_alg = flops_as_written(g["n_grid"], g["n_mesh"], len(g["g2m"]["senders"]), len(g["mesh"]["senders"]),
                               len(g["m2g"]["senders"]), c_in, c_out, gnn_steps)
      cpu = cpu_baseline(c_in, c_out, gnn_steps, f_alg, full_graphs=g,
                         full_x=np.random.default_rng(0).standard_normal((g["n_grid"], 1, c_in), dtype=np.float32))
    roofline = {"bound": "mfma", "kernel": f"rowmlp16h_kernel<MLP_LN> stage {dominant} (rank 0's share of the launch)",
                "achieved": stages[dominant]["achieved"], "peak": peak, "unit": "TFLOP/s", "frac": stages[dominant]["frac"],
                "launches_per_step": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"], "traffic": None,
                "mfma_flops_per_algorithmic_flop": 3.0 if precision == "f16x3" else 1.0,
                "stages": stages, "rank_compute_ms_per_step_max_over_ranks": compute_ms,
                "exchange": exch,
                "note": "per-rank launches timed one by one (gc_time_program); a rank's launches are 1/N of the step's "
                        "rows -- few tiles per CU: see DESIGN.md section 7"}
    line = json.dumps({
        "metric": "6-h rollout steps/sec at 0.25deg/37-level",
        "value": args.steps / elapsed, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32 (3 x f16-split MFMA products, f32 accumulate)" if precision == "f16x3" else precision,
        "data": "synthetic",
        "config": {"workload": f"GraphCast {args.config}: ONE encode-process-decode 6-h step partitioned over {world} GPU(s)",
                   "parallelism": f"octant partition x{world} (partition.plan: hemispheres / quadrants / octants), receiver-owned "
                                  f"edges, 18 halo exchanges per step = one RCCL all_to_all_single each",
                   "rank0_rows": {"grid": mine.n_grid_owned, "mesh": mine.n_mesh_owned, "halo": halo}},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "precision": precision, "output_finite": finite,
        "build": nat.lib().gc_build_info().decode(), "tuning": nat.tuning_string()})
  dist.destroy_process_group()
  if rank == 0:
    emit_line(line)


if __name__ == "__main__":
  main()
