"""latent_size / hidden_layers other than 512 / 1 AT the headline graph (0.25 deg / 37 levels / M6): the launches
push_mlp adds (csrc/gcast_plan.inc) reach sizes here that the 4 deg GPU tests do not -- the forms the launcher picks
for big launches (wide two-pass edge updates, helper-form node launches) with addend-only first layers.  No oracle
runs at this size in seconds, so the check is the one bench.py's `cross_check` makes for the published model: the
f16x3 step against the EXACT-fp32 chunked kernels (another kernel family: one workgroup per CU, no forms), which the
4 deg tests pin to the fp64 oracle for the same sizes (tests/test_general_sizes_gpu.py); plus the bf16 tier against
f16x3 at the tier's tolerance, bitwise repeatability, and ms per step.

  python scripts/general_sizes_fullsize.py [--config 0.25deg_37L_M6] [--out profiles/...json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                       # noqa: E402  (CONFIGS)
from graphcast_amd import graphcast as gc          # noqa: E402


def params_of(c_in, c_out, latent, steps, hidden_layers, seed=7):
  from graphcast_amd import params as gparams
  rng = np.random.default_rng(seed)
  out = {}
  for stem, k, n, ln in gparams.mlp_table(c_in, c_out, latent, steps):
    sizes = [k] + [latent] * hidden_layers + [n]
    for layer in range(len(sizes) - 1):
      w = rng.standard_normal((sizes[layer], sizes[layer + 1]), dtype=np.float32)
      np.clip(w, -2, 2, out=w)
      w /= np.float32(np.sqrt(sizes[layer]))
      out[f"{stem}_mlp/~/linear_{layer}"] = {"w": w, "b": (0.1 * rng.standard_normal(sizes[layer + 1])).astype(np.float32)}
    if ln:
      out[f"{stem}_layer_norm"] = {"scale": (1 + 0.1 * rng.standard_normal(n)).astype(np.float32),
                                   "offset": (0.1 * rng.standard_normal(n)).astype(np.float32)}
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--config", default="0.25deg_37L_M6")
  ap.add_argument("--cases", default="256x1,512x2,128x3")
  ap.add_argument("--steps", type=int, default=5)
  ap.add_argument("--out", default="")
  args = ap.parse_args()
  res, mesh_size, levels, gnn_steps = bench.CONFIGS[args.config]
  task = {37: gc.TASK, 13: gc.TASK_13}[levels]
  c_out = gc.num_output_channels(task)
  c_in = 2 * (5 + 6 * levels) + 2 * 5 + 2 + 5
  lat = np.arange(-90, 90 + res / 2, res)
  lon = np.arange(0, 360, res)
  rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
  results = []
  base = None
  for case in args.cases.split(","):
    latent, hidden = (int(v) for v in case.split("x"))
    cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=latent, gnn_msg_steps=gnn_steps,
                         hidden_layers=hidden, radius_query_fraction_edge_length=0.6)
    params = params_of(c_in, c_out, latent, gnn_steps, hidden)
    model = gc.GraphCast(cfg, task, params=params, device="cuda:0", precision="f16x3")
    if base is None:
      model.init_from_coordinates(lat, lon)
      base = model
    else:                                    # (the static graphs are the same: built once)
      model = base.replica("cuda:0")
      model._model_config = cfg
      model.load_params(params)
    n_grid = model.graph_arrays()["n_grid"]
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((n_grid, 1, c_in), dtype=np.float32)).to("cuda:0")
    out = {}
    row = dict(config=args.config, latent_size=latent, hidden_layers=hidden)
    for prec in ("f16x3", "f32", "bf16"):
      model.set_precision(prec)
      y = model.forward_grid_node_features(x).clone()
      y2 = model.forward_grid_node_features(x).clone()
      torch.cuda.synchronize()
      eng = model._engine
      yb = torch.empty_like(y)
      eng(x, yb)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(args.steps):
        eng(x, yb)
      torch.cuda.synchronize()
      ms = (time.perf_counter() - t0) / args.steps * 1e3
      eng.check_range()
      out[prec] = y
      row[prec] = dict(ms_per_step=round(ms, 3), finite=bool(torch.isfinite(y).all().item()),
                       repeatable_bitwise=bool(torch.equal(y, y2)), launches=len(eng.bind(x)[0]))
      # (engines hold the packed weights + workspace of one precision: released before the next one is built)
      model._engines.pop(prec, None)
      model._engine = None
      model._precision = None
      del eng
      torch.cuda.empty_cache()
    row["f16x3_vs_exact_fp32_kernels"] = rel(out["f16x3"], out["f32"])
    row["bf16_vs_f16x3"] = rel(out["bf16"], out["f16x3"])
    print(json.dumps(row), flush=True)
    results.append(row)
    del out, model
    torch.cuda.empty_cache()
  ok = all(r["f16x3_vs_exact_fp32_kernels"] <= 5e-6 and r["bf16_vs_f16x3"] <= 3e-2 and
           all(r[p]["finite"] and r[p]["repeatable_bitwise"] for p in ("f16x3", "f32", "bf16")) for r in results)
  if args.out:
    with open(args.out, "w") as f:
      json.dump(dict(results=results, ok=ok), f, indent=1)
  print("GENERAL_SIZES_FULLSIZE", "ok" if ok else "FAILED")
  return 0 if ok else 1


if __name__ == "__main__":
  sys.exit(main())
