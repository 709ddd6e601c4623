#!/usr/bin/env python
"""A/B of single gc_rowmlp launches (shapes of the 0.25 deg step): the chunked kernels (one
workgroup per CU) against the half-N kernels (GC_LAYOUT_HALF, two workgroups per CU), interleaved
ABAB in one process so that both see the same clocks.  Extra builds of the half kernel can be
compiled on the spot: HALF_BUILDS="tag:-DX=1,-DY=2;tag2:..." (or tag:@path/to/prebuilt.so).  GPU box only.

    python scripts/half_probe.py [--out gpurun_out/half_probe.json] [--iters 20]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphcast_amd import _native as nat      # noqa: E402
from graphcast_amd import packing             # noqa: E402

D = 512


def build(tag, defines):
  out = f"/tmp/libgcast_{tag}.so"
  cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-inline-asm", "-DGC_PROFILING_BUILD",
         "-DGC_PIPE=2", *defines, "-I", os.path.join(ROOT, "include"), "-shared", "-fPIC",
         os.path.join(ROOT, "graphcast_amd", "csrc", "gcast.hip"), "-o", out]
  subprocess.run(cmd, check=True)
  return load(out)


def load(path):
  lib = ctypes.CDLL(path)
  lib.gc_rowmlp.argtypes = [ctypes.POINTER(nat.RowMlpDesc), ctypes.c_void_p]
  lib.gc_rowmlp.restype = ctypes.c_int
  lib.gc_last_error.restype = ctypes.c_char_p
  return lib


def time_launch(lib, d, iters):
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  for _ in range(2):
    rc = lib.gc_rowmlp(ctypes.byref(d), stream)
    assert rc == 0, lib.gc_last_error()
  torch.cuda.synchronize()
  t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  t0.record()
  for _ in range(iters):
    lib.gc_rowmlp(ctypes.byref(d), stream)
  t1.record()
  torch.cuda.synchronize()
  return t0.elapsed_time(t1) / iters


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "half_probe.json"))
  ap.add_argument("--iters", type=int, default=20)
  ap.add_argument("--rounds", type=int, default=3)
  args = ap.parse_args()
  dev = torch.device("cuda:0")
  rng = np.random.default_rng(0)
  up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
  w = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
  sc = packing.choose_weight_scale(w)
  w1 = up(packing.pack_weight_split(w, scale=sc).view(np.int16))
  w2 = up(packing.pack_weight_split(w, chained=True, scale=sc).view(np.int16))
  w1b = up(packing.pack_weight_split(np.concatenate([w, w]), scale=sc).view(np.int16))
  wo = (rng.standard_normal((D, 227)) / np.sqrt(D)).astype(np.float32)
  w2o = up(packing.pack_weight_split(wo, np_cols=256, chained=True, scale=sc).view(np.int16))
  vec = up(0.1 * rng.standard_normal(D).astype(np.float32))
  one = up(np.ones(D, np.float32))
  n_mesh, n_g = 40962, 1038240
  recv = np.repeat(np.arange(n_mesh), 8)
  send = rng.integers(0, n_mesh, len(recv))
  pk = packing.pack_edges(send, recv, n_mesh)
  n_e = pk.n_rows
  e = torch.randn((n_e, D), device=dev)
  tab_s, tab_r = torch.randn((n_mesh, D), device=dev), torch.randn((n_mesh, D), device=dev)
  agg = torch.empty((n_mesh, D), device=dev)
  partial = torch.empty((2 * n_e // 64, D), device=dev)
  snd, rcv, flags = up(pk.senders), up(pk.receivers), up(pk.tile_flags)
  hg = torch.randn((n_g, D), device=dev)
  og = torch.empty((n_g, D), device=dev)
  yg = torch.empty((n_g, 227), device=dev)
  # mesh2grid-like edges: 3 per grid node (uniform degree 3 -> 63 rows + 1 pad per tile), senders on the mesh
  n_gd = n_g // 4                                   # a quarter of the grid: 0.78 M edges, 12 k tiles
  recv3 = np.repeat(np.arange(n_gd), 3)
  send3 = rng.integers(0, n_mesh, len(recv3))
  pk3 = packing.pack_edges(send3, recv3, n_gd)
  n_e3 = pk3.n_rows
  d3 = torch.randn((n_e3, D), device=dev)
  agg3 = torch.empty((n_gd, D), device=dev)
  partial3 = torch.empty((2 * n_e3 // 64, D), device=dev)
  snd3, rcv3, flags3 = up(pk3.senders), up(pk3.receivers), up(pk3.tile_flags)
  # the persistent workgroups' parking slots (HALF_BIG_SCRATCH=1: 1 KiB per row, what a round-2 library
  # given as HALF_BUILDS=r02:@ab_libs/libgcast_r02.so needs; the shipped kernels use its first 32 MiB)
  big = os.environ.get("HALF_BIG_SCRATCH") == "1"
  scratch = torch.empty(((max(n_g, n_e, n_e3) + 128) * 256 if big else nat.SCRATCH_FLOATS,), device=dev)
  prec = nat.PRECISIONS["f16x3"]
  s1 = s2 = float(sc)

  # the dynamic tile queue (gc_rowmlp_desc.tile_queue); PROBE_QUEUE=0: the static walk b, b + grid, ...
  queue = torch.zeros((2,), dtype=torch.int32, device=dev)
  use_queue = os.environ.get("PROBE_QUEUE", "1") != "0"

  def desc(mode, n_rows, layout):
    d = nat.RowMlpDesc()
    d.mode, d.n_rows, d.prec, d.layout = mode, n_rows, prec, layout
    d.w1_scale, d.w2_scale = s1, s2
    d.scratch = scratch.data_ptr()
    if use_queue and layout == nat.LAYOUT_HALF:
      d.tile_queue = queue.data_ptr()
    d.flags |= int(os.environ.get("PROBE_FLAGS", "0"))        # e.g. 256 = GC_WG_WIDE: every shape in the wide form
    return d

  def proc_edge(layout):
    d = desc(nat.MODE_MLP_LN, n_e, layout)
    d.a0, d.lda0, d.k0, d.w1p, d.b1 = e.data_ptr(), D, D, w1.data_ptr(), vec.data_ptr()
    d.g0, d.idx0, d.g1, d.idx1 = tab_s.data_ptr(), snd.data_ptr(), tab_r.data_ptr(), rcv.data_ptr()
    d.w2p, d.b2, d.n2 = w2.data_ptr(), vec.data_ptr(), D
    d.ln_scale, d.ln_offset = one.data_ptr(), vec.data_ptr()
    d.res, d.ldres, d.out, d.ldo = e.data_ptr(), D, og.data_ptr(), D       # (out != res: e stays put across iterations)
    d.seg, d.tile_flags, d.agg, d.partial = rcv.data_ptr(), flags.data_ptr(), agg.data_ptr(), partial.data_ptr()
    return d, n_e, 2.0 * n_e * 2 * D * D

  def gemm_only_mlp(layout):
    d = desc(nat.MODE_MLP_LN, n_e, layout)
    d.a0, d.lda0, d.k0, d.w1p, d.b1 = e.data_ptr(), D, D, w1.data_ptr(), vec.data_ptr()
    d.w2p, d.b2, d.n2 = w2.data_ptr(), vec.data_ptr(), D
    d.ln_scale, d.ln_offset = one.data_ptr(), vec.data_ptr()
    d.out, d.ldo = agg.data_ptr(), 0
    return d, n_e, 2.0 * n_e * 2 * D * D

  def gemm_only_cached_rows(layout):      # the same, every tile reading THE SAME 64 input rows (L2 / L1 hits):
    d, rows, flop = gemm_only_mlp(layout)  # what the layer-1 row loads' memory latency costs
    d.n_rows = n_e
    d.a0 = e.data_ptr()
    d.lda0 = 0
    return d, rows, flop

  def dec_edge(layout):       # k0 = 0: addends only -> swish -> W2 -> LN -> segment-sum, nothing stored
    d = desc(nat.MODE_MLP_LN, n_e3, layout)
    d.w1_scale = 1.0
    d.d, d.ldd = d3.data_ptr(), D
    d.g0, d.idx0, d.g1, d.idx1 = tab_s.data_ptr(), snd3.data_ptr(), hg.data_ptr(), rcv3.data_ptr()
    d.w2p, d.b2, d.n2 = w2.data_ptr(), vec.data_ptr(), D
    d.ln_scale, d.ln_offset = one.data_ptr(), vec.data_ptr()
    d.seg, d.tile_flags, d.agg, d.partial = rcv3.data_ptr(), flags3.data_ptr(), agg3.data_ptr(), partial3.data_ptr()
    return d, n_e3, 2.0 * n_e3 * D * D

  w2n = up(packing.pack_weight_split(w, chained=False, scale=sc).view(np.int16))

  def dec_edge_onepass(layout):       # the same launch in the ONE-PASS formulation (GC_W2_NATURAL)
    d, rows, flop = dec_edge(layout)
    if layout == nat.LAYOUT_HALF:
      d.w2p = w2n.data_ptr()
      d.flags |= nat.W2_NATURAL
    return d, rows, flop

  def linear_grid(layout):
    d = desc(nat.MODE_LINEAR, n_g, layout)
    d.a0, d.lda0, d.k0, d.w1p = hg.data_ptr(), D, D, w1.data_ptr()
    d.out, d.ldo = og.data_ptr(), D
    return d, n_g, 2.0 * n_g * D * D

  def node_grid(layout):
    d = desc(nat.MODE_MLP_LN, n_g, layout)
    d.a0, d.lda0, d.k0, d.a1, d.lda1, d.k1 = hg.data_ptr(), D, D, og.data_ptr(), D, D
    d.w1p, d.b1 = w1b.data_ptr(), vec.data_ptr()
    d.w2p, d.b2, d.n2 = w2.data_ptr(), vec.data_ptr(), D
    d.ln_scale, d.ln_offset = one.data_ptr(), vec.data_ptr()
    d.res, d.ldres, d.out, d.ldo = hg.data_ptr(), D, og.data_ptr(), D
    return d, n_g, 2.0 * n_g * 3 * D * D

  def dec_out(layout):
    d = desc(nat.MODE_MLP_OUT, n_g, layout)
    d.a0, d.lda0, d.k0, d.w1p, d.b1 = hg.data_ptr(), D, D, w1.data_ptr(), vec.data_ptr()
    d.w2p, d.b2, d.n2 = w2o.data_ptr(), vec.data_ptr(), 227
    d.out, d.ldo = yg.data_ptr(), 227
    return d, n_g, 2.0 * n_g * (D * D + D * 240)

  # ---- the same shapes in the GC_PREC_BF16 tier (bfloat16 rows in pi order, bf16 weight images)
  bf = lambda t: t.to(torch.bfloat16)
  wb1 = up(packing.pack_weight_bf16(w, chained=True).view(np.int16))
  wb1b = up(packing.pack_weight_bf16(np.concatenate([w, w]), chained=True).view(np.int16))
  e_bf, tab_s_bf, tab_r_bf, hg_bf, d3_bf = bf(e), bf(tab_s), bf(tab_r), bf(hg), bf(d3)
  og_bf, agg_bf, agg3_bf = torch.empty_like(hg_bf), torch.empty((n_mesh, D), dtype=torch.bfloat16, device=dev), torch.empty((n_gd, D), dtype=torch.bfloat16, device=dev)
  eo_bf = torch.empty_like(e_bf)

  def bdesc(n_rows):
    d = nat.RowMlpDesc()
    d.mode, d.n_rows, d.prec, d.layout, d.n2 = nat.MODE_MLP_LN, n_rows, nat.PREC_BF16, nat.LAYOUT_HALF, D
    d.w2p, d.b2 = wb1.data_ptr(), vec.data_ptr()
    d.ln_scale, d.ln_offset, d.b1 = one.data_ptr(), vec.data_ptr(), vec.data_ptr()
    d.scratch = scratch.data_ptr()            # (only the trace build looks at it)
    if use_queue:
      d.tile_queue = queue.data_ptr()
    return d

  def proc_edge_bf16(layout):
    d = bdesc(n_e)
    d.a0, d.lda0, d.k0, d.w1p = e_bf.data_ptr(), D, D, wb1.data_ptr()
    d.g0, d.idx0, d.g1, d.idx1 = tab_s_bf.data_ptr(), snd.data_ptr(), tab_r_bf.data_ptr(), rcv.data_ptr()
    d.res, d.ldres, d.out, d.ldo = e_bf.data_ptr(), D, eo_bf.data_ptr(), D
    d.seg, d.tile_flags, d.agg, d.partial = rcv.data_ptr(), flags.data_ptr(), agg_bf.data_ptr(), partial.data_ptr()
    return d, n_e, 2.0 * n_e * 2 * D * D

  def dec_edge_bf16(layout):
    d = bdesc(n_e3)
    d.d, d.ldd = d3_bf.data_ptr(), D
    d.g0, d.idx0, d.g1, d.idx1 = tab_s_bf.data_ptr(), snd3.data_ptr(), hg_bf.data_ptr(), rcv3.data_ptr()
    d.seg, d.tile_flags, d.agg, d.partial = rcv3.data_ptr(), flags3.data_ptr(), agg3_bf.data_ptr(), partial3.data_ptr()
    return d, n_e3, 2.0 * n_e3 * D * D

  def node_grid_bf16(layout):
    d = bdesc(n_g)
    d.a0, d.lda0, d.k0, d.a1, d.lda1, d.k1 = hg_bf.data_ptr(), D, D, og_bf.data_ptr(), D, D
    d.w1p = wb1b.data_ptr()
    d.res, d.ldres, d.out, d.ldo = hg_bf.data_ptr(), D, og_bf.data_ptr(), D
    return d, n_g, 2.0 * n_g * 3 * D * D

  shapes = dict(proc_edge_bf16=proc_edge_bf16, dec_edge_bf16=dec_edge_bf16, node_grid_bf16=node_grid_bf16, proc_edge=proc_edge, gemm_only_mlp=gemm_only_mlp, gemm_only_cached_rows=gemm_only_cached_rows, dec_edge=dec_edge,
                dec_edge_onepass=dec_edge_onepass, linear_grid=linear_grid,
                node_grid=node_grid, dec_out=dec_out)
  only = os.environ.get("PROBE_SHAPES")
  if only:
    shapes = {k: v for k, v in shapes.items() if k in only.split(",")}
  if os.environ.get("HALF_TRACE"):
    # phase timeline of wave 0 of every workgroup (GC_H_TRACE build): shader cycles per phase, wall time
    # per workgroup, how many workgroups ran on each CU
    lib = build("htrace", ["-DGC_H_TRACE=" + os.environ.get("HALF_TRACE", "1")] + [x for x in os.environ.get("HALF_TRACE_DEFS", "").split(",") if x])
    names = ["prologue_gather", "first_barrier", "layer1", "hidden_pack", "layer2_passes", "scratch_reload_ln",
             "segsum", "residual_store"]
    out = {}
    for name, make in shapes.items():
      d, rows, flop = make(nat.LAYOUT_HALF)
      if d.mode != nat.MODE_MLP_LN:
        continue
      tiles = (rows + 63) // 64
      buf = torch.zeros((nat.SCRATCH_FLOATS + tiles * 48,), dtype=torch.float32, device=dev)    # slots | 24 x int64 per tile
      d.scratch = buf.data_ptr()
      ms = time_launch(lib, d, 3)
      t = buf[nat.SCRATCH_FLOATS:].view(torch.int64).view(tiles, 24).cpu().numpy()
      t = t[t[:, 0] != 0]                                # (the wide form marks one 64-row tile number in two: PROBE_FLAGS=256)
      tiles = len(t)
      ph = np.diff(t[:, :9], axis=1).astype(np.float64)
      row = {"ms": round(ms, 4), "tiles": int(tiles), "wave0_cycles_total_mean": float((t[:, 8] - t[:, 0]).mean())}
      for j, nme in enumerate(names):
        row[nme + "_cycles_mean"] = round(float(ph[:, j].mean()), 0)
      row["wg_wall_us_mean"] = round(float((t[:, 14] - t[:, 12]).mean()) / 100.0, 2)
      per_cu = np.unique(t[:, 13], return_counts=True)[1]
      row["workgroups_per_cu_min_max"] = [int(per_cu.min()), int(per_cu.max())]
      row["shader_ghz_implied"] = round(row["wave0_cycles_total_mean"] / (row["wg_wall_us_mean"] * 1e3), 3)
      # how evenly the persistent workgroups finish (static schedule: workgroup b walks tiles b, b + grid, ...): the
      # launch lasts until the LAST one is done; (max - mean) finish time is what a dynamic tile queue could recover
      wg = t[:, 15].astype(np.int64)                   # which workgroup ran the tile (dynamic queue: not tile % grid)
      grid = int(wg.max()) + 1
      row["wg_schedule_queue"] = bool(use_queue)
      t_first = float(t[:, 12].min())
      # where the speed differences sit: per XCD (smid = xcc << 6 | se << 4 | cu on gfx94x / gfx950) and per CU
      smid = t[:, 13].astype(np.int64)
      wall = (t[:, 14] - t[:, 12]).astype(np.float64) / 100.0
      row["by_xcd"] = {int(x): {"tiles": int((smid >> 6 == x).sum()), "tile_us_mean": round(float(wall[smid >> 6 == x].mean()), 2)}
                       for x in np.unique(smid >> 6)}
      row["by_shader_engine_tile_us_mean"] = {int(x): round(float(wall[((smid >> 4) & 3) == x].mean()), 2) for x in np.unique((smid >> 4) & 3)}
      cu_mean = np.array([wall[smid == c].mean() for c in np.unique(smid)])
      row["per_cu_tile_us_mean_p5_p50_p95"] = [round(float(np.percentile(cu_mean, q)), 2) for q in (5, 50, 95)]
      row["tiles_per_workgroup_min_max"] = [int(np.bincount(wg, minlength=grid).min()), int(np.bincount(wg, minlength=grid).max())]
      finish = np.array([t[wg == b, 14].max() for b in range(grid)], dtype=np.float64)
      start = np.array([t[wg == b, 12].min() for b in range(grid)], dtype=np.float64)
      busy = np.array([(t[wg == b, 14] - t[wg == b, 12]).sum() for b in range(grid)], dtype=np.float64)
      us = lambda v: round(float(v) / 100.0, 2)
      row["wg_schedule"] = {"grid": int(grid), "span_us": us(finish.max() - t_first),
                            "first_start_us_max": us(start.max() - t_first),
                            "finish_us_min_mean_max": [us(finish.min() - t_first), us(finish.mean() - t_first), us(finish.max() - t_first)],
                            "busy_us_min_mean_max": [us(busy.min()), us(busy.mean()), us(busy.max())],
                            "tail_frac_of_span": round(float((finish.max() - finish.mean()) / (finish.max() - t_first)), 4)}
      out[name] = row
      print("htrace", name, json.dumps(row), flush=True)
    with open(args.out, "w") as f:
      json.dump(out, f, indent=1)
    return
  # (rounds 2-4 also timed the chunked f16x3 kernels of round 1 here: retired in round 5, GC_PREC_F16X3 == GC_LAYOUT_HALF)
  libs = [("half", load(nat.library_path()), nat.LAYOUT_HALF)]
  for spec in filter(None, os.environ.get("HALF_BUILDS", "").split(";")):
    tag, _, defs = spec.partition(":")
    if defs.startswith("@"):          # a library compiled beforehand (travels with the snapshot): tag:@relative/path.so
      libs.append((tag, load(os.path.join(ROOT, defs[1:])), nat.LAYOUT_HALF))
    else:
      libs.append((tag, build(tag, [x for x in defs.split(",") if x]), nat.LAYOUT_HALF))
  results = {}
  for name, make in shapes.items():
    row = {}
    for r in range(args.rounds):                      # ABAB...: every build once per round
      for tag, lib, layout in libs:
        if name.endswith("_bf16") and layout != nat.LAYOUT_HALF:
          continue                                   # (the tier exists in the persistent half-N form only)
        d, rows, flop = make(layout)
        ms = time_launch(lib, d, args.iters)
        row.setdefault(tag, []).append(ms)
    # results of every extra build against the in-tree half-N library (same inputs): max |diff| of `og` / `agg`
    check = {}
    if name in ("proc_edge", "node_grid") and len(libs) > 1:
      def run_once(lib, layout):
        d, _, _ = make(layout)
        og.zero_(); agg.zero_()
        assert lib.gc_rowmlp(ctypes.byref(d), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0, lib.gc_last_error()
        torch.cuda.synchronize()
        return og.clone(), agg.clone()
      ref_o, ref_a = run_once(libs[0][1], libs[0][2])
      for tag, lib, layout in libs[1:]:
        o, a = run_once(lib, layout)
        check[tag] = {"out_max_abs_diff": float((o - ref_o).abs().max()), "agg_max_abs_diff": float((a - ref_a).abs().max()),
                      "out_abs_max": float(ref_o.abs().max())}
    out = {"_check_vs_in_tree_half": check} if check else {}
    for tag, _, _ in libs:
      if tag not in row:
        continue
      ms = float(np.median(row[tag]))
      d, rows, flop = make(nat.LAYOUT_HALF)
      out[tag] = {"ms": round(ms, 4), "ms_all": [round(v, 4) for v in row[tag]],
                  "us_per_tile_per_cu": round(ms * 1e3 / (((rows + 63) // 64) / 256.0), 2),
                  "algorithmic_tflops": round(flop / ms / 1e9, 1)}
    results[name] = out
    print(name, json.dumps(out), flush=True)
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  with open(args.out, "w") as f:
    json.dump(results, f, indent=1)


if __name__ == "__main__":
  main()
