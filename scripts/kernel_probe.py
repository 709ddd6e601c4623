#!/usr/bin/env python
"""Times single gc_rowmlp launches (shapes of the 0.25 deg step) for several builds of
csrc/gcast.hip, including profiling-only experiment builds (-DGC_EXP=...: results wrong,
timing informative) compiled on the spot into /tmp.  GPU box only.

    python scripts/kernel_probe.py [--out gpurun_out/probe.json]
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
  cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-inline-asm", "-DGC_PROFILING_BUILD", *defines,
         "-I", os.path.join(ROOT, "include"), "-shared", "-fPIC",
         os.path.join(ROOT, "graphcast_amd", "csrc", "gcast.hip"), "-o", out]
  subprocess.run(cmd, check=True)
  lib = ctypes.CDLL(out)
  lib.gc_rowmlp.argtypes = [ctypes.POINTER(nat.RowMlpDesc), ctypes.c_void_p]
  lib.gc_rowmlp.restype = ctypes.c_int
  lib.gc_last_error.restype = ctypes.c_char_p
  return lib


def time_launch(lib, d, iters=8):
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
  ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "probe.json"))
  ap.add_argument("--prec", default="f16x3")
  ap.add_argument("--layout", default="chunked", choices=["chunked", "co"],
                  help="co: MLP_LN shapes run the column-owner kernel (GC_LAYOUT_COLOWN); builds = CO_EXP variants")
  args = ap.parse_args()
  dev = torch.device("cuda:0")
  rng = np.random.default_rng(0)
  split = args.prec == "f16x3"
  up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
  w = (rng.standard_normal((D, D)) / np.sqrt(D)).astype(np.float32)
  co = args.layout == "co"
  if co:
    w1 = up(packing.pack_weight_split_co(w).view(np.int16))
    w2 = up(packing.pack_weight_split_co(w).view(np.int16))
    w1b = up(packing.pack_weight_split_co(np.concatenate([w, w])).view(np.int16))
    w1lin = up(packing.pack_weight_split(w).view(np.int16))
  elif split:
    w1 = up(packing.pack_weight_split(w).view(np.int16))
    w2 = up(packing.pack_weight_split(w, chained=True).view(np.int16))
    w1b = up(packing.pack_weight_split(np.concatenate([w, w])).view(np.int16))
  else:
    w1, w2, w1b = up(packing.pack_weight(w)), up(packing.pack_weight(w)), up(packing.pack_weight(np.concatenate([w, w])))
  vec = up(np.zeros(D, np.float32))
  one = up(np.ones(D, np.float32))
  n_mesh = 40962
  recv = np.repeat(np.arange(n_mesh), 8)
  send = rng.integers(0, n_mesh, len(recv))
  pk = packing.pack_edges(send, recv, n_mesh)
  n_e = pk.n_rows
  e = torch.randn((n_e, D), device=dev)
  tab_s, tab_r = torch.randn((n_mesh, D), device=dev), torch.randn((n_mesh, D), device=dev)
  agg = torch.empty((n_mesh, D), device=dev)
  partial = torch.empty((2 * n_e // 64, D), device=dev)
  snd, rcv, flags = up(pk.senders), up(pk.receivers), up(pk.tile_flags)
  n_g = 1038240
  hg = torch.randn((n_g, D), device=dev)
  og = torch.empty((n_g, D), device=dev)
  prec = nat.PRECISIONS[args.prec]

  def desc(mode, n_rows):
    d = nat.RowMlpDesc()
    d.mode, d.n_rows, d.prec = mode, n_rows, prec
    d.layout = nat.LAYOUT_COLOWN if (co and mode == nat.MODE_MLP_LN) else nat.LAYOUT_CHUNKED
    return d

  def proc_edge():
    d = desc(nat.MODE_MLP_LN, n_e)
    d.a0, d.lda0, d.k0, d.w1p, d.b1 = e.data_ptr(), D, D, w1.data_ptr(), vec.data_ptr()
    d.g0, d.idx0, d.g1, d.idx1 = tab_s.data_ptr(), snd.data_ptr(), tab_r.data_ptr(), rcv.data_ptr()
    d.w2p, d.b2, d.n2 = w2.data_ptr(), vec.data_ptr(), D
    d.ln_scale, d.ln_offset = one.data_ptr(), vec.data_ptr()
    d.res, d.ldres, d.out, d.ldo = e.data_ptr(), D, e.data_ptr(), D
    d.seg, d.tile_flags, d.agg, d.partial = rcv.data_ptr(), flags.data_ptr(), agg.data_ptr(), partial.data_ptr()
    return d, 32, n_e

  def gemm_only_mlp():        # MLP_LN on the same rows without gathers / residual / segment-sum
    d = desc(nat.MODE_MLP_LN, n_e)
    d.a0, d.lda0, d.k0, d.w1p, d.b1 = e.data_ptr(), D, D, w1.data_ptr(), vec.data_ptr()
    d.w2p, d.b2, d.n2 = w2.data_ptr(), vec.data_ptr(), D
    d.ln_scale, d.ln_offset = one.data_ptr(), vec.data_ptr()
    d.out, d.ldo = agg.data_ptr(), 0          # every row stores to the same 2 KiB: no HBM write stream
    return d, 32, n_e

  def linear_grid():
    d = desc(nat.MODE_LINEAR, n_g)
    d.a0, d.lda0, d.k0, d.w1p = hg.data_ptr(), D, D, (w1lin if co else w1).data_ptr()
    d.out, d.ldo = og.data_ptr(), D
    return d, 16, n_g

  def node_grid():            # dec_node-like: [h | agg] (K = 1024) -> MLP -> LN -> residual
    d = desc(nat.MODE_MLP_LN, n_g)
    d.a0, d.lda0, d.k0, d.a1, d.lda1, d.k1 = hg.data_ptr(), D, D, og.data_ptr(), D, D
    d.w1p, d.b1 = w1b.data_ptr(), vec.data_ptr()
    d.w2p, d.b2, d.n2 = w2.data_ptr(), vec.data_ptr(), D
    d.ln_scale, d.ln_offset = one.data_ptr(), vec.data_ptr()
    d.res, d.ldres, d.out, d.ldo = hg.data_ptr(), D, hg.data_ptr(), D
    return d, 48, n_g

  shapes = dict(proc_edge=proc_edge, gemm_only_mlp=gemm_only_mlp, linear_grid=linear_grid, node_grid=node_grid)
  builds = [("pipe2", ["-DGC_PIPE=2"]), ("pipe2_c_flush", ["-DGC_PIPE=2", "-DGC_ASM_FLUSH=0"]),
            ("pipe2_dma_builtin", ["-DGC_PIPE=2", "-DGC_DMA_ASM=0"]),
            ("pipe2_noride", ["-DGC_PIPE=2", "-DGC_RIDE=0"]),
            ("pipe2_nodmawait", ["-DGC_PIPE=2", "-DGC_EXP=8"]),
            ("pipe1", ["-DGC_PIPE=1"]),
            ("pipe2_nodma", ["-DGC_PIPE=2", "-DGC_EXP=1"]),
            ("pipe2_noreads", ["-DGC_PIPE=2", "-DGC_EXP=2"]),
            ("pipe2_nomfma", ["-DGC_PIPE=2", "-DGC_EXP=4"]),
            ("pipe2_nodma_noreads", ["-DGC_PIPE=2", "-DGC_EXP=3"]),
            ("pipe2_nosched", ["-DGC_PIPE=2", "-DGC_SCHED_PIN=0"])]
  if co:
    builds = [("co", []), ("co_stg8", ["-DCO_STAGGER_US=8"]), ("co_stg15", ["-DCO_STAGGER_US=15"]),
              ("co_stg22", ["-DCO_STAGGER_US=22"]), ("co_a3", ["-DCO_AHEAD=3"]), ("co_a4", ["-DCO_AHEAD=4"]), ("co_a5", ["-DCO_AHEAD=5"]),
              ("co_stage_cached", ["-DCO_EXP=64"]), ("co_add_cached", ["-DCO_EXP=128"]),
              ("co_all_cached_nostore", ["-DCO_EXP=208"]),
              ("co_noadd", ["-DCO_EXP=1"]), ("co_nostage", ["-DCO_EXP=2"]),
              ("co_noadd_nostage", ["-DCO_EXP=3"]), ("co_noweights", ["-DCO_EXP=4"]),
              ("co_nomfma", ["-DCO_EXP=8"]), ("co_nostore", ["-DCO_EXP=16"]),
              ("co_nohbm", ["-DCO_EXP=19"]), ("co_nohbm_nobarrier", ["-DCO_EXP=51"]),
              ("co_noweights_nohbm", ["-DCO_EXP=23"])]
    shapes.pop("linear_grid")
  if co and os.environ.get("PROBE_TRACE"):
    lib = build("cotrace", ["-DCO_TRACE=1"] + os.environ.get("PROBE_DEFINES", "").split())
    out = {}
    names = ["prologue", "layer1", "boundary", "layer2", "ln_stats", "ytile", "colpass", "segsum"]
    for name in ("proc_edge", "gemm_only_mlp", "node_grid"):
      d, chunks, rows = shapes[name]()
      tiles = (rows + 63) // 64
      if d.seg:      # trace behind the partial rows (CO_MARK)
        buf = torch.zeros((2 * tiles * D + tiles * 32,), dtype=torch.float32, device=dev)
        d.partial = buf.data_ptr()
        ms = time_launch(lib, d, iters=1)
        t = buf[2 * tiles * D:].view(torch.int64).view(tiles, 16).cpu().numpy()
      else:
        trace = torch.zeros((tiles, 16), dtype=torch.int64, device=dev)
        d.partial = trace.data_ptr()
        ms = time_launch(lib, d, iters=1)
        t = trace.cpu().numpy()
      ph = np.diff(t[:, :9], axis=1).astype(np.float64)
      row = {"ms": round(ms, 4), "tiles": int(tiles), "total_ticks_mean": float((t[:, 8] - t[:, 0]).mean())}
      for j, nme in enumerate(names):
        row[nme + "_ticks_mean"] = round(float(ph[:, j].mean()), 1)
      row["bulk_issue_to_landed"] = round(float((t[:, 10] - t[:, 9]).mean()), 1)
      row["bulk_convert_write"] = round(float((t[:, 11] - t[:, 10]).mean()), 1)
      row["bulk_barrier"] = round(float((t[:, 12] - t[:, 11]).mean()), 1)
      gaps, per_cu = [], []
      for cu in np.unique(t[:, 15]):
        sel = t[t[:, 15] == cu]
        sel = sel[np.argsort(sel[:, 13])]
        gaps.extend((sel[1:, 13] - sel[:-1, 14]).tolist())
        per_cu.append(len(sel))
      row["wg_gap_us_mean"] = round(float(np.mean(gaps)) / 100.0, 2) if gaps else None
      row["wg_us_mean"] = round(float((t[:, 14] - t[:, 13]).mean()) / 100.0, 2)
      row["launch_span_us"] = round(float(t[:, 14].max() - t[:, 13].min()) / 100.0, 1)
      row["tiles_per_cu_min_max"] = [int(min(per_cu)), int(max(per_cu))]
      row["n_cus"] = len(per_cu)
      row["start_to_bulk_issued"] = round(float((t[:, 9] - t[:, 0]).mean()), 1)
      out[name] = row
      print("cotrace", name, json.dumps(row), flush=True)
    with open(args.out, "w") as f:
      json.dump(out, f, indent=1)
    return
  if os.environ.get("PROBE_TRACE"):
    # phase timeline of one wave per workgroup (GC_TRACE build): cycles per phase + dispatch gaps
    lib = build("trace", ["-DGC_PIPE=2", "-DGC_TRACE=1"])
    out = {}
    for name in ("gemm_only_mlp", "linear_grid", "node_grid"):
      d, chunks, rows = shapes[name]()
      tiles = (rows + 63) // 64
      trace = torch.zeros((tiles, 16), dtype=torch.int64, device=dev)
      d.partial = trace.data_ptr()
      ms = time_launch(lib, d, iters=1)
      t = trace.cpu().numpy()
      ph = np.diff(t[:, :6], axis=1).astype(np.float64)
      names = ["prologue", "layer1", "swish_split", "layer2", "finish"]
      row = {"ms": round(ms, 4), "tiles": int(tiles), "wave_cycles_total_mean": float((t[:, 5] - t[:, 0]).mean())}
      for j, nme in enumerate(names):
        row[nme + "_cycles_mean"] = float(ph[:, j].mean())
      # dispatch gap: consecutive workgroups on the same CU (smid), wall clock 100 MHz
      gaps = []
      for cu in np.unique(t[:, 9]):
        sel = t[t[:, 9] == cu]
        sel = sel[np.argsort(sel[:, 8])]
        gaps.extend((sel[1:, 8] - sel[:-1, 10]).tolist())
      row["wg_gap_wallclock_ticks_mean"] = float(np.mean(gaps)) if gaps else None
      row["wg_wallclock_ticks_mean"] = float((t[:, 10] - t[:, 8]).mean())
      row["n_cus_seen"] = int(len(np.unique(t[:, 9])))
      out[name] = row
      print("trace", name, json.dumps(row), flush=True)
    with open(args.out, "w") as f:
      json.dump(out, f, indent=1)
    return
  only_b = os.environ.get("PROBE_BUILDS")
  only_s = os.environ.get("PROBE_SHAPES")
  if only_b:
    builds = [b for b in builds if b[0] in only_b.split(",")]
  if only_s:
    shapes = {k: v for k, v in shapes.items() if k in only_s.split(",")}
  results = {}
  for tag, defines in builds:
    lib = build(tag, defines)
    row = {}
    for name, make in shapes.items():
      d, chunks, rows = make()
      ms = time_launch(lib, d)
      tiles_per_cu = (rows + 63) // 64 / 256.0
      row[name] = {"ms": round(ms, 4), "chunks_per_tile": chunks,
                   "kcycles_per_chunk_at_2.4GHz": round(ms * 1e-3 * 2.4e9 / tiles_per_cu / chunks / 1e3, 2)}
    results[tag] = row
    print(tag, json.dumps(row), flush=True)
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  with open(args.out, "w") as f:
    json.dump(results, f, indent=1)


if __name__ == "__main__":
  main()
