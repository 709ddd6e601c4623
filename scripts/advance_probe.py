#!/usr/bin/env python
"""A/B of gc_advance_state launches at the 0.25 deg / 37-level shape (rows 1,038,240, 474 -> 227 channels,
5 forcings): the in-tree library against others given as ADV_LIBS="tag:@relative/path.so;...".  Tables
shaped like rollout_device.build_tables' (window roll + residual update + forcing picks).  GPU box only.

    python scripts/advance_probe.py [--out gpurun_out/advance_probe.json]
"""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from graphcast_amd import _native as nat      # noqa: E402


def load(path):
  lib = ctypes.CDLL(path)
  lib.gc_advance_state.argtypes = [ctypes.POINTER(nat.AdvanceDesc), ctypes.c_void_p]
  lib.gc_advance_state.restype = ctypes.c_int
  lib.gc_last_error.restype = ctypes.c_char_p
  return lib


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "advance_probe.json"))
  ap.add_argument("--iters", type=int, default=20)
  args = ap.parse_args()
  dev = torch.device("cuda:0")
  n, c_in, c_out, nf = 1038240, 474, 227, 5
  rng = np.random.default_rng(0)
  x = torch.randn((n, c_in), device=dev)
  y = torch.randn((n, c_out), device=dev)
  f0, f1 = torch.randn((n, nf), device=dev), torch.randn((n, nf), device=dev)
  # state = [frame t-1 (227) | frame t (227) | forcings (10) | statics (2) | pad]: next frame t-1 <- frame t,
  # next frame t <- frame t + scale * y, forcings from f_cur / f_next, statics copied
  src_x, ax = np.full(c_in, -1, np.int32), np.zeros(c_in, np.float32)
  src_y, ay = np.full(c_in, -1, np.int32), np.zeros(c_in, np.float32)
  src_f = np.full(c_in, -1, np.int32)
  for c in range(227):
    src_x[c], ax[c] = 227 + c, 1.0
    src_x[227 + c], ax[227 + c] = 227 + c, 1.0
    src_y[227 + c], ay[227 + c] = c, rng.uniform(0.5, 2.0)
  for k in range(10):
    src_f[454 + k] = k
  for c in range(464, c_in):
    src_x[c], ax[c] = c, 1.0
  p_src_x = np.arange(227, 454, dtype=np.int32)
  p_ax = rng.uniform(0.5, 2.0, c_out).astype(np.float32)
  p_ay = rng.uniform(0.5, 2.0, c_out).astype(np.float32)
  p_b = rng.standard_normal(c_out).astype(np.float32)
  up = lambda a: torch.from_numpy(a).to(dev)
  tabs = [up(a) for a in (src_x, ax, src_y, ay, src_f, p_src_x, p_ax, p_ay, p_b)]
  xn = torch.empty_like(x)
  pred = torch.empty_like(y)
  d = nat.AdvanceDesc()
  d.n_rows, d.c_in, d.c_out, d.n_forc = n, c_in, c_out, nf
  d.x, d.y, d.f_cur, d.f_next, d.x_next, d.pred = (t.data_ptr() for t in (x, y, f0, f1, xn, pred))
  for name, t in zip(("src_x", "ax", "src_y", "ay", "src_f", "p_src_x", "p_ax", "p_ay", "p_b"), tabs):
    setattr(d, name, t.data_ptr())
  libs = [("in_tree", load(nat.library_path()))]
  for spec in filter(None, os.environ.get("ADV_LIBS", "").split(";")):
    tag, _, path = spec.partition(":@")
    libs.append((tag, load(os.path.join(ROOT, path))))
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  gbytes = 4.0 * n * (2 * c_in + 2 * c_out + 2 * nf) / 1e9
  out, first = {}, None
  for rnd in range(2):
    for tag, lib in libs:
      xn.zero_(); pred.zero_()
      assert lib.gc_advance_state(ctypes.byref(d), stream) == 0, lib.gc_last_error()
      torch.cuda.synchronize()
      if first is None:
        first = (xn.clone(), pred.clone())
      same = bool(torch.equal(xn, first[0]) and torch.equal(pred, first[1]))
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(args.iters):
        lib.gc_advance_state(ctypes.byref(d), stream)
      e1.record()
      torch.cuda.synchronize()
      ms = e0.elapsed_time(e1) / args.iters
      out.setdefault(tag, []).append(dict(ms=round(ms, 4), gb_per_s=round(gbytes / ms * 1e3, 1), bit_identical_to_first=same))
  print(json.dumps(out))
  os.makedirs(os.path.dirname(args.out), exist_ok=True)
  with open(args.out, "w") as f:
    json.dump(dict(shape=dict(rows=n, c_in=c_in, c_out=c_out, n_forc=nf), gbytes_per_launch=gbytes, results=out), f, indent=1)


if __name__ == "__main__":
  main()
