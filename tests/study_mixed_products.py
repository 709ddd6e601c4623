"""Numerics study (CPU, numpy; not collected by pytest): can the two CORRECTION products of the
f16x3 arithmetic run in a narrower MFMA format?

The shipped fp32-grade step forms every product as  xh.wh + xh.wl + xl.wh  with x = xh + xl,
w = wh + wl split into float16 halves: three v_mfma_f32_16x16x32_f16 per 32 k.  The corrections are
2^-11 of the main product, so THEY only need ~2^-9 relative accuracy for an fp32-grade sum.  gfx950
has block-scaled MX MFMAs (v_mfma_scale_f32_16x16x128_f8f6f4: fp8 at 2x, fp6 / fp4 at ~3.7x the f16
rate, /opt/skills/guides/cdna_hip_programming.md section 3): both corrections of 64 k fit ONE
K = 128 instruction  [q(xh) | q(xl)] . [q(wl) ; q(wh)].

This script emulates that arithmetic inside the oracle (every `gnn.linear`) and reports the
whole-step relative RMSE against the float64 run, next to the emulated f16x3 and bf16-operand runs:

    python tests/study_mixed_products.py [--res 2.0 --mesh 4 --steps 16]

Test infrastructure only (imports oracle/).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gnn, graphcast as og, params as op      # noqa: E402


def f16(a):
  return a.astype(np.float16).astype(np.float64)


def minifloat(a, mbits, emin, vmax):
  """Round to nearest even on a (1, e, mbits) grid with gradual underflow below 2^emin, saturating."""
  a = np.asarray(a, np.float64)
  mag = np.abs(a)
  e = np.floor(np.log2(np.where(mag > 0, mag, 1.0)))
  e = np.maximum(e, emin)
  step = np.exp2(e - mbits)
  q = np.rint(mag / step) * step
  return np.sign(a) * np.minimum(q, vmax)


FORMATS = {           # mantissa bits, exponent of the smallest normal, largest value, exponent of the largest binade
    "fp8": (3, -6, 448.0, 8),          # OCP e4m3fn
    "fp6": (3, 0, 7.5, 2),             # e2m3
    "fp4": (1, 0, 6.0, 2),             # e2m1
}


def mx_quant(a, fmt, axis):
  """OCP MX: blocks of 32 along `axis` share a power-of-two scale 2^(floor(log2 max) - emax)."""
  mbits, emin, vmax, emax = FORMATS[fmt]
  a = np.moveaxis(np.asarray(a, np.float64), axis, -1)
  k = a.shape[-1]
  pad = (-k) % 32
  if pad:
    a = np.concatenate([a, np.zeros(a.shape[:-1] + (pad,))], -1)
  blk = a.reshape(a.shape[:-1] + (-1, 32))
  m = np.abs(blk).max(-1, keepdims=True)
  scale = np.exp2(np.floor(np.log2(np.where(m > 0, m, 1.0))) - emax)
  q = minifloat(blk / scale, mbits, emin, vmax) * scale
  q = q.reshape(a.shape)[..., :k]
  return np.moveaxis(q, -1, axis)


def make_linear(mode):
  def linear(x, w, b):
    x2 = np.asarray(x, np.float64).reshape(-1, x.shape[-1])
    w = np.asarray(w, np.float64)
    if mode == "bf16":
      y = gnn._bf16(x2.astype(np.float32)).astype(np.float64) @ gnn._bf16(w.astype(np.float32)).astype(np.float64)
    elif mode == "f16":
      y = f16(x2) @ f16(w)
    else:
      xh, wh = f16(x2), f16(w)
      xl, wl = f16(x2 - xh), f16(w - wh)
      y = xh @ wh
      if mode == "f16x3":
        y = y + (xh @ wl + xl @ wh)
      elif mode == "f16x2_w":          # weights split, activations one half
        y = y + xh @ wl
      else:                            # corrections in an MX format
        y = y + (mx_quant(xh, mode, 1) @ mx_quant(wl, mode, 0) + mx_quant(xl, mode, 1) @ mx_quant(wh, mode, 0))
    y = y.astype(np.float32).astype(np.float64)        # fp32 accumulator
    return y.reshape(x.shape[:-1] + (w.shape[1],)) + b
  return linear


def rel_rmse(a, b):
  return float(np.sqrt(np.sum((a - b) ** 2) / np.sum(b ** 2)))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--res", type=float, default=2.0)
  ap.add_argument("--mesh", type=int, default=4)
  ap.add_argument("--steps", type=int, default=16)
  ap.add_argument("--c-in", type=int, default=186)
  ap.add_argument("--c-out", type=int, default=83)
  ap.add_argument("--modes", default="f16x3,fp8,fp6,fp4,f16x2_w,f16,bf16")
  ap.add_argument("--out", default=None)
  args = ap.parse_args()
  lat = np.arange(-90, 90 + args.res / 2, args.res, dtype=np.float32)
  lon = np.arange(0, 360, args.res, dtype=np.float32)
  graphs = og.build_graphs(lat, lon, args.mesh)
  prm = op.init_params(args.c_in, args.c_out, 512, args.steps, seed=1, nontrivial=True)
  x = np.random.default_rng(0).standard_normal((lat.size * lon.size, 1, args.c_in)).astype(np.float32)
  t0 = time.time()
  truth = og.forward(prm, graphs, x, args.steps, dtype=np.float64)
  print(f"float64 run: {time.time() - t0:.1f} s, grid {lat.size}x{lon.size}, mesh M{args.mesh}", flush=True)
  res = {}
  plain = gnn.linear
  # one GEMM in isolation (K = 512, unit-variance rows, 1/sqrt(K) weights)
  rng = np.random.default_rng(3)
  xa = rng.standard_normal((2048, 512))
  wa = rng.standard_normal((512, 512)) / np.sqrt(512)
  ya = xa @ wa
  for mode in args.modes.split(","):
    lin = make_linear(mode)
    one = rel_rmse(lin(xa, wa, 0.0), ya)
    gnn.linear = lin
    try:
      t0 = time.time()
      y = og.forward(prm, graphs, x, args.steps, dtype=np.float64)
    finally:
      gnn.linear = plain
    res[mode] = dict(one_gemm=one, whole_step=rel_rmse(y, truth))
    print(mode, json.dumps(res[mode]), f"({time.time() - t0:.0f} s)", flush=True)
  if args.out:
    with open(args.out, "w") as f:
      json.dump(dict(config=vars(args), results=res), f, indent=1)


if __name__ == "__main__":
  main()
