"""GPU parity of the fused row-MLP kernel family (through the C-ABI) against the
oracle's numpy primitives in float64, in BOTH arithmetic modes of include/gcast.h:
"f32" (exact fp32 MFMA: an fp32 fma chain) and "f16x3" (operands split into two
halves, three f16 MFMAs per product, fp32 accumulation).  Per-element error is fp32
round-off class in both; we require rel-RMSE <= 2e-6 (f32) / 3e-6 (f16x3) -- two
orders inside the 1e-4 budget of BASELINE.json -- and max-abs <= 2e-4."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from graphcast_amd import _native as nat          # noqa: E402
from graphcast_amd import packing                 # noqa: E402
from oracle import gnn as ognn                    # noqa: E402

D = 512
# bf16: against the oracle that rounds the same GEMM operands.  A single fused MLP already
# differs from it by ~2e-5: the hidden activations are re-rounded to bf16, and an fp32-level
# difference d in a pre-rounding value becomes an rms difference sqrt(d * ulp_bf16) after it.
REL_RMSE_TOL = {"f32": 2e-6, "f16x3": 3e-6}
MAX_ABS_TOLS = {"f32": 2e-4, "f16x3": 2e-4}
MAX_ABS_TOL = 2e-4
_PREC = "f32"          # set per test by the autouse fixture below
_HALF = False          # f16x3: every launch runs the half-N formulation, two workgroups per CU (GC_LAYOUT_HALF)
_SCRATCH = {}          # keeps the GC_LAYOUT_HALF scratch slots alive until the launch has run


@pytest.fixture(autouse=True, params=["f32", "f16x3"])
def prec(request):
  # (round 5 retired the chunked f16x3 kernels and the "bf16gemm" tier: "f16x3" = GC_LAYOUT_HALF, "f32" = the chunked
  #  exact-fp32 kernel)
  global _PREC, _HALF
  _PREC = request.param
  _HALF = _PREC == "f16x3"
  return _PREC


class Image(np.ndarray):
  """A packed weight image that remembers the power of two it was multiplied by."""
  scale = 1.0


_SCALES = {}          # device pointer of an uploaded image -> its scale (see `up` / `apply_scales`)


def pw1(w):
  """Layer-1 weight image for the current arithmetic mode."""
  if _PREC == "f16x3":
    sc = packing.choose_weight_scale(w)
    img = packing.pack_weight_split(w, scale=sc).view(np.int16).view(Image)
    img.scale = sc
    return img
  return packing.pack_weight(w)


def r16(a):
  """What the current arithmetic mode does to a GEMM operand before multiplying (float64 out)."""
  a = np.asarray(a)
  return a.astype(np.float64)


def pw2(w, np_cols=D):
  """Layer-2 weight image (chained K order in split mode)."""
  if _PREC == "f16x3":
    sc = packing.choose_weight_scale(w)
    img = packing.pack_weight_split(w, np_cols=np_cols, chained=True, scale=sc).view(np.int16).view(Image)
    img.scale = sc
    return img
  return packing.pack_weight(w, np_cols=np_cols)


def apply_scales(d):
  d.w1_scale = _SCALES.get(d.w1p, 1.0)
  d.w2_scale = _SCALES.get(d.w2p, 1.0)


def new_desc(mode, n_rows):
  d = nat.RowMlpDesc()
  d.mode, d.n_rows, d.prec = mode, n_rows, nat.PRECISIONS[_PREC]
  d.layout = nat.LAYOUT_CHUNKED
  if _HALF:
    d.layout = nat.LAYOUT_HALF
    if mode == nat.MODE_MLP_LN:
      if "t" not in _SCRATCH:
        _SCRATCH["t"] = torch.empty((nat.SCRATCH_FLOATS,), dtype=torch.float32, device="cuda:0")
      _SCRATCH["t"].fill_(float("nan"))
      d.scratch = _SCRATCH["t"].data_ptr()
  return d


@pytest.fixture(scope="module")
def dev():
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  return torch.device("cuda:0")


def up(a, dev, dtype=np.float32):
  scale = getattr(a, "scale", None)
  a = np.asarray(a)
  if a.dtype == np.int16:          # split-f16 weight image: raw bits
    dtype = np.int16
  t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dtype))).to(dev)
  _SCALES[t.data_ptr()] = 1.0 if scale is None else scale    # (the allocator recycles addresses)
  return t


def run(desc):
  lib = nat.lib()
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  apply_scales(desc)
  nat.check(lib.gc_rowmlp(ctypes.byref(desc), stream), "gc_rowmlp")
  torch.cuda.synchronize()


def assert_close(got, want, what):
  got = np.asarray(got, dtype=np.float64)
  err = np.linalg.norm(got - want) / np.linalg.norm(want)
  assert np.isfinite(got).all(), what
  assert err <= REL_RMSE_TOL[_PREC], f"{what} [{_PREC}]: rel-RMSE {err:.3e}"
  assert np.abs(got - want).max() <= MAX_ABS_TOLS[_PREC] * max(1.0, np.abs(want).max()), what


def asymmetric_weight(rng, k, n):
  # random + a deterministic asymmetric ramp: a transposed or permuted K/N index shows up
  return (rng.standard_normal((k, n)) / np.sqrt(k) + 1e-3 * np.arange(k)[:, None] / k
          - 2e-3 * np.arange(n)[None, :] / n).astype(np.float32)


@pytest.mark.parametrize("n_rows", [64, 100, 1000])
@pytest.mark.parametrize("k", [32, 480, 512])
def test_linear_mode(dev, n_rows, k):
  rng = np.random.default_rng(k + n_rows)
  a = rng.standard_normal((n_rows, k)).astype(np.float32)
  w = asymmetric_weight(rng, k, D)
  b1 = rng.standard_normal(D).astype(np.float32)
  ta, tw, tb = up(a, dev), up(pw1(w), dev), up(b1, dev)
  out = torch.zeros((n_rows, D), device=dev)
  d = new_desc(nat.MODE_LINEAR, n_rows)
  d.a0, d.lda0, d.k0, d.w1p, d.b1 = ta.data_ptr(), k, k, tw.data_ptr(), tb.data_ptr()
  d.out, d.ldo = out.data_ptr(), D
  run(d)
  assert_close(out.cpu().numpy(), r16(a) @ r16(w) + b1, f"linear k={k}")


def test_linear_identity_weight_detects_transposes(dev):
  rng = np.random.default_rng(5)
  a = rng.standard_normal((64, D)).astype(np.float32)
  ta, tw = up(a, dev), up(pw1(np.eye(D, dtype=np.float32)), dev)
  out = torch.zeros((64, D), device=dev)
  d = new_desc(nat.MODE_LINEAR, 64)
  d.a0, d.lda0, d.k0, d.w1p = ta.data_ptr(), D, D, tw.data_ptr()
  d.out, d.ldo = out.data_ptr(), D
  run(d)
  if _PREC == "f32":
    np.testing.assert_array_equal(out.cpu().numpy(), a)   # exact: one product per output
  else:                                                    # x_hi + x_lo: 22 of x's 24 bits
    np.testing.assert_allclose(out.cpu().numpy(), a, rtol=2.0 ** -21, atol=2.0 ** -24)


def test_linear_with_gathers_and_direct_addend(dev):
  rng = np.random.default_rng(7)
  n_rows, n_src = 300, 50
  a = rng.standard_normal((n_rows, D)).astype(np.float32)
  w = asymmetric_weight(rng, D, D)
  dd = rng.standard_normal((n_rows, D)).astype(np.float32)
  g0 = rng.standard_normal((n_src, D)).astype(np.float32)
  g1 = rng.standard_normal((n_src, D)).astype(np.float32)
  i0 = rng.integers(0, n_src, n_rows).astype(np.int32)
  i1 = rng.integers(0, n_src, n_rows).astype(np.int32)
  t = [up(x, dev) for x in (a, pw1(w), dd, g0, g1)]
  ti0, ti1 = up(i0, dev, np.int32), up(i1, dev, np.int32)
  out = torch.zeros((n_rows, D), device=dev)
  d = new_desc(nat.MODE_LINEAR, n_rows)
  d.a0, d.lda0, d.k0, d.w1p = t[0].data_ptr(), D, D, t[1].data_ptr()
  d.d, d.ldd = t[2].data_ptr(), D
  d.g0, d.idx0, d.g1, d.idx1 = t[3].data_ptr(), ti0.data_ptr(), t[4].data_ptr(), ti1.data_ptr()
  d.out, d.ldo = out.data_ptr(), D
  run(d)
  want = r16(a) @ r16(w) + dd + g0[i0] + g1[i1]
  assert_close(out.cpu().numpy(), want, "linear+gathers")


def _mlp_ln_case(rng, n_rows, k0, k1):
  p = dict(
      a0=rng.standard_normal((n_rows, k0)).astype(np.float32) if k0 else None,
      a1=rng.standard_normal((n_rows, k1)).astype(np.float32) if k1 else None,
      w1=asymmetric_weight(rng, max(k0 + k1, 1), D) if k0 else None,
      b1=(0.3 * rng.standard_normal(D)).astype(np.float32),
      w2=asymmetric_weight(rng, D, D),
      b2=(0.3 * rng.standard_normal(D)).astype(np.float32),
      scale=(1 + 0.2 * rng.standard_normal(D)).astype(np.float32),
      offset=(0.2 * rng.standard_normal(D)).astype(np.float32))
  return p


def _mlp_ln_want(p, extra=0.0):
  z = extra + p["b1"].astype(np.float64)
  if p["a0"] is not None:
    a = p["a0"] if p["a1"] is None else np.concatenate([p["a0"], p["a1"]], axis=1)
    z = z + r16(a) @ r16(p["w1"])
  y = r16(ognn.swish(z)) @ r16(p["w2"]) + p["b2"]
  return ognn.layer_norm(y, p["scale"].astype(np.float64), p["offset"].astype(np.float64))


@pytest.mark.parametrize("n_rows,k0,k1", [(64, 512, 0), (130, 480, 0), (257, 512, 512), (64, 32, 0)])
def test_mlp_ln_mode_with_residual(dev, n_rows, k0, k1):
  rng = np.random.default_rng(n_rows + k0 + k1)
  p = _mlp_ln_case(rng, n_rows, k0, k1)
  res = rng.standard_normal((n_rows, D)).astype(np.float32)
  keep = [up(p["a0"], dev), up(pw1(p["w1"]), dev), up(p["b1"], dev),
          up(pw2(p["w2"]), dev), up(p["b2"], dev), up(p["scale"], dev),
          up(p["offset"], dev), up(res, dev)]
  ta1 = up(p["a1"], dev) if k1 else None
  out = torch.zeros((n_rows, D), device=dev)
  d = new_desc(nat.MODE_MLP_LN, n_rows)
  d.a0, d.lda0, d.k0, d.w1p, d.b1 = keep[0].data_ptr(), k0, k0, keep[1].data_ptr(), keep[2].data_ptr()
  if k1:
    d.a1, d.lda1, d.k1 = ta1.data_ptr(), k1, k1
  d.w2p, d.b2, d.n2 = keep[3].data_ptr(), keep[4].data_ptr(), D
  d.ln_scale, d.ln_offset = keep[5].data_ptr(), keep[6].data_ptr()
  d.res, d.ldres, d.out, d.ldo = keep[7].data_ptr(), D, out.data_ptr(), D
  run(d)
  assert_close(out.cpu().numpy(), _mlp_ln_want(p) + res, f"mlp_ln k0={k0} k1={k1}")
  # in place (out aliases res) must give the same bits
  inplace = keep[7].clone()
  d.res, d.out = inplace.data_ptr(), inplace.data_ptr()
  run(d)
  assert torch.equal(inplace, out)


@pytest.mark.parametrize("case", ["mesh_like", "skewed", "uniform3", "with_empty", "many_tiles"])
def test_edge_block_with_segment_sum(dev, case):
  """The fused edge kernel: gathers -> MLP -> LN -> segment-sum (+fixup, +zero rows)."""
  rng = np.random.default_rng(11)
  if case == "mesh_like":
    n_recv, deg = 400, rng.integers(5, 37, 400)
  elif case == "skewed":
    n_recv, deg = 40, rng.integers(1, 30, 40)
    deg[3], deg[4] = 700, 131
  elif case == "uniform3":
    n_recv, deg = 500, np.full(500, 3)
  elif case == "many_tiles":          # more tiles than persistent workgroups (GC_SCRATCH_SLOTS)
    n_recv, deg = 9000, rng.integers(2, 9, 9000)
    deg[17] = 300
  else:
    n_recv, deg = 200, rng.integers(0, 9, 200)
    deg[0] = deg[199] = 0
  receivers = rng.permutation(np.repeat(np.arange(n_recv), deg))
  n_send = 90
  senders = rng.integers(0, n_send, len(receivers))
  pk = packing.pack_edges(senders, receivers, n_recv)
  p = _mlp_ln_case(rng, pk.n_rows, 0, 0)
  dd = rng.standard_normal((pk.n_rows, D)).astype(np.float32)
  gs = rng.standard_normal((n_send, D)).astype(np.float32)
  gr = rng.standard_normal((n_recv, D)).astype(np.float32)
  t = dict(d=up(dd, dev), gs=up(gs, dev), gr=up(gr, dev), b1=up(p["b1"], dev),
           w2=up(pw2(p["w2"]), dev), b2=up(p["b2"], dev),
           sc=up(p["scale"], dev), of=up(p["offset"], dev),
           snd=up(pk.senders, dev, np.int32), rcv=up(pk.receivers, dev, np.int32),
           flags=up(pk.tile_flags, dev, np.int32))
  agg = torch.full((n_recv, D), float("nan"), device=dev)
  partial = torch.full((2 * pk.n_rows // 64, D), float("nan"), device=dev)
  out = torch.zeros((pk.n_rows, D), device=dev)
  d = new_desc(nat.MODE_MLP_LN, pk.n_rows)
  d.d, d.ldd = t["d"].data_ptr(), D
  d.g0, d.idx0, d.g1, d.idx1 = t["gs"].data_ptr(), t["snd"].data_ptr(), t["gr"].data_ptr(), t["rcv"].data_ptr()
  d.b1, d.w2p, d.b2, d.n2 = t["b1"].data_ptr(), t["w2"].data_ptr(), t["b2"].data_ptr(), D
  d.ln_scale, d.ln_offset = t["sc"].data_ptr(), t["of"].data_ptr()
  d.out, d.ldo = out.data_ptr(), D
  d.seg, d.tile_flags = t["rcv"].data_ptr(), t["flags"].data_ptr()
  d.agg, d.partial = agg.data_ptr(), partial.data_ptr()
  lib = nat.lib()
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

  def pipeline():
    apply_scales(d)
    nat.check(lib.gc_rowmlp(ctypes.byref(d), stream), "gc_rowmlp")
    if len(pk.fix_recv):
      f = [up(x, dev, np.int32) for x in (pk.fix_recv, pk.fix_t0, pk.fix_t1)]
      nat.check(lib.gc_seg_fixup(len(pk.fix_recv), f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr(),
                                 partial.data_ptr(), agg.data_ptr(), stream), "gc_seg_fixup")
    if len(pk.empty_receivers):
      z = up(pk.empty_receivers, dev, np.int32)
      nat.check(lib.gc_zero_rows(len(z), z.data_ptr(), agg.data_ptr(), stream), "gc_zero_rows")
    torch.cuda.synchronize()

  pipeline()
  ok = pk.receivers >= 0
  extra = dd.astype(np.float64) + gs[np.maximum(pk.senders, 0)] + gr[np.maximum(pk.receivers, 0)]
  e_new = _mlp_ln_want(p, extra)
  assert_close(out.cpu().numpy()[ok], e_new[ok], f"edge rows {case}")
  want = ognn.segment_sum(e_new[ok], pk.receivers[ok], n_recv)
  got = agg.cpu().numpy()
  assert np.isfinite(got).all(), "segment-sum left poisoned rows"
  scale = max(1.0, np.abs(want).max())
  assert np.abs(got - want).max() <= MAX_ABS_TOLS[_PREC] * scale
  assert np.linalg.norm(got - want) <= 2 * REL_RMSE_TOL[_PREC] * np.linalg.norm(want)
  # deterministic: a second run gives identical bits (no float atomics anywhere)
  first = agg.clone()
  agg.fill_(float("nan"))
  pipeline()
  assert torch.equal(first, agg)
  if not _HALF:
    return
  # round 4: the helper-wave form (GC_WG_HELPERS: four multiplying + four staging waves, one workgroup per CU)
  # runs the segment-sum's barriers with eight waves -- the same bits, rows and aggregate
  first_out = out.clone()
  d.flags = nat.WG_HELPERS
  agg.fill_(float("nan")); out.zero_()
  pipeline()
  assert torch.equal(first, agg) and torch.equal(first_out, out)
  # round 6: the WIDE form takes launches with a segment-sum too (GC_WG_WIDE: eight multiplying waves on one weight ring;
  # the 128-row tile's two 64-row sub-tiles run their segment-sums side by side; "skewed" / "with_empty" have an ODD
  # number of 64-row tiles -- the last wide tile's second half has no rows): the same bits, rows and aggregate
  d.flags = nat.WG_WIDE
  agg.fill_(float("nan")); out.zero_()
  pipeline()
  assert torch.equal(first, agg) and torch.equal(first_out, out)
  d.flags = 0
  # round 4: with the dynamic tile queue (more tiles than workgroups in "many_tiles"; a no-op otherwise) the
  # segment-sum's partial rows are still addressed by the TILE: the same bits whichever workgroup ran it
  queue = torch.zeros((2,), dtype=torch.int32, device=dev)
  d.tile_queue, d.flags = queue.data_ptr(), nat.TILE_QUEUE_ANY
  agg.fill_(float("nan")); out.zero_()
  pipeline()
  assert torch.equal(first, agg) and torch.equal(first_out, out) and queue.tolist() == [0, 0]
  d.flags = 0
  # ---- the ONE-PASS formulation of the same launch (GC_W2_NATURAL: no layer-1 GEMM, so every K chunk's
  #      hidden columns are formed on the fly from the addend rows; W2 in the natural K order; no scratch),
  #      with three addend sources and with two (the encoder edge update has no receiver term)
  sc2 = packing.choose_weight_scale(p["w2"])
  w2n_img = packing.pack_weight_split(p["w2"], chained=False, scale=sc2).view(np.int16).view(Image)
  w2n_img.scale = sc2
  w2n = up(w2n_img, dev)
  for with_g1 in (True, False):
    d.w2p, d.scratch = w2n.data_ptr(), None
    if not with_g1:
      d.g1, d.idx1 = None, None
    d.flags = nat.W2_NATURAL | nat.WG_HELPERS | nat.TILE_QUEUE_ANY      # (first in the helper-wave form: must give the same bits)
    agg.fill_(float("nan")); out.zero_()
    pipeline()
    helper_bits = (agg.clone(), out.clone())
    d.flags = nat.W2_NATURAL | nat.TILE_QUEUE_ANY
    agg.fill_(float("nan"))
    out.zero_()
    pipeline()
    extra1 = dd.astype(np.float64) + gs[np.maximum(pk.senders, 0)] + (gr[np.maximum(pk.receivers, 0)] if with_g1 else 0.0)
    e1 = _mlp_ln_want(p, extra1)
    assert_close(out.cpu().numpy()[ok], e1[ok], f"one-pass edge rows {case} g1={with_g1}")
    want1 = ognn.segment_sum(e1[ok], pk.receivers[ok], n_recv)
    got1 = agg.cpu().numpy()
    assert np.isfinite(got1).all()
    assert np.linalg.norm(got1 - want1) <= 2 * REL_RMSE_TOL[_PREC] * np.linalg.norm(want1)
    again = agg.clone()
    pipeline()
    assert torch.equal(again, agg)
    assert torch.equal(helper_bits[0], agg) and torch.equal(helper_bits[1], out)
    # round 6: the one-pass launch in the WIDE form (128 rows per weight ring), statically walked and through the queue
    for wide_flags in (nat.W2_NATURAL | nat.WG_WIDE, nat.W2_NATURAL | nat.WG_WIDE | nat.TILE_QUEUE_ANY):
      d.flags = wide_flags
      agg.fill_(float("nan")); out.zero_()
      pipeline()
      assert torch.equal(helper_bits[0], agg) and torch.equal(helper_bits[1], out), (case, with_g1, wide_flags)
      assert queue.tolist() == [0, 0]


@pytest.mark.parametrize("n_rows,n2,batch", [(64, 227, 1), (500, 83, 2), (70, 240, 1)])
def test_mlp_out_mode(dev, n_rows, n2, batch):
  rng = np.random.default_rng(n2)
  a = rng.standard_normal((n_rows, D)).astype(np.float32)
  w1, b1 = asymmetric_weight(rng, D, D), (0.3 * rng.standard_normal(D)).astype(np.float32)
  w2, b2 = asymmetric_weight(rng, D, n2), (0.3 * rng.standard_normal(n2)).astype(np.float32)
  t = [up(a, dev), up(pw1(w1), dev), up(b1, dev),
       up(pw2(w2, np_cols=256), dev), up(packing.pad_vector(b2, 256), dev)]
  out = torch.full((n_rows, batch, n2), 7.0, device=dev)
  d = new_desc(nat.MODE_MLP_OUT, n_rows)
  d.a0, d.lda0, d.k0, d.w1p, d.b1 = t[0].data_ptr(), D, D, t[1].data_ptr(), t[2].data_ptr()
  d.w2p, d.b2, d.n2 = t[3].data_ptr(), t[4].data_ptr(), n2
  b = batch - 1
  d.out, d.ldo = out.data_ptr() + 4 * b * n2, batch * n2
  run(d)
  want = r16(ognn.swish(r16(a) @ r16(w1) + b1)) @ r16(w2) + b2
  got = out.cpu().numpy()
  assert_close(got[:, b, :], want, f"mlp_out n2={n2}")
  if batch > 1:
    assert (got[:, 0, :] == 7.0).all()          # other batch element untouched


def test_split_mode_small_weights_and_large_rows(dev, prec):
  """What only the split mode can get wrong: (a) weights whose lo half is an fp16 SUBNORMAL
  (|w| ~ 1e-3: a flush-to-zero matrix core would leave an 11-bit weight, error ~2e-4),
  (b) rows beyond the fp16 range: the split saturates (hi clamps at 65504, lo carries the
  rest with 11 bits) instead of producing inf -- degraded but finite and small."""
  if prec != "f16x3":
    pytest.skip("split-mode specific")
  rng = np.random.default_rng(21)
  n_rows = 128
  a = rng.standard_normal((n_rows, D)).astype(np.float32)
  w = (1e-3 * rng.standard_normal((D, D))).astype(np.float32)
  ta, tw = up(a, dev), up(pw1(w), dev)
  out = torch.zeros((n_rows, D), device=dev)
  d = new_desc(nat.MODE_LINEAR, n_rows)
  d.a0, d.lda0, d.k0, d.w1p = ta.data_ptr(), D, D, tw.data_ptr()
  d.out, d.ldo = out.data_ptr(), D
  run(d)
  assert_close(out.cpu().numpy(), a.astype(np.float64) @ w, "subnormal lo halves")
  # fp16-subnormal ROW values as well (|x| ~ 1e-6: hi itself is subnormal)
  tiny = (a * np.float32(1e-6)).astype(np.float32)
  w2 = asymmetric_weight(rng, D, D)
  tt, tw2 = up(tiny, dev), up(pw1(w2), dev)
  d.a0, d.w1p = tt.data_ptr(), tw2.data_ptr()
  run(d)
  got, want = out.cpu().numpy().astype(np.float64), tiny.astype(np.float64) @ w2
  # absolute floor of the split: 2^-25 per element (fp16 subnormal spacing / 2)
  assert np.abs(got - want).max() <= 2.0 ** -24 * np.sqrt(D), "subnormal rows"
  big = a * np.float32(3.0e4)                  # |x| up to ~1.2e5 > 65504
  assert 7e4 < np.abs(big).max() < 1.3e5
  tb = up(big, dev)
  d.a0 = tb.data_ptr()
  run(d)
  got, want = out.cpu().numpy().astype(np.float64), big.astype(np.float64) @ w2
  assert np.isfinite(got).all()
  assert np.linalg.norm(got - want) / np.linalg.norm(want) < 5e-4


def test_prep_grid_input(dev):
  rng = np.random.default_rng(3)
  n, batch, c_in, kp = 1000, 3, 471, 480
  x = rng.standard_normal((n, batch, c_in)).astype(np.float32)
  st = rng.standard_normal((n, 3)).astype(np.float32)
  tx, ts = up(x, dev), up(st, dev)
  xin = torch.full((n, kp), float("nan"), device=dev)
  lib = nat.lib()
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  nat.check(lib.gc_prep_grid_input(n, batch, 1, c_in, tx.data_ptr(), 3, ts.data_ptr(), kp,
                                   xin.data_ptr(), stream), "gc_prep_grid_input")
  torch.cuda.synchronize()
  got = xin.cpu().numpy()
  np.testing.assert_array_equal(got[:, :c_in], x[:, 1])
  np.testing.assert_array_equal(got[:, c_in:c_in + 3], st)
  assert (got[:, c_in + 3:] == 0).all()


@pytest.mark.parametrize("n,batch,b,c_in,c0,n_struct,kt,offset", [
    (1000, 3, 1, 471, 448, 3, 32, 0),        # the 0.25 deg shape: [x[:, b, 448:] | struct | 0]
    (1037, 2, 0, 183, 160, 3, 32, 0),        # 1 deg; rows not a multiple of the block's 32
    (513, 1, 0, 20, 0, 3, 32, 0),            # fewer than 32 input channels: the tail is the whole input
    (257, 2, 1, 96, 64, 0, 32, 0),           # no structural features
    (300, 2, 1, 100, 64, 3, 64, 0),          # a 64-column tail
    (300, 2, 1, 100, 64, 3, 64, 1),          # xt not 16-byte aligned: the one-wave-per-row kernel
])
def test_prep_grid_tail(dev, n, batch, b, c_in, c0, n_struct, kt, offset):
  """gc_prep_grid_tail (round 6: one float4 of the output per thread where xt is 16-byte aligned) against the
  concat it replaces, graphcast.py:561-568, for the columns c0 .. c0 + kt - 1."""
  rng = np.random.default_rng(n + kt)
  x = rng.standard_normal((n, batch, c_in)).astype(np.float32)
  st = rng.standard_normal((n, max(n_struct, 1))).astype(np.float32)
  tx, ts = up(x, dev), up(st[:, :n_struct] if n_struct else st, dev)
  buf = torch.full((n * kt + 4,), float("nan"), device=dev)
  xt = buf[offset:offset + n * kt].view(n, kt)
  lib = nat.lib()
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  nat.check(lib.gc_prep_grid_tail(n, batch, b, c_in, c0, tx.data_ptr(), n_struct, ts.data_ptr() if n_struct else None,
                                  kt, xt.data_ptr(), stream), "gc_prep_grid_tail")
  torch.cuda.synchronize()
  want = np.zeros((n, kt), np.float32)
  want[:, :c_in - c0] = x[:, b, c0:]
  want[:, c_in - c0:c_in - c0 + n_struct] = st[:, :n_struct]
  np.testing.assert_array_equal(xt.cpu().numpy(), want)
  assert torch.isnan(buf[:offset]).all() and torch.isnan(buf[offset + n * kt:]).all()      # nothing written outside


# ---- GC_LAYOUT_HALF only: chained stages and in-place (unaligned) layer-1 rows ------------------
def _half_only():
  if not _HALF:
    pytest.skip("chained stages / unaligned rows exist in the half-N formulation only")


def _ln(y, scale, offset):
  mean = y.mean(axis=1, keepdims=True)
  var = np.square(y - mean).mean(axis=1, keepdims=True)
  return (y - mean) / np.sqrt(var + 1e-5) * scale + offset


def _chain_stage(d, k, w_img, kind, b=None, out=None, ldo=0, n=0):
  c = d.chain[k]
  c.wp, c.kind = w_img.data_ptr(), kind
  c.w_scale = _SCALES.get(w_img.data_ptr(), 1.0)
  c.b = b.data_ptr() if b is not None else None
  c.out = out.data_ptr() if out is not None else None
  c.ldo, c.n = ldo, n


@pytest.mark.parametrize("n_rows", [64, 200, 1000])
def test_half_chain_two_row_stages(dev, n_rows):
  """Node update followed by the next edge update's sender / receiver products (engine: proc_node
  -> h.W_s, h.W_r; reference typed_graph_net.py:431-453 after the pre-gather split): both chained
  Linear layers consume the rows the launch stores (LayerNorm output + residual)."""
  _half_only()
  rng = np.random.default_rng(n_rows)
  p = _mlp_ln_case(rng, n_rows, D, D)
  res = rng.standard_normal((n_rows, D)).astype(np.float32)
  ws, wr = asymmetric_weight(rng, D, D), asymmetric_weight(rng, D, D)
  br = (0.2 * rng.standard_normal(D)).astype(np.float32)
  t = {k: up(v, dev) for k, v in dict(a0=p["a0"], a1=p["a1"], b1=p["b1"], b2=p["b2"], scale=p["scale"],
                                       offset=p["offset"], res=res, br=br).items()}
  tw1, tw2, tws, twr = up(pw1(p["w1"]), dev), up(pw2(p["w2"]), dev), up(pw2(ws), dev), up(pw2(wr), dev)
  out = torch.zeros((n_rows, D), device=dev)
  o_s = torch.full((n_rows, D + 32), float("nan"), device=dev)           # a wider row stride
  o_r = torch.zeros((n_rows, D), device=dev)
  d = new_desc(nat.MODE_MLP_LN, n_rows)
  d.a0, d.lda0, d.k0, d.a1, d.lda1, d.k1 = t["a0"].data_ptr(), D, D, t["a1"].data_ptr(), D, D
  d.w1p, d.b1, d.w2p, d.b2, d.n2 = tw1.data_ptr(), t["b1"].data_ptr(), tw2.data_ptr(), t["b2"].data_ptr(), D
  d.ln_scale, d.ln_offset = t["scale"].data_ptr(), t["offset"].data_ptr()
  d.res, d.ldres, d.out, d.ldo = t["res"].data_ptr(), D, out.data_ptr(), D
  d.n_chain = 2
  _chain_stage(d, 0, tws, nat.CHAIN_ROWS, out=o_s, ldo=D + 32)
  _chain_stage(d, 1, twr, nat.CHAIN_ROWS, b=t["br"], out=o_r, ldo=D)
  run(d)
  h = _mlp_ln_want(p) + res
  assert_close(out.cpu().numpy(), h, "chain: stored rows")
  h32 = out.cpu().numpy().astype(np.float64)          # the chain consumes the rows AS STORED (fp32)
  assert_close(o_s.cpu().numpy()[:, :D], h32 @ ws.astype(np.float64), "chain stage 0 (rows . W_s)")
  assert torch.isnan(o_s[:, D:]).all()                # nothing written beyond the 512 columns
  assert_close(o_r.cpu().numpy(), h32 @ wr.astype(np.float64) + br, "chain stage 1 (rows . W_r + b)")


def test_node_update_of_one_and_a_half_rounds_is_split_into_a_wide_round_and_a_helper_tail(dev):
  """Round 6 (gc_tuning.split_tail): the 0.25 deg processor's node update -- 40,962 rows = 641 tiles, K = 512 + 512,
  residual, the next edge update's two products chained on -- is 1.6 rounds of four-wave pairs.  With split_tail set the
  launcher runs it as TWO launches: rows 0 .. 32,767 as one full round of 256 wide tiles, the last 129 tiles (the final one partial) in
  the helper form.  Same bits as the pinned four-wave form, in every output -- stored rows and both chained products --
  and against the float64 oracle on a row sample."""
  _half_only()
  n_rows = 40962
  rng = np.random.default_rng(7)
  p = _mlp_ln_case(rng, n_rows, D, D)
  res = rng.standard_normal((n_rows, D)).astype(np.float32)
  ws, wr = asymmetric_weight(rng, D, D), asymmetric_weight(rng, D, D)
  t = {k: up(v, dev) for k, v in dict(a0=p["a0"], a1=p["a1"], b1=p["b1"], b2=p["b2"], scale=p["scale"],
                                       offset=p["offset"], res=res).items()}
  tw1, tw2, tws, twr = up(pw1(p["w1"]), dev), up(pw2(p["w2"]), dev), up(pw2(ws), dev), up(pw2(wr), dev)

  def launch(flags):
    out = torch.full((n_rows, D), float("nan"), device=dev)
    o_s = torch.full((n_rows, D + 32), float("nan"), device=dev)
    o_r = torch.full((n_rows, D), float("nan"), device=dev)
    d = new_desc(nat.MODE_MLP_LN, n_rows)
    d.a0, d.lda0, d.k0, d.a1, d.lda1, d.k1 = t["a0"].data_ptr(), D, D, t["a1"].data_ptr(), D, D
    d.w1p, d.b1, d.w2p, d.b2, d.n2 = tw1.data_ptr(), t["b1"].data_ptr(), tw2.data_ptr(), t["b2"].data_ptr(), D
    d.ln_scale, d.ln_offset = t["scale"].data_ptr(), t["offset"].data_ptr()
    d.res, d.ldres, d.out, d.ldo = t["res"].data_ptr(), D, out.data_ptr(), D
    d.n_chain, d.flags = 2, flags
    _chain_stage(d, 0, tws, nat.CHAIN_ROWS, out=o_s, ldo=D + 32)
    _chain_stage(d, 1, twr, nat.CHAIN_ROWS, out=o_r, ldo=D)
    run(d)
    return out, o_s, o_r

  assert nat.get_tuning().helpers == -1
  want = launch(nat.WG_NO_HELPERS)
  prev = nat.set_tuning(split_tail=1)                  # (off by default: that stage 2 % faster, the power-limited step not)
  try:
    got = launch(0)                                    # no form pinned: the split rule applies
  finally:
    nat.set_tuning(prev)
  for a, b in zip(got, want):
    assert torch.equal(a[:, :D], b[:, :D])             # (NaN padding of the wider stride excluded)
    assert torch.isfinite(a[:, :D]).all()
  off = launch(0)
  for a, b in zip(off, want):
    assert torch.equal(a[:, :D], b[:, :D])
  pick = np.r_[0:3, 32766:32770, n_rows - 3:n_rows]    # the seam between the two launches and both ends
  sub = {k: (v[pick] if k in ("a0", "a1") else v) for k, v in p.items()}
  assert_close(got[0].cpu().numpy()[pick], _mlp_ln_want(sub) + res[pick], "split node update: stored rows")


@pytest.mark.parametrize("n_rows,n_out", [(64, 227), (333, 83), (500, 240)])
def test_half_chain_output_mlp(dev, n_rows, n_out):
  """Decoder node update with the output MLP chained on (engine: dec_node -> swish(h.W1 + b1).W2 + b2,
  reference deep_typed_graph_net.py:313-322); the node rows themselves are not stored."""
  _half_only()
  rng = np.random.default_rng(n_out)
  p = _mlp_ln_case(rng, n_rows, D, 0)
  res = rng.standard_normal((n_rows, D)).astype(np.float32)
  wo1, wo2 = asymmetric_weight(rng, D, D), asymmetric_weight(rng, D, n_out)
  bo1 = (0.2 * rng.standard_normal(D)).astype(np.float32)
  bo2 = (0.2 * rng.standard_normal(n_out)).astype(np.float32)
  t = {k: up(v, dev) for k, v in dict(a0=p["a0"], b1=p["b1"], b2=p["b2"], scale=p["scale"], offset=p["offset"],
                                       res=res, bo1=bo1, bo2=packing.pad_vector(bo2, 256)).items()}
  tw1, tw2 = up(pw1(p["w1"]), dev), up(pw2(p["w2"]), dev)
  two1, two2 = up(pw2(wo1), dev), up(pw2(wo2, np_cols=256), dev)
  y = torch.full((n_rows, 2, n_out), float("nan"), device=dev)       # batch-strided output rows
  d = new_desc(nat.MODE_MLP_LN, n_rows)
  d.a0, d.lda0, d.k0 = t["a0"].data_ptr(), D, D
  d.w1p, d.b1, d.w2p, d.b2, d.n2 = tw1.data_ptr(), t["b1"].data_ptr(), tw2.data_ptr(), t["b2"].data_ptr(), D
  d.ln_scale, d.ln_offset = t["scale"].data_ptr(), t["offset"].data_ptr()
  d.res, d.ldres = t["res"].data_ptr(), D                               # out stays NULL
  d.n_chain = 2
  _chain_stage(d, 0, two1, nat.CHAIN_SWISH, b=t["bo1"])
  c = d.chain[1]
  c.wp, c.kind, c.w_scale, c.b = two2.data_ptr(), nat.CHAIN_NARROW, _SCALES.get(two2.data_ptr(), 1.0), t["bo2"].data_ptr()
  c.out, c.ldo, c.n = y.data_ptr() + 4 * n_out, 2 * n_out, n_out       # batch element 1
  run(d)
  h = (_mlp_ln_want(p) + res).astype(np.float32).astype(np.float64)
  hid = ognn.swish(h @ wo1.astype(np.float64) + bo1)
  want = hid @ wo2.astype(np.float64) + bo2
  got = y.cpu().numpy()
  assert np.isnan(got[:, 0]).all()                    # batch element 0 untouched
  assert_close(got[:, 1], want, "chain: output MLP")


def test_half_reads_unaligned_rows_in_place(dev):
  """The grid embedder reads x[:, b, :448] where it lies (row stride B * 471 floats: 4-byte aligned
  rows) plus the 32-column tail [x[:, b, 448:] | struct | 0] of gc_prep_grid_tail, instead of the
  [N, 480] copy of gc_prep_grid_input (reference concat: graphcast.py:561-568)."""
  _half_only()
  rng = np.random.default_rng(12)
  n_rows, batch, c_in, n_struct, b = 333, 2, 471, 3, 1
  kp, k_full = 480, 448
  kt = kp - k_full
  x = rng.standard_normal((n_rows, batch, c_in)).astype(np.float32)
  st = rng.standard_normal((n_rows, n_struct)).astype(np.float32)
  w = asymmetric_weight(rng, c_in + n_struct, D)
  p = dict(b1=(0.3 * rng.standard_normal(D)).astype(np.float32), w2=asymmetric_weight(rng, D, D),
           b2=(0.3 * rng.standard_normal(D)).astype(np.float32),
           scale=(1 + 0.2 * rng.standard_normal(D)).astype(np.float32),
           offset=(0.2 * rng.standard_normal(D)).astype(np.float32))
  tx, ts = up(x, dev), up(st, dev)
  xt = torch.full((n_rows, kt), float("nan"), device=dev)
  lib = nat.lib()
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  nat.check(lib.gc_prep_grid_tail(n_rows, batch, b, c_in, k_full, tx.data_ptr(), n_struct, ts.data_ptr(), kt,
                                  xt.data_ptr(), stream), "gc_prep_grid_tail")
  torch.cuda.synchronize()
  want_tail = np.zeros((n_rows, kt), np.float32)
  want_tail[:, :c_in - k_full] = x[:, b, k_full:]
  want_tail[:, c_in - k_full:c_in - k_full + n_struct] = st
  np.testing.assert_array_equal(xt.cpu().numpy(), want_tail)
  t = {k: up(v, dev) for k, v in p.items() if k not in ("w2",)}
  tw1, tw2 = up(pw1(w), dev), up(pw2(p["w2"]), dev)
  out = torch.zeros((n_rows, D), device=dev)
  d = new_desc(nat.MODE_MLP_LN, n_rows)
  d.a0, d.lda0, d.k0 = tx.data_ptr() + 4 * b * c_in, batch * c_in, k_full
  d.a1, d.lda1, d.k1 = xt.data_ptr(), kt, kt
  d.w1p, d.b1, d.w2p, d.b2, d.n2 = tw1.data_ptr(), t["b1"].data_ptr(), tw2.data_ptr(), t["b2"].data_ptr(), D
  d.ln_scale, d.ln_offset = t["scale"].data_ptr(), t["offset"].data_ptr()
  d.out, d.ldo = out.data_ptr(), D
  run(d)
  z = np.concatenate([x[:, b], st], axis=1).astype(np.float64) @ w.astype(np.float64) + p["b1"]
  want = _ln(ognn.swish(z) @ p["w2"].astype(np.float64) + p["b2"], p["scale"], p["offset"])
  assert_close(out.cpu().numpy(), want, "in-place unaligned rows + tail")
  # round 5: the same launch (the grid embedder's shape: 4-byte aligned external rows, K = 448 + 32) in the wide form
  first = out.clone()
  out.fill_(float("nan"))
  d.flags = nat.WG_WIDE
  run(d)
  assert torch.equal(first, out)


def test_half_persistent_loop_revisits_scratch_slots(dev):
  """GC_LAYOUT_HALF launches are persistent: at most GC_SCRATCH_SLOTS workgroups walk the tiles
  (tile, tile + slots, ...), each parking its layer-2 / chain pass-0 accumulators in ITS slot of
  `scratch`.  More tiles than slots: every slot is rewritten by later tiles (stale parked rows,
  LayerNorm parameters and biases staged once per workgroup, the weight ring restarted per tile)."""
  _half_only()
  rng = np.random.default_rng(5)
  n_rows, n_out = 64 * (nat.SCRATCH_SLOTS + 90) + 5, 227
  p = _mlp_ln_case(rng, n_rows, D, D)
  res = rng.standard_normal((n_rows, D)).astype(np.float32)
  ws, wo = asymmetric_weight(rng, D, D), asymmetric_weight(rng, D, n_out)
  bo = np.zeros(256, np.float32)
  bo[:n_out] = 0.2 * rng.standard_normal(n_out)
  t = {k: up(v, dev) for k, v in dict(a0=p["a0"], a1=p["a1"], b1=p["b1"], b2=p["b2"], scale=p["scale"],
                                       offset=p["offset"], res=res, bo=bo).items()}
  tw1, tw2, tws, two = up(pw1(p["w1"]), dev), up(pw2(p["w2"]), dev), up(pw2(ws), dev), up(pw2(wo, np_cols=256), dev)
  out = torch.zeros((n_rows, D), device=dev)
  y = torch.zeros((n_rows, n_out), device=dev)
  d = new_desc(nat.MODE_MLP_LN, n_rows)
  d.a0, d.lda0, d.k0, d.a1, d.lda1, d.k1 = t["a0"].data_ptr(), D, D, t["a1"].data_ptr(), D, D
  d.w1p, d.b1, d.w2p, d.b2, d.n2 = tw1.data_ptr(), t["b1"].data_ptr(), tw2.data_ptr(), t["b2"].data_ptr(), D
  d.ln_scale, d.ln_offset = t["scale"].data_ptr(), t["offset"].data_ptr()
  d.res, d.ldres, d.out, d.ldo = t["res"].data_ptr(), D, out.data_ptr(), D
  d.n_chain = 2
  _chain_stage(d, 0, tws, nat.CHAIN_SWISH)
  _chain_stage(d, 1, two, nat.CHAIN_NARROW, b=t["bo"], out=y, ldo=n_out, n=n_out)
  run(d)
  first = (out.clone(), y.clone())
  run(d)                                              # bitwise repeatable (no atomics, fixed tile -> slot map)
  assert torch.equal(first[0], out) and torch.equal(first[1], y)
  # round 4: the same launch as ONE eight-wave workgroup per CU (four multiplying + four staging waves,
  # GC_WG_HELPERS) and with the XCD-contiguous tile map (GC_TILE_XCD) -- speed choices: the same bits
  # round 5: ... and as ONE workgroup of eight MULTIPLYING waves per CU on one weight ring (GC_WG_WIDE: 128-row tiles,
  # two scratch slots per workgroup; 301 tiles on 256 workgroups, the last one half empty)
  for flags in (nat.WG_HELPERS, nat.TILE_MAP_XCD, nat.WG_HELPERS | nat.TILE_MAP_XCD, nat.WG_WIDE):
    out.zero_(); y.zero_()
    d.flags = flags
    run(d)
    assert torch.equal(first[0], out) and torch.equal(first[1], y), flags
  # round 4: the dynamic tile queue (gc_rowmlp_desc.tile_queue: a workgroup takes its first tile by index, every
  # further one from a device counter) -- which workgroup runs which tile changes nothing; every launch leaves the
  # two words zero, so the next launch (and the next form) can reuse them
  queue = torch.zeros((2,), dtype=torch.int32, device=dev)
  d.tile_queue = queue.data_ptr()
  for flags in (nat.WG_NO_HELPERS, nat.WG_HELPERS, nat.WG_WIDE, nat.WG_NO_HELPERS):
    out.zero_(); y.zero_()
    d.flags = flags | nat.TILE_QUEUE_ANY      # (by default only launches of >= 4 tiles per workgroup use the queue)
    run(d)
    assert torch.equal(first[0], out) and torch.equal(first[1], y), ("tile queue", flags)
    assert queue.tolist() == [0, 0], queue.tolist()
  d.tile_queue = None
  d.flags = nat.WG_NO_HELPERS
  assert_close(out.cpu().numpy(), res.astype(np.float64) + _mlp_ln_want(p), "rows over many tiles per slot")
  h32 = out.cpu().numpy().astype(np.float64)
  want = ognn.swish(h32 @ ws.astype(np.float64)) @ wo.astype(np.float64) + bo[:n_out]
  assert_close(y.cpu().numpy(), want, "chained output MLP over many tiles per slot")


def test_small_launches_give_the_same_bits_in_both_kernel_forms(dev):
  """a node-side GC_LAYOUT_HALF launch of at most one tile per CU runs in the eight-wave helper form BY
  DEFAULT (gcast.hip: the `small` rule) -- LINEAR, MLP_OUT and chained launches of callers who pass no flags (DeepGNN,
  the conditioned encoder / decoder, every 1 deg launch).  Each of those shapes, with the four-wave form pinned
  (GC_WG_NO_HELPERS), the eight-wave form asked for (GC_WG_HELPERS) and no flag at all: the same bits."""
  _half_only()
  rng = np.random.default_rng(77)
  n_rows = 64 * 3 + 5

  def run_all_forms(d, outs):
    got = []
    for flags in (nat.WG_NO_HELPERS, nat.WG_HELPERS, 0):
      for o in outs:
        o.fill_(float("nan"))
      d.flags = flags
      run(d)
      got.append([o.clone() for o in outs])
    for form in got[1:]:
      for a, b in zip(got[0], form):
        assert torch.equal(a, b)
    assert all(torch.isfinite(o).all() for o in got[0])

  # LINEAR
  a = rng.standard_normal((n_rows, D)).astype(np.float32)
  w = asymmetric_weight(rng, D, D)
  ta, tw = up(a, dev), up(pw1(w), dev)
  out = torch.zeros((n_rows, D), device=dev)
  d = new_desc(nat.MODE_LINEAR, n_rows)
  d.a0, d.lda0, d.k0, d.w1p = ta.data_ptr(), D, D, tw.data_ptr()
  d.out, d.ldo = out.data_ptr(), D
  run_all_forms(d, [out])
  # MLP_OUT
  n2 = 227
  w1, b1 = asymmetric_weight(rng, D, D), (0.3 * rng.standard_normal(D)).astype(np.float32)
  w2, b2 = asymmetric_weight(rng, D, n2), (0.3 * rng.standard_normal(n2)).astype(np.float32)
  t = [up(pw1(w1), dev), up(b1, dev), up(pw2(w2, np_cols=256), dev), up(packing.pad_vector(b2, 256), dev)]
  y = torch.zeros((n_rows, n2), device=dev)
  d = new_desc(nat.MODE_MLP_OUT, n_rows)
  d.a0, d.lda0, d.k0, d.w1p, d.b1 = ta.data_ptr(), D, D, t[0].data_ptr(), t[1].data_ptr()
  d.w2p, d.b2, d.n2 = t[2].data_ptr(), t[3].data_ptr(), n2
  d.out, d.ldo = y.data_ptr(), n2
  run_all_forms(d, [y])
  # MLP_LN + residual + two chained row stages
  p = _mlp_ln_case(rng, n_rows, D, D)
  res = rng.standard_normal((n_rows, D)).astype(np.float32)
  ws, wr = asymmetric_weight(rng, D, D), asymmetric_weight(rng, D, D)
  tt = {k: up(v, dev) for k, v in dict(a0=p["a0"], a1=p["a1"], b1=p["b1"], b2=p["b2"], scale=p["scale"],
                                        offset=p["offset"], res=res).items()}
  tw1, tw2, tws, twr = up(pw1(p["w1"]), dev), up(pw2(p["w2"]), dev), up(pw2(ws), dev), up(pw2(wr), dev)
  o, o_s, o_r = (torch.zeros((n_rows, D), device=dev) for _ in range(3))
  d = new_desc(nat.MODE_MLP_LN, n_rows)
  d.a0, d.lda0, d.k0, d.a1, d.lda1, d.k1 = tt["a0"].data_ptr(), D, D, tt["a1"].data_ptr(), D, D
  d.w1p, d.b1, d.w2p, d.b2, d.n2 = tw1.data_ptr(), tt["b1"].data_ptr(), tw2.data_ptr(), tt["b2"].data_ptr(), D
  d.ln_scale, d.ln_offset = tt["scale"].data_ptr(), tt["offset"].data_ptr()
  d.res, d.ldres, d.out, d.ldo = tt["res"].data_ptr(), D, o.data_ptr(), D
  d.n_chain = 2
  _chain_stage(d, 0, tws, nat.CHAIN_ROWS, out=o_s, ldo=D)
  _chain_stage(d, 1, twr, nat.CHAIN_ROWS, out=o_r, ldo=D)
  run_all_forms(d, [o, o_s, o_r])
  # ... and in the wide form (GC_WG_WIDE; round 5): two 128-row tiles, the second one's waves 4.3 .. 7 beyond the rows
  want = [x.clone() for x in (o, o_s, o_r)]
  for x in (o, o_s, o_r):
    x.fill_(float("nan"))
  d.flags = nat.WG_WIDE
  run(d)
  for a, b in zip(want, (o, o_s, o_r)):
    assert torch.equal(a, b)


@pytest.mark.parametrize("n_recv", [300, 9000, 16500])
def test_processor_edge_update_in_both_kernel_forms(dev, n_recv):
  """The processor's edge update as the step launches it from step 1 on (engine: proc_edge -- typed_graph_net.py:431-453
  after the pre-gather split): e.W_e (a layer-1 GEMM over the edge rows) + b1 + (h.W_s)[senders] + (h.W_r)[receivers]
  -> MLP -> LayerNorm -> segment-sum, residual and output IN PLACE on the edge rows.  Round 5: in the eight-wave form
  the STAGING waves take residual + store (from the segment-sum's staging tile) and gather the NEXT tile's addends
  while the multiplying waves are in this tile's GEMMs, handing them over through LDS at the top of the next tile
  (csrc/rowmlp_half.inc: HST == 2).  Same bits as the four-wave form -- rows and aggregate -- with one tile per
  workgroup (n_recv 300) and with several tiles per workgroup, statically walked and handed out by the tile queue
  (9000: 700+ tiles on 256 workgroups); and against the float64 oracle."""
  _half_only()
  rng = np.random.default_rng(n_recv)
  deg = rng.integers(2, 9, n_recv)
  deg[7] = 300
  receivers = rng.permutation(np.repeat(np.arange(n_recv), deg))
  n_send = 120
  senders = rng.integers(0, n_send, len(receivers))
  pk = packing.pack_edges(senders, receivers, n_recv)
  p = _mlp_ln_case(rng, pk.n_rows, D, 0)
  gs = rng.standard_normal((n_send, D)).astype(np.float32)
  gr = rng.standard_normal((n_recv, D)).astype(np.float32)
  t = dict(gs=up(gs, dev), gr=up(gr, dev), b1=up(p["b1"], dev), w1=up(pw1(p["w1"]), dev),
           w2=up(pw2(p["w2"]), dev), b2=up(p["b2"], dev), sc=up(p["scale"], dev), of=up(p["offset"], dev),
           snd=up(pk.senders, dev, np.int32), rcv=up(pk.receivers, dev, np.int32), flags=up(pk.tile_flags, dev, np.int32))
  e0 = up(p["a0"], dev)
  e = torch.empty_like(e0)
  agg = torch.empty((n_recv, D), device=dev)
  partial = torch.empty((2 * pk.n_rows // 64, D), device=dev)
  queue = torch.zeros((2,), dtype=torch.int32, device=dev)
  d = new_desc(nat.MODE_MLP_LN, pk.n_rows)
  d.a0, d.lda0, d.k0, d.w1p = e.data_ptr(), D, D, t["w1"].data_ptr()
  d.g0, d.idx0, d.g1, d.idx1 = t["gs"].data_ptr(), t["snd"].data_ptr(), t["gr"].data_ptr(), t["rcv"].data_ptr()
  d.b1, d.w2p, d.b2, d.n2 = t["b1"].data_ptr(), t["w2"].data_ptr(), t["b2"].data_ptr(), D
  d.ln_scale, d.ln_offset = t["sc"].data_ptr(), t["of"].data_ptr()
  d.res, d.ldres, d.out, d.ldo = e.data_ptr(), D, e.data_ptr(), D
  d.seg, d.tile_flags = t["rcv"].data_ptr(), t["flags"].data_ptr()
  d.agg, d.partial = agg.data_ptr(), partial.data_ptr()
  lib = nat.lib()
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  fix = [up(x, dev, np.int32) for x in (pk.fix_recv, pk.fix_t0, pk.fix_t1)] if len(pk.fix_recv) else None

  def pipeline(flags, with_queue):
    e.copy_(e0)
    agg.fill_(float("nan")); partial.fill_(float("nan"))
    d.flags = flags
    d.tile_queue = queue.data_ptr() if with_queue else None
    apply_scales(d)
    nat.check(lib.gc_rowmlp(ctypes.byref(d), stream), "gc_rowmlp")
    if fix is not None:
      nat.check(lib.gc_seg_fixup(len(pk.fix_recv), fix[0].data_ptr(), fix[1].data_ptr(), fix[2].data_ptr(),
                                 partial.data_ptr(), agg.data_ptr(), stream), "gc_seg_fixup")
    torch.cuda.synchronize()
    assert queue.tolist() == [0, 0]
    return e.clone(), agg.clone()

  ref_e, ref_agg = pipeline(nat.WG_NO_HELPERS, False)
  ok = pk.receivers >= 0
  extra = gs[np.maximum(pk.senders, 0)].astype(np.float64) + gr[np.maximum(pk.receivers, 0)]
  e_new = _mlp_ln_want(p, extra)
  assert_close(ref_e.cpu().numpy()[ok], (p["a0"].astype(np.float64) + e_new)[ok], "processor edge rows (four-wave form)")
  want = ognn.segment_sum(e_new[ok], pk.receivers[ok], n_recv)
  has = np.bincount(pk.receivers[ok], minlength=n_recv) > 0
  got = ref_agg.cpu().numpy()
  assert np.linalg.norm(got[has] - want[has]) <= 2 * REL_RMSE_TOL[_PREC] * np.linalg.norm(want[has])
  # (round 6: ... and in the WIDE form, whose two-pass launches park ten of sixteen n-blocks in LDS)
  for flags, with_queue in ((nat.WG_HELPERS, False), (nat.WG_HELPERS | nat.TILE_QUEUE_ANY, True),
                            (nat.WG_NO_HELPERS | nat.TILE_QUEUE_ANY, True), (nat.WG_HELPERS, False),
                            (nat.WG_WIDE, False), (nat.WG_WIDE | nat.TILE_QUEUE_ANY, True)):
    got_e, got_agg = pipeline(flags, with_queue)
    assert torch.equal(got_e, ref_e), (flags, with_queue, float((got_e - ref_e).abs().max()))
    assert torch.equal(got_agg[torch.from_numpy(has).to(dev)], ref_agg[torch.from_numpy(has).to(dev)]), (flags, with_queue)
  # round 6 (gc_tuning.split_edges): NO form pinned.  Above 512 tiles (n_recv 9000: 708 = 512 + 196; 16500: 1,295 = 1,024 +
  # 271) the launcher runs the full rounds of 256 wide tiles in the wide form and a remainder of at most 256 tiles as a
  # SECOND launch in the helper form (a bigger remainder: one wide launch) -- pointers, receiver ids, straddle flags and
  # partial rows of the second launch offset by the first's tiles: the same bits, and the same with the rule off
  if n_recv >= 9000:
    assert pk.n_rows // 64 > 512 and (pk.n_rows // 64 % 512 <= 256) == (n_recv == 9000)
  # (off by default: measured slower at the two sizes it was built for, profiles/r06_s16_*)
  assert nat.get_tuning().helpers == -1
  prev = nat.set_tuning(split_edges=1)
  try:
    for flags, with_queue in ((0, False), (nat.TILE_QUEUE_ANY, True)):
      got_e, got_agg = pipeline(flags, with_queue)
      assert torch.equal(got_e, ref_e), ("split_edges", flags, float((got_e - ref_e).abs().max()))
      assert torch.equal(got_agg[torch.from_numpy(has).to(dev)], ref_agg[torch.from_numpy(has).to(dev)]), ("split_edges", flags)
  finally:
    nat.set_tuning(prev)
  got_e, got_agg = pipeline(0, False)                   # (the rule off: the helpers_edge rule's form)
  assert torch.equal(got_e, ref_e) and torch.equal(got_agg[torch.from_numpy(has).to(dev)], ref_agg[torch.from_numpy(has).to(dev)])
  # round 6 (GC_LATE_ADDENDS; gc_tuning.wide_late sets it on the launches the wide_edges rule makes wide): the gathered rows
  # added when the hidden layer is formed -- ((b1 + products) + g0) + g1 instead of (b1 + g0 + g1) + products: another fp32
  # ASSOCIATION, so not the unflagged launch's bits (1e-7 apart) -- but the SAME bits in both forms that honour the flag
  # (four-wave and wide; a flagged launch never runs in the helper form), statically walked and through the queue, and
  # as close to the float64 oracle as the unflagged launch
  late_e, late_agg = pipeline(nat.WG_NO_HELPERS | nat.LATE_ADDENDS, False)
  for flags, with_queue in ((nat.WG_WIDE | nat.LATE_ADDENDS, False), (nat.WG_WIDE | nat.LATE_ADDENDS | nat.TILE_QUEUE_ANY, True),
                            (nat.LATE_ADDENDS | nat.TILE_QUEUE_ANY, True), (nat.WG_HELPERS | nat.LATE_ADDENDS, False)):
    again_e, again_agg = pipeline(flags, with_queue)
    assert torch.equal(late_e, again_e), (flags, with_queue)
    assert torch.equal(late_agg[torch.from_numpy(has).to(dev)], again_agg[torch.from_numpy(has).to(dev)]), (flags, with_queue)
  assert not torch.equal(late_e, ref_e)
  assert float((late_e - ref_e).norm() / ref_e.norm()) < 5e-7
  assert_close(late_e.cpu().numpy()[ok], (p["a0"].astype(np.float64) + e_new)[ok], "processor edge rows (late addends)")
  got = late_agg.cpu().numpy()
  assert np.linalg.norm(got[has] - want[has]) <= 2 * REL_RMSE_TOL[_PREC] * np.linalg.norm(want[has])
