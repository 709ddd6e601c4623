"""GPU parity of the GC_PREC_BF16 tier -- the reference's casting.Bfloat16Cast run
(utils/casting.py:31-65,155-205) -- through the C-ABI.

Oracle: oracle/gnn.py with ACTIVATIONS = "bf16": the reference's Python restated op by op, the
result of every jnp / lax operation rounded to bfloat16 (dots and the reductions of jnp.mean /
jnp.var accumulate in float32).  **Parity unpinned** against XLA's fusion choices and its bfloat16
scatter order (stated in oracle/gnn.py); so the bars are
  * per launch: <= 1 bfloat16 ulp rms -- rms(got - want) in units of the bfloat16 spacing at the rms magnitude
    of the reference values, 2^(floor(log2 rms(want)) - 7) (the HIP launch rounds where arrays are
    materialised and keeps LayerNorm's internals / the segment-sum in fp32: it differs from the
    op-by-op restatement by one ulp on a fraction of the elements -- the restatement's LayerNorm alone
    carries five roundings -- rarely by two); the rel-RMSE is printed next to it;
  * whole step: the distance of the HIP path to the float64 truth must not exceed the distance of
    the op-by-op bfloat16 restatement of the reference to that truth by more than 25 %.
Both distances are printed (-s) and recorded in profiles/ by the round's GPU session."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from graphcast_amd import _native as nat          # noqa: E402
from graphcast_amd import graphcast as gc         # noqa: E402
from graphcast_amd import packing                 # noqa: E402
from oracle import gnn as ognn                    # noqa: E402
from oracle import graphcast as ogc               # noqa: E402
from oracle import params as oparams              # noqa: E402

D = 512
ULP = 2.0 ** -8


@pytest.fixture(scope="module")
def dev():
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  return torch.device("cuda:0")


def rb(a):
  """float -> nearest bfloat16, as float64."""
  return packing.bf16_round(np.asarray(a, dtype=np.float32)).astype(np.float64)


def up_rows(a, dev):
  """[rows, 512] values -> the device tensor a GC_PREC_BF16 launch reads: bfloat16, pi order."""
  t = torch.from_numpy(np.ascontiguousarray(packing.to_pi(packing.bf16_round(np.asarray(a, np.float32)))))
  return t.to(torch.bfloat16).to(dev)


def down_rows(t):
  return packing.from_pi(t.float().cpu().numpy()).astype(np.float64)


def up(a, dev, dtype=np.float32):
  return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dtype))).to(dev)


def wimg(w, dev, np_cols=D, chained=True):
  return up(packing.pack_weight_bf16(w, np_cols=np_cols, chained=chained).view(np.int16), dev, np.int16)


def weight(rng, k, n):
  return (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)


def rel_rmse(got, want):
  return float(np.linalg.norm(np.asarray(got, np.float64) - want) / np.linalg.norm(want))


def ulp_rms(got, want):
  """rms error in units of the bfloat16 spacing (8 significand bits) at the rms magnitude of `want`."""
  got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
  scale = np.sqrt(np.mean(want ** 2))
  return float(np.sqrt(np.mean((got - want) ** 2)) / 2.0 ** (np.floor(np.log2(scale)) - 7))


def run(desc):
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  nat.check(nat.lib().gc_rowmlp(ctypes.byref(desc), stream), "gc_rowmlp")
  torch.cuda.synchronize()


def new_desc(n_rows):
  d = nat.RowMlpDesc()
  d.mode, d.n_rows, d.prec, d.layout, d.n2 = nat.MODE_MLP_LN, n_rows, nat.PREC_BF16, nat.LAYOUT_HALF, D
  return d


def mlp_ln_want(z_terms, p):
  """The launch restated op by op in bfloat16: z -> swish -> linear -> LayerNorm."""
  with ognn.activations("bf16"):
    z = ognn.act(np.asarray(z_terms, np.float32))
    h = ognn.swish(z)
    y = ognn.linear(h, p["w2"], p["b2"])
    return ognn.layer_norm(y, p["scale"], p["offset"]).astype(np.float64)


def case(rng, n_rows, k):
  return dict(a=rng.standard_normal((n_rows, k)).astype(np.float32) if k else None,
              w1=weight(rng, k, D) if k else None, b1=(0.1 * rng.standard_normal(D)).astype(np.float32),
              w2=weight(rng, D, D), b2=(0.1 * rng.standard_normal(D)).astype(np.float32),
              scale=(1.0 + 0.2 * rng.standard_normal(D)).astype(np.float32),
              offset=(0.2 * rng.standard_normal(D)).astype(np.float32))


def fill_common(d, t):
  d.b1, d.w2p, d.b2 = t["b1"].data_ptr(), t["w2"].data_ptr(), t["b2"].data_ptr()
  d.ln_scale, d.ln_offset = t["scale"].data_ptr(), t["offset"].data_ptr()


def vecs(p, dev):
  return {k: up(packing.bf16_round(p[k]), dev) for k in ("b1", "b2", "scale", "offset")}


@pytest.mark.parametrize("n_rows,k0,k1", [(64, 512, 0), (200, 512, 512), (64 * 600 + 7, 512, 0), (90, 32, 0)])
def test_node_like_launch_with_residual(dev, n_rows, k0, k1):
  """bfloat16 rows in, K = k0 + k1 (two sources = jraph.concatenated_args), residual, rows out."""
  rng = np.random.default_rng(n_rows + k0 + k1)
  p = case(rng, n_rows, k0 + k1)
  res = rng.standard_normal((n_rows, D)).astype(np.float32)
  # bfloat16 rows are [rows, 512] in pi order: K chunk c of a row is its positions 32c .. 32c + 31 = its
  # logical columns 32c .. 32c + 31 in the chained K order, so a K = 32 launch reads logical columns 0 .. 31
  a0 = np.zeros((n_rows, D), np.float32)
  a0[:, :min(k0, D)] = p["a"][:, :min(k0, D)]
  t = vecs(p, dev)
  t.update(a0=up_rows(a0, dev), res=up_rows(res, dev), w1=wimg(p["w1"], dev), w2=wimg(p["w2"], dev))
  if k1:
    t["a1"] = up_rows(p["a"][:, D:], dev)
  out = torch.zeros((n_rows, D), dtype=torch.bfloat16, device=dev)
  d = new_desc(n_rows)
  d.a0, d.lda0, d.k0, d.w1p = t["a0"].data_ptr(), D, k0, t["w1"].data_ptr()
  if k1:
    d.a1, d.lda1, d.k1 = t["a1"].data_ptr(), D, k1
  fill_common(d, t)
  d.res, d.ldres, d.out, d.ldo = t["res"].data_ptr(), D, out.data_ptr(), D
  run(d)
  first = out.clone()
  run(d)
  assert torch.equal(first, out)                                  # bitwise repeatable
  for pin in (nat.WG_ROWS_64, nat.WG_ROWS_128):                   # 64- and 128-row workgroups: the same bits
    out.zero_()
    d.flags = pin
    run(d)
    assert torch.equal(first, out), pin
  d.flags = 0
  z = rb(p["a"]) @ rb(p["w1"]) + rb(p["b1"])
  e = mlp_ln_want(z, p)
  want = rb(rb(res) + e)
  err, ulps = rel_rmse(down_rows(out), want), ulp_rms(down_rows(out), want)
  print(f"bf16 node-like launch n={n_rows} K={k0 + k1}: {ulps:.2f} ulp rms, rel-RMSE {err:.2e} vs the op-by-op bf16 restatement")
  assert ulps <= 1.0 and err <= 2 * ULP


@pytest.mark.parametrize("kind", ["mesh_like", "uniform3", "with_empty_and_skew", "many_tiles", "many_tiles_no_rows"])
def test_edge_launch_with_gathers_and_segment_sum(dev, kind):
  """d + g0[snd] + g1[rcv] (+ rows . W1) -> MLP -> LN -> residual / rows out + receiver segment-sum
  (fp32 run sums, bfloat16 aggregate rows; straddling runs through fp32 partials + gc_seg_fixup_bf16)."""
  rng = np.random.default_rng(7)
  if kind == "mesh_like":
    n_recv, deg = 400, rng.integers(5, 37, 400)
  elif kind == "uniform3":
    n_recv, deg = 500, np.full(500, 3)
  elif kind in ("many_tiles", "many_tiles_no_rows"):
    n_recv, deg = 9000, rng.integers(2, 9, 9000)
  else:
    n_recv, deg = 60, rng.integers(0, 12, 60)
    deg[0] = deg[59] = 0
    deg[7] = 500
  receivers = rng.permutation(np.repeat(np.arange(n_recv), deg))
  n_send = 90
  senders = rng.integers(0, n_send, len(receivers))
  pk = packing.pack_edges(senders, receivers, n_recv)
  use_rows = kind not in ("uniform3", "many_tiles_no_rows")       # the encoder / decoder edge updates have no GEMM-1 rows
  p = case(rng, pk.n_rows, D if use_rows else 0)
  dd = rng.standard_normal((pk.n_rows, D)).astype(np.float32)
  gs = rng.standard_normal((n_send, D)).astype(np.float32)
  gr = rng.standard_normal((n_recv, D)).astype(np.float32)
  t = vecs(p, dev)
  t.update(d=up_rows(dd, dev), gs=up_rows(gs, dev), gr=up_rows(gr, dev), w2=wimg(p["w2"], dev),
           snd=up(pk.senders, dev, np.int32), rcv=up(pk.receivers, dev, np.int32), flags=up(pk.tile_flags, dev, np.int32))
  agg = torch.full((n_recv, D), float("nan"), dtype=torch.bfloat16, device=dev)
  partial = torch.full((2 * pk.n_rows // 64, D), float("nan"), device=dev)
  out = torch.zeros((pk.n_rows, D), dtype=torch.bfloat16, device=dev)
  d = new_desc(pk.n_rows)
  if use_rows:
    t["a0"], t["w1"] = up_rows(p["a"], dev), wimg(p["w1"], dev)
    d.a0, d.lda0, d.k0, d.w1p = t["a0"].data_ptr(), D, D, t["w1"].data_ptr()
    d.res, d.ldres = t["a0"].data_ptr(), D
  d.d, d.ldd = t["d"].data_ptr(), D
  d.g0, d.idx0, d.g1, d.idx1 = t["gs"].data_ptr(), t["snd"].data_ptr(), t["gr"].data_ptr(), t["rcv"].data_ptr()
  fill_common(d, t)
  d.out, d.ldo = out.data_ptr(), D
  d.seg, d.tile_flags, d.agg, d.partial = t["rcv"].data_ptr(), t["flags"].data_ptr(), agg.data_ptr(), partial.data_ptr()
  lib = nat.lib()
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

  def launch(flags):
    agg.fill_(float("nan")); partial.fill_(float("nan")); out.zero_()
    d.flags = flags
    nat.check(lib.gc_rowmlp(ctypes.byref(d), stream), "gc_rowmlp")
    if len(pk.fix_recv):
      f = [up(x, dev, np.int32) for x in (pk.fix_recv, pk.fix_t0, pk.fix_t1)]
      nat.check(lib.gc_seg_fixup_bf16(len(pk.fix_recv), f[0].data_ptr(), f[1].data_ptr(), f[2].data_ptr(),
                                      partial.data_ptr(), agg.data_ptr(), stream), "gc_seg_fixup_bf16")
    if len(pk.empty_receivers):
      zr = up(pk.empty_receivers, dev, np.int32)
      nat.check(lib.gc_zero_rows_bf16(len(zr), zr.data_ptr(), agg.data_ptr(), stream), "gc_zero_rows_bf16")
    torch.cuda.synchronize()
    return out.clone(), agg.clone()

  wide = launch(nat.WG_ROWS_128)                    # eight-wave workgroups (two packed edge tiles each; odd tile counts too)
  narrow = launch(nat.WG_ROWS_64)
  assert torch.equal(wide[0], narrow[0]) and torch.equal(wide[1].view(torch.int16), narrow[1].view(torch.int16))
  launch(0)
  assert torch.equal(out, narrow[0]) and torch.equal(agg.view(torch.int16), narrow[1].view(torch.int16))
  if not use_rows:
    # round 6 (gc_tuning.bf16_stream, the default): a launch without GEMM-1 rows forms every K step's hidden pair on the
    # fly from addend loads four K steps ahead instead of gathering up front -- the same sums in the same order: the same
    # bits as the unstreamed launch, in both workgroup sizes, with three addend sources and with two (the encoder's edge
    # update has no receiver term: the absent source reads the zero row)
    assert nat.get_tuning().bf16_stream == 1
    prev = nat.set_tuning(bf16_stream=0)
    try:
      for pin in (nat.WG_ROWS_64, nat.WG_ROWS_128):
        unstreamed = launch(pin)
        assert torch.equal(unstreamed[0], narrow[0]) and torch.equal(unstreamed[1].view(torch.int16), narrow[1].view(torch.int16))
      d.g1, d.idx1 = None, None
      two_unstreamed = launch(nat.WG_ROWS_64)
    finally:
      nat.set_tuning(prev)
    for pin in (nat.WG_ROWS_64, nat.WG_ROWS_128):
      two = launch(pin)
      assert torch.equal(two[0], two_unstreamed[0]) and torch.equal(two[1].view(torch.int16), two_unstreamed[1].view(torch.int16))
    assert not torch.equal(two_unstreamed[0], narrow[0])          # (the receiver term does matter)
    d.g1, d.idx1 = t["gr"].data_ptr(), t["rcv"].data_ptr()
    launch(0)
  ok = pk.receivers >= 0
  z = rb(dd) + rb(gs)[np.maximum(pk.senders, 0)] + rb(gr)[np.maximum(pk.receivers, 0)] + rb(p["b1"])
  if use_rows:
    z = z + rb(p["a"]) @ rb(p["w1"])
  e = mlp_ln_want(z, p)
  rows_want = rb(rb(p["a"]) + e) if use_rows else e
  err_rows, ulp_rows = rel_rmse(down_rows(out)[ok], rows_want[ok]), ulp_rms(down_rows(out)[ok], rows_want[ok])
  agg_want = rb(ognn.segment_sum(e[ok], pk.receivers[ok], n_recv))
  got = down_rows(agg)
  assert np.isfinite(got).all(), "segment-sum left poisoned rows"
  err_agg, ulp_agg = rel_rmse(got, agg_want), ulp_rms(got, agg_want)
  print(f"bf16 edge launch {kind}: rows {ulp_rows:.2f} ulp rms (rel-RMSE {err_rows:.2e}), aggregate {ulp_agg:.2f} ulp rms "
        f"(rel-RMSE {err_agg:.2e})")
  assert ulp_rows <= 1.0 and err_rows <= 2 * ULP
  # an aggregate sums up to hundreds of rows that each differ by their own ulps: bounded relative to its size
  assert err_agg <= 2 * ULP


def _chain(d, k, w_img, kind, b=None, out=None, ldo=0, n=0):
  c = d.chain[k]
  c.wp, c.kind, c.w_scale = w_img.data_ptr(), kind, 1.0
  c.b = b.data_ptr() if b is not None else None
  c.out = out.data_ptr() if out is not None else None
  c.ldo, c.n = ldo, n


@pytest.mark.parametrize("n_rows,c_in", [(64, 471), (333, 183), (64 * 530, 471), (70, 20)])
def test_external_rows_and_chained_stages(dev, n_rows, c_in):
  """The grid embedder's shape: external fp32 rows read in place (natural order, any alignment) +
  a 32-column tail, rounded in registers (GC_ROWS_F32); then chained stages on the rows while they
  are in registers: ROWS (bfloat16 pi rows), SWISH -> NARROW (the decoder's output MLP, fp32 out)."""
  rng = np.random.default_rng(c_in)
  batch, b, n_struct, n_out = 2, 1, 3, 227
  k_full = (c_in // 32) * 32
  kt = -(-(c_in + n_struct) // 32) * 32 - k_full
  x = rng.standard_normal((n_rows, batch, c_in)).astype(np.float32)
  st = rng.standard_normal((n_rows, n_struct)).astype(np.float32)
  p = case(rng, n_rows, c_in + n_struct)
  ws, w_hid, w_o = weight(rng, D, D), weight(rng, D, D), weight(rng, D, n_out)
  bh = (0.1 * rng.standard_normal(D)).astype(np.float32)
  bo = np.zeros(256, np.float32)
  bo[:n_out] = 0.2 * rng.standard_normal(n_out)
  tx, ts = up(x, dev), up(st, dev)
  xt = torch.full((n_rows, kt), float("nan"), device=dev)
  lib = nat.lib()
  stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  nat.check(lib.gc_prep_grid_tail(n_rows, batch, b, c_in, k_full, tx.data_ptr(), n_struct, ts.data_ptr(), kt,
                                  xt.data_ptr(), stream), "gc_prep_grid_tail")
  t = vecs(p, dev)
  t.update(w1=wimg(p["w1"], dev, chained=False), w2=wimg(p["w2"], dev), ws=wimg(ws, dev), wh=wimg(w_hid, dev),
           wo=wimg(w_o, dev, np_cols=256), bh=up(packing.bf16_round(bh), dev), bo=up(packing.bf16_round(bo), dev))
  out = torch.zeros((n_rows, D), dtype=torch.bfloat16, device=dev)
  pre = torch.zeros((n_rows, D), dtype=torch.bfloat16, device=dev)
  y = torch.zeros((n_rows, n_out), device=dev)

  def base():
    d = new_desc(n_rows)
    d.flags = nat.ROWS_F32
    if k_full:
      d.a0, d.lda0, d.k0 = tx.data_ptr() + 4 * b * c_in, batch * c_in, k_full
      d.a1, d.lda1, d.k1 = xt.data_ptr(), kt, kt
    else:
      d.a0, d.lda0, d.k0 = xt.data_ptr(), kt, kt
    d.w1p = t["w1"].data_ptr()
    fill_common(d, t)
    return d

  d = base()
  d.out, d.ldo = out.data_ptr(), D
  d.n_chain = 1
  _chain(d, 0, t["ws"], nat.CHAIN_ROWS, out=pre, ldo=D)
  d.flags = nat.ROWS_F32 | nat.WG_ROWS_128
  run(d)
  wide = (out.clone(), pre.clone())
  out.zero_(); pre.zero_()
  d.flags = nat.ROWS_F32 | nat.WG_ROWS_64
  run(d)
  assert torch.equal(wide[0], out) and torch.equal(wide[1], pre)         # 128- and 64-row workgroups: the same bits
  xin = np.concatenate([x[:, b], st], axis=1)
  rows_want = mlp_ln_want(rb(xin) @ rb(p["w1"]) + rb(p["b1"]), p)
  err, ulps = rel_rmse(down_rows(out), rows_want), ulp_rms(down_rows(out), rows_want)
  got_rows = down_rows(out)
  with ognn.activations("bf16"):
    pre_want = ognn.linear(got_rows.astype(np.float32), ws, np.zeros(D, np.float32)).astype(np.float64)
  err_pre, ulp_pre = rel_rmse(down_rows(pre), pre_want), ulp_rms(down_rows(pre), pre_want)
  d = base()
  d.n_chain = 2
  _chain(d, 0, t["wh"], nat.CHAIN_SWISH, b=t["bh"])
  _chain(d, 1, t["wo"], nat.CHAIN_NARROW, b=t["bo"], out=y, ldo=n_out, n=n_out)
  d.flags = nat.ROWS_F32 | nat.WG_ROWS_128
  run(d)
  y_wide = y.clone()
  y.zero_()
  d.flags = nat.ROWS_F32 | nat.WG_ROWS_64
  run(d)
  assert torch.equal(y_wide, y)
  with ognn.activations("bf16"):
    y_want = ognn.linear(ognn.swish(ognn.linear(got_rows.astype(np.float32), w_hid, bh)), w_o, bo[:n_out]).astype(np.float64)
  err_y, ulp_y = rel_rmse(y.cpu().numpy(), y_want), ulp_rms(y.cpu().numpy(), y_want)
  print(f"bf16 external rows c_in={c_in} n={n_rows}: rows {ulps:.2f} ulp rms ({err:.2e}), chained product {ulp_pre:.2f} "
        f"({err_pre:.2e}), output MLP {ulp_y:.2f} ({err_y:.2e})")
  assert ulps <= 1.0 and ulp_pre <= 1.0 and ulp_y <= 1.0
  assert max(err, err_pre, err_y) <= 2 * ULP
  yb = y.cpu().numpy()
  np.testing.assert_array_equal(yb, packing.bf16_round(yb))        # the narrow output holds bfloat16 values


def test_argument_validation(dev):
  d = new_desc(64)
  d.mode = nat.MODE_LINEAR
  lib = nat.lib()
  assert lib.gc_rowmlp(ctypes.byref(d), None) == -1
  assert b"GC_PREC_BF16 is built for" in lib.gc_last_error()


@pytest.fixture(scope="module")
def small():
  res, mesh_size, steps = 4.0, 3, 3
  lat = np.arange(-90, 90 + res / 2, res)
  lon = np.arange(0, 360, res)
  cfg = gc.ModelConfig(resolution=res, mesh_size=mesh_size, latent_size=512, gnn_msg_steps=steps,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_in, c_out = 183, gc.num_output_channels(gc.TASK_13)
  params = oparams.init_params(c_in, c_out, 512, steps, seed=1, nontrivial=True)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params, precision="bf16").init_from_coordinates(lat, lon)
  return dict(model=model, graphs=ogc.build_graphs(lat, lon, mesh_size), params=params, steps=steps, c_in=c_in)


@pytest.mark.parametrize("batch", [1, 2])
def test_whole_step_against_truth_and_the_bf16_restatement(small, batch):
  x = np.random.default_rng(batch).standard_normal((small["graphs"]["n_grid"], batch, small["c_in"])).astype(np.float32)
  got = small["model"].forward_grid_node_features(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
  truth = ogc.forward(small["params"], small["graphs"], x, steps=small["steps"], dtype=np.float64)
  with ognn.activations("bf16"):
    ref_bf16 = ogc.forward(small["params"], small["graphs"], x, steps=small["steps"], dtype=np.float32,
                           f32_aggregation=True)
  e_hip, e_ref, e_between = rel_rmse(got, truth), rel_rmse(ref_bf16, truth), rel_rmse(got, ref_bf16.astype(np.float64))
  print(f"BF16_TIER_PARITY batch={batch}: HIP vs fp64 truth {e_hip:.3e}; op-by-op bf16 restatement of the reference vs "
        f"truth {e_ref:.3e}; HIP vs restatement {e_between:.3e}")
  assert np.isfinite(got).all()
  np.testing.assert_array_equal(got, packing.bf16_round(got))      # predictions are bfloat16 values (cast back to fp32)
  assert e_hip <= 1.25 * e_ref
  assert e_between <= 2.0 * e_ref
  # ---- what a DIFFERENTLY-WRONG step of the same overall size could not pass (the two norms above
  # cannot tell a correct bf16 step from one that is off by as much in some other way).  The step's error against the
  # fp64 truth must LOOK like the restated reference's error -- rounding noise of the same bf16 pipeline: spread
  # evenly over the output channels, unbiased, without outlier rows.  A bug (a dropped addend, a channel mapping or
  # tile-edge error, a wrong bias) concentrates its error in some channels / rows or shifts their mean.
  err_hip = (got.astype(np.float64) - truth).reshape(-1, truth.shape[-1])
  err_ref = (ref_bf16.astype(np.float64) - truth).reshape(-1, truth.shape[-1])
  scale = np.sqrt((truth.reshape(-1, truth.shape[-1]) ** 2).mean(0))
  rms_hip, rms_ref = np.sqrt((err_hip ** 2).mean(0)) / scale, np.sqrt((err_ref ** 2).mean(0)) / scale
  ratio = rms_hip / rms_ref                                        # per output channel
  bias_hip = np.abs(err_hip.mean(0)) / np.sqrt((err_hip ** 2).mean(0))
  bias_ref = np.abs(err_ref.mean(0)) / np.sqrt((err_ref ** 2).mean(0))
  row_hip, row_ref = np.sqrt((err_hip ** 2).mean(1)), np.sqrt((err_ref ** 2).mean(1))       # per grid row (x batch)
  tail = lambda r: (float(np.quantile(r, 0.999) / np.sqrt((r ** 2).mean())), float(r.max() / np.sqrt((r ** 2).mean())))
  (q_hip, m_hip), (q_ref, m_ref) = tail(row_hip), tail(row_ref)
  print(f"BF16_TIER_ERROR_SHAPE batch={batch}: per-channel rms ratio HIP / restatement min {ratio.min():.2f} median "
        f"{np.median(ratio):.2f} max {ratio.max():.2f}; |mean| / rms per channel: HIP max {bias_hip.max():.3f}, restatement max "
        f"{bias_ref.max():.3f}; worst rows / rms (99.9 %, max): HIP {q_hip:.2f} {m_hip:.2f}, restatement {q_ref:.2f} {m_ref:.2f}")
  # (measured on the MI355X, round 4: ratio 0.67 .. 0.90, median 0.78 -- the HIP step rounds less often than the op-by-op
  #  restatement; the bf16 pipeline's error IS biased in some channels, common-mode operand rounding: |mean| / rms up to
  #  0.77 for the HIP step, 0.69 for the restatement; worst rows 1.35 / 1.39 vs 1.35 / 1.40 times the rms)
  assert 0.5 <= ratio.min() and ratio.max() <= 1.25                # no channel carries a foreign error
  assert bias_hip.max() <= 1.25 * bias_ref.max() + 0.05            # no mean shift beyond the pipeline's own
  assert q_hip <= 1.25 * q_ref and m_hip <= 1.5 * m_ref            # no outlier rows (tile edges, high-degree receivers)
