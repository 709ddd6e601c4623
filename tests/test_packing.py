"""Host-side layout logic (CPU): packed weights, receiver-sorted tile packing and the
segment-sum metadata.  The device algorithm (tile kernel + fixup + zero rows) is
emulated step by step in numpy and compared with a plain scatter-add."""
import numpy as np
import pytest

from graphcast_amd import packing


def test_pack_weight_roundtrip_and_layout():
  rng = np.random.default_rng(0)
  w = rng.standard_normal((474, 512)).astype(np.float32)
  wp = packing.pack_weight(w)
  assert wp.shape == (480 // 4, 512, 4) and wp.flags["C_CONTIGUOUS"]
  np.testing.assert_array_equal(packing.unpack_weight(wp, 474, 512), w)
  assert wp[5, 17, 2] == w[4 * 5 + 2, 17]
  assert (wp[474 // 4 + 1:] == 0).all()           # zero padded K rows
  w2 = rng.standard_normal((512, 227)).astype(np.float32)
  wp2 = packing.pack_weight(w2, np_cols=256)
  assert wp2.shape == (128, 256, 4)
  np.testing.assert_array_equal(packing.unpack_weight(wp2, 512, 227), w2)
  assert (wp2[:, 227:, :] == 0).all()
  with pytest.raises(ValueError):
    packing.pack_weight(w, np_cols=256)


def emulate_device_segment_sum(rows, pk, n_receivers):
  """Mirror of rowmlp_kernel's segment epilogue + seg_fixup_kernel + zero_rows_kernel."""
  n_tiles = pk.n_rows // packing.TILE
  agg = np.full((n_receivers, rows.shape[1]), np.nan, dtype=rows.dtype)   # poison
  partial = np.full((2 * n_tiles, rows.shape[1]), np.nan, dtype=rows.dtype)
  for t in range(n_tiles):
    segs = pk.receivers[t * 64:(t + 1) * 64]
    flags = pk.tile_flags[t]
    cur, run_start, acc = -1, 0, 0.0
    for r in range(65):
      sid = segs[r] if r < 64 else -2
      if sid != cur:
        if cur >= 0:
          if run_start == 0 and (flags & 1):
            partial[2 * t] = acc
          elif r == 64 and (flags & 2):
            partial[2 * t + 1] = acc
          else:
            assert np.isnan(agg[cur]).all(), "receiver written twice"
            agg[cur] = acc
        cur, run_start, acc = sid, r, 0.0
      if sid >= 0:
        acc = acc + rows[t * 64 + r]
  for rcv, t0, t1 in zip(pk.fix_recv, pk.fix_t0, pk.fix_t1):
    s = partial[2 * t0 + 1].copy()
    for t in range(t0 + 1, t1 + 1):
      s += partial[2 * t]
    assert np.isnan(agg[rcv]).all()
    agg[rcv] = s
  for rcv in pk.empty_receivers:
    agg[rcv] = 0
  return agg


@pytest.mark.parametrize("case", ["mesh_like", "skewed", "uniform3", "uniform4", "with_empty",
                                  "single_giant"])
def test_pack_edges_segment_sum(case):
  rng = np.random.default_rng(1)
  if case == "mesh_like":
    n_recv, deg = 500, rng.integers(5, 37, 500)
  elif case == "skewed":
    n_recv, deg = 60, rng.integers(1, 40, 60)
    deg[7], deg[8] = 700, 131                      # pole-like receivers spanning many tiles
  elif case == "uniform3":
    n_recv, deg = 1000, np.full(1000, 3)
  elif case == "uniform4":
    n_recv, deg = 100, np.full(100, 4)             # 64 % 4 == 0 -> plain packing
  elif case == "with_empty":
    n_recv, deg = 300, rng.integers(0, 9, 300)
    deg[0] = deg[299] = 0
  else:
    n_recv, deg = 3, np.array([1, 1000, 2])
  receivers = rng.permutation(np.repeat(np.arange(n_recv), deg))
  senders = rng.integers(0, 77, len(receivers))
  pk = packing.pack_edges(senders, receivers, n_recv)
  assert pk.n_rows % 64 == 0 and pk.n_edges == len(receivers)
  ok = pk.perm >= 0
  # packed rows are a permutation of the original edges with consistent indices
  np.testing.assert_array_equal(np.sort(pk.perm[ok]), np.arange(len(receivers)))
  np.testing.assert_array_equal(pk.senders[ok], senders[pk.perm[ok]])
  np.testing.assert_array_equal(pk.receivers[ok], receivers[pk.perm[ok]])
  assert (pk.senders[~ok] == -1).all() and (pk.receivers[~ok] == -1).all()
  valid_rcv = pk.receivers[ok]
  assert (np.diff(valid_rcv) >= 0).all()           # receiver-sorted
  # stable: original order is kept inside a segment
  same = valid_rcv[1:] == valid_rcv[:-1]
  assert (np.diff(pk.perm[ok])[same] > 0).all()
  if case == "uniform3":
    assert len(pk.fix_recv) == 0 and (pk.tile_flags == 0).all()
    assert pk.n_rows == -(-len(receivers) // 63) * 64
  rows = rng.standard_normal((pk.n_rows, 8))
  rows[~ok] = 1e30                                  # padding rows must never be read
  want = packing.segment_sum_packed_reference(rows, pk, n_recv)
  got = emulate_device_segment_sum(rows, pk, n_recv)
  assert not np.isnan(got).any()
  np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
  np.testing.assert_array_equal(pk.empty_receivers, np.flatnonzero(deg == 0))


def test_pack_edges_rejects_bad_input():
  with pytest.raises(ValueError):
    packing.pack_edges(np.array([], dtype=int), np.array([], dtype=int), 3)
  with pytest.raises(ValueError):
    packing.pack_edges(np.array([0]), np.array([3]), 3)


# ----------------------------------------------------------------------------- split-f16 layout
def test_split_f16_keeps_22_bits_and_subnormal_floor():
  rng = np.random.default_rng(0)
  x = (rng.standard_normal(100000) * np.exp(rng.uniform(-12, 8, 100000))).astype(np.float32)
  x = x[np.abs(x) < 6e4]
  hi, lo = packing.split_f16(x)
  err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - x)
  assert (err <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -25)).all()
  with pytest.raises(ValueError):
    packing.split_f16(np.array([7e4], np.float32))


@pytest.mark.parametrize("chained", [False, True])
@pytest.mark.parametrize("k,n,np_cols", [(474, 512, 512), (512, 227, 256), (4, 512, 512)])
def test_pack_weight_split_round_trip_and_lane_map(chained, k, n, np_cols):
  rng = np.random.default_rng(k + n)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  wp = packing.pack_weight_split(w, np_cols=np_cols, chained=chained)
  kp = packing.round_up(k, 32)
  assert wp.shape == (kp // 32, np_cols // 16, 2, 64, 8) and wp.dtype == np.uint16
  assert wp.nbytes == kp // 32 * np_cols * 128              # same chunk size as the fp32 layout
  hi, lo = packing.unpack_weight_split(wp, k, n, chained=chained)
  want_hi, want_lo = packing.split_f16(w)
  np.testing.assert_array_equal(hi, want_hi.astype(np.float32))
  np.testing.assert_array_equal(lo, want_lo.astype(np.float32))
  # explicit lane map of include/gcast.h: lane 16 g + n, element j
  v = wp.view(np.float16)
  for (c, nb, g, nn, j) in [(0, 0, 0, 0, 0), (kp // 32 - 1, np_cols // 16 - 1, 3, 15, 7), (0, 1, 2, 5, 3),
                            (0, 1, 2, 5, 4)]:
    kk = 32 * c + ((4 * g + j if j < 4 else 16 + 4 * g + j - 4) if chained else 8 * g + j)
    col = 16 * nb + nn
    want = want_hi[kk, col] if (kk < k and col < n) else np.float16(0)
    assert v[c, nb, 0, 16 * g + nn, j] == want


def test_split_product_error_is_fp32_class():
  """Numpy emulation of the GC_PREC_F16X3 arithmetic (three half products, wide accumulation):
  the representation error of x_hi.w_hi + x_lo.w_hi + x_hi.w_lo is ~1e-7 relative."""
  rng = np.random.default_rng(1)
  x = rng.standard_normal((64, 512)).astype(np.float32)
  w = (rng.standard_normal((512, 512)) / np.sqrt(512)).astype(np.float32)
  xh, xl = [a.astype(np.float64) for a in packing.split_f16(x)]
  wh, wl = [a.astype(np.float64) for a in packing.split_f16(w)]
  got = xh @ wh + xl @ wh + xh @ wl
  want = x.astype(np.float64) @ w
  assert np.linalg.norm(got - want) / np.linalg.norm(want) < 6e-7   # fp32 sgemm: 3e-7


def test_bf16_rounding_and_layout():
  torch = pytest.importorskip("torch")
  rng = np.random.default_rng(3)
  x = (rng.standard_normal(50000) * np.exp(rng.uniform(-20, 20, 50000))).astype(np.float32)
  want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
  np.testing.assert_array_equal(packing.bf16_round(x), want)            # round to nearest even
  w = (rng.standard_normal((474, 227)) / 20).astype(np.float32)
  for chained in (False, True):
    wp = packing.pack_weight_bf16(w, np_cols=256, chained=chained)
    assert wp.shape == (15, 16, 64, 8) and wp.dtype == np.uint16
    v = (wp.astype(np.uint32) << 16).view(np.float32)
    for (c, nb, g, n, j) in [(0, 0, 0, 0, 0), (14, 14, 3, 2, 7), (3, 7, 2, 5, 3), (3, 7, 2, 5, 4)]:
      kk = 32 * c + ((4 * g + j if j < 4 else 16 + 4 * g + j - 4) if chained else 8 * g + j)
      col = 16 * nb + n
      expect = packing.bf16_round(w)[kk, col] if (kk < 474 and col < 227) else 0.0
      assert v[c, nb, 16 * g + n, j] == expect
