"""The C++ host packers behind the plan API (gc_host_pack_weight / gc_host_pack_edges,
csrc/gcast_plan.inc) against the numpy packers of graphcast_amd/packing.py, bit for bit; and the
plan API's argument checking.  No GPU needed: nothing here launches a kernel."""
import ctypes

import numpy as np
import pytest

from graphcast_amd import _native as nat
from graphcast_amd import packing


@pytest.fixture(scope="module")
def lib():
  return nat.lib()


# gc_host_pack_weight's image ids: the launch precisions + the plain bfloat16 image the GC_PREC_BF16 tier packs its
# parameters in (include/gcast.h: GC_PREC_BF16_IMAGE)
_PACK_ID = {"f32": nat.PREC_F32, "f16x3": nat.PREC_F16X3, "bf16image": nat.PREC_BF16_IMAGE}


@pytest.mark.parametrize("k,n,np_cols,chained", [(512, 512, 512, False), (474, 512, 512, False),
                                                 (512, 227, 256, True), (4, 512, 512, False),
                                                 (1024, 512, 512, False), (512, 512, 512, True)])
@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16image"])
def test_host_pack_weight_equals_numpy_packer(lib, prec, k, n, np_cols, chained):
  rng = np.random.default_rng(k + n)
  w = (rng.standard_normal((k, n)) / np.sqrt(k)).astype(np.float32)
  if prec == "f32":
    if chained:
      pytest.skip("the fp32 layout has one K order")
    want, want_scale = packing.pack_weight(w, np_cols=np_cols), 1.0
  elif prec == "f16x3":
    want_scale = packing.choose_weight_scale(w)
    want = packing.pack_weight_split(w, np_cols=np_cols, chained=chained, scale=want_scale)
  else:
    want, want_scale = packing.pack_weight_bf16(w, np_cols=np_cols, chained=chained), 1.0
  scale = ctypes.c_float(0)
  size = lib.gc_host_pack_weight(_PACK_ID[prec], int(chained), w.ctypes.data, k, n, np_cols, None,
                                 ctypes.byref(scale))
  assert size == want.nbytes
  got = np.empty(size, dtype=np.uint8)
  assert lib.gc_host_pack_weight(_PACK_ID[prec], int(chained), w.ctypes.data, k, n, np_cols,
                                 got.ctypes.data, ctypes.byref(scale)) == size
  assert scale.value == want_scale
  np.testing.assert_array_equal(got, np.ascontiguousarray(want).view(np.uint8).ravel())


def test_host_pack_weight_rejects_bad_arguments(lib):
  w = np.zeros((8, 8), np.float32)
  assert lib.gc_host_pack_weight(7, 0, w.ctypes.data, 8, 8, 512, None, None) == 0       # unknown precision
  assert lib.gc_host_pack_weight(0, 0, w.ctypes.data, 8, 600, 512, None, None) == 0     # wider than the layout
  # (huge weights are fine in split mode: the power-of-two scale keeps max |s w| <= 2^14)
  big = np.full((8, 8), 1e6, np.float32)
  scale = ctypes.c_float(0)
  assert lib.gc_host_pack_weight(1, 0, big.ctypes.data, 8, 8, 512, None, ctypes.byref(scale)) > 0
  assert 0 < scale.value * 1e6 <= 2.0 ** 14


def _cases():
  rng = np.random.default_rng(3)
  deg = rng.integers(0, 40, 300)
  deg[7] = 300
  recv = rng.permutation(np.repeat(np.arange(300), deg)).astype(np.int32)
  yield "skewed_with_empty", rng.integers(0, 50, len(recv)).astype(np.int32), recv, 300
  recv = np.repeat(np.arange(200), 3).astype(np.int32)               # uniform degree 3: padded tiles
  yield "uniform3", rng.integers(0, 9, len(recv)).astype(np.int32), rng.permutation(recv).astype(np.int32), 200
  recv = np.repeat(np.arange(50), 8).astype(np.int32)                # uniform degree 8: no padding
  yield "uniform8", rng.integers(0, 9, len(recv)).astype(np.int32), recv, 50


@pytest.mark.parametrize("name,senders,receivers,n_recv", list(_cases()), ids=[c[0] for c in _cases()])
def test_host_pack_edges_equals_numpy_packer(lib, name, senders, receivers, n_recv):
  want = packing.pack_edges(senders, receivers, n_recv)
  cap = 64 * (len(receivers) // 21 + 1)
  perm = np.full(cap, -9, np.int64)
  snd, rcv = np.full(cap, -9, np.int32), np.full(cap, -9, np.int32)
  flags = np.full(cap // 64, -9, np.int32)
  fix, empty = np.full(3 * n_recv, -9, np.int32), np.full(n_recv, -9, np.int32)
  n_fix, n_empty = ctypes.c_int(-1), ctypes.c_int(-1)
  n_rows = lib.gc_host_pack_edges(len(receivers), senders.ctypes.data, receivers.ctypes.data, n_recv,
                                  perm.ctypes.data, snd.ctypes.data, rcv.ctypes.data, flags.ctypes.data,
                                  fix.ctypes.data, ctypes.byref(n_fix), empty.ctypes.data, ctypes.byref(n_empty))
  assert n_rows == want.n_rows
  np.testing.assert_array_equal(perm[:n_rows], want.perm)
  np.testing.assert_array_equal(snd[:n_rows], want.senders)
  np.testing.assert_array_equal(rcv[:n_rows], want.receivers)
  np.testing.assert_array_equal(flags[:n_rows // 64], want.tile_flags)
  assert n_fix.value == len(want.fix_recv) and n_empty.value == len(want.empty_receivers)
  f = fix[:3 * n_fix.value].reshape(-1, 3)
  np.testing.assert_array_equal(f[:, 0], want.fix_recv)
  np.testing.assert_array_equal(f[:, 1], want.fix_t0)
  np.testing.assert_array_equal(f[:, 2], want.fix_t1)
  np.testing.assert_array_equal(empty[:n_empty.value], want.empty_receivers)


def test_host_pack_edges_rejects_bad_input(lib):
  s = np.zeros(4, np.int32)
  r = np.array([0, 1, 5, 1], np.int32)
  assert lib.gc_host_pack_edges(4, s.ctypes.data, r.ctypes.data, 3, *([None] * 8)) < 0      # receiver out of range
  assert lib.gc_host_pack_edges(0, s.ctypes.data, r.ctypes.data, 3, *([None] * 8)) < 0      # empty
  assert b"gc_host_pack_edges" in lib.gc_last_error()


def test_plan_api_argument_checks_without_gpu(lib):
  handle = ctypes.c_void_p()
  assert lib.gc_plan_create(None, None, 0, None, ctypes.byref(handle)) == -1
  assert lib.gc_plan_workspace_bytes(None, 1) == 0
  assert lib.gc_step_forward(None, None, None, 1, None, 0, None) == -1
  lib.gc_plan_destroy(None)                         # a no-op, like free(NULL)
  m = nat.ModelDesc()
  m.n_grid, m.n_mesh, m.c_in, m.c_out, m.num_steps = 10, 4, 5, 300, 1       # c_out beyond the output tile
  t = (nat.TensorDesc * 1)(nat.TensorDesc(b"x/w", None, 1, 1))
  assert lib.gc_plan_create(ctypes.byref(m), t, 1, None, ctypes.byref(handle)) == -1
  assert b"gc_plan_create" in lib.gc_last_error()


# ---- property tests (hypothesis): random graphs / matrices, C++ packers == numpy packers ----------
hypothesis = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st          # noqa: E402


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 60), st.integers(1, 400), st.integers(0, 2 ** 31 - 1), st.booleans())
def test_property_pack_edges_native_equals_numpy(n_recv, n_edges, seed, uniform):
  rng = np.random.default_rng(seed)
  lib = nat.lib()
  if uniform:                      # every receiver the same degree (exercises the padded-tile layout)
    deg = int(rng.integers(1, 9))
    receivers = rng.permutation(np.repeat(np.arange(n_recv), deg)).astype(np.int32)
  else:
    receivers = rng.integers(0, n_recv, n_edges).astype(np.int32)
  senders = rng.integers(0, 50, len(receivers)).astype(np.int32)
  want = packing.pack_edges(senders, receivers, n_recv)
  cap = 64 * (len(receivers) // 21 + 1)
  perm, snd, rcv = np.empty(cap, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int32)
  flags, fix, empty = np.empty(cap // 64, np.int32), np.empty(3 * n_recv, np.int32), np.empty(n_recv, np.int32)
  n_fix, n_empty = ctypes.c_int(), ctypes.c_int()
  n_rows = lib.gc_host_pack_edges(len(receivers), senders.ctypes.data, receivers.ctypes.data, n_recv,
                                  perm.ctypes.data, snd.ctypes.data, rcv.ctypes.data, flags.ctypes.data,
                                  fix.ctypes.data, ctypes.byref(n_fix), empty.ctypes.data, ctypes.byref(n_empty))
  assert n_rows == want.n_rows <= cap
  np.testing.assert_array_equal(perm[:n_rows], want.perm)
  np.testing.assert_array_equal(snd[:n_rows], want.senders)
  np.testing.assert_array_equal(rcv[:n_rows], want.receivers)
  np.testing.assert_array_equal(flags[:n_rows // 64], want.tile_flags)
  np.testing.assert_array_equal(fix[:3 * n_fix.value].reshape(-1, 3)[:, 0], want.fix_recv)
  np.testing.assert_array_equal(empty[:n_empty.value], want.empty_receivers)
  # invariants of the packed order itself: a permutation of the edges, receiver-sorted, padding trails
  ok = want.perm >= 0
  assert sorted(want.perm[ok]) == list(range(len(receivers)))
  assert (np.diff(want.receivers[ok]) >= 0).all()
  for t in range(n_rows // 64):
    tile = want.perm[64 * t:64 * t + 64] >= 0
    assert not (np.diff(tile.astype(int)) > 0).any()          # once padding starts, it stays


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 70), st.integers(1, 40), st.integers(0, 2 ** 31 - 1), st.sampled_from(["f32", "f16x3", "bf16image"]),
       st.booleans(), st.floats(1e-4, 30.0))
def test_property_pack_weight_native_equals_numpy(k, n, seed, prec, chained, magnitude):
  rng = np.random.default_rng(seed)
  lib = nat.lib()
  w = (magnitude * rng.standard_normal((k, n))).astype(np.float32)
  np_cols = 64
  if prec == "f32":
    want, want_scale = packing.pack_weight(w, np_cols=np_cols), 1.0
    chained = False
  elif prec == "f16x3":
    want_scale = packing.choose_weight_scale(w)
    want = packing.pack_weight_split(w, np_cols=np_cols, chained=chained, scale=want_scale)
  else:
    want, want_scale = packing.pack_weight_bf16(w, np_cols=np_cols, chained=chained), 1.0
  scale = ctypes.c_float(0)
  got = np.empty(want.nbytes, dtype=np.uint8)
  assert lib.gc_host_pack_weight(_PACK_ID[prec], int(chained), w.ctypes.data, k, n, np_cols,
                                 got.ctypes.data, ctypes.byref(scale)) == want.nbytes
  assert scale.value == want_scale
  np.testing.assert_array_equal(got, np.ascontiguousarray(want).view(np.uint8).ravel())
