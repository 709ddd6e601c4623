"""GPU parity of the Dataset-level path: GraphCast.__call__ (reference graphcast.py:298-329)
under normalization.InputsAndResiduals + rollout.chunked_prediction, with the state kept in
HBM between steps (torch-backed datasets), against the same wrappers around a Predictor whose
step is the float64 CPU oracle.  Tolerance: rel-RMSE <= 1e-4 (BASELINE.json) after 3
autoregressive steps; asserted 5e-5."""
import contextlib
import dataclasses
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from graphcast_amd import graphcast as gc          # noqa: E402
from graphcast_amd import model_utils              # noqa: E402
from graphcast_amd import normalization            # noqa: E402
from graphcast_amd import predictor_base           # noqa: E402
from graphcast_amd import rollout                  # noqa: E402
from graphcast_amd import synthetic                # noqa: E402
from graphcast_amd import xarray_lite as xarray    # noqa: E402
from oracle import graphcast as ogc                # noqa: E402
from oracle import params as oparams               # noqa: E402

RES, MESH, STEPS = 6.0, 2, 2
LAT = np.arange(-90, 90 + RES / 2, RES)
LON = np.arange(0, 360, RES)


@contextlib.contextmanager
def generic_loop():
  """rollout.chunked_prediction* with the predictor called chunk by chunk through its Dataset interface, as the
  reference does -- not the fused device loop that round 5 runs underneath a recognised stack."""
  old = os.environ.get("GCAST_ROLLOUT_FUSED")
  os.environ["GCAST_ROLLOUT_FUSED"] = "0"
  try:
    yield
  finally:
    if old is None:
      del os.environ["GCAST_ROLLOUT_FUSED"]
    else:
      os.environ["GCAST_ROLLOUT_FUSED"] = old


class OraclePredictor(predictor_base.Predictor):
  """The reference's GraphCast.__call__ with the float64 oracle as the step -- and the ORACLE's stacking
  (oracle/stacking.py: plain dims / arrays, independent of graphcast_amd.model_utils and xarray_lite) on both sides of
  it, so that a channel-order bug in the product's stacking shows up HERE too (this side used to
  stack with the product's own helpers)."""

  def __init__(self, params, graphs):
    self.params, self.graphs = params, graphs

  def __call__(self, inputs, targets_template, forcings, **kw):
    from oracle import gnn as ognn
    from oracle import stacking as ostack
    plain = lambda ds: {k: (tuple(ds[k].dims), np.asarray(ds[k].values, np.float64)) for k in ds.keys()}
    sizes = dict(inputs.sizes)
    x = ostack.grid_node_features(plain(inputs), plain(forcings), sizes)
    bf16 = ognn.ACTIVATIONS == "bf16"          # (the op-by-op bfloat16 restatement runs in float32 containers)
    y = ogc.forward(self.params, self.graphs, x, steps=STEPS, dtype=np.float32 if bf16 else np.float64,
                    f32_aggregation=bf16)
    template = {k: (tuple(targets_template[k].dims), tuple(targets_template[k].shape)) for k in targets_template.keys()}
    out = ostack.prediction_from_grid_nodes(y, template, len(LAT), len(LON))
    coords = {k: v.variable for k, v in targets_template.coords.items()}
    return xarray.Dataset({k: (template[k][0], out[k]) for k in sorted(out)}, coords=coords)


@pytest.fixture(scope="module")
def setup():
  if not torch.cuda.is_available():
    pytest.fail("GPU test selected but no GPU is visible")
  cfg = gc.ModelConfig(resolution=RES, mesh_size=MESH, latent_size=512, gnn_msg_steps=STEPS,
                       hidden_layers=1, radius_query_fraction_edge_length=0.6)
  c_in, c_out = 183, gc.num_output_channels(gc.TASK_13)
  params = oparams.init_params(c_in, c_out, 512, STEPS, seed=1, nontrivial=True)
  model = gc.GraphCast(cfg, gc.TASK_13, params=params)
  graphs = ogc.build_graphs(LAT, LON, MESH)
  return model, OraclePredictor(params, graphs)


def _rel(a, b):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_dataset_call_host_and_device_inputs(setup):
  model, oracle = setup
  inputs, template, forcings = synthetic.make_example(gc.TASK_13, LAT, LON, batch=2, seed=5)
  want = oracle(inputs, template, forcings)
  got_host = model(inputs, template, forcings)
  dev = lambda ds: synthetic.to_device(ds, "cuda:0")
  got_dev = model(dev(inputs), template, dev(forcings))
  for k in template.keys():
    assert got_host[k].dims == template[k].dims
    assert isinstance(got_host[k].data, np.ndarray)
    assert torch.is_tensor(got_dev[k].data) and got_dev[k].data.is_cuda
    assert _rel(got_host[k].values, want[k].values) < 2e-5, k
    np.testing.assert_array_equal(got_dev[k].values, got_host[k].values)


def test_normalised_rollout_resident_in_hbm(setup):
  model, oracle = setup
  n_steps = 3
  inputs, template, forcings = synthetic.make_example(gc.TASK_13, LAT, LON, num_target_steps=n_steps,
                                                      seed=7)
  mean, std, dstd = synthetic.make_stats(gc.TASK_13)
  wrap = lambda p: normalization.InputsAndResiduals(p, std, mean, dstd)
  ref = wrap(oracle)
  want = rollout.chunked_prediction(lambda rng, **kw: ref(**kw), None, inputs, template, forcings)
  dut = wrap(model)
  put = lambda ds: synthetic.to_device(ds, "cuda:0")
  for name, ctx in (("generic loop", generic_loop()), ("fused loop", contextlib.nullcontext())):
    with ctx:
      chunks = list(rollout.chunked_prediction_generator(
          lambda rng, **kw: dut(**kw), None, inputs, template, 1, forcings, device_put_fn=put))
      assert len(chunks) == n_steps
      assert all(c["temperature"].data.is_cuda for c in chunks)          # state never left the device
      got = rollout.chunked_prediction(lambda rng, **kw: dut(**kw), None, inputs, template, forcings,
                                       device_put_fn=put)
    worst = 0.0
    for k in template.keys():
      assert got[k].shape == want[k].shape
      for t in range(n_steps):
        tax = got[k].dims.index("time")
        e = _rel(np.take(got[k].values, t, axis=tax), np.take(want[k].values, t, axis=tax))
        worst = max(worst, e)
    print(f"3-step normalised rollout ({name}): worst per-variable per-step rel-RMSE {worst:.2e}")
    assert worst < 5e-5
    np.testing.assert_array_equal(got.coords["time"].values, template.coords["time"].values)


def test_wrapper_chain_on_host_datasets_runs_on_the_device(setup):
  """Reference-style use -- host (numpy) Datasets into InputsAndResiduals(Bfloat16Cast-less GraphCast) through the
  autoregressive Predictor: the OUTERMOST wrapper uploads once, every wrapper's Dataset arithmetic and the feedback
  run on device tensors (the GraphCast call sees torch-backed inputs), and host Datasets come back with the same
  values as the device-resident path (predictor_base.host_datasets_on_device)."""
  from graphcast_amd import autoregressive
  model, _ = setup
  n_steps = 2
  inputs, template, forcings = synthetic.make_example(gc.TASK_13, LAT, LON, num_target_steps=n_steps, seed=11)
  mean, std, dstd = synthetic.make_stats(gc.TASK_13)
  seen = []

  class Spy(predictor_base.Predictor):
    _device = model._device
    def __call__(self, inputs, targets_template, forcings, **kw):
      seen.append(all(torch.is_tensor(v.data) and v.data.is_cuda for v in inputs._vars.values()))
      return model(inputs, targets_template, forcings, **kw)

  assert predictor_base.device_of(autoregressive.Predictor(normalization.InputsAndResiduals(Spy(), std, mean, dstd))) == model._device
  dut = autoregressive.Predictor(normalization.InputsAndResiduals(Spy(), std, mean, dstd))
  got_host = dut(inputs, template, forcings)
  assert seen == [True] * n_steps
  put = lambda ds: synthetic.to_device(ds, "cuda:0")
  got_dev = dut(put(inputs), template, put(forcings))
  for k in template.keys():
    assert isinstance(got_host[k].data, np.ndarray) and got_host[k].dims == got_dev[k].dims
    assert torch.is_tensor(got_dev[k].data) and got_dev[k].data.is_cuda
    np.testing.assert_array_equal(got_host[k].values, got_dev[k].values)


def test_device_rollout_matches_dataset_rollout(setup):
  """rollout_device.DeviceRollout (gc_advance_state fusing normalisation + residual + window
  roll) against rollout.chunked_prediction(InputsAndResiduals(GraphCast)) on the same device
  step: the two differ only by fp32 rounding of the normalisation algebra."""
  from graphcast_amd import rollout_device
  model, _ = setup
  n_steps = 4
  inputs, template, forcings = synthetic.make_example(gc.TASK_13, LAT, LON, batch=2,
                                                      num_target_steps=n_steps, seed=13)
  mean, std, dstd = synthetic.make_stats(gc.TASK_13)
  dut = normalization.InputsAndResiduals(model, std, mean, dstd)
  put = lambda ds: synthetic.to_device(ds, "cuda:0")
  with generic_loop():
    want = rollout.chunked_prediction(lambda rng, **kw: dut(**kw), None, inputs, template, forcings,
                                      device_put_fn=put)
  roll = rollout_device.DeviceRollout(model, std, mean, dstd)
  traj = roll.run(inputs, template, forcings)
  assert traj.is_cuda and traj.shape[0] == n_steps
  got = roll.to_dataset(traj, template)
  worst = 0.0
  for k in template.keys():
    assert got[k].dims == want[k].dims and got[k].shape == want[k].shape
    worst = max(worst, _rel(got[k].values, want[k].values))
  print(f"device rollout vs dataset rollout, {n_steps} steps: worst per-variable rel diff {worst:.2e}")
  assert worst < 2e-5
  last = roll.run(inputs, template, forcings, keep_trajectory=False)
  assert torch.equal(last[0], traj[-1])
  np.testing.assert_array_equal(got.coords["time"].values, template.coords["time"].values)


def test_disabled_bfloat16_cast_is_the_wrapped_predictor(setup):
  """casting.Bfloat16Cast(enabled=False) -- the reference's switch (utils/casting.py:40-47) -- passes the call through."""
  from graphcast_amd import casting
  model, _ = setup
  inputs, template, forcings = synthetic.make_example(gc.TASK_13, LAT, LON, seed=21)
  full = model(inputs, template, forcings)
  same = casting.Bfloat16Cast(model, enabled=False)(inputs, template, forcings)
  np.testing.assert_array_equal(same["temperature"].values, full["temperature"].values)
  assert model._precision is None


def test_bfloat16_cast_wrapper(setup):
  """casting.Bfloat16Cast -- the reference's wrapper (utils/casting.py:31-65) -- in the reference's
  standard chain position: inputs / forcings rounded to bfloat16, the inner GraphCast in its "bf16"
  arithmetic (GC_PREC_BF16), predictions bfloat16 values in the targets' dtype.  Checked against
  the op-by-op bfloat16 restatement of the reference run (oracle ACTIVATIONS = "bf16"; parity
  unpinned vs XLA's fusion choices): the HIP path must be at least as close to the fp32-grade
  result as that restatement is (within 25 %)."""
  from graphcast_amd import casting
  from graphcast_amd import packing
  from oracle import gnn as ognn
  model, oracle = setup
  inputs, template, forcings = synthetic.make_example(gc.TASK_13, LAT, LON, seed=22)
  full = model(inputs, template, forcings)
  got = casting.Bfloat16Cast(model)(inputs, template, forcings)          # constructible with the default enabled=True
  assert model._precision is None                      # restored
  with ognn.activations("bf16"):
    want = oracle(casting.to_bfloat16_values(inputs), template, casting.to_bfloat16_values(forcings))
  d_hip = d_ref = between = 0.0
  for k in template.keys():
    d_hip = max(d_hip, _rel(got[k].values, full[k].values))
    d_ref = max(d_ref, _rel(want[k].values, full[k].values))
    between = max(between, _rel(got[k].values, want[k].values))
    np.testing.assert_array_equal(got[k].values, packing.bf16_round(got[k].values))
  print(f"Bfloat16Cast: worst per-variable distance to the fp32-grade prediction: HIP {d_hip:.2e}, op-by-op "
        f"restatement of the reference's bf16 run {d_ref:.2e}; HIP vs restatement {between:.2e}")
  assert 1e-4 < d_hip <= 1.25 * d_ref
  assert between <= 2.0 * d_ref


def test_device_rollout_in_the_bfloat16_tier(setup):
  """The reference's standard chain -- rollout(InputsAndResiduals(Bfloat16Cast(GraphCast))) -- against
  rollout_device.DeviceRollout with the model switched to its "bf16" arithmetic: the kernels round
  the fp32 state to bfloat16 on read (GC_ROWS_F32) and emit bfloat16-valued fp32 predictions, which is
  exactly what the wrapper's casts do around the step; the two differ by the fp32 rounding of the
  normalisation algebra, which can move an input across a bfloat16 rounding boundary (hence the
  tolerance at bfloat16 resolution)."""
  from graphcast_amd import casting
  from graphcast_amd import rollout_device
  model, _ = setup
  n_steps = 3
  inputs, template, forcings = synthetic.make_example(gc.TASK_13, LAT, LON, batch=2,
                                                      num_target_steps=n_steps, seed=17)
  mean, std, dstd = synthetic.make_stats(gc.TASK_13)
  dut = normalization.InputsAndResiduals(casting.Bfloat16Cast(model), std, mean, dstd)
  put = lambda ds: synthetic.to_device(ds, "cuda:0")
  with generic_loop():
    want = rollout.chunked_prediction(lambda rng, **kw: dut(**kw), None, inputs, template, forcings,
                                      device_put_fn=put)
    fp32 = rollout.chunked_prediction(
        lambda rng, **kw: normalization.InputsAndResiduals(model, std, mean, dstd)(**kw), None, inputs, template,
        forcings, device_put_fn=put)
  with casting.precision_view(model, "bf16"):
    roll = rollout_device.DeviceRollout(model, std, mean, dstd)
    got = roll.to_dataset(roll.run(inputs, template, forcings), template)
  assert model._precision is None
  worst = tier = 0.0
  for k in template.keys():
    worst = max(worst, _rel(got[k].values, want[k].values))
    tier = max(tier, _rel(want[k].values, fp32[k].values))
  print(f"bf16 tier, {n_steps}-step device rollout vs the wrapper chain: worst per-variable rel diff {worst:.2e} "
        f"(the tier's own distance to the fp32-grade rollout: {tier:.2e})")
  assert worst <= tier              # closer to the wrapper chain than the tier is to fp32
  assert worst < 2e-2


def _equal_datasets(a, b):
  assert sorted(a.keys()) == sorted(b.keys())
  for k in a.keys():
    assert a[k].dims == b[k].dims and a[k].shape == b[k].shape, (k, a[k].dims, b[k].dims)
    np.testing.assert_array_equal(np.asarray(a[k].values), np.asarray(b[k].values), err_msg=k)
  for c in ("time", "lat", "lon", "level"):
    if c in b.coords:
      np.testing.assert_array_equal(np.asarray(a.coords[c].values), np.asarray(b.coords[c].values))


def test_chunked_prediction_runs_the_fused_device_loop_under_a_recognised_stack(setup, caplog):
  """the REFERENCE's entry point -- rollout.chunked_prediction (utils/rollout.py:326-364) on host
  Datasets -- around the demo stack InputsAndResiduals(GraphCast) runs DeviceRollout's fused loop underneath:
  bit-identical to DeviceRollout.run, host Datasets back, same dims / coordinates as the generic loop; a closure
  around the stack is cross-checked on its first chunk, and one that does something ELSE to the predictions falls
  back to being called chunk by chunk."""
  from graphcast_amd import autoregressive, rollout_device
  model, _ = setup
  n_steps = 4
  inputs, template, forcings = synthetic.make_example(gc.TASK_13, LAT, LON, batch=2, num_target_steps=n_steps, seed=23)
  mean, std, dstd = synthetic.make_stats(gc.TASK_13)
  stack = normalization.InputsAndResiduals(model, std, mean, dstd)
  roll = rollout_device.DeviceRollout(model, std, mean, dstd)
  want = roll.to_dataset(roll.run(inputs, template, forcings), template)          # host Dataset of the fused loop
  with generic_loop():
    generic = rollout.chunked_prediction(lambda rng, **kw: stack(**kw), None, inputs, template, forcings)
  # (1) the trusted form: as_predictor_fn keeps the stack visible
  got = rollout.chunked_prediction(rollout.as_predictor_fn(stack), None, inputs, template, forcings)
  _equal_datasets(got, want)
  assert all(isinstance(got[k].data, np.ndarray) for k in got.keys())             # host in -> host out
  for k in got.keys():                                                            # ... and it is the same rollout
    assert got[k].dims == generic[k].dims and _rel(got[k].values, generic[k].values) < 2e-5
  np.testing.assert_array_equal(got.coords["time"].values, generic.coords["time"].values)
  if "datetime" in generic.coords:
    np.testing.assert_array_equal(got.coords["datetime"].values, generic.coords["datetime"].values)
  # (2) a closure around the stack (how the reference's users write predictor_fn) is OPAQUE by default (round 6;
  #     reference utils/rollout.py:78-87, :534-538): it is called for EVERY chunk, whatever it closes over -- a closure
  #     that logs or saves every chunk sees all of them
  calls = []
  def fn(rng, inputs, targets_template, forcings):
    calls.append(1)
    return stack(inputs, targets_template, forcings)
  got_opaque = rollout.chunked_prediction(fn, None, inputs, template, forcings)
  assert len(calls) == n_steps
  for k in got_opaque.keys():
    assert _rel(got_opaque[k].values, want[k].values) < 2e-5
  #     ... rollout.fuse(fn) is the caller's opt-in: found, cross-checked on the FIRST and the LAST chunk, fused between
  calls.clear()
  _equal_datasets(rollout.chunked_prediction(rollout.fuse(fn), None, inputs, template, forcings), want)
  assert len(calls) == 2
  assert rollout.last_fused_stats["verify"] and rollout.last_fused_stats["stats"]["cross_check_s"] > 0
  #     ... and so is GCAST_ROLLOUT_FUSED=closures for every closure of the process (round 5's behaviour)
  calls.clear()
  os.environ["GCAST_ROLLOUT_FUSED"] = "closures"
  try:
    _equal_datasets(rollout.chunked_prediction(fn, None, inputs, template, forcings), want)
  finally:
    del os.environ["GCAST_ROLLOUT_FUSED"]
  assert len(calls) == 2
  # (3) an opted-in closure that ALTERS the predictions is not short-circuited: the first chunk's cross-check catches it,
  #     its own first chunk is what is yielded (it is not called twice for one chunk)
  calls.clear()
  def doubled(rng, inputs, targets_template, forcings):
    calls.append(1)
    out = stack(inputs, targets_template, forcings)
    return xarray.Dataset({k: out[k] * np.float32(2.0) for k in out.keys()}, coords=dict(out._coords))
  want2 = rollout.chunked_prediction(doubled, None, inputs, template, forcings)       # (opaque: the generic loop)
  assert len(calls) == n_steps
  calls.clear()
  got2 = rollout.chunked_prediction(rollout.fuse(doubled), None, inputs, template, forcings)
  assert len(calls) == n_steps
  for k in got2.keys():
    assert got2[k].dims == want2[k].dims and _rel(got2[k].values, want2[k].values) < 2e-5
  assert all(isinstance(got2[k].data, np.ndarray) for k in got2.keys())           # host in -> host out on this path too
  # (3b) ... and one whose extra work only bites LATE (here: from the third chunk on) passes the first cross-check and is
  #     caught by the last one: RuntimeError, not a silently different trajectory
  calls.clear()
  def late(rng, inputs, targets_template, forcings):
    calls.append(1)
    out = stack(inputs, targets_template, forcings)
    if len(calls) < 2:
      return out
    return xarray.Dataset({k: out[k] * np.float32(1.5) for k in out.keys()}, coords=dict(out._coords))
  with pytest.raises(RuntimeError, match="disagrees on the last"):
    rollout.chunked_prediction(rollout.fuse(late), None, inputs, template, forcings)
  # (3c) several steps per chunk over a ONE-step stack are not fused -- the generic path hands the predictor a multi-time
  #     template, which a one-step GraphCast rejects (its forcings then have more channels than the model was built for);
  #     the fused loop used to autoregress silently instead: both forms raise now
  for form in (fn, rollout.as_predictor_fn(stack)):
    with pytest.raises(ValueError, match="expected"):
      rollout.chunked_prediction(form, None, inputs, template, forcings, num_steps_per_chunk=2)
  # (4) the reference's full chain with the autoregressive wrapper outermost, two steps per chunk: time-leading
  #     variables, as that wrapper returns them
  ar = autoregressive.Predictor(stack)
  with generic_loop():
    want4 = rollout.chunked_prediction(lambda rng, **kw: ar(**kw), None, inputs, template, forcings, num_steps_per_chunk=2)
  got4 = rollout.chunked_prediction(rollout.as_predictor_fn(ar), None, inputs, template, forcings, num_steps_per_chunk=2)
  for k in got4.keys():
    assert got4[k].dims == want4[k].dims and got4[k].dims[0] == "time"
    assert _rel(got4[k].values, want4[k].values) < 2e-5
    np.testing.assert_array_equal(np.moveaxis(got4[k].values, 0, 1), want[k].values)       # the same bits, time-leading
  # (5) device-resident inputs: device-backed chunks, nothing crosses PCIe
  put = lambda ds: synthetic.to_device(ds, "cuda:0")
  chunks = list(rollout.chunked_prediction_generator(rollout.as_predictor_fn(stack), None, inputs, template, 1, forcings,
                                                     device_put_fn=put))
  assert len(chunks) == n_steps and all(c["temperature"].data.is_cuda for c in chunks)
  for s, c in enumerate(chunks):
    np.testing.assert_array_equal(c["temperature"].values, want["temperature"].isel(time=slice(s, s + 1)).values)
  # (6) the switch: GCAST_ROLLOUT_FUSED=0 fuses nobody, opted in or not
  with generic_loop():
    calls.clear()
    rollout.chunked_prediction(rollout.fuse(fn), None, inputs, template, forcings)
    assert len(calls) == n_steps
  # (7) a consumer that stops after the first chunk still hears of an out-of-range input state (the range
  #     flag used to be read only when the generator was exhausted)
  from graphcast_amd import _native as nat
  huge = xarray.Dataset({k: inputs[k] * np.float32(3e6) for k in inputs.keys()}, coords=dict(inputs._coords))
  gen = rollout.chunked_prediction_generator(rollout.as_predictor_fn(stack), None, huge, template, 1, forcings)
  with pytest.raises(nat.GcastRangeError):
    next(gen)


def test_fused_rollout_in_the_bfloat16_tier(setup):
  """rollout.chunked_prediction around the reference's demo chain InputsAndResiduals(Bfloat16Cast(GraphCast)): the
  fused loop runs the step in the "bf16" arithmetic -- the same bits as DeviceRollout under casting.precision_view --
  and the model's precision is restored afterwards."""
  from graphcast_amd import casting, rollout_device
  model, _ = setup
  inputs, template, forcings = synthetic.make_example(gc.TASK_13, LAT, LON, num_target_steps=3, seed=29)
  mean, std, dstd = synthetic.make_stats(gc.TASK_13)
  stack = normalization.InputsAndResiduals(casting.Bfloat16Cast(model), std, mean, dstd)
  with casting.precision_view(model, "bf16"):
    roll = rollout_device.DeviceRollout(model, std, mean, dstd)
    want = roll.to_dataset(roll.run(inputs, template, forcings), template)
  got = rollout.chunked_prediction(rollout.fuse(lambda rng, **kw: stack(**kw)), None, inputs, template, forcings)     # (cross-checked)
  assert model._precision is None
  _equal_datasets(got, want)


def test_pmap_devices_runs_one_engine_per_listed_device_from_one_process(setup):
  """VERDICT r5 missing #2 / next #5: the reference's single-process multi-device rollout (utils/rollout.py:196-283,
  :471-487).  `pmap_devices=["cuda:0", "cuda:0"]`: TWO engines (GraphCast.replica: shared parameters and graphs, own
  plan + workspace), two streams, one GPU -- the most one lease allows; members in groups of two, per chunk the steps of
  both engines enqueued before the host waits on either, the group's chunk yielded once with a leading "sample" axis of
  views into ONE pinned [2, N_grid, B, C_out] buffer.  Bitwise equal to the sequential (un-pmapped) branch, through
  `chunked_prediction_generator_multiple_runs` and through `chunked_prediction_generator(pmap_devices=...)` itself."""
  from graphcast_amd import autoregressive
  model, _ = setup
  n_steps, members = 3, 4
  per_member = [synthetic.make_example(gc.TASK_13, LAT, LON, num_target_steps=n_steps, seed=40 + m) for m in range(members)]
  inputs0, template, forcings0 = per_member[0]
  stack_sample = lambda dss: xarray.Dataset(
      {k: ((("sample",) + dss[0][k].dims), np.stack([ds[k].values for ds in dss])) if "time" in dss[0][k].dims
       else (dss[0][k].dims, dss[0][k].values) for k in dss[0].keys()}, coords=dict(dss[0]._coords))
  inputs = stack_sample([p[0] for p in per_member])
  forcings = stack_sample([p[2] for p in per_member])
  mean, std, dstd = synthetic.make_stats(gc.TASK_13)
  stack = normalization.InputsAndResiduals(model, std, mean, dstd)
  fn = rollout.as_predictor_fn(stack)
  seq = list(rollout.chunked_prediction_generator_multiple_runs(fn, [None] * members, inputs, template, forcings,
                                                                num_samples=None, num_steps_per_chunk=1))
  assert len(seq) == members * n_steps
  got = list(rollout.chunked_prediction_generator_multiple_runs(fn, [None] * members, inputs, template, forcings,
                                                                num_samples=None, num_steps_per_chunk=1,
                                                                pmap_devices=["cuda:0", "cuda:0"]))
  assert len(got) == (members // 2) * n_steps
  assert rollout.last_fused_stats.get("pmap_devices") == ["cuda:0", "cuda:0"]
  replicas = model.__dict__.get("_pmap_replicas", {})
  assert list(replicas) == [("cuda:0", 1)] and replicas[("cuda:0", 1)]._engine is not model._engine    # a second engine
  assert replicas[("cuda:0", 1)]._params is model._params
  for gi, c in enumerate(got):
    group, k = divmod(gi, n_steps)
    assert list(c.coords["sample"].values) == [2 * group, 2 * group + 1]
    for j in range(2):
      want = seq[(2 * group + j) * n_steps + k]
      assert int(want.coords["sample"].values) == 2 * group + j
      for name in want.keys():
        assert c[name].dims == ("sample",) + tuple(want[name].dims)
        assert isinstance(c[name].data, np.ndarray)
        np.testing.assert_array_equal(c[name].isel(sample=j).values, np.asarray(want[name].values), err_msg=name)
      np.testing.assert_array_equal(c.coords["time"].values, want.coords["time"].values)
  # the generator itself, with the reference's own keyword set (replica_axis + replicate_fn); inputs WITHOUT the axis
  import functools
  rep = functools.partial(rollout.replicate_dataset, replica_dim="sample", num_replicas=2)
  direct = list(rollout.chunked_prediction_generator(fn, None, inputs0, template, 1, forcings0, pmap_devices=["cuda:0", "cuda:0"],
                                                     replica_axis="sample", replicate_fn=rep))
  assert len(direct) == n_steps
  for k, c in enumerate(direct):
    for name in c.keys():
      np.testing.assert_array_equal(c[name].isel(sample=0).values, np.asarray(seq[k][name].values))
      np.testing.assert_array_equal(c[name].isel(sample=1).values, np.asarray(seq[k][name].values))
  # time-leading stacks (autoregressive.Predictor outermost), two steps per chunk
  ar = rollout.as_predictor_fn(autoregressive.Predictor(stack))
  template2, forc2 = template.isel(time=slice(0, 2)), forcings.isel(time=slice(0, 2))
  seq2 = list(rollout.chunked_prediction_generator_multiple_runs(ar, [None] * 2, inputs.isel(sample=slice(0, 2)), template2,
                                                                 forc2.isel(sample=slice(0, 2)), num_samples=None, num_steps_per_chunk=2))
  got2 = list(rollout.chunked_prediction_generator_multiple_runs(ar, [None] * 2, inputs.isel(sample=slice(0, 2)), template2,
                                                                 forc2.isel(sample=slice(0, 2)), num_samples=None, num_steps_per_chunk=2,
                                                                 pmap_devices=[0, 0]))
  assert len(got2) == 1 and len(seq2) == 2
  for j in range(2):
    for name in seq2[j].keys():
      assert got2[0][name].dims == ("sample",) + tuple(seq2[j][name].dims) and seq2[j][name].dims[0] == "time"
      np.testing.assert_array_equal(got2[0][name].isel(sample=j).values, np.asarray(seq2[j][name].values))
