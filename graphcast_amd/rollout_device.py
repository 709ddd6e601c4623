"""HBM-resident autoregressive rollout: normalisation + residual + rolling window fused
into one state-advance kernel per step.

``rollout.chunked_prediction`` around ``normalization.InputsAndResiduals(GraphCast)``
(the reference's demo stack: ``rollout.py:367-565`` -> ``normalization.py:148-160`` ->
``graphcast.py:298-329``) does, per 6-h step, a Dataset round trip: normalise ~18 variables,
stack them into channels, run the step, unstack 11 variables, un-normalise, add the last
input frame, roll the 2-frame window, re-normalise ... all of it affine *per channel*.

``DeviceRollout`` keeps the NORMALISED stacked state ``x [N_grid * B, C_in]`` in HBM and
replaces that round trip by ``gc_advance_state`` (include/gcast.h): per input channel one
pick from {old state, step output, forcings} with two coefficients, per output channel one
affine de-normalisation.  The channel tables are derived by pushing *index-valued* datasets
through the very same stacking code (``model_utils.dataset_to_stacked``) the Dataset path
uses, so the layout cannot drift from ``graphcast.py:680-723``.

The trajectory (de-normalised predictions, ``[T, N_grid, B, C_out]`` fp32: 0.94 GB per
0.25 deg step, 38 GB for 10 days) stays in HBM -- 288 GB is plenty -- and is turned into a
Dataset only on request (``to_dataset``).

In normalised space the residual connection is
    x'_last = x_last + y * (diffs_stddev / stddev)
which differs from the reference's ((y * dstd + x_raw) - mean) / std only by fp32 rounding
(tests compare the two paths at 1e-5).
"""
import ctypes
from typing import Optional

import numpy as np
import torch

from graphcast_amd import _native as nat
from graphcast_amd import model_utils
from graphcast_amd import normalization
from graphcast_amd import xarray_lite as xarray


def _coded_like(ds: xarray.Dataset, base: int, time_scale: int = 1000) -> xarray.Dataset:
  """Dataset of the same structure whose values encode (variable id, time idx, level idx)."""
  out = {}
  for vid, name in enumerate(sorted(ds.keys())):
    v = ds[name].variable
    code = np.full(v.shape, float(base + vid * 1_000_000), dtype=np.float64)
    for d, n in zip(v.dims, v.shape):
      shape = [1] * len(v.dims)
      shape[v.dims.index(d)] = n
      if d == "time":
        code = code + time_scale * np.arange(n).reshape(shape)
      elif d == "level":
        code = code + np.arange(n).reshape(shape)
    out[name] = (v.dims, code)
  return xarray.Dataset(out, coords={k: v for k, v in ds._coords.items()
                                     if k in ("lat", "lon", "level", "time", "batch")})


def _channel_codes(ds: xarray.Dataset):
  """[(name, time idx or -1, level idx or -1)] per stacked channel, in the stacking order."""
  names = sorted(ds.keys())
  # (ONE grid point is enough to read the channel order off: at the full 0.25 deg grid the index-valued float64
  #  twins of the inputs are 5 GB and building + stacking them cost 2.7 s per rollout, profiles/r05_s3_*)
  ds = ds.isel({d: slice(0, 1) for d in ("lat", "lon") if d in ds.sizes})
  stacked = model_utils.dataset_to_stacked(_coded_like(ds, 0))
  codes = np.asarray(stacked.values).reshape(-1, stacked.shape[-1])[0]
  out = []
  for c in codes:
    c = int(round(c))
    vid, rest = divmod(c, 1_000_000)
    t, l = divmod(rest, 1000)
    v = ds[names[vid]]
    out.append((names[vid], t if "time" in v.dims else -1, l if "level" in v.dims else -1))
  return out


def _stat(stats: Optional[xarray.Dataset], name: str, level, default: float) -> float:
  """Statistic of `name` at pressure level LABEL `level` (None: a level-less variable).  Looked
  up by label like xarray's alignment in normalization.py:29-48: published statistics carry 37
  levels whatever the task uses, in whatever order the file has them."""
  if stats is None or name not in stats.keys():
    return default
  da = stats[name]
  v = np.asarray(da.values, dtype=np.float64)
  if not v.ndim:
    return float(v)
  if level is None:
    raise ValueError(f"statistic {name!r} has a level axis but the variable has none")
  labels = np.asarray(stats.coords["level"].values)
  hit = np.nonzero(labels == level)[0]
  if len(hit) != 1:
    raise KeyError(f"level {level!r} not (uniquely) among the levels of statistic {name!r}")
  return float(v.reshape(-1)[hit[0]])


def build_tables(inputs, targets_template, forcings, stddev_by_level, mean_by_level,
                 diffs_stddev_by_level) -> dict:
  """Per-channel pick tables of ``gc_advance_desc`` (numpy), see include/gcast.h."""
  one_f = forcings.isel(time=slice(0, 1))
  one_t = targets_template.isel(time=slice(0, 1))
  in_ch = _channel_codes(inputs)              # state channels from `inputs` ...
  f_ch = _channel_codes(one_f)                # ... followed by the target-time forcings
  t_ch = _channel_codes(one_t)
  n_in_frames = inputs.sizes["time"]
  pos_in = {c: i for i, c in enumerate(in_ch)}
  pos_f = {(n, l): i for i, (n, _, l) in enumerate(f_ch)}
  pos_t = {(n, l): i for i, (n, _, l) in enumerate(t_ch)}
  c_in, n_forc, c_out = len(in_ch) + len(f_ch), len(f_ch), len(t_ch)
  level_labels = (np.asarray(inputs.coords["level"].values) if "level" in inputs.coords
                  else np.asarray(targets_template.coords["level"].values) if "level" in targets_template.coords
                  else None)
  lab = lambda l: None if l < 0 else level_labels[l]      # level index of a channel -> its label

  src_x = np.full(c_in, -1, np.int32)
  src_y = np.full(c_in, -1, np.int32)
  src_f = np.full(c_in, -1, np.int32)
  ax = np.zeros(c_in, np.float32)
  ay = np.zeros(c_in, np.float32)
  for c, (name, t, l) in enumerate(in_ch):
    if t < 0:                                   # static input: carried over
      src_x[c], ax[c] = c, 1.0
    elif t < n_in_frames - 1:                   # window shift: frame t <- frame t+1
      src_x[c], ax[c] = pos_in[(name, t + 1, l)], 1.0
    elif (name, l) in pos_t:                    # predicted: last frame + residual
      std = _stat(stddev_by_level, name, lab(l), 1.0)
      dstd = _stat(diffs_stddev_by_level, name, lab(l), 1.0)
      src_x[c], ax[c] = c, 1.0
      src_y[c], ay[c] = pos_t[(name, l)], dstd / std
    elif (name, l) in pos_f:                    # forced: value at the time just predicted
      src_f[c] = pos_f[(name, l)]
    else:
      raise ValueError("Found an input with a time index that is not predicted or forced.")
  for j in range(n_forc):                       # target-time forcings of the NEXT step
    src_f[len(in_ch) + j] = n_forc + j

  p_src_x = np.full(c_out, -1, np.int32)
  p_ax = np.zeros(c_out, np.float32)
  p_ay = np.zeros(c_out, np.float32)
  p_b = np.zeros(c_out, np.float32)
  for k, (name, _, l) in enumerate(t_ch):
    std, mean = _stat(stddev_by_level, name, lab(l), 1.0), _stat(mean_by_level, name, lab(l), 0.0)
    if (name, n_in_frames - 1, l) in pos_in:    # residual target: y*dstd + (x_norm*std + mean)
      p_src_x[k] = pos_in[(name, n_in_frames - 1, l)]
      p_ax[k], p_ay[k], p_b[k] = std, _stat(diffs_stddev_by_level, name, lab(l), 1.0), mean
    else:                                       # direct target: y*std + mean
      p_ay[k], p_b[k] = std, mean
  return dict(c_in=c_in, c_out=c_out, n_forc=n_forc, src_x=src_x, src_y=src_y, src_f=src_f,
              ax=ax, ay=ay, p_src_x=p_src_x, p_ax=p_ax, p_ay=p_ay, p_b=p_b)


class DeviceRollout:
  """Rolls a ``GraphCast`` forward with the state, forcings and trajectory resident in HBM."""

  def __init__(self, model, stddev_by_level: xarray.Dataset, mean_by_level: xarray.Dataset,
               diffs_stddev_by_level: xarray.Dataset):
    self._model = model
    self._std, self._mean, self._dstd = (xarray.from_xarray(stddev_by_level), xarray.from_xarray(mean_by_level),
                                         xarray.from_xarray(diffs_stddev_by_level))
    self._lib = nat.lib()
    self._tables = None

  def _build_tables(self, inputs, targets_template, forcings):
    dev = torch.device(self._model._device)
    tb = build_tables(inputs, targets_template, forcings, self._std, self._mean, self._dstd)
    self._tables = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v)
                    for k, v in tb.items()}

  # ---------------------------------------------------------------- run
  def _normalised_forcing_rows(self, forcings, t):
    """[N_grid * B, n_forc] normalised forcings of target time index t (device tensor)."""
    f = normalization.normalize(forcings.isel(time=slice(t, t + 1)), self._std, self._mean)
    stacked = model_utils.dataset_to_stacked(f, sizes=self._sizes)
    data = model_utils.lat_lon_to_leading_axes(stacked).data
    data = data if torch.is_tensor(data) else torch.from_numpy(np.ascontiguousarray(data))
    return data.to(device=self._model._device, dtype=torch.float32).reshape(-1, data.shape[-1]).contiguous()

  def _prepare(self, inputs, targets_template, forcings):
    """Everything before the first step: validation, the channel tables, the initial normalised state (through the
    Dataset path once) and every target time's normalised forcing rows, uploaded once."""
    model = self._model
    dev = model._device
    inputs, targets_template, forcings = (xarray.from_xarray(inputs), xarray.from_xarray(targets_template),
                                          xarray.from_xarray(forcings))
    n_steps = targets_template.sizes["time"]
    if forcings.sizes.get("time") != n_steps:
      raise ValueError("forcings must cover every target time")
    if len(np.unique(np.diff(np.asarray(targets_template.coords["time"].values)))) > 1:
      raise ValueError("The targets time coordinates must be evenly spaced")
    model._maybe_init(np.asarray(inputs.coords["lat"].values), np.asarray(inputs.coords["lon"].values))
    # Host Datasets: every variable is uploaded as the caller holds it, ONCE per rollout (inputs 2 GB, the forcings of
    # all lead times 20 MB each at 0.25 deg), so that normalisation and stacking below run on the device -- on the host
    # they cost 4-5 s per 40-step rollout, single-threaded numpy (profiles/r05_s1_*: 246 ms per step through
    # rollout.chunked_prediction with them, the loop itself 54)
    if str(dev).startswith("cuda"):
      if xarray.is_host(inputs):
        inputs = xarray.to_device(inputs, dev)
      if xarray.is_host(forcings):
        forcings = xarray.to_device(forcings, dev)
    self._sizes = dict(inputs.sizes)
    self._build_tables(inputs, targets_template, forcings)
    tb = self._tables
    norm_inputs = normalization.normalize(inputs, self._std, self._mean)
    norm_f0 = normalization.normalize(forcings.isel(time=slice(0, 1)), self._std, self._mean)
    feats = model._inputs_to_grid_node_features(norm_inputs, norm_f0)
    feats = feats if torch.is_tensor(feats) else torch.from_numpy(np.ascontiguousarray(feats, dtype=np.float32))
    x = feats.to(device=dev, dtype=torch.float32).contiguous()
    n_grid, batch, c_in = x.shape
    if c_in != tb["c_in"]:
      raise ValueError(f"state has {c_in} channels, tables describe {tb['c_in']}")
    # forcings of every target time, normalised, uploaded once (20 MB per 0.25 deg step)
    f_rows = [self._normalised_forcing_rows(forcings, t) for t in range(n_steps)]
    return dict(x=x, x_next=torch.empty_like(x), n_steps=n_steps, f_rows=f_rows,
                y=torch.empty((n_grid, batch, tb["c_out"]), dtype=torch.float32, device=dev),
                out_shape=(n_grid, batch, tb["c_out"]))

  def _loop(self, st, out=None):
    """The step loop on a prepared state: yields ``(s, prediction)``; ``prediction`` = ``out[s]`` (``out[0]`` when the
    buffer holds one slot) or a fresh ``[N_grid, B, C_out]`` tensor per step.  Enqueues only; never synchronises."""
    model, tb = self._model, self._tables
    dev = model._device
    x, x_next, y, f_rows, n_steps = st["x"], st["x_next"], st["y"], st["f_rows"], st["n_steps"]
    n_grid, batch, c_out = st["out_shape"]
    stream = lambda: ctypes.c_void_p(torch.cuda.current_stream(torch.device(dev)).cuda_stream)
    for s in range(n_steps):
      model.forward_grid_node_features(x, y)
      pred = (out[s if out.shape[0] > 1 else 0] if out is not None
              else torch.empty(st["out_shape"], dtype=torch.float32, device=dev))
      d = nat.AdvanceDesc()
      d.n_rows, d.c_in, d.c_out, d.n_forc = n_grid * batch, x.shape[-1], c_out, tb["n_forc"]
      d.x, d.y, d.x_next = x.data_ptr(), y.data_ptr(), x_next.data_ptr()
      d.f_cur = f_rows[s].data_ptr()
      d.f_next = f_rows[min(s + 1, n_steps - 1)].data_ptr()
      for k in ("src_x", "ax", "src_y", "ay", "src_f", "p_src_x", "p_ax", "p_ay", "p_b"):
        setattr(d, k, tb[k].data_ptr())
      d.pred = pred.data_ptr()
      nat.check(self._lib.gc_advance_state(ctypes.byref(d), stream()), "gc_advance_state")
      x, x_next = x_next, x
      self._last_advance = (d, x, x_next, y, pred)    # (keeps the buffers of the descriptor alive)
      self.final_state = x
      yield s, pred

  def steps(self, inputs: xarray.Dataset, targets_template: xarray.Dataset, forcings: xarray.Dataset):
    """The rollout as a generator: ``(s, prediction [N_grid, B, C_out])`` per lead time, a fresh de-normalised device
    tensor each (the consumer may keep it or drop it -- nothing but the 2-frame state is retained here).  This is the
    loop ``run`` executes -- the same kernels in the same order, bit for bit -- and what
    ``rollout.chunked_prediction_generator`` runs underneath a recognised predictor stack."""
    yield from self._loop(self._prepare(inputs, targets_template, forcings))

  def run(self, inputs: xarray.Dataset, targets_template: xarray.Dataset,
          forcings: xarray.Dataset, keep_trajectory: bool = True) -> torch.Tensor:
    """Rolls out ``targets_template.sizes['time']`` steps.  Returns the trajectory
    ``[T, N_grid, B, C_out]`` (de-normalised, device) -- or only the last step ``[1, ...]``
    when ``keep_trajectory`` is False."""
    st = self._prepare(inputs, targets_template, forcings)
    traj = torch.empty((st["n_steps"] if keep_trajectory else 1,) + st["out_shape"], dtype=torch.float32,
                       device=self._model._device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in self._loop(st, out=traj):
      pass
    ev1.record()
    self._loop_events = (ev0, ev1)
    self._model._engine.check_range()        # (once per run: an out-of-range input state raises instead of a wrong trajectory)
    return traj

  def advance_ms(self, iters: int = 20) -> float:
    """Device time of one gc_advance_state launch (the last step's descriptor replayed), milliseconds.

    The replay REWRITES the last step's outputs (x_next and the last trajectory slot) with the same values;
    call it right after run(), before those buffers are handed on."""
    if getattr(self, "_last_advance", None) is None:
      raise RuntimeError("DeviceRollout.advance_ms: no step has run yet (call run() with at least one step first)")
    d = self._last_advance[0]
    s = ctypes.c_void_p(torch.cuda.current_stream(torch.device(self._model._device)).cuda_stream)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nat.check(self._lib.gc_advance_state(ctypes.byref(d), s), "gc_advance_state")
    ev0.record()
    for _ in range(iters):
      nat.check(self._lib.gc_advance_state(ctypes.byref(d), s), "gc_advance_state")
    ev1.record()
    ev1.synchronize()
    return ev0.elapsed_time(ev1) / iters

  def last_loop_ms(self) -> float:
    """Device time of the last run()'s step loop (steps + state advances), milliseconds."""
    ev0, ev1 = self._loop_events
    ev1.synchronize()
    return ev0.elapsed_time(ev1)

  def to_dataset(self, traj: torch.Tensor, targets_template: xarray.Dataset, host: bool = True):
    """Trajectory tensor -> Dataset shaped like ``targets_template`` (reference
    ``_grid_node_outputs_to_prediction`` per step, concatenated in time)."""
    model = self._model
    steps = []
    for s in range(traj.shape[0]):
      data = traj[s].cpu().numpy() if host else traj[s]
      steps.append(model._grid_node_outputs_to_prediction(
          data, targets_template.isel(time=slice(s, s + 1))))
    out = xarray.concat(steps, dim="time")
    return out.assign_coords({k: v.variable for k, v in targets_template.coords.items()
                              if "time" in v.dims})
