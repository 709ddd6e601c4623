"""Generates tests/golden/data_utils_ref.npz by executing the REFERENCE's own

    weathernext/utils/data_utils.py        (derived forcings, inputs / targets / forcings split)
    weathernext/utils/solar_radiation.py   (top-of-atmosphere incident solar radiation)

UNMODIFIED, in this container (needs /root/reference; outputs are committed).  Stand-ins:
``xarray`` is graphcast_amd.xarray_lite; ``jax.numpy`` is numpy behind a thin layer that reproduces
what matters of JAX's default configuration here -- EVERYTHING IS float32 (x64 disabled): arrays
are created as float32 and a float64 operand (the numpy offsets added to the J2000 day count) is
demoted, not the other way round.  That quantisation (2^-10 day on today's J2000 day counts) is
part of the reference's observable behaviour.  Run:  python tests/golden/make_golden_data_utils.py
"""
import os
import sys
import types

import numpy as np
import pandas as pd

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(HERE, "ref_shims"))
sys.path.insert(0, REF)

from graphcast_amd import xarray_lite                        # noqa: E402

sys.modules["xarray"] = xarray_lite


class F32(np.ndarray):
  """ndarray that stays float32: float64 operands of a ufunc are demoted (JAX with x64 off)."""
  __array_priority__ = 100

  def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
    cast = []
    for x in inputs:
      if isinstance(x, np.ndarray):
        x = np.asarray(x)
        if x.dtype == np.float64:
          x = x.astype(np.float32)
      elif isinstance(x, (np.floating, float)):
        x = np.float32(x)
      cast.append(x)
    out = getattr(ufunc, method)(*cast, **kwargs)
    return out.view(F32) if isinstance(out, np.ndarray) and out.dtype == np.float32 else out


def f32(x):
  return np.asarray(x, dtype=np.float32).view(F32)


jnp = types.ModuleType("jax.numpy")
jnp.pi = np.pi
jnp.array = f32
jnp.asarray = f32
jnp.ones_like = lambda x: f32(np.ones_like(np.asarray(x)))
jnp.radians = lambda x: f32(np.radians(f32(x)))
for name in ("sin", "cos", "sqrt"):
  setattr(jnp, name, (lambda fn: lambda x: f32(fn(np.asarray(f32(x)))))(getattr(np, name)))
jnp.maximum = lambda a, b: f32(np.maximum(np.asarray(f32(a)), np.asarray(f32(b))))
jnp.expand_dims = lambda x, axis: f32(np.expand_dims(np.asarray(f32(x)), axis))
jnp.stack = lambda xs, axis=0: f32(np.stack([np.asarray(f32(x)) for x in xs], axis=axis))
jnp.dot = lambda a, b: f32(np.dot(np.asarray(f32(a)), np.asarray(f32(b))))


def _trapezoid(y, x=None, dx=1.0, axis=-1):      # jax/_src/scipy/integrate.py: trapezoid, x is None
  assert x is None and axis == -1
  y = np.asarray(f32(y))
  return f32(np.float32(0.5) * np.sum(np.float32(dx) * (y[..., 1:] + y[..., :-1]), axis=-1, dtype=np.float32))


jax = types.ModuleType("jax")
jax.numpy = jnp
jax.jit = lambda f, **kw: f
jax.scipy = types.SimpleNamespace(integrate=types.SimpleNamespace(trapezoid=_trapezoid))
sys.modules["jax"] = jax
sys.modules["jax.numpy"] = jnp

from weathernext.utils import data_utils as ref_du            # noqa: E402
from weathernext.utils import solar_radiation as ref_sr       # noqa: E402


def main():
  out = {}
  # ---- solar radiation: a coarse global grid, timestamps around the year and the clock ----------
  lat = np.linspace(-90.0, 90.0, 13)
  lon = np.linspace(0.0, 360.0, 16, endpoint=False)
  stamps = pd.DatetimeIndex(["2019-12-31T18:00", "2020-03-20T06:00", "2020-06-21T12:00", "2022-09-23T00:00",
                             "1988-11-07T02:45:34"])
  tisr = ref_sr.get_toa_incident_solar_radiation(stamps, lat, lon, use_jit=True)
  assert np.asarray(tisr).dtype == np.float32
  out.update(sr_lat=lat, sr_lon=lon, sr_stamps=stamps.values.astype("datetime64[s]").astype(np.int64),
             sr_tisr=np.asarray(tisr))
  short = ref_sr.get_toa_incident_solar_radiation(stamps[:2], lat, lon, integration_period="6h", num_integration_bins=12)
  out["sr_tisr_6h_12bins"] = np.asarray(short)
  out["sr_tsi"] = np.asarray(ref_sr.get_tsi(stamps, ref_sr.era5_tsi_data()))

  # ---- a synthetic ERA5-like sample through extract_inputs_targets_forcings ---------------------
  rng = np.random.default_rng(0)
  nt, levels = 6, np.array([50, 500, 850, 1000])
  glat, glon = np.linspace(-90, 90, 5), np.linspace(0, 360, 8, endpoint=False)
  t0 = np.datetime64("2021-12-31T12:00:00")
  time = (np.arange(nt) * np.timedelta64(6, "h")).astype("timedelta64[ns]")
  datetime = (t0 + time).astype("datetime64[ns]")
  ds = xarray_lite.Dataset(
      data_vars={
          "2m_temperature": (("batch", "time", "lat", "lon"), rng.standard_normal((1, nt, 5, 8)).astype(np.float32)),
          "temperature": (("batch", "time", "level", "lat", "lon"), rng.standard_normal((1, nt, 4, 5, 8)).astype(np.float32)),
          "geopotential_at_surface": (("lat", "lon"), rng.standard_normal((5, 8)).astype(np.float32)),
      },
      coords={"lat": glat, "lon": glon, "level": levels, "time": time,
              "datetime": (("batch", "time"), datetime[None])})
  kw = dict(input_variables=("2m_temperature", "temperature", "geopotential_at_surface", "toa_incident_solar_radiation",
                             "year_progress_sin", "day_progress_cos"),
            target_variables=("2m_temperature", "temperature"),
            forcing_variables=("toa_incident_solar_radiation", "year_progress_sin", "year_progress_cos",
                               "day_progress_sin", "day_progress_cos"),
            pressure_levels=(50, 850), input_duration="12h", target_lead_times=slice("6h", "18h"))
  inputs, targets, forcings = ref_du.extract_inputs_targets_forcings(ds.copy(), **kw)
  for tag, d in (("in", inputs), ("tg", targets), ("fc", forcings)):
    out[f"du_{tag}_time"] = np.asarray(d.coords["time"].data).astype("timedelta64[ns]").astype(np.int64)
    out[f"du_{tag}_level"] = np.asarray(d.coords["level"].data) if "level" in d.coords else np.zeros(0)
    for name in d.data_vars:
      v = d[name]
      out[f"du_{tag}/{name}"] = np.asarray(v.data)
      out[f"du_{tag}_dims/{name}"] = np.array("|".join(v.dims))
  # single lead times in a list, unordered
  _, targets2, _ = ref_du.extract_inputs_targets_forcings(ds.copy(), **dict(kw, target_lead_times=("18h", "6h")))
  out["du_tg2_time"] = np.asarray(targets2.coords["time"].data).astype("timedelta64[ns]").astype(np.int64)
  out["du_tg2/2m_temperature"] = np.asarray(targets2["2m_temperature"].data)
  # the raw sample, so that the test feeds exactly the same arrays
  for k in ("2m_temperature", "temperature", "geopotential_at_surface"):
    out[f"raw/{k}"] = np.asarray(ds[k].data)
  out.update(raw_lat=glat, raw_lon=glon, raw_level=levels, raw_time=time.astype(np.int64),
             raw_datetime=datetime.astype(np.int64))
  # progress features on their own (float64 in, float32 out)
  secs = np.array([0, 123, 86399, 1640995200, 1640995200 + 6 * 3600], dtype=np.int64)
  out["prog_secs"] = secs
  out["prog_year"] = ref_du.get_year_progress(secs)
  out["prog_day"] = ref_du.get_day_progress(secs, glon)
  path = os.path.join(HERE, "data_utils_ref.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, {k: np.shape(v) for k, v in out.items() if k.startswith(("sr_", "du_in/"))})


if __name__ == "__main__":
  main()
