"""Generates tests/golden/rollout40_1deg_rows.npz: the ORACLE's 40-step autoregressive rollout
(BASELINE.json configs[2]) at 1 deg / 13 levels / M5 with 16 processor steps, through the
reference's demo stack -- rollout.chunked_prediction (utils/rollout.py:326-364) around
normalization.InputsAndResiduals (utils/normalization.py:148-160) around a Predictor whose step is
the fp32 torch-CPU restatement (oracle/torch_cpu.py, pinned to the numpy oracle) -- sampled at 256
fixed grid rows per lead time (the full trajectory is 0.87 GB).

Why a fixture: 40 oracle steps are ~4 TFLOP each, 10+ minutes of host time; measured once live on
the GPU box (profiles/r02_s1_rollout40_parity_1deg_live_oracle.json: rel-RMSE 3.3e-7 at step 1,
6.6e-7 at step 40 over the FULL fields), afterwards the GPU test compares with these rows.
Inputs, statistics and parameters are seeded (synthetic.make_example / make_stats,
params.random_params): the test regenerates them and checks the stored digests.

    python tests/golden/make_golden_rollout40.py          # ~10 minutes on 8 cores

`--config 0p25deg` writes tests/golden/rollout3_0p25deg_rows.npz instead: the SAME stack at the headline
size (0.25 deg / 37 levels / M6, BASELINE.json configs[1] geometry), 3 autoregressive steps (one full
fp32 oracle step is ~2 minutes on the GPU box's 128 host cores; generated there by a round-3 session,
see tests/test_rollout3_fullsize_gpu.py).
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from graphcast_amd import graphcast as gc          # noqa: E402
from graphcast_amd import model_utils              # noqa: E402
from graphcast_amd import normalization            # noqa: E402
from graphcast_amd import params as gparams        # noqa: E402
from graphcast_amd import predictor_base           # noqa: E402
from graphcast_amd import rollout                  # noqa: E402
from graphcast_amd import synthetic                # noqa: E402
from graphcast_amd import xarray_lite as xarray    # noqa: E402
from oracle import graphcast as ogc                # noqa: E402
from oracle import torch_cpu                       # noqa: E402

RES, MESH, GNN_STEPS, N_STEPS, N_ROWS = 1.0, 5, 16, 40, 256
SEEDS = dict(params=7, example=11, rows=3)
LAT = np.arange(-90, 90 + RES / 2, RES)
LON = np.arange(0, 360, RES)


class Config:
  """One fixture = one (grid, task, number of autoregressive steps)."""

  def __init__(self, name, res, mesh, task, n_steps, fixture):
    self.name, self.res, self.mesh, self.task, self.n_steps, self.fixture = name, res, mesh, task, n_steps, fixture
    self.lat = np.arange(-90, 90 + res / 2, res)
    self.lon = np.arange(0, 360, res)
    self.c_in = 2 * (5 + 6 * len(task.pressure_levels)) + 2 * 5 + 2 + 5
    self.c_out = gc.num_output_channels(task)


CONFIGS = {"1deg": Config("1deg", RES, MESH, gc.TASK_13, N_STEPS, "rollout40_1deg_rows.npz"),
           "0p25deg": Config("0p25deg", 0.25, 6, gc.TASK, 3, "rollout3_0p25deg_rows.npz"),
           # BASELINE.json configs[2] itself: 40 autoregressive steps AT the headline size (round 4)
           "0p25deg40": Config("0p25deg40", 0.25, 6, gc.TASK, 40, "rollout40_0p25deg_rows.npz")}


def setup(config="1deg"):
  cfg = CONFIGS[config]
  params = gparams.random_params(cfg.c_in, cfg.c_out, 512, GNN_STEPS, seed=SEEDS["params"])
  inputs, template, forcings = synthetic.make_example(cfg.task, cfg.lat, cfg.lon, num_target_steps=cfg.n_steps,
                                                      seed=SEEDS["example"])
  stats = synthetic.make_stats(cfg.task)
  rows = np.sort(np.random.default_rng(SEEDS["rows"]).choice(len(cfg.lat) * len(cfg.lon), N_ROWS, replace=False))
  return params, inputs, template, forcings, stats, rows


def digest(params, inputs, forcings):
  h = hashlib.sha256()
  for mod in sorted(params):
    for leaf in sorted(params[mod]):
      h.update(np.ascontiguousarray(params[mod][leaf], dtype=np.float32).tobytes())
  for ds in (inputs, forcings):
    for k in sorted(ds.keys()):
      h.update(np.ascontiguousarray(ds[k].values, dtype=np.float32).tobytes())
  return h.hexdigest()


def stacked_rows(ds, template, s, rows):
  """[len(rows), C_out] of lead time s, channels in the stacking order of graphcast.py:680-723."""
  one = xarray.Dataset({k: ds[k].isel(time=slice(s, s + 1)) for k in sorted(template.keys())})
  st = model_utils.lat_lon_to_leading_axes(model_utils.dataset_to_stacked(one))
  data = np.asarray(st.data)
  return data.reshape((-1,) + data.shape[2:])[rows, 0]


class TorchOraclePredictor(predictor_base.Predictor):
  def __init__(self, params, graphs, n_lat=len(LAT), n_lon=len(LON)):
    self.params, self.graphs, self.n_lat, self.n_lon = params, graphs, n_lat, n_lon

  def __call__(self, inputs, targets_template, forcings, **kw):
    x = xarray.concat([model_utils.dataset_to_stacked(inputs),
                       model_utils.dataset_to_stacked(forcings)], dim="channels")
    x = np.asarray(model_utils.lat_lon_to_leading_axes(x).data, np.float32)
    y = torch_cpu.forward(self.params, self.graphs, x.reshape((-1,) + x.shape[2:]), GNN_STEPS)
    y = xarray.DataArray(y.reshape((self.n_lat, self.n_lon) + y.shape[1:]),
                         dims=("lat", "lon", "batch", "channels"))
    return model_utils.stacked_to_dataset(model_utils.restore_leading_axes(y).variable, targets_template)


def main(config="1deg", out_dir=HERE):
  cfg = CONFIGS[config]
  params, inputs, template, forcings, (mean, std, dstd), rows = setup(config)
  t0 = time.perf_counter()
  graphs = ogc.build_graphs(cfg.lat, cfg.lon, cfg.mesh)
  t_graphs = time.perf_counter() - t0
  torch_cpu.set_threads()
  ref = normalization.InputsAndResiduals(TorchOraclePredictor(params, graphs, len(cfg.lat), len(cfg.lon)), std, mean, dstd)
  t0 = time.perf_counter()
  os.makedirs(out_dir, exist_ok=True)
  sha = np.array(digest(params, inputs, forcings))
  one_step = xarray.Dataset({k: template[k].isel(time=slice(0, 1)) for k in sorted(template.keys())})
  traj = []
  # chunk by chunk (= what chunked_prediction concatenates, utils/rollout.py:352-364): only the sampled
  # rows of a lead time are kept (40 full 0.25 deg frames are 38 GB), and the fixture is rewritten after
  # every step so that an interrupted run leaves a shorter, still valid trajectory (the test reads the
  # number of lead times from the file).
  for s, chunk in enumerate(rollout.chunked_prediction_generator(
      lambda rng, **kw: ref(**kw), None, inputs, template, 1, forcings)):
    traj.append(stacked_rows(rollout._to_host(chunk), one_step, 0, rows).astype(np.float32))
    del chunk
    np.savez_compressed(os.path.join(out_dir, cfg.fixture), rows=rows, traj=np.stack(traj), inputs_sha256=sha,
                        config=np.array([cfg.res, cfg.mesh, GNN_STEPS, cfg.n_steps]))
    print(f"step {s + 1}/{cfg.n_steps}: {time.perf_counter() - t0:.0f} s", flush=True)
  dt = time.perf_counter() - t0
  print(f"wrote {cfg.fixture}: traj {np.stack(traj).shape}, oracle graphs {t_graphs:.0f} s, oracle rollout {dt:.0f} s")


if __name__ == "__main__":
  import argparse
  ap = argparse.ArgumentParser()
  ap.add_argument("--config", default="1deg", choices=sorted(CONFIGS))
  ap.add_argument("--out-dir", default=HERE)
  a = ap.parse_args()
  main(a.config, a.out_dir)
