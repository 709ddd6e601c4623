"""Multi-process (world_size 2, gloo, CPU) test of the ensemble path of BASELINE.json
config 4: members are sharded over ranks with no data-path collective; results are
collected with one all_gather; the benchmark clock is the max over ranks."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    return s.getsockname()[1]


@pytest.mark.parametrize("num_members", [2, 5])
def test_two_rank_ensemble(tmp_path, num_members):
  sys.path.insert(0, HERE)
  import _gloo_worker as w
  port = _free_port()
  out = str(tmp_path / "res")
  procs = []
  for rank in range(2):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), out,
                                   str(num_members)], env=env))
  for p in procs:
    assert p.wait(timeout=300) == 0
  res = [np.load(out + f".rank{r}.npz") for r in range(2)]
  assert sorted(res[0]["owned"].tolist() + res[1]["owned"].tolist()) == list(range(num_members))
  want = np.stack([w.step(w.step(w.member_state(m))).numpy() for m in range(num_members)])
  for r in res:
    np.testing.assert_array_equal(r["gathered"], want)     # bit-identical to a single process
  slow = max(float(r["elapsed"]) for r in res)
  assert float(res[0]["slowest"]) == pytest.approx(slow) == float(res[1]["slowest"])


@pytest.mark.parametrize("num_members", [2, 5])
def test_two_rank_ensemble_through_the_product_rollout(tmp_path, num_members):
  """the two gloo ranks drive the PRODUCT's path -- rollout.chunked_prediction_generator_multiple_runs(
  rank=, world_size=) (reference utils/rollout.py:158-307: the members' autoregressive rollouts, `sample` coordinates,
  _get_next_inputs feedback) around a stub Predictor, trajectories collected with ensemble.gather_member_arrays -- and
  every rank ends up with exactly what ONE process rolling out all members computes."""
  sys.path.insert(0, HERE)
  import _gloo_worker as w
  port = _free_port()
  out = str(tmp_path / "res")
  procs = []
  for rank in range(2):
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), out, str(num_members), "rollout"],
                                  env=env))
  for p in procs:
    assert p.wait(timeout=300) == 0
  res = [np.load(out + f".rank{r}.npz") for r in range(2)]
  assert res[0]["owned"].tolist() == list(range(0, num_members, 2)) and res[1]["owned"].tolist() == list(range(1, num_members, 2))
  single = w.member_trajectories(num_members, 0, 1)
  want = np.stack([single[m].numpy() for m in range(num_members)])
  assert want.shape[1] == w.R_STEPS                       # [members, lead times, lat, lon, channels]
  assert not np.array_equal(want[0], want[1])             # the members really differ
  for r in res:
    np.testing.assert_array_equal(r["gathered"], want)    # bit-identical to a single process
