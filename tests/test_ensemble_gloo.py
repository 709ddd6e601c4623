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
