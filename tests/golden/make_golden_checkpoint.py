"""tests/golden/checkpoint_ref.npz: written by the REFERENCE's weathernext/utils/checkpoint.py
(``dump``, numpy only, runs unmodified here) for a small tree shaped like graphcast.CheckPoint;
graphcast_amd.checkpoint.load must read it (tests/test_host_api.py), and the reference's ``load``
must read what graphcast_amd.checkpoint.dump writes (checked here, at generation time).

    python tests/golden/make_golden_checkpoint.py
"""
import dataclasses
import io
import os
import sys
from typing import Any, Optional

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from weathernext.utils import checkpoint as ref_ckpt       # noqa: E402
from graphcast_amd import checkpoint as our_ckpt            # noqa: E402


@dataclasses.dataclass(frozen=True)
class ModelConfig:
  resolution: float
  mesh_size: int
  latent_size: int
  gnn_msg_steps: int
  hidden_layers: int
  radius_query_fraction_edge_length: float
  mesh2grid_edge_normalization_factor: Optional[float] = None


@dataclasses.dataclass(frozen=True)
class TaskConfig:
  input_variables: tuple[str, ...]
  target_variables: tuple[str, ...]
  forcing_variables: tuple[str, ...]
  pressure_levels: tuple[int, ...]
  input_duration: str


@dataclasses.dataclass(frozen=True)
class CheckPoint:
  params: dict[str, Any]
  model_config: ModelConfig
  task_config: TaskConfig
  description: str
  license: str


def main():
  rng = np.random.default_rng(0)
  params = {
      "grid2mesh_gnn/~_networks_builder/encoder_edges_grid2mesh_mlp/~/linear_0":
          {"w": rng.standard_normal((4, 8)).astype(np.float32), "b": np.zeros(8, np.float32)},
      "mesh_gnn/~_networks_builder/processor_nodes_0_mesh_nodes_layer_norm":
          {"scale": np.ones(8, np.float32), "offset": rng.standard_normal(8).astype(np.float32)}}
  ck = CheckPoint(params=params, model_config=ModelConfig(1.0, 5, 512, 16, 1, 0.6),
                  task_config=TaskConfig(("a", "b"), ("a",), ("b",), (50, 100, 1000), "12h"),
                  description="golden", license="none")
  buf = io.BytesIO()
  ref_ckpt.dump(buf, ck)
  with open(os.path.join(HERE, "checkpoint_ref.npz"), "wb") as f:
    f.write(buf.getvalue())
  # the other direction, checked right here: the reference reads what we write
  ours = io.BytesIO()
  our_ckpt.dump(ours, ck)
  ours.seek(0)
  back = ref_ckpt.load(ours, CheckPoint)
  assert back.model_config == ck.model_config and back.task_config == ck.task_config
  for mod in params:
    for leaf in params[mod]:
      np.testing.assert_array_equal(back.params[mod][leaf], params[mod][leaf])
  print("wrote checkpoint_ref.npz; reference.load(graphcast_amd.dump(...)) round trip OK")


if __name__ == "__main__":
  main()
