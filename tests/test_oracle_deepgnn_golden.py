"""(CPU) oracle/deep_gnn.py against tests/golden/gnn_deepgnn512.npz = the reference's own utils/deep_gnn.py +
utils/dense.py + utils/typed_graph_net.py executed on the numpy stand-ins (tests/golden/make_golden_deepgnn.py):
concat form, pre-gather-matmul form, and two processor repetitions.  Pins the WN2 processor's wiring (SURVEY.md
section 8 f4, second half) for the oracle the GPU test then uses at full precision."""
import os

import numpy as np
import pytest

from oracle import deep_gnn as odg
from oracle import params as oparams
from tests.golden import deepgnn_case as G

CASES = [("concat", False, 1), ("pregather", True, 1), ("pregather_x2", True, 2)]


@pytest.fixture(scope="module")
def gold(golden_dir):
  return np.load(os.path.join(golden_dir, "gnn_deepgnn512.npz"))


@pytest.mark.parametrize("tag,pre,reps", CASES)
def test_oracle_matches_reference_execution(gold, tag, pre, reps):
  x = G.inputs()
  params = oparams.init_deep_gnn_params(G.LATENT, G.STEPS, {"mesh_nodes": 1}, ["mesh"], pre_gather_matmul=pre, seed=G.SEED)
  assert oparams.digest({k: {l: np.asarray(v, np.float64) for l, v in m.items()} for k, m in params.items()}) \
      == str(gold[f"{tag}_params_sha256"])
  nodes, edges = odg.forward(
      params, {"mesh_nodes": x["h"]},
      {"mesh": dict(senders_set="mesh_nodes", receivers_set="mesh_nodes", senders=x["senders"],
                    receivers=x["receivers"], features=x["e"])},
      num_message_passing_steps=G.STEPS, num_processor_repetitions=reps, pre_gather_matmul=pre)
  rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
  assert rel(nodes["mesh_nodes"][gold["node_rows"]], gold[f"{tag}_nodes_f64"]) < 1e-12
  assert rel(edges["mesh"][gold["edge_rows"]], gold[f"{tag}_edges_f64"]) < 1e-12
  assert rel(nodes["mesh_nodes"], gold[f"{tag}_nodes"].astype(np.float64)) < 2e-7       # (float32 storage)


def test_pre_gather_form_is_the_concat_form_with_a_split_matrix():
  """dense.summed_args + drop_first_matmul (deep_gnn.py:224-262): same function as the concat MLP whose first
  matrix is [W_edge; W_sender; W_receiver]."""
  x = G.inputs()
  pre = oparams.init_deep_gnn_params(G.LATENT, 1, {"mesh_nodes": 1}, ["mesh"], pre_gather_matmul=True, seed=3)
  cat = {k: dict(v) for k, v in pre.items() if "_edge_" not in k and "_sender_" not in k and "_receiver_" not in k}
  cat["DeepGNN/processor_edges_0_mesh/mlp/linear_0"] = {
      "w": np.concatenate([pre[f"DeepGNN/processor_edges_0_{p}_mesh"]["w"] for p in ("edge", "sender", "receiver")]),
      "b": pre["DeepGNN/processor_edges_0_mesh/mlp/linear_0"]["b"]}
  kw = dict(num_message_passing_steps=1)
  g = {"mesh": dict(senders_set="mesh_nodes", receivers_set="mesh_nodes", senders=x["senders"],
                    receivers=x["receivers"], features=x["e"][:, :1])}
  a, _ = odg.forward(pre, {"mesh_nodes": x["h"][:, :1]}, g, pre_gather_matmul=True, **kw)
  b, _ = odg.forward(cat, {"mesh_nodes": x["h"][:, :1]}, g, pre_gather_matmul=False, **kw)
  assert np.linalg.norm(a["mesh_nodes"] - b["mesh_nodes"]) / np.linalg.norm(b["mesh_nodes"]) < 1e-12
