"""graphcast_amd: GraphCast's encode-process-decode step, MI355X (gfx950) native.

Only the hot path of google-deepmind/graphcast (`weathernext`) is built here,
behind the reference's own Python API:
  graphcast.GraphCast / ModelConfig / TaskConfig / CheckPoint / TASK*,
  rollout.chunked_prediction*, predictor_base.Predictor, typed_graph.*,
  icosahedral_mesh.*, grid_mesh_connectivity.*, model_utils.* (structural part).
Device code: csrc/gcast.hip behind the C-ABI of include/gcast.h.
"""
__version__ = "0.1.0"
