"""Fingerprints of the mesh2grid index arrays the ORACLE builds (oracle/connectivity.py's
restatement of trimesh's nearest.on_surface rule; reference call site
utils/legacy/grid_mesh_connectivity.py:114-134).

NOT reference-produced: trimesh is absent here, so `in_mesh_triangle_indices` cannot run (SURVEY.md
8c).  The file pins the PRODUCT (graphcast_amd/grid_mesh_connectivity.py, an independent second
restatement) to the oracle at sizes where building both in a test would cost minutes; the tie
candidates (grid points exactly on a mesh edge: 63 at 1 deg, 254 at 0.25 deg) stay unpinned
against trimesh -- `GraphCast(..., mesh2grid_face_indices=...)` is the escape hatch for a host
that has it.

  python tests/golden/make_m2g_hashes.py      # ~1 minute, writes m2g_restated_hashes.json
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import connectivity, mesh     # noqa: E402


def h16(a):
  return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def main():
  out = {}
  for name, res, mesh_size in (("1deg_M5", 1.0, 5), ("0p25deg_M6", 0.25, 6)):
    lat = np.arange(-90, 90 + res / 2, res).astype(np.float32)
    lon = np.arange(0, 360, res).astype(np.float32)
    verts, faces = mesh.mesh_hierarchy(mesh_size)[-1]
    grid_idx, mesh_idx = connectivity.containing_triangle_query(lat, lon, verts, faces)
    out[name] = {
        "m2g_grid_idx": {"dtype": "int64", "shape": [int(len(grid_idx))],
                         "sha256_16": h16(np.asarray(grid_idx, np.int64))},
        "m2g_mesh_idx": {"dtype": "int64", "shape": [int(len(mesh_idx))],
                         "sha256_16": h16(np.asarray(mesh_idx, np.int64))},
        "source": "oracle/connectivity.py (restated trimesh rule; not reference-produced)"}
  with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "m2g_restated_hashes.json"), "w") as f:
    json.dump(out, f, indent=1, sort_keys=True)
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main()
