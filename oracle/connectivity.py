"""Oracle: grid<->mesh connectivity (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``/root/reference/weathernext/utils/legacy/grid_mesh_connectivity.py``:
  * ``grid_xyz``                  <- ``_grid_lat_lon_to_coordinates`` (:22-37)
  * ``radius_query``              <- ``radius_query_indices`` (:40-86)
  * ``containing_triangle_query`` <- ``in_mesh_triangle_indices`` (:89-134)

``in_mesh_triangle_indices`` delegates to ``trimesh.Trimesh.nearest.on_surface``
(trimesh is an un-vendored, un-pinned dependency, reference ``setup.py:48``,
absent here).  Its published algorithm (trimesh/proximity.py ``closest_point``)
is restated below: for every query point take the candidate triangles, compute
the closest point on each (Ericson, Real-Time Collision Detection 5.1.5), keep
the minimum Euclidean distance; among candidates whose distance ties with the
minimum within ``tol.merge`` = 1e-8 keep the one whose face normal is best
aligned with (query - closest point) (largest ``|n . d|``); remaining exact
ties go to the lowest face index.  Parity for exact ties (points ON a mesh
edge) is unpinned; see oracle/__init__.py.
"""
import numpy as np
import scipy.spatial

_TIE_TOL = 1e-8   # trimesh.constants.tol.merge


def grid_xyz(lat_deg, lon_deg):
  """[n_lat, n_lon, 3] unit-sphere coordinates (dtype follows the inputs)."""
  phi, theta = np.meshgrid(np.deg2rad(lon_deg), np.deg2rad(90 - lat_deg))
  return np.stack([np.cos(phi) * np.sin(theta),
                   np.sin(phi) * np.sin(theta),
                   np.cos(theta)], axis=-1)


def radius_query(lat_deg, lon_deg, mesh_vertices, radius):
  """(grid_idx, mesh_idx) int64, grid-major, neighbours ascending per grid point."""
  pts = grid_xyz(lat_deg, lon_deg).reshape([-1, 3])
  tree = scipy.spatial.cKDTree(mesh_vertices)
  hits = tree.query_ball_point(x=pts, r=radius)
  g = [np.repeat(i, len(h)) for i, h in enumerate(hits)]
  return (np.concatenate(g, axis=0).astype(int),
          np.concatenate(list(hits), axis=0).astype(int))


def closest_point_on_triangles(p, a, b, c):
  """Closest points on triangles (a,b,c)[k] to points p[k]; all [K,3] float64.

  Region classification of Ericson 5.1.5 (the routine trimesh uses).
  """
  ab, ac, ap = b - a, c - a, p - a
  d1 = np.einsum("ij,ij->i", ab, ap)
  d2 = np.einsum("ij,ij->i", ac, ap)
  bp = p - b
  d3 = np.einsum("ij,ij->i", ab, bp)
  d4 = np.einsum("ij,ij->i", ac, bp)
  cp = p - c
  d5 = np.einsum("ij,ij->i", ab, cp)
  d6 = np.einsum("ij,ij->i", ac, cp)
  vc = d1 * d4 - d3 * d2
  vb = d5 * d2 - d1 * d6
  va = d3 * d6 - d5 * d4
  out = np.empty_like(p)
  done = np.zeros(len(p), dtype=bool)

  def assign(mask, value):
    m = mask & ~done
    out[m] = value[m]
    done[m] = True

  with np.errstate(divide="ignore", invalid="ignore"):
    assign((d1 <= 0) & (d2 <= 0), a)
    assign((d3 >= 0) & (d4 <= d3), b)
    assign((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + (d1 / (d1 - d3))[:, None] * ab)
    assign((d6 >= 0) & (d5 <= d6), c)
    assign((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + (d2 / (d2 - d6))[:, None] * ac)
    w = (d4 - d3) / ((d4 - d3) + (d5 - d6))
    assign((va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0), b + w[:, None] * (c - b))
    denom = 1.0 / (va + vb + vc)
    inside = a + ab * (vb * denom)[:, None] + ac * (vc * denom)[:, None]
  assign(np.ones(len(p), dtype=bool), inside)
  return out


def containing_triangle_faces(points, vertices, faces, n_candidate_vertices=2):
  """Face index per query point following the restated trimesh rule."""
  vertices = np.asarray(vertices, dtype=np.float64)
  points = np.asarray(points, dtype=np.float64)
  faces = np.asarray(faces)
  # Candidate faces = every face incident to the k nearest mesh vertices.
  nv = len(vertices)
  order = np.argsort(faces.reshape(-1), kind="stable")
  owner = (np.arange(faces.size) // 3)[order]
  starts = np.searchsorted(faces.reshape(-1)[order], np.arange(nv + 1))
  max_deg = int(np.diff(starts).max())
  tree = scipy.spatial.cKDTree(vertices)
  _, near = tree.query(points, k=n_candidate_vertices)
  near = near.reshape(len(points), -1)
  normals = np.cross(vertices[faces[:, 1]] - vertices[faces[:, 0]],
                     vertices[faces[:, 2]] - vertices[faces[:, 0]])
  normals /= np.linalg.norm(normals, axis=1, keepdims=True)

  n_pts = len(points)
  cand = np.full((n_pts, near.shape[1] * max_deg), -1, dtype=np.int64)
  for j in range(near.shape[1]):
    v = near[:, j]
    for d in range(max_deg):
      pos = starts[v] + d
      ok = pos < starts[v + 1]
      cand[ok, j * max_deg + d] = owner[np.minimum(pos, len(owner) - 1)][ok]
  # Duplicates and "no face" slots are masked with +inf distance; candidates
  # are sorted ascending so that an exact tie resolves to the lowest face id.
  cand.sort(axis=1)
  dup = np.zeros_like(cand, dtype=bool)
  dup[:, 1:] = cand[:, 1:] == cand[:, :-1]
  valid = (cand >= 0) & ~dup
  dist = np.full(cand.shape, np.inf)
  align = np.full(cand.shape, -np.inf)
  for col in range(cand.shape[1]):
    idx = np.nonzero(valid[:, col])[0]
    if len(idx) == 0:
      continue
    fi = cand[idx, col]
    q = closest_point_on_triangles(points[idx], vertices[faces[fi, 0]],
                                   vertices[faces[fi, 1]], vertices[faces[fi, 2]])
    d = points[idx] - q
    dist[idx, col] = np.linalg.norm(d, axis=1)
    align[idx, col] = np.abs(np.einsum("ij,ij->i", normals[fi], d))
  near_min = dist <= dist.min(axis=1, keepdims=True) + _TIE_TOL
  pick = np.argmax(np.where(near_min, align, -np.inf), axis=1)
  best_face = cand[np.arange(n_pts), pick]
  assert (best_face >= 0).all()
  return best_face


def containing_triangle_query(lat_deg, lon_deg, mesh_vertices, mesh_faces):
  """(grid_idx, mesh_idx): 3 edges per grid point, receiver(grid)-sorted (:119-134)."""
  pts = grid_xyz(lat_deg, lon_deg).reshape([-1, 3])
  face = containing_triangle_faces(pts, mesh_vertices, mesh_faces)
  mesh_idx = np.asarray(mesh_faces)[face].reshape([-1])
  grid_idx = np.tile(np.arange(len(pts)).reshape([-1, 1]), [1, 3]).reshape([-1])
  return grid_idx, mesh_idx
