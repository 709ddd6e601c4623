"""Icosahedral multi-mesh construction (host side, runs once per Predictor).

Same public names and results as the reference's
``weathernext/utils/icosahedral_mesh.py`` (``TriangularMesh`` :46-57,
``get_hierarchy_of_triangular_meshes_for_sphere`` :98-133, ``get_icosahedron``
:136-222, ``merge_meshes`` :79-95, ``faces_to_edges`` :366-388), but built
level-at-a-time with array operations instead of a per-face Python loop with a
dict: all 3F parent pairs of a level are keyed at once, de-duplicated with
``np.unique`` and numbered by first use, which reproduces the reference's
vertex numbering exactly (checked bit-for-bit in tests against fixtures made
by the reference code).
"""
import itertools
from typing import List, NamedTuple, Sequence, Tuple

import numpy as np


class TriangularMesh(NamedTuple):
  """vertices [V, 3] float32 on the unit sphere; faces [F, 3] int32, CCW from outside."""
  vertices: np.ndarray
  faces: np.ndarray


_BASE_FACES = np.array([
    (0, 1, 2), (0, 6, 1), (8, 0, 2), (8, 4, 0), (3, 8, 2), (3, 2, 7), (7, 2, 1),
    (0, 4, 6), (4, 11, 6), (6, 11, 5), (1, 5, 7), (4, 10, 11), (4, 8, 10),
    (10, 8, 3), (10, 3, 9), (11, 10, 9), (11, 9, 5), (5, 9, 7), (9, 3, 7),
    (1, 6, 5)], dtype=np.int32)


def get_icosahedron(pole_parallel_faces: bool = True) -> TriangularMesh:
  """Regular icosahedron inscribed in the unit sphere."""
  phi = (1 + np.sqrt(5)) / 2
  signs = np.array([1.0, -1.0])
  c1 = np.repeat(signs, 2)                     # 1, 1, -1, -1
  c2 = np.tile(signs * phi, 2)                 # phi, -phi, phi, -phi
  zero = np.zeros(4)
  # per (c1, c2): (c1, c2, 0), (0, c1, c2), (c2, 0, c1)
  verts = np.stack([np.stack([c1, c2, zero], -1), np.stack([zero, c1, c2], -1),
                    np.stack([c2, zero, c1], -1)], axis=1).reshape(12, 3)
  verts = verts.astype(np.float32)
  verts /= np.linalg.norm([1.0, phi])
  if pole_parallel_faces:
    # Tilt about y by half the supplement of the dihedral angle: a face, not an
    # edge, ends up on top, so no vertex sits on a pole.
    angle = (np.pi - 2 * np.arcsin(phi / np.sqrt(3))) / 2
    c, s = np.cos(angle), np.sin(angle)
    rot_y = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
    verts = np.dot(verts, rot_y)
  return TriangularMesh(vertices=verts.astype(np.float32), faces=_BASE_FACES.copy())


def _row_norms_like_numpy_1d(x: np.ndarray) -> np.ndarray:
  # np.linalg.norm of a 1-D vector is sqrt(dot(x, x)); the BLAS dot rounds
  # differently from an elementwise sum of squares, and vertex bits must match.
  return np.sqrt(np.array([r.dot(r) for r in x], dtype=x.dtype))


def _split_level(mesh: TriangularMesh) -> TriangularMesh:
  v, f = mesh.vertices, mesh.faces.astype(np.int64)
  nv = v.shape[0]
  # parent pairs in creation order: face-major, then (1,2), (2,3), (3,1)
  pairs = np.stack([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=1).reshape(-1, 2)
  key = pairs.min(axis=1) * nv + pairs.max(axis=1)
  _, first, inverse = np.unique(key, return_index=True, return_inverse=True)
  by_first_use = np.argsort(first, kind="stable")
  rank = np.empty_like(by_first_use)
  rank[by_first_use] = np.arange(by_first_use.size)
  child = (nv + rank[inverse]).reshape(-1, 3)          # [F, 3]: m12, m23, m31
  parents = pairs[first[by_first_use]]
  mid = (v[parents[:, 0]] + v[parents[:, 1]]) / v.dtype.type(2)
  mid /= _row_norms_like_numpy_1d(mid)[:, None]
  v1, v2, v3 = f[:, 0], f[:, 1], f[:, 2]
  m12, m23, m31 = child[:, 0], child[:, 1], child[:, 2]
  faces = np.stack([v1, m12, m31, m12, v2, m23, m31, m23, v3, m12, m23, m31], axis=1)
  return TriangularMesh(vertices=np.concatenate([v, mid], axis=0),
                        faces=faces.reshape(-1, 3).astype(np.int32))


def get_hierarchy_of_triangular_meshes_for_sphere(
    splits: int, pole_parallel_faces: bool = True) -> List[TriangularMesh]:
  """Meshes for refinement levels 0..splits (each level's vertices prefix the next's)."""
  meshes = [get_icosahedron(pole_parallel_faces)]
  for _ in range(splits):
    meshes.append(_split_level(meshes[-1]))
  return meshes


def get_last_triangular_mesh_for_sphere(splits: int) -> TriangularMesh:
  return get_hierarchy_of_triangular_meshes_for_sphere(splits)[-1]


def assert_all_meshes_compatible(mesh_list: Sequence[TriangularMesh]) -> None:
  for coarse, fine in itertools.pairwise(mesh_list):
    n = coarse.vertices.shape[0]
    assert np.allclose(coarse.vertices, fine.vertices[:n])


def merge_meshes(mesh_list: Sequence[TriangularMesh]) -> TriangularMesh:
  """Finest vertices + the faces of every level, coarse to fine (the multi-mesh)."""
  assert_all_meshes_compatible(mesh_list)
  return TriangularMesh(vertices=mesh_list[-1].vertices,
                        faces=np.concatenate([m.faces for m in mesh_list], axis=0))


def faces_to_edges(faces: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
  """Each face (a, b, c) contributes a->b, b->c, c->a; senders = [a; b; c], receivers = [b; c; a]."""
  assert faces.ndim == 2 and faces.shape[-1] == 3
  return faces.T.reshape(-1), np.roll(faces, -1, axis=1).T.reshape(-1)
