"""Oracle: icosahedral multi-mesh (TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``/root/reference/weathernext/utils/icosahedral_mesh.py``:
  * ``icosahedron``          <- ``get_icosahedron`` (:136-222)
  * ``split_faces_once``     <- ``_two_split_unit_sphere_triangle_faces`` (:225-263)
                                + ``_ChildVerticesBuilder`` (:321-363)
  * ``mesh_hierarchy``       <- ``get_hierarchy_of_triangular_meshes_for_sphere`` (:98-133)
  * ``merged_faces``         <- ``merge_meshes`` (:79-95)
  * ``faces_to_edges``       <- ``faces_to_edges`` (:366-388)

Deliberately written as the slow, literal per-face loop so that it is easy to
audit against the reference; the product has its own vectorised builder.
"""
import numpy as np
from scipy.spatial import transform

# Face table of the regular icosahedron, counter-clockwise seen from outside
# (icosahedral_mesh.py:173-193).  This is data of the algorithm, not code.
_ICO_FACES = (
    (0, 1, 2), (0, 6, 1), (8, 0, 2), (8, 4, 0), (3, 8, 2), (3, 2, 7), (7, 2, 1),
    (0, 4, 6), (4, 11, 6), (6, 11, 5), (1, 5, 7), (4, 10, 11), (4, 8, 10),
    (10, 8, 3), (10, 3, 9), (11, 10, 9), (11, 9, 5), (5, 9, 7), (9, 3, 7),
    (1, 6, 5))


def icosahedron():
  """12 float32 unit vertices + 20 int32 faces, top/bottom faces pole-parallel."""
  golden = (1 + np.sqrt(5)) / 2
  verts = []
  for a in (1.0, -1.0):                      # icosahedral_mesh.py:163-167
    for b in (golden, -golden):
      verts += [(a, b, 0.0), (0.0, a, b), (b, 0.0, a)]
  verts = np.array(verts, dtype=np.float32)
  verts /= np.linalg.norm([1.0, golden])     # :170 (float32 / float64 scalar -> float32)
  # Rotate about y so that a face (not an edge) is on top (:215-219).
  dihedral = 2 * np.arcsin(golden / np.sqrt(3))
  rot = transform.Rotation.from_euler(seq="y", angles=(np.pi - dihedral) / 2)
  verts = np.dot(verts, rot.as_matrix())
  return verts.astype(np.float32), np.array(_ICO_FACES, dtype=np.int32)


def split_faces_once(vertices, faces):
  """One 4-way split; new vertices appended in order of first use (:335-359)."""
  verts = list(vertices)
  child_of = {}

  def midpoint(i, j):
    key = (i, j) if i < j else (j, i)
    if key not in child_of:
      p = vertices[[i, j]].mean(0)           # float32 mean
      p /= np.linalg.norm(p)
      child_of[key] = len(verts)
      verts.append(p)
    return child_of[key]

  out = []
  for v1, v2, v3 in faces:
    v1, v2, v3 = int(v1), int(v2), int(v3)
    m12 = midpoint(v1, v2)                   # creation order 12, 23, 31 (:250-252)
    m23 = midpoint(v2, v3)
    m31 = midpoint(v3, v1)
    out += [[v1, m12, m31], [m12, v2, m23], [m31, m23, v3], [m12, m23, m31]]
  return np.array(verts), np.array(out, dtype=np.int32)


def mesh_hierarchy(splits):
  """List of (vertices, faces) for refinement levels 0..splits."""
  v, f = icosahedron()
  levels = [(v, f)]
  for _ in range(splits):
    v, f = split_faces_once(v, f)
    levels.append((v, f))
  return levels


def merged_faces(levels):
  """Faces of all levels concatenated coarse -> fine (merge_meshes :93-95)."""
  return np.concatenate([f for _, f in levels], axis=0)


def faces_to_edges(faces):
  """senders=[f0;f1;f2], receivers=[f1;f2;f0] (:386-387)."""
  faces = np.asarray(faces)
  assert faces.ndim == 2 and faces.shape[1] == 3
  return (np.concatenate([faces[:, 0], faces[:, 1], faces[:, 2]]),
          np.concatenate([faces[:, 1], faces[:, 2], faces[:, 0]]))


def max_edge_length(vertices, faces):
  """graphcast.py:733-737 (_get_max_edge_distance); float32 result."""
  s, r = faces_to_edges(faces)
  return np.linalg.norm(vertices[s] - vertices[r], axis=-1).max()
