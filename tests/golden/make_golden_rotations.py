"""Generates tests/golden/rotation_ref.npz by calling the REFERENCE's own local-coordinate helpers
(weathernext/utils/model_utils.py: get_rotation_matrices_to_local_coordinates,
rotate_with_matrices, get_relative_position_in_receiver_local_coordinates and its bipartite form;
numpy + scipy only -- jax / xarray are inert stand-ins that these functions never touch).

    python tests/golden/make_golden_rotations.py
"""
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


class _Inert(types.ModuleType):
  def __getattr__(self, name):
    if name.startswith("__"):
      raise AttributeError(name)
    sub = _Inert(f"{self.__name__}.{name}")
    setattr(self, name, sub)
    return sub

  def __call__(self, *a, **k):                  # (typing accepts any callable as a TypeVar constraint)
    raise RuntimeError(f"inert stand-in {self.__name__} was called")


for name in ("jax", "jax.numpy", "xarray", "xarray.ufuncs"):
  sys.modules.setdefault(name, _Inert(name))
sys.path.insert(0, REF)
from weathernext.utils import model_utils as ref                 # noqa: E402


def main():
  rng = np.random.default_rng(4)
  n, n2, e = 9, 6, 25
  phi = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
  theta = rng.uniform(0.05, np.pi - 0.05, n).astype(np.float32)
  phi2 = rng.uniform(0, 2 * np.pi, n2).astype(np.float32)
  theta2 = rng.uniform(0.05, np.pi - 0.05, n2).astype(np.float32)
  senders, receivers = rng.integers(0, n, e), rng.integers(0, n, e)
  b_send, b_recv = rng.integers(0, n, e), rng.integers(0, n2, e)
  pos = rng.standard_normal((n, 3))
  out = dict(phi=phi, theta=theta, phi2=phi2, theta2=theta2, senders=senders, receivers=receivers,
             b_send=b_send, b_recv=b_recv, pos=pos)
  for lat, lon in ((True, True), (False, True), (True, False)):
    tag = f"lat{int(lat)}lon{int(lon)}"
    m = ref.get_rotation_matrices_to_local_coordinates(phi, theta, rotate_latitude=lat, rotate_longitude=lon)
    out[f"mat_{tag}"] = m
    out[f"rot_{tag}"] = ref.rotate_with_matrices(m, pos)
    out[f"rel_{tag}"] = ref.get_relative_position_in_receiver_local_coordinates(
        phi, theta, senders, receivers, latitude_local_coordinates=lat, longitude_local_coordinates=lon)
    out[f"brel_{tag}"] = ref.get_bipartite_relative_position_in_receiver_local_coordinates(
        phi, theta, b_send, phi2, theta2, b_recv, latitude_local_coordinates=lat, longitude_local_coordinates=lon)
  path = os.path.join(HERE, "rotation_ref.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, {k: (v.shape, str(v.dtype)) for k, v in out.items() if k.endswith("lat1lon1")})


if __name__ == "__main__":
  main()
