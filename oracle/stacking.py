"""Oracle: Dataset <-> [node, batch, channel] stacking on plain dicts of numpy arrays
(TEST INFRASTRUCTURE, see oracle/__init__.py).

Restates ``weathernext/utils/model_utils.py``: ``variable_to_stacked`` (:645-674),
``dataset_to_stacked`` (:677-710), ``stacked_to_dataset`` (:713-776),
``lat_lon_to_leading_axes`` / ``restore_leading_axes`` (:155-177) and their use in
``graphcast.py:680-723`` -- without any labelled-array container: a variable is
``(dims, array)``.  Independent of ``graphcast_amd.xarray_lite`` on purpose.
"""
import numpy as np

PRESERVED = ("batch", "lat", "lon")


def to_channels(dims, array, sizes):
  """(dims, array) -> [batch, lat, lon, channels]; folded dims in C order (time-major, level)."""
  array = np.asarray(array)
  folded = [d for d in dims if d not in PRESERVED]
  kept = [d for d in dims if d in PRESERVED]
  perm = [dims.index(d) for d in kept + folded]
  a = np.transpose(array, perm)
  n_ch = int(np.prod([array.shape[dims.index(d)] for d in folded])) if folded else 1
  a = a.reshape([array.shape[dims.index(d)] for d in kept] + [n_ch])
  # insert missing preserved dims (broadcast), then order them batch, lat, lon
  for d in PRESERVED:
    if d not in kept:
      a = np.broadcast_to(a[None], (sizes[d],) + a.shape)
      kept = [d] + kept
  order = [kept.index(d) for d in PRESERVED] + [len(kept)]
  return np.transpose(a, order)


def grid_node_features(inputs, forcings, sizes):
  """dicts name -> (dims, array)  ->  [lat*lon, batch, channels]  (graphcast.py:680-699)."""
  blocks = [to_channels(*inputs[k], sizes) for k in sorted(inputs)]
  blocks += [to_channels(*forcings[k], sizes) for k in sorted(forcings)]
  stacked = np.concatenate(blocks, axis=-1)                 # [batch, lat, lon, C]
  leading = np.transpose(stacked, (1, 2, 0, 3))             # [lat, lon, batch, C]
  return leading.reshape((-1,) + leading.shape[2:])


def prediction_from_grid_nodes(outputs, template, n_lat, n_lon):
  """[lat*lon, batch, C_out] -> dict name -> array shaped like template[name] = (dims, shape)
  (graphcast.py:701-723)."""
  y = np.asarray(outputs).reshape((n_lat, n_lon) + outputs.shape[1:])
  y = np.transpose(y, (2, 0, 1, 3))                         # [batch, lat, lon, C]
  out, start = {}, 0
  for name in sorted(template):
    dims, shape = template[name]
    folded = [d for d in dims if d not in PRESERVED]
    n = int(np.prod([shape[dims.index(d)] for d in folded])) if folded else 1
    piece = y[..., start:start + n]
    start += n
    piece = piece.reshape(piece.shape[:3] + tuple(shape[dims.index(d)] for d in folded))
    cur = list(PRESERVED) + folded
    out[name] = np.transpose(piece, [cur.index(d) for d in dims])
  if start != y.shape[-1]:
    raise ValueError(f"Expected {start} channels but found {y.shape[-1]}")
  return out
