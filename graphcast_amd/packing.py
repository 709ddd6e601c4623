"""Host-side layout transforms for the device plan (pure numpy, CPU-testable).

* ``pack_weight``: haiku ``w`` [K, N] -> the k4-interleaved layout the kernels
  stream through LDS (see include/gcast.h).
* ``pack_edges``: reorders an edge set into receiver-sorted rows packed into
  64-row tiles, with the metadata the fused edge kernel needs to perform
  ``jraph.segment_sum`` (reference ``typed_graph_net.py:532-538``)
  deterministically: per-row receiver ids, per-tile straddle flags and the list
  of receivers whose edges span more than one tile.

The reference keeps edges in construction order (sender-sorted for grid2mesh,
unsorted for the multi-mesh, receiver-sorted for mesh2grid,
``SURVEY.md`` appendix A.1) and lets XLA scatter-add.  Edge latents are internal
to the step, so the engine is free to keep them in packed order; ``perm``
records packed row -> original edge id so that indices stay checkable
bit-for-bit against the reference's.
"""
from typing import NamedTuple

import numpy as np

TILE = 64
K_CHUNK = 32
LATENT = 512


def round_up(x, m):
  return (x + m - 1) // m * m


def pack_weight(w, np_cols=LATENT):
  """[K, N] -> [ceil32(K)/4, np_cols, 4] float32 with Wp[q, n, j] = w[4q + j, n]."""
  w = np.asarray(w, dtype=np.float32)
  k, n = w.shape
  if n > np_cols:
    raise ValueError(f"weight has {n} columns, packed layout holds {np_cols}")
  kp = round_up(k, K_CHUNK)
  padded = np.zeros((kp, np_cols), dtype=np.float32)
  padded[:k, :n] = w
  return np.ascontiguousarray(padded.reshape(kp // 4, 4, np_cols).transpose(0, 2, 1))


def split_f16(x):
  """x (float32) -> (hi, lo) float16 with hi = fp16(x), lo = fp16(x - hi): 22 mantissa bits.

  Round-to-nearest both times; |x - (hi + lo)| <= max(2^-22 |x|, 2^-25) (the second term is
  the fp16 subnormal spacing: tiny weights keep an ABSOLUTE error of 3e-8)."""
  x = np.asarray(x, dtype=np.float32)
  if np.abs(x).max(initial=0.0) > 65504.0:
    raise ValueError("weight magnitude above the fp16 range (65504)")
  hi = x.astype(np.float16)
  lo = (x - hi.astype(np.float32)).astype(np.float16)
  return hi, lo


def _kmap(chained):
  """k index inside a 32-chunk for (g, j): lane group g (4) supplies 8 k values j."""
  g = np.arange(4)[:, None]
  j = np.arange(8)[None, :]
  if chained:
    return np.where(j < 4, 4 * g + j, 16 + 4 * g + (j - 4))       # [4, 8]
  return 8 * g + j


def choose_weight_scale(w):
  """Power of two s such that rms(s w) ~ 0.5 and max|s w| <= 2^14: the lo half fp16(sw - hi) of
  a typical weight is then a NORMAL fp16 number (fp16 subnormals below 2^-14 have an absolute
  spacing of 2^-24, which would leave weights of ~1e-3 with only ~17 good bits)."""
  w = np.asarray(w, dtype=np.float64)
  rms = float(np.sqrt(np.mean(w * w))) if w.size else 0.0
  top = float(np.abs(w).max()) if w.size else 0.0
  if not np.isfinite(top) or rms == 0.0:
    return 1.0
  k = int(np.round(-1.0 - np.log2(rms)))
  k = min(k, int(np.floor(14.0 - np.log2(top))))
  return float(2.0 ** max(min(k, 24), -24))


def pack_weight_split(w, np_cols=LATENT, chained=False, scale=1.0):
  """[K, N] float32 -> uint16 [ceil32(K)/32, np_cols/16, 2, 64, 8]: the GC_PREC_F16X3 layout of
  include/gcast.h.  Entry [c, nb, part, 16 g + n, j] = part(hi|lo) of s*w[32 c + kmap(g, j)][16 nb + n];
  ``chained`` selects the K permutation of a layer fed by the previous layer's accumulator
  registers (layer 2) instead of by rows read from memory (layer 1); ``scale`` (a power of two,
  see ``choose_weight_scale``) goes into gc_rowmlp_desc.w1_scale / w2_scale."""
  w = np.asarray(w, dtype=np.float32) * np.float32(scale)
  k, n = w.shape
  if n > np_cols:
    raise ValueError(f"weight has {n} columns, packed layout holds {np_cols}")
  kp = round_up(k, K_CHUNK)
  padded = np.zeros((kp, np_cols), dtype=np.float32)
  padded[:k, :n] = w
  hi, lo = split_f16(padded)
  km = _kmap(chained)                                      # [4, 8]
  out = np.empty((kp // K_CHUNK, np_cols // 16, 2, 4, 16, 8), dtype=np.float16)
  for part, src in enumerate((hi, lo)):
    blk = src.reshape(kp // K_CHUNK, K_CHUNK, np_cols // 16, 16)          # [c, k, nb, n]
    gathered = blk[:, km, :, :]                                          # [c, g, j, nb, n]
    out[:, :, part] = gathered.transpose(0, 3, 1, 4, 2)                  # [c, nb, g, n, j]
  return np.ascontiguousarray(out.reshape(kp // K_CHUNK, np_cols // 16, 2, 64, 8)).view(np.uint16)


def bf16_bits(x):
  """float32 -> bfloat16 bit patterns (uint16), round to nearest even (what v_cvt_pk_bf16_f32 does)."""
  u = np.ascontiguousarray(np.asarray(x, dtype=np.float32)).view(np.uint32)
  return ((u + (((u >> 16) & 1) + np.uint32(0x7FFF))) >> 16).astype(np.uint16)


def bf16_round(x):
  """float32 -> the nearest bfloat16, returned as float32."""
  return (bf16_bits(x).astype(np.uint32) << 16).view(np.float32).reshape(np.shape(x))


def _pi_perm():
  """perm[p] = the logical column stored at position p of a GC_PREC_BF16 row (include/gcast.h "pi order":
  position 32 m + 8 g + 4 b + r holds column 16 (2 m + b) + 4 g + r)."""
  p = np.arange(LATENT)
  m, g, b, r = p // 32, (p // 8) % 4, (p // 4) % 2, p % 4
  return 16 * (2 * m + b) + 4 * g + r


PI_PERM = _pi_perm()
PI_INV = np.argsort(PI_PERM)


def to_pi(a):
  """[..., 512] logical columns -> pi order (works on numpy arrays and torch tensors)."""
  return a[..., PI_PERM]


def from_pi(a):
  """[..., 512] pi order -> logical columns."""
  return a[..., PI_INV]


def pack_weight_bf16(w, np_cols=LATENT, chained=False):
  """[K, N] float32 -> uint16 [ceil32(K)/32, np_cols/16, 64, 8]: the GC_PREC_BF16_IMAGE layout of
  include/gcast.h (the hi-only analogue of ``pack_weight_split``, same K maps)."""
  w = np.asarray(w, dtype=np.float32)
  k, n = w.shape
  if n > np_cols:
    raise ValueError(f"weight has {n} columns, packed layout holds {np_cols}")
  kp = round_up(k, K_CHUNK)
  padded = np.zeros((kp, np_cols), dtype=np.float32)
  padded[:k, :n] = w
  bits = bf16_bits(padded)
  km = _kmap(chained)
  blk = bits.reshape(kp // K_CHUNK, K_CHUNK, np_cols // 16, 16)              # [c, k, nb, n]
  out = blk[:, km, :, :].transpose(0, 3, 1, 4, 2)                            # [c, nb, g, n, j]
  return np.ascontiguousarray(out.reshape(kp // K_CHUNK, np_cols // 16, 64, 8))


def unpack_weight_split(wp, k, n, chained=False):
  """Inverse of pack_weight_split -> (hi, lo) float32 [k, n] (tests)."""
  wp = np.asarray(wp).view(np.float16)
  chunks, nblk = wp.shape[0], wp.shape[1]
  km = _kmap(chained)
  parts = []
  for part in range(2):
    v = wp[:, :, part].reshape(chunks, nblk, 4, 16, 8).astype(np.float32)   # [c, nb, g, n, j]
    full = np.zeros((chunks, K_CHUNK, nblk, 16), dtype=np.float32)
    full[:, km, :, :] = v.transpose(0, 2, 4, 1, 3)                        # [c, g, j, nb, n]
    parts.append(full.reshape(chunks * K_CHUNK, nblk * 16)[:k, :n])
  return parts[0], parts[1]


def unpack_weight(wp, k, n):
  """Inverse of pack_weight (tests)."""
  q, np_cols, _ = wp.shape
  return wp.transpose(0, 2, 1).reshape(q * 4, np_cols)[:k, :n]


def pad_vector(v, n=LATENT):
  out = np.zeros(n, dtype=np.float32)
  out[:len(v)] = v
  return out


class PackedEdges(NamedTuple):
  n_edges: int              # real edges
  n_rows: int               # packed rows (multiple of 64)
  perm: np.ndarray          # [n_rows] int64: original edge id per packed row, -1 = padding
  senders: np.ndarray       # [n_rows] int32, -1 = padding
  receivers: np.ndarray     # [n_rows] int32, -1 = padding  (the kernel's `seg`)
  tile_flags: np.ndarray    # [n_rows/64] int32, bit0/bit1 = first/last run straddles
  fix_recv: np.ndarray      # [n_fix] int32 receivers spanning several tiles
  fix_t0: np.ndarray        # [n_fix] int32 first tile
  fix_t1: np.ndarray        # [n_fix] int32 last tile
  empty_receivers: np.ndarray   # [n_empty] int32 receivers with no incoming edge


def pack_edges(senders, receivers, n_receivers):
  """Receiver-sorted (stable), tile-packed edge order + segment-sum metadata."""
  senders = np.asarray(senders)
  receivers = np.asarray(receivers)
  n_edges = len(receivers)
  if n_edges == 0:
    raise ValueError("empty edge set")
  if receivers.min() < 0 or receivers.max() >= n_receivers:
    raise ValueError("receiver index out of range")
  order = np.argsort(receivers, kind="stable")
  r_sorted = receivers[order]
  degree = np.bincount(r_sorted, minlength=n_receivers)
  nonzero = degree[degree > 0]
  uniform = int(nonzero[0]) if (nonzero == nonzero[0]).all() else 0

  if uniform and uniform <= TILE and TILE % uniform:
    # whole segments per tile, the tail rows of each tile are padding: no straddling
    rows_per_tile = TILE // uniform * uniform
    e = np.arange(n_edges)
    pos = e // rows_per_tile * TILE + e % rows_per_tile
    n_rows = round_up(int(pos[-1]) + 1, TILE)
  else:
    pos = np.arange(n_edges)
    n_rows = round_up(n_edges, TILE)

  perm = np.full(n_rows, -1, dtype=np.int64)
  perm[pos] = order
  snd = np.full(n_rows, -1, dtype=np.int32)
  rcv = np.full(n_rows, -1, dtype=np.int32)
  snd[pos] = senders[order]
  rcv[pos] = r_sorted

  n_tiles = n_rows // TILE
  rt = rcv.reshape(n_tiles, TILE)
  first = rt[:, 0]
  # last valid receiver of every tile (padding only ever trails a tile)
  n_valid = (rt >= 0).sum(axis=1)
  last = rt[np.arange(n_tiles), np.maximum(n_valid - 1, 0)]
  full = n_valid == TILE
  cont = np.zeros(n_tiles, dtype=bool)            # tile t continues tile t-1's last run
  cont[1:] = full[:-1] & (first[1:] == last[:-1]) & (first[1:] >= 0)
  flags = cont.astype(np.int32)
  flags[:-1] |= cont[1:].astype(np.int32) << 1

  # receivers whose packed rows span more than one tile
  seg_start = np.flatnonzero(np.r_[True, r_sorted[1:] != r_sorted[:-1]])
  seg_end = np.r_[seg_start[1:], n_edges] - 1
  t0 = pos[seg_start] // TILE
  t1 = pos[seg_end] // TILE
  span = t1 > t0
  return PackedEdges(
      n_edges=n_edges, n_rows=n_rows, perm=perm, senders=snd, receivers=rcv,
      tile_flags=flags,
      fix_recv=r_sorted[seg_start][span].astype(np.int32),
      fix_t0=t0[span].astype(np.int32), fix_t1=t1[span].astype(np.int32),
      empty_receivers=np.flatnonzero(degree == 0).astype(np.int32))


def segment_sum_packed_reference(rows, packed: PackedEdges, n_receivers):
  """What the device pipeline (tile kernel + fixup + zero rows) must produce (tests)."""
  out = np.zeros((n_receivers, rows.shape[1]), dtype=rows.dtype)
  ok = packed.receivers >= 0
  np.add.at(out, packed.receivers[ok], rows[ok])
  return out
