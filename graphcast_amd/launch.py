"""Launch-level helpers over the C-ABI (include/gcast.h: gc_rowmlp_desc / gc_op / gc_run_program): packed device
copies of MLPs and edge sets, and a base class that fills descriptors and enqueues op lists.

Who builds programs from these: ``deep_gnn.DeepGNN`` (WN2's processor) and ``conditioned.ConditionedEncoderDecoder``
(GenCast's encoder / decoder) -- models of their own, recorded in Python through the launch-level API.  The GraphCast
step itself is NOT recorded here: its one launch program lives in csrc/gcast_plan.inc (gc_plan_program) and
``engine.StepEngine`` drives that.  All arithmetic happens in libgcast_hip.so; torch only owns the memory.
"""
import ctypes
import weakref
from typing import Optional

import numpy as np
import torch

from graphcast_amd import _native as nat
from graphcast_amd import packing

D = packing.LATENT
DEFAULT_PRECISION = "f16x3"

# stage tags of a program's ops (include/gcast.h: enum gc_stage_tag), reported by gc_time_program / used by bench.py
TAGS = dict(prep=0, enc_embed_grid=1, enc_pre=2, enc_edge=3, enc_node_mesh=4, enc_node_grid=5,
            proc_pre=6, proc_edge=7, proc_node=8, dec_pre=9, dec_edge=10, dec_node=11,
            dec_out=12, fixup=13)


# Engines whose last range check was the non-blocking kind (LaunchBase.check_range(wait=False)): its verdict is still
# pending.  flush_range_checks() settles them -- called wherever the host synchronises anyway.
_pending_range_checks = weakref.WeakSet()


def flush_range_checks():
  """The blocking range check on every engine that has a non-blocking one pending (raises GcastRangeError)."""
  for engine in list(_pending_range_checks):
    engine.check_range(wait=True)


class _PW:
  """A packed weight image on the device + the power of two it was multiplied by."""
  __slots__ = ("t", "scale")

  def __init__(self, t, scale=1.0):
    self.t, self.scale = t, float(scale)

  def data_ptr(self):
    return self.t.data_ptr()


class _Mlp:
  """Packed device copy of one `<stem>_mlp` (+ `<stem>_layer_norm`)."""

  def __init__(self, params, stem, dev, split=None, np2=D, prec=nat.PREC_F32, k_natural=False):
    w1 = np.asarray(params[f"{stem}_mlp/~/linear_0"]["w"], dtype=np.float32)
    b1 = np.asarray(params[f"{stem}_mlp/~/linear_0"]["b"], dtype=np.float32)
    w2 = np.asarray(params[f"{stem}_mlp/~/linear_1"]["w"], dtype=np.float32)
    b2 = np.asarray(params[f"{stem}_mlp/~/linear_1"]["b"], dtype=np.float32)
    if f"{stem}_mlp/~/linear_2" in params:
      raise NotImplementedError("only mlp_num_hidden_layers == 1 (GraphCast's value) is built")
    if w1.shape[1] != D or w2.shape[0] != D:
      raise NotImplementedError(f"latent/hidden size must be {D}, got {w1.shape}, {w2.shape}")
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    if prec == nat.PREC_F16X3:
      # (hi, lo) fp16 images; layer 1 reads rows from memory (natural K order), layer 2 is fed by
      # layer 1's accumulator registers (chained K order) -- include/gcast.h.  Stored as int16
      # bit patterns: the kernels only ever see the raw chunk image.
      def pack1(w):
        sc = packing.choose_weight_scale(w)
        return _PW(up(packing.pack_weight_split(w, scale=sc).view(np.int16)), sc)

      def pack2(w, np_cols):
        sc = packing.choose_weight_scale(w)
        return _PW(up(packing.pack_weight_split(w, np_cols=np_cols, chained=True, scale=sc)
                      .view(np.int16)), sc)
    elif prec == nat.PREC_BF16:
      # GC_PREC_BF16: the bfloat16 view of the fp32-stored parameters (reference casting.py:155-205).  A
      # matrix whose K operand is a bfloat16 row tensor (pi order == the chained K order) is packed
      # chained; `k_natural` marks the one fed by external fp32 rows (the grid embedder's first layer).
      pack1 = lambda w: _PW(up(packing.pack_weight_bf16(w, chained=not k_natural).view(np.int16)))
      pack2 = lambda w, np_cols: _PW(up(packing.pack_weight_bf16(w, np_cols=np_cols, chained=True)
                                       .view(np.int16)))
      b1, b2 = packing.bf16_round(b1), packing.bf16_round(b2)
    else:
      pack1 = lambda w: _PW(up(packing.pack_weight(w)))
      pack2 = lambda w, np_cols: _PW(up(packing.pack_weight(w, np_cols=np_cols)))
    self.k_in = w1.shape[0]
    self.n_out = w2.shape[1]
    self._w1_raw, self._pack2, self._chained = w1, pack2, {}
    # f16x3: W2 once more in the NATURAL K order, for the one-pass launches (GC_W2_NATURAL) of the
    # edge updates that have no layer-1 GEMM (include/gcast.h); same scale as the chained image
    self.w2_natural = None
    if prec == nat.PREC_F16X3 and split is not None and len(split) == 3 and np2 == D:
      sc = packing.choose_weight_scale(w2)
      self.w2_natural = _PW(up(packing.pack_weight_split(w2, np_cols=D, chained=False, scale=sc).view(np.int16)), sc)
    # W1 either whole, or split into named row blocks of 512 (concat order)
    if split is None:
      self.w1 = pack1(w1)
      self.k1p = packing.round_up(w1.shape[0], packing.K_CHUNK)
    else:
      assert w1.shape[0] == D * len(split), (stem, w1.shape, split)
      self.w1 = {name: pack1(w1[j * D:(j + 1) * D]) for j, name in enumerate(split)}
    self.b1 = up(b1)
    self.w2 = pack2(w2, np2)
    self.b2 = up(packing.pad_vector(b2, np2))
    self.scale = self.offset = None
    if f"{stem}_layer_norm" in params:
      vec = packing.bf16_round if prec == nat.PREC_BF16 else (lambda a: a)
      self.scale = up(vec(np.asarray(params[f"{stem}_layer_norm"]["scale"], dtype=np.float32)))
      self.offset = up(vec(np.asarray(params[f"{stem}_layer_norm"]["offset"], dtype=np.float32)))


class _Edges:
  """Device copy of a packed edge set."""

  def __init__(self, pk: packing.PackedEdges, dev):
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    self.pk = pk
    self.n_rows = pk.n_rows
    self.snd, self.rcv = up(pk.senders), up(pk.receivers)
    self.flags = up(pk.tile_flags)
    self.fix = (up(pk.fix_recv), up(pk.fix_t0), up(pk.fix_t1)) if len(pk.fix_recv) else None
    self.empty = up(pk.empty_receivers) if len(pk.empty_receivers) else None
    self.partial = torch.empty((2 * pk.n_rows // packing.TILE, D), dtype=torch.float32, device=dev)


def _chained(mlp: _Mlp, block=None):
  """First-layer matrix of `mlp` (or its 512-row block `block` of a split one) packed like a
  layer-2 matrix (chained K order): what a GC_CHAIN stage needs, because its K operand is the
  producing launch's rows as they sit in the accumulator registers (include/gcast.h)."""
  if block not in mlp._chained:
    w = mlp._w1_raw
    if block is not None:
      j = {"e": 0, "s": 1, "r": 2, "h": 0, "a": 1}[block]
      w = w[j * D:(j + 1) * D]
    mlp._chained[block] = mlp._pack2(w, D)
  return mlp._chained[block]


class LaunchBase:
  """Descriptor filling + program enqueueing shared by the Python-recorded models (DeepGNN, the conditioned encoder /
  decoder).  A subclass sets: dev, lib, prec / precision, half, onepass, scratch (None), range_flag, _keep."""

  helpers_min_rows = 0
  range_flag = None
  tile_queue = None
  check_all_rows = False        # True: EVERY launch with layer-1 rows carries the range flag (their latents are external)

  # ---------------------------------------------------------------- helpers
  def _new(self, rows, cols=D):
    t = torch.empty((rows, cols), dtype=torch.float32, device=self.dev)
    self._keep.append(t)
    return t

  def _up(self, a, dtype=np.float32):
    t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dtype))).to(self.dev)
    self._keep.append(t)
    return t

  def _stream_ptr(self):
    return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

  def _desc(self, mode, n_rows, *, a0=None, k0=0, lda0=None, a1=None, k1=0, lda1=None, w1p=None,
            d=None, g0=None, idx0=None, g1=None, idx1=None, b1=None, w2p=None, b2=None, n2=0,
            ln=None, res=None, out=None, ldo=None, out_ptr=None, edges: Optional[_Edges] = None,
            agg=None, chain=(), rows_f32=False, w2_natural=None, check_range=False):
    ds = nat.RowMlpDesc()
    if (check_range or (self.check_all_rows and a0 is not None)) and self.range_flag is not None:
      ds.range_flag = self.range_flag.data_ptr()
    ds.flags = nat.ROWS_F32 if (rows_f32 and self.prec == nat.PREC_BF16) else 0
    ds.mode, ds.n_rows, ds.prec = mode, n_rows, self.prec
    ds.a0, ds.k0, ds.lda0 = nat.ptr(a0), k0, (lda0 if lda0 is not None else (a0.shape[1] if a0 is not None else 0))
    ds.a1, ds.k1, ds.lda1 = nat.ptr(a1), k1, (lda1 if lda1 is not None else (a1.shape[1] if a1 is not None else 0))
    ds.layout = nat.LAYOUT_HALF if self.half else nat.LAYOUT_CHUNKED
    if self.half and mode == nat.MODE_MLP_LN and self.prec != nat.PREC_BF16:
      ds.scratch = self._scratch_slots().data_ptr()
    if self.half:
      ds.tile_queue = self._tile_queue().data_ptr()
    ds.n_chain = len(chain)
    for k, st in enumerate(chain):
      c = ds.chain[k]
      c.wp, c.w_scale, c.kind = st["w"].data_ptr(), st["w"].scale, st["kind"]
      c.b = nat.ptr(st.get("b"))
      if st.get("out_ptr") is not None:
        c.out = st["out_ptr"]
      else:
        c.out = nat.ptr(st.get("out"))
      c.ldo = st.get("ldo", st["out"].shape[1] if st.get("out") is not None else 0)
      c.n = st.get("n", 0)
    ds.w1p = nat.ptr(w1p)
    ds.w1_scale = w1p.scale if w1p is not None else 1.0
    ds.d, ds.ldd = nat.ptr(d), (d.shape[1] if d is not None else 0)
    ds.g0, ds.idx0, ds.g1, ds.idx1 = nat.ptr(g0), nat.ptr(idx0), nat.ptr(g1), nat.ptr(idx1)
    ds.b1 = nat.ptr(b1)
    ds.w2p, ds.b2, ds.n2 = nat.ptr(w2p), nat.ptr(b2), n2
    ds.w2_scale = w2p.scale if w2p is not None else 1.0
    if (w2_natural is not None and self.onepass and self.half and self.prec == nat.PREC_F16X3
        and mode == nat.MODE_MLP_LN and k0 + k1 == 0 and d is not None and g0 is not None and not chain):
      # an edge update whose first layer was folded into addends: ONE pass (csrc/rowmlp_half.inc ONEPASS)
      ds.w2p, ds.flags = w2_natural.data_ptr(), ds.flags | nat.W2_NATURAL
    if (self.helpers_min_rows and n_rows >= self.helpers_min_rows and self.half and self.prec == nat.PREC_F16X3
        and g0 is None and edges is None):
      # the big node-side launches (no gather, no segment-sum) in the eight-wave form: four multiplying + four
      # weight-staging waves, parked accumulators in LDS (csrc/rowmlp_half.inc: rowmlp16d_kernel); same bits
      ds.flags |= nat.WG_HELPERS
    if ln is not None:
      ds.ln_scale, ds.ln_offset = nat.ptr(ln[0]), nat.ptr(ln[1])
    ds.res, ds.ldres = nat.ptr(res), (res.shape[1] if res is not None else 0)
    ds.out = out_ptr if out_ptr is not None else nat.ptr(out)
    ds.ldo = ldo if ldo is not None else (out.shape[1] if out is not None else 0)
    if edges is not None:
      ds.seg, ds.tile_flags = nat.ptr(edges.rcv), nat.ptr(edges.flags)
      ds.agg, ds.partial = nat.ptr(agg), nat.ptr(edges.partial)
    return ds

  def _scratch_slots(self):
    """GC_LAYOUT_HALF: the parking slots of the persistent workgroups (include/gcast.h:
    gc_rowmlp_desc.scratch) -- 32 MiB whatever the launch sizes are, rewritten by every tile and
    therefore cache resident; shared by all launches of the engine (they run one after another)."""
    if self.scratch is None:
      self.scratch = torch.empty((nat.SCRATCH_FLOATS,), dtype=torch.float32, device=self.dev)
      self._keep.append(self.scratch)
    return self.scratch

  def _tile_queue(self):
    """The persistent kernels' dynamic tile queue (include/gcast.h: gc_rowmlp_desc.tile_queue): two device words,
    zero here and left zero by every launch; shared by all launches of the engine like the parking slots (they run
    one after another on one stream)."""
    if self.tile_queue is None:
      self.tile_queue = torch.zeros((2,), dtype=torch.int32, device=self.dev)
    return self.tile_queue

  def _op_mlp(self, tag, desc):
    op = nat.Op()
    op.kind, op.tag, op.mlp = nat.OP_ROWMLP, TAGS[tag], desc
    return op

  def _ops_after_segsum(self, edges: _Edges, agg, zero=True):
    ops = []
    if edges.fix is not None:
      op = nat.Op()
      op.mlp.prec = self.prec           # (GC_PREC_BF16: bfloat16 aggregate rows)
      op.kind, op.tag, op.n = nat.OP_FIXUP, TAGS["fixup"], edges.fix[0].numel()
      op.i0, op.i1, op.i2 = (nat.ptr(t) for t in edges.fix)
      op.src, op.dst = nat.ptr(edges.partial), nat.ptr(agg)
      ops.append(op)
    if zero and edges.empty is not None:
      op = nat.Op()
      op.mlp.prec = self.prec
      op.kind, op.tag, op.n = nat.OP_ZERO, TAGS["fixup"], edges.empty.numel()
      op.i0, op.dst = nat.ptr(edges.empty), nat.ptr(agg)
      ops.append(op)
    return ops

  def _run(self, ops):
    arr = (nat.Op * len(ops))(*ops)
    with torch.cuda.device(self.dev):
      if self.tile_queue is not None:     # (see _clear_tile_queue; DeepGNN / ConditionedEncoderDecoder run through here)
        self.tile_queue.zero_()
      nat.check(self.lib.gc_run_program(arr, len(ops), self._stream_ptr()), "gc_run_program")

  def _mlp_ln(self, n_rows, mlp: _Mlp, **kw):
    return self._desc(nat.MODE_MLP_LN, n_rows, w2p=mlp.w2, b2=mlp.b2, n2=D,
                      ln=(mlp.scale, mlp.offset), w2_natural=getattr(mlp, "w2_natural", None), **kw)

  _range_pending = None

  def check_range(self, wait: bool = True):
    """Raises GcastRangeError if a step since the last call read an input value, or an AGGREGATE (the layer-1 operand
    of the encoder's mesh-node update, the processor's node updates and the decoder's grid-node update: a sum over
    up to 3,753 edges, which the reference up-casts to fp32 for this reason, graphcast.py:215), outside the exact range of the
    f16x3 arithmetic (|x| > 65504: the split halves saturate -- 5e-4 errors up to 1.3e5, garbage beyond -- where
    the reference's fp32 does not care; un-normalised geopotential is ~5e5).  SYNCHRONISES the launch stream:
    call it where the host waits for the step anyway (GraphCast.__call__ on host Datasets, DeviceRollout.run,
    bench.py do); ``wait=False`` never blocks (a device-resident Dataset rollout must not wait on the host
    once per step): the verdict of such a call is PENDING until the next call on this engine or -- round 6, ADVICE r5 --
    the host's next real synchronisation point, whichever comes first: ``flush_range_checks()`` (module level) runs the
    blocking check on every engine with a pending one, and ``rollout._to_host`` (the reference's ``jax.device_get``), the
    end of ``rollout.chunked_prediction_generator`` and ``GraphCast.set_precision`` call it."""
    if self.range_flag is None:
      return
    if not wait:
      # device-resident callers (torch-backed Datasets: nothing else makes the host wait): the word is copied to pinned
      # memory behind the step and tested at the NEXT call -- the launches never clear it, so an out-of-range step is
      # reported one call late at worst (and at the latest by the first blocking check: flush_range_checks above)
      _pending_range_checks.add(self)
      if self._range_pending is not None:
        host, done = self._range_pending
        if not done.query():
          return                         # (still in flight: test it next time)
        self._range_pending = None
        hit = int(host.item()) != 0
      else:
        hit = False
      if not hit:
        host = torch.empty((1,), dtype=torch.int32, pin_memory=True)
        host.copy_(self.range_flag, non_blocking=True)
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.dev))
        self._range_pending = (host, done)
        return
    else:
      _pending_range_checks.discard(self)
      self._range_pending = None
      if int(self.range_flag.item()) == 0:
        return
    _pending_range_checks.discard(self)
    self._range_pending = None
    self.range_flag.zero_()
    raise nat.GcastRangeError(
        f"an input value -- or a per-receiver sum of edge messages (a node update's aggregate operand) -- exceeds "
        f"{nat.F16X3_MAX:g} in magnitude: outside the exact range of the f16x3 arithmetic (precision='f16x3').  "
        "Normalise the inputs (normalization.InputsAndResiduals, as the reference's demo stack does) or run with "
        "precision='f32'.")

