"""Device plan for one GraphCast encode-process-decode step on an MI355X.

Takes the three static graphs (``graphcast.py`` builds them exactly like the
reference's ``_init_*_graph``, weathernext1_graph/graphcast.py:408-548) and the
haiku parameter tree, and turns them into
  * packed, receiver-sorted edge sets (``packing.pack_edges``),
  * k4-interleaved weights (``packing.pack_weight``) with the first edge-MLP
    matrix split into its edge / sender / receiver row blocks
    (W1 = [W_e; W_s; W_r] in concat order, deep_typed_graph_net.py:209 +
    typed_graph_net.py:448-453), so the sender/receiver products are taken per
    NODE before the gather ((x[idx]).W == (x.W)[idx]),
  * input-independent terms folded once at load time ON THE DEVICE with the
    same kernels (mesh-node embedding of [0 | struct], the three edge
    embedders, and every first-layer term that only depends on them),
  * a fixed program of fused launches (``_native.Op`` array) replayed by
    ``gc_run_program`` for every step.

All arithmetic happens in libgcast_hip.so; torch only owns the memory.
"""
import ctypes
import os
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from graphcast_amd import _native as nat
from graphcast_amd import packing

D = packing.LATENT

# Arithmetic of the GEMMs (include/gcast.h `gc_precision`): "f16x3" = fp32 operands split into
# two halves in registers, three f16 MFMAs per product, fp32 accumulation (fp32-grade results);
# "f32" = exact fp32 MFMA (the chunked round-1 kernel: bench.py's cross-check); "bf16" = the reference's Bfloat16Cast
# run (casting.py).  Overridable with GCAST_PRECISION.  (The chunked f16x3 kernel -- GCAST_HALF=0 -- and the
# "bf16gemm" operand-rounding tier of rounds 1-4 were retired in round 5: f16x3 IS the half-N formulation.)
DEFAULT_PRECISION = "f16x3"
DEFAULT_HELPERS_MIN_ROWS = "65536"     # = GC_HELPERS_MIN_ROWS_DEFAULT (include/gcast.h); see StepEngine.helpers_min_rows

# stage tags reported by gc_time_program / used by bench.py
TAGS = dict(prep=0, enc_embed_grid=1, enc_pre=2, enc_edge=3, enc_node_mesh=4, enc_node_grid=5,
            proc_pre=6, proc_edge=7, proc_node=8, dec_pre=9, dec_edge=10, dec_node=11,
            dec_out=12, fixup=13)


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


class _HaloPart:
  """The halo-sender edges of one edge set of a partitioned graph (see StepEngine._build: split): packed edges,
  folded first-layer term `d`, embedded latents `e0` / latent buffer `lat` (multi-mesh), receiver rows."""

  def __init__(self, e, d, rows, e0=None):
    self.e, self.d, self.rows, self.e0, self.lat = e, d, rows, e0, None


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


class StepEngine:
  """x [N_grid, B, C_in] fp32 (device) -> y [N_grid, B, C_out] fp32 (device)."""

  # defaults for the classes that reuse this one's launch helpers without running its constructor
  # (deep_gnn.DeepGNN, conditioned.ConditionedEncoderDecoder)
  helpers_min_rows = 0
  range_flag = None
  tile_queue = None
  check_all_rows = False        # True: EVERY launch with layer-1 rows carries the range flag (their latents are external)

  def __init__(self, graphs: Mapping, params: Mapping, *, num_steps: int, c_in: int, c_out: int,
               device="cuda:0", precision: Optional[str] = None, half: Optional[bool] = None,
               fold_only: bool = False):
    self.dev = torch.device(device)
    self.lib = nat.lib()
    precision = precision or os.environ.get("GCAST_PRECISION", DEFAULT_PRECISION)
    if precision not in nat.PRECISIONS:
      raise ValueError(f"precision must be one of {sorted(nat.PRECISIONS)}, got {precision!r}")
    self.precision = precision
    self.prec = nat.PRECISIONS[precision]
    # f16x3 and bf16 run the half-N formulation (csrc/rowmlp_half.inc, rowmlp_bf16.inc: <= 256 VGPRs and 75 KiB of
    # LDS per workgroup, two workgroups per CU, so one tile's non-GEMM phases run under the other's MFMAs); f32 the
    # chunked round-1 kernel.  `half=False` with f16x3 asked for the chunked f16x3 kernel of rounds 1-4: retired.
    if half is False and self.prec == nat.PREC_F16X3:
      raise ValueError("half=False: the chunked f16x3 kernels were retired in round 5 (f16x3 runs the half-N kernels)")
    self.half = self.prec in (nat.PREC_F16X3, nat.PREC_BF16)
    self.scratch = None
    # f16x3 half-N kernels: the device word the launches fed by EXTERNAL rows set when a value exceeds the exact
    # range of the split halves (include/gcast.h: gc_rowmlp_desc.range_flag); read by check_range()
    self.range_flag = (torch.zeros((1,), dtype=torch.int32, device=self.dev)
                       if self.half and self.prec == nat.PREC_F16X3 else None)
    # chained Linear layers + in-place grid input (only the half-N kernels have them); GCAST_FUSE=0
    # keeps one launch per reference layer group for A/B runs
    # GCAST_HELPERS_MIN_ROWS=<n>: launches without gather / segment-sum from n rows on run in the helper-wave form
    # (0 = never; DESIGN.md section 9.7: the grid-sized node launches are 3-5 % faster in it, profiles/r04_s7_*)
    self.helpers_min_rows = int(os.environ.get("GCAST_HELPERS_MIN_ROWS", DEFAULT_HELPERS_MIN_ROWS))
    self.onepass = os.environ.get("GCAST_ONEPASS", "1") == "1"     # (0: the two-pass launches everywhere, for A/B runs)
    self.fuse = self.half and (os.environ.get("GCAST_FUSE", "1") == "1" or self.prec == nat.PREC_BF16)
    self.n_grid, self.n_mesh = int(graphs["n_grid"]), int(graphs["n_mesh"])
    # Spatially partitioned graphs (partition.plan): node tables that edges GATHER from carry a
    # halo suffix of remote sender rows behind the owned rows; kernels run over the owned prefix
    # and a halo exchange (see `segments`) fills the suffix.  Unpartitioned: no suffix.
    self.ng_tab = int(graphs.get("n_grid_senders", self.n_grid))
    self.nm_tab = max(int(graphs.get("n_mesh_senders", self.n_mesh)),
                      int(graphs.get("n_mesh_senders_dec", self.n_mesh)))
    self.c_in, self.c_out, self.num_steps = c_in, c_out, num_steps
    self.n_struct = graphs["grid_node_feat"].shape[1]
    self.kp = packing.round_up(c_in + self.n_struct, packing.K_CHUNK)
    if c_out > 240:
      raise NotImplementedError("decoder width above 240 needs a wider output tile")
    self._stream = None
    self._keep = []            # keeps every tensor referenced by raw pointer alive
    if self.prec == nat.PREC_BF16:
      self._build_bf16(graphs, params)
    else:
      self._build(graphs, params, fold_only)
    self._programs: Dict[int, tuple] = {}
    self._cuts: Dict[int, list] = {}

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

  def _halo_ops(self, name, tag, desc, agg, agg2):
    """The second launch of a split edge update (`desc`: its descriptor, aggregating into `agg2`) + its fix-ups
    + the join of its aggregate rows into `agg`."""
    h = self.halo[name]
    ops = [self._op_mlp(tag, desc)] + self._ops_after_segsum(h.e, agg2, zero=False)
    op = nat.Op()
    op.kind, op.tag, op.n = nat.OP_ADD, TAGS["fixup"], h.rows.numel()
    op.i0, op.src, op.dst = nat.ptr(h.rows), nat.ptr(agg2), nat.ptr(agg)
    return ops + [op]

  def _run(self, ops):
    arr = (nat.Op * len(ops))(*ops)
    with torch.cuda.device(self.dev):
      if self.tile_queue is not None:     # (see _clear_tile_queue; DeepGNN / ConditionedEncoderDecoder run through here)
        self.tile_queue.zero_()
      nat.check(self.lib.gc_run_program(arr, len(ops), self._stream_ptr()), "gc_run_program")

  def _mlp_ln(self, n_rows, mlp: _Mlp, **kw):
    return self._desc(nat.MODE_MLP_LN, n_rows, w2p=mlp.w2, b2=mlp.b2, n2=D,
                      ln=(mlp.scale, mlp.offset), w2_natural=getattr(mlp, "w2_natural", None), **kw)

  # ---------------------------------------------------------------- build
  def _build(self, graphs, params, fold_only=False):
    dev = self.dev
    G = "grid2mesh_gnn/~_networks_builder/"
    M = "mesh_gnn/~_networks_builder/"
    X = "mesh2grid_gnn/~_networks_builder/"
    esr = ("e", "s", "r")
    self.m_enc_grid = _Mlp(params, G + "encoder_nodes_grid_nodes", dev, prec=self.prec)
    m_enc_mesh = _Mlp(params, G + "encoder_nodes_mesh_nodes", dev, prec=self.prec)
    m_enc_e_g2m = _Mlp(params, G + "encoder_edges_grid2mesh", dev, prec=self.prec)
    self.m_g2m_edge = _Mlp(params, G + "processor_edges_0_grid2mesh", dev, split=esr, prec=self.prec)
    self.m_g2m_mesh = _Mlp(params, G + "processor_nodes_0_mesh_nodes", dev, split=("h", "a"), prec=self.prec)
    self.m_g2m_grid = _Mlp(params, G + "processor_nodes_0_grid_nodes", dev, prec=self.prec)
    m_enc_e_mesh = _Mlp(params, M + "encoder_edges_mesh", dev, prec=self.prec)
    self.m_proc_edge = [_Mlp(params, M + f"processor_edges_{i}_mesh", dev, split=esr, prec=self.prec)
                        for i in range(self.num_steps)]
    self.m_proc_node = [_Mlp(params, M + f"processor_nodes_{i}_mesh_nodes", dev, prec=self.prec)
                        for i in range(self.num_steps)]
    m_enc_e_m2g = _Mlp(params, X + "encoder_edges_mesh2grid", dev, prec=self.prec)
    self.m_m2g_edge = _Mlp(params, X + "processor_edges_0_mesh2grid", dev, split=esr, prec=self.prec)
    self.m_m2g_grid = _Mlp(params, X + "processor_nodes_0_grid_nodes", dev, prec=self.prec)
    self.m_out = _Mlp(params, X + "decoder_nodes_grid_nodes", dev, np2=256, prec=self.prec)
    if self.m_out.n_out != self.c_out:
      raise ValueError(f"decoder produces {self.m_out.n_out} channels, task needs {self.c_out}")
    if self.m_enc_grid.k_in != self.c_in + self.n_struct:
      raise ValueError(f"grid embedder expects {self.m_enc_grid.k_in} input channels, "
                       f"got {self.c_in} + {self.n_struct} structural")
    self._keep += [self.m_enc_grid, self.m_g2m_edge, self.m_g2m_mesh, self.m_g2m_grid,
                   self.m_proc_edge, self.m_proc_node, self.m_m2g_edge, self.m_m2g_grid, self.m_out]

    self.grid_struct = self._up(graphs["grid_node_feat"])

    # ---- load-time constant folding, on the device --------------------------
    nm, ng = self.n_mesh, self.n_grid

    def edge_feat_rows(edges: _Edges, feat):
      rows = np.zeros((edges.n_rows, packing.K_CHUNK), dtype=np.float32)
      ok = edges.pk.perm >= 0
      rows[ok, :feat.shape[1]] = np.asarray(feat)[edges.pk.perm[ok]].astype(np.float32)
      return torch.from_numpy(rows).to(dev)

    def embed(mlp: _Mlp, rows_in):
      out = torch.empty((rows_in.shape[0], D), dtype=torch.float32, device=dev)
      self._run([self._op_mlp("enc_pre", self._mlp_ln(
          rows_in.shape[0], mlp, a0=rows_in, k0=rows_in.shape[1], w1p=mlp.w1, b1=mlp.b1, out=out))])
      return out

    def linear(rows_in, wp, n_rows=None, **kw):
      n_rows = rows_in.shape[0] if n_rows is None else n_rows
      out = torch.empty((n_rows, D), dtype=torch.float32, device=dev)
      self._run([self._op_mlp("enc_pre", self._desc(
          nat.MODE_LINEAR, n_rows, a0=rows_in, k0=D, w1p=wp, out=out, **kw))])
      return out

    # Spatially partitioned graphs: an edge set whose sender table carries a halo suffix is run as TWO
    # launches -- the edges whose sender row this rank owns (the halo exchange runs under that launch) and
    # the edges whose sender row arrives with the exchange; the second launch's aggregate rows are added
    # to the first's (gc_add_rows).  OPT-IN (GCAST_OVERLAP=1): measured on the 0.25 deg graphs at 8-way
    # (profiles/r03_s9_*), the 18 extra small launches + joins cost 1.6 ms per rank -- about what 18 exchanges of
    # 0.86 MB cost over xGMI -- so the default stays one launch behind a blocking exchange.
    overlap = os.environ.get("GCAST_OVERLAP", "0") == "1"

    def split(g, n_owned):
      snd = np.asarray(g["senders"])
      halo = snd >= n_owned
      if not overlap or not halo.any() or halo.all():
        return g, None
      pick = lambda m: dict(senders=snd[m], receivers=np.asarray(g["receivers"])[m], feat=np.asarray(g["feat"])[m])
      return pick(~halo), pick(halo)

    def recv_rows(edges: _Edges):
      r = np.unique(edges.pk.receivers[edges.pk.receivers >= 0]).astype(np.int32)
      return torch.from_numpy(r).to(dev)

    mesh_in = np.zeros((nm, self.kp), dtype=np.float32)
    mesh_in[:, self.c_in:self.c_in + self.n_struct] = graphs["mesh_node_feat"]
    self.h_mesh0 = embed(m_enc_mesh, torch.from_numpy(mesh_in).to(dev))            # [N_m, 512]
    pre_r = linear(self.h_mesh0, self.m_g2m_edge.w1["r"])

    def fold_g2m(gd):       # grid2mesh edge first layer: e0.We + b1 + (h_mesh0.Wr)[receivers]   (per packed edge)
      e = _Edges(packing.pack_edges(gd["senders"], gd["receivers"], nm), dev)
      e0 = embed(m_enc_e_g2m, edge_feat_rows(e, gd["feat"]))
      return e, linear(e0, self.m_g2m_edge.w1["e"], b1=self.m_g2m_edge.b1, g1=pre_r, idx1=e.rcv)

    def fold_mesh(gd):      # multi-mesh: embedded edges (kept: residual of step 0) and step-0 first-layer edge term
      e = _Edges(packing.pack_edges(gd["senders"], gd["receivers"], nm), dev)
      e0 = embed(m_enc_e_mesh, edge_feat_rows(e, gd["feat"]))
      return e, e0, linear(e0, self.m_proc_edge[0].w1["e"], b1=self.m_proc_edge[0].b1)

    def fold_m2g(gd):       # mesh2grid edge first layer: e0.We + b1
      e = _Edges(packing.pack_edges(gd["senders"], gd["receivers"], ng), dev)
      e0 = embed(m_enc_e_m2g, edge_feat_rows(e, gd["feat"]))
      return e, linear(e0, self.m_m2g_edge.w1["e"], b1=self.m_m2g_edge.b1)

    (g_g2m, h_g2m), (g_mesh, h_mesh), (g_m2g, h_m2g) = (split(graphs["g2m"], ng), split(graphs["mesh"], nm),
                                                        split(graphs["m2g"], nm))
    self.e_g2m, self.d_g2m = fold_g2m(g_g2m)
    # encoder mesh-node update first layer: h_mesh0.Wh + b1
    self.d_enc_mesh = linear(self.h_mesh0, self.m_g2m_mesh.w1["h"], b1=self.m_g2m_mesh.b1)
    self.e_mesh, self.e_mesh0, self.d_mesh0 = fold_mesh(g_mesh)
    self.e_m2g, self.d_m2g = fold_m2g(g_m2g)
    self.halo = dict(g2m=None, mesh=None, m2g=None)
    if h_g2m is not None:
      e, d = fold_g2m(h_g2m)
      self.halo["g2m"] = _HaloPart(e=e, d=d, rows=recv_rows(e))
    if h_mesh is not None:
      e, e0, d0 = fold_mesh(h_mesh)
      self.halo["mesh"] = _HaloPart(e=e, e0=e0, d=d0, rows=recv_rows(e))
    if h_m2g is not None:
      e, d = fold_m2g(h_m2g)
      self.halo["m2g"] = _HaloPart(e=e, d=d, rows=recv_rows(e))
    del pre_r
    torch.cuda.synchronize(dev)
    self._keep += [self.h_mesh0, self.d_g2m, self.d_enc_mesh, self.e_mesh0, self.d_mesh0, self.d_m2g,
                   self.e_g2m, self.e_mesh, self.e_m2g, self.halo]

    if fold_only:             # (Bf16StepEngine's helper: only the folded constants and packed edges are wanted)
      return
    # ---- per-step workspace ---------------------------------------------------
    self.xin = self._new(ng, self.kp)
    self.h_grid = self._new(ng)         # embedded grid latents, later reused for the decoder update
    self.pre_grid = self._new(self.ng_tab)   # h_grid.Ws (encoder; + halo rows) / h_grid2.Wr (decoder)
    self.h_grid2 = self._new(ng)        # grid latents after the encoder's node update
    self.agg_grid = self._new(ng)
    self.h_mesh = self._new(nm)
    self.agg_mesh = self._new(nm)
    self.pre_s_mesh = self._new(self.nm_tab)   # h_mesh.Ws (+ halo rows of remote senders)
    self.pre_r_mesh = self._new(nm)
    self.e_mesh_lat = self._new(self.e_mesh.n_rows)
    if self.halo["mesh"] is not None:
      self.halo["mesh"].lat = self._new(self.halo["mesh"].e.n_rows)
    if self.halo["g2m"] is not None or self.halo["mesh"] is not None:
      self.agg_mesh2 = self._new(nm)       # the halo-sender edges' aggregate rows, joined by gc_add_rows
    if self.halo["m2g"] is not None:
      self.agg_grid2 = self._new(ng)

  def _build_bf16(self, graphs, params):
    """GC_PREC_BF16 (the reference's Bfloat16Cast run, casting.py:31-65): bfloat16 weights and
    bfloat16 row tensors in pi order (include/gcast.h).  The input-independent terms are folded by an
    fp32-grade helper engine (same kernels as the default step) and rounded ONCE to bfloat16 -- the
    reference recomputes them in bfloat16 every step; ours are the more accurate constants."""
    dev = self.dev
    G = "grid2mesh_gnn/~_networks_builder/"
    M = "mesh_gnn/~_networks_builder/"
    X = "mesh2grid_gnn/~_networks_builder/"
    esr = ("e", "s", "r")
    P = dict(prec=self.prec)
    self.m_enc_grid = _Mlp(params, G + "encoder_nodes_grid_nodes", dev, k_natural=True, **P)
    self.m_g2m_edge = _Mlp(params, G + "processor_edges_0_grid2mesh", dev, split=esr, **P)
    self.m_g2m_mesh = _Mlp(params, G + "processor_nodes_0_mesh_nodes", dev, split=("h", "a"), **P)
    self.m_g2m_grid = _Mlp(params, G + "processor_nodes_0_grid_nodes", dev, **P)
    self.m_proc_edge = [_Mlp(params, M + f"processor_edges_{i}_mesh", dev, split=esr, **P) for i in range(self.num_steps)]
    self.m_proc_node = [_Mlp(params, M + f"processor_nodes_{i}_mesh_nodes", dev, **P) for i in range(self.num_steps)]
    self.m_m2g_edge = _Mlp(params, X + "processor_edges_0_mesh2grid", dev, split=esr, **P)
    self.m_m2g_grid = _Mlp(params, X + "processor_nodes_0_grid_nodes", dev, **P)
    self.m_out = _Mlp(params, X + "decoder_nodes_grid_nodes", dev, np2=256, **P)
    if self.m_out.n_out != self.c_out:
      raise ValueError(f"decoder produces {self.m_out.n_out} channels, task needs {self.c_out}")
    if self.m_enc_grid.k_in != self.c_in + self.n_struct:
      raise ValueError(f"grid embedder expects {self.m_enc_grid.k_in} input channels, "
                       f"got {self.c_in} + {self.n_struct} structural")
    self._keep += [self.m_enc_grid, self.m_g2m_edge, self.m_g2m_mesh, self.m_g2m_grid,
                   self.m_proc_edge, self.m_proc_node, self.m_m2g_edge, self.m_m2g_grid, self.m_out]
    base = StepEngine(graphs, params, num_steps=self.num_steps, c_in=self.c_in, c_out=self.c_out,
                      device=dev, precision="f16x3", half=True, fold_only=True)
    pi = torch.from_numpy(packing.PI_PERM).to(dev)
    to_bf = lambda t: t.index_select(1, pi).to(torch.bfloat16).contiguous()
    if any(v is not None for v in base.halo.values()):
      raise NotImplementedError("the GC_PREC_BF16 tier runs partitioned graphs behind blocking exchanges only "
                                "(GCAST_OVERLAP=1 splits edge updates and joins fp32 aggregate rows)")
    self.halo = dict(g2m=None, mesh=None, m2g=None)
    self.e_g2m, self.e_mesh, self.e_m2g = base.e_g2m, base.e_mesh, base.e_m2g
    self.grid_struct = base.grid_struct
    self.h_mesh0, self.d_g2m, self.d_enc_mesh = to_bf(base.h_mesh0), to_bf(base.d_g2m), to_bf(base.d_enc_mesh)
    self.e_mesh0, self.d_mesh0, self.d_m2g = to_bf(base.e_mesh0), to_bf(base.d_mesh0), to_bf(base.d_m2g)
    del base
    torch.cuda.synchronize(dev)
    torch.cuda.empty_cache()
    self._keep += [self.h_mesh0, self.d_g2m, self.d_enc_mesh, self.e_mesh0, self.d_mesh0, self.d_m2g,
                   self.e_g2m, self.e_mesh, self.e_m2g, self.grid_struct]
    nm, ng = self.n_mesh, self.n_grid
    new = lambda rows: self._new_bf16(rows)
    self.xin = self._new(ng, self.kp)            # fp32: the external rows' 32-column tail (gc_prep_grid_tail)
    self.h_grid, self.pre_grid, self.h_grid2, self.agg_grid = new(ng), new(self.ng_tab), new(ng), new(ng)
    self.h_mesh, self.agg_mesh = new(nm), new(nm)
    self.pre_s_mesh, self.pre_r_mesh = new(self.nm_tab), new(nm)
    self.e_mesh_lat = new(self.e_mesh.n_rows)

  def _new_bf16(self, rows):
    t = torch.empty((rows, D), dtype=torch.bfloat16, device=self.dev)
    self._keep.append(t)
    return t

  # ---------------------------------------------------------------- program
  def _program(self, batch):
    """Op list for all batch elements; x / y pointers are patched per call."""
    if batch in self._programs:
      return self._programs[batch]
    build = self._program_fused if self.fuse else self._program_plain
    ops, x_slots, y_slots, cuts = build(batch)
    arr = (nat.Op * len(ops))(*ops)
    self._programs[batch] = (arr, x_slots, y_slots)
    self._cuts[batch] = cuts
    return self._programs[batch]

  def _program_plain(self, batch):
    """One launch per reference layer group (every formulation / precision)."""
    ng, nm = self.n_grid, self.n_mesh
    ops, x_slots, y_slots = [], [], []
    cuts = []          # (op index, table, "start" | "wait"): see segments()
    for b in range(batch):
      op = nat.Op()
      op.kind, op.tag, op.n = nat.OP_PREP, TAGS["prep"], ng
      op.batch, op.b, op.c_in, op.n_struct, op.kp = batch, b, self.c_in, self.n_struct, self.kp
      op.node_struct, op.dst = nat.ptr(self.grid_struct), nat.ptr(self.xin)
      x_slots.append((len(ops), "prep"))
      ops.append(op)
      # ---- encoder (grid2mesh GNN) ----
      m = self.m_enc_grid
      ops.append(self._op_mlp("enc_embed_grid", self._mlp_ln(
          ng, m, a0=self.xin, k0=self.kp, w1p=m.w1, b1=m.b1, out=self.h_grid, check_range=True)))
      m = self.m_g2m_edge
      ops.append(self._op_mlp("enc_pre", self._desc(
          nat.MODE_LINEAR, ng, a0=self.h_grid, k0=D, w1p=m.w1["s"], out=self.pre_grid)))
      self._enc_edge_site(ops, cuts)
      m = self.m_g2m_mesh
      ops.append(self._op_mlp("enc_node_mesh", self._mlp_ln(
          nm, m, a0=self.agg_mesh, k0=D, w1p=m.w1["a"], d=self.d_enc_mesh, res=self.h_mesh0,
          out=self.h_mesh, check_range=True)))
      m = self.m_g2m_grid
      ops.append(self._op_mlp("enc_node_grid", self._mlp_ln(
          ng, m, a0=self.h_grid, k0=D, w1p=m.w1, b1=m.b1, res=self.h_grid, out=self.h_grid2)))
      # ---- processor (multi-mesh GNN) ----
      for i in range(self.num_steps):
        me, mn = self.m_proc_edge[i], self.m_proc_node[i]
        last = i == self.num_steps - 1
        ops.append(self._op_mlp("proc_pre", self._desc(
            nat.MODE_LINEAR, nm, a0=self.h_mesh, k0=D, w1p=me.w1["s"], out=self.pre_s_mesh)))
        ops.append(self._op_mlp("proc_pre", self._desc(
            nat.MODE_LINEAR, nm, a0=self.h_mesh, k0=D, w1p=me.w1["r"], out=self.pre_r_mesh)))
        self._proc_edge_site(ops, cuts, i)
        ops.append(self._op_mlp("proc_node", self._mlp_ln(
            nm, mn, a0=self.h_mesh, k0=D, a1=self.agg_mesh, k1=D, w1p=mn.w1, b1=mn.b1,
            res=self.h_mesh, out=self.h_mesh, check_range=True)))
      # ---- decoder (mesh2grid GNN) ----
      m = self.m_m2g_edge
      ops.append(self._op_mlp("dec_pre", self._desc(
          nat.MODE_LINEAR, nm, a0=self.h_mesh, k0=D, w1p=m.w1["s"], out=self.pre_s_mesh)))
      ops.append(self._op_mlp("dec_pre", self._desc(
          nat.MODE_LINEAR, ng, a0=self.h_grid2, k0=D, w1p=m.w1["r"], out=self.pre_grid)))
      self._dec_edge_site(ops, cuts)
      m = self.m_m2g_grid
      ops.append(self._op_mlp("dec_node", self._mlp_ln(
          ng, m, a0=self.h_grid2, k0=D, a1=self.agg_grid, k1=D, w1p=m.w1, b1=m.b1,
          res=self.h_grid2, out=self.h_grid, check_range=True)))
      m = self.m_out
      y_slots.append((len(ops), "out"))
      ops.append(self._op_mlp("dec_out", self._desc(
          nat.MODE_MLP_OUT, ng, a0=self.h_grid, k0=D, w1p=m.w1, b1=m.b1, w2p=m.w2, b2=m.b2,
          n2=self.c_out, out_ptr=0, ldo=batch * self.c_out)))
    return ops, x_slots, y_slots, cuts

  def _proc_edge_desc(self, i, part=None):
    """Processor edge update of step i over the sender-local edges (part None) or the halo-sender ones."""
    me = self.m_proc_edge[i]
    last = i == self.num_steps - 1
    e, e0, d0, lat, agg = ((self.e_mesh, self.e_mesh0, self.d_mesh0, self.e_mesh_lat, self.agg_mesh) if part is None
                           else (part.e, part.e0, part.d, part.lat, self.agg_mesh2))
    common = dict(g0=self.pre_s_mesh, idx0=e.snd, g1=self.pre_r_mesh, idx1=e.rcv, edges=e, agg=agg)
    if i == 0:
      desc = self._mlp_ln(e.n_rows, me, d=d0, res=e0, out=None if last else lat, **common)
      if last:
        desc.res, desc.ldres = None, 0
      return desc
    return self._mlp_ln(e.n_rows, me, a0=lat, k0=D, w1p=me.w1["e"], b1=me.b1, res=None if last else lat,
                        out=None if last else lat, **common)

  def _edge_site(self, ops, cuts, name, tag, main_desc, halo_desc, main_edges, agg, agg2):
    """One edge update of the step: the launch over the sender-local edges (+ fix-ups), and -- partitioned graphs --
    the launch over the halo-sender edges behind the exchange of table `name`.  `cuts` records where the exchange may
    START (the producing launch is enqueued) and where it must have FINISHED."""
    h = self.halo[name]
    cuts.append((len(ops), name, "start"))
    if h is None:
      cuts.append((len(ops), name, "wait"))
    ops.append(self._op_mlp(tag, main_desc()))
    ops += self._ops_after_segsum(main_edges, agg)
    if h is not None:
      cuts.append((len(ops), name, "wait"))
      ops += self._halo_ops(name, tag, halo_desc(h), agg, agg2)

  def _enc_edge_site(self, ops, cuts):
    m = self.m_g2m_edge
    self._edge_site(
        ops, cuts, "g2m", "enc_edge",
        lambda: self._mlp_ln(self.e_g2m.n_rows, m, d=self.d_g2m, g0=self.pre_grid, idx0=self.e_g2m.snd,
                             edges=self.e_g2m, agg=self.agg_mesh),
        lambda h: self._mlp_ln(h.e.n_rows, m, d=h.d, g0=self.pre_grid, idx0=h.e.snd, edges=h.e, agg=self.agg_mesh2),
        self.e_g2m, self.agg_mesh, getattr(self, "agg_mesh2", None))

  def _proc_edge_site(self, ops, cuts, i):
    self._edge_site(ops, cuts, "mesh", "proc_edge", lambda: self._proc_edge_desc(i),
                    lambda h: self._proc_edge_desc(i, h), self.e_mesh, self.agg_mesh, getattr(self, "agg_mesh2", None))

  def _dec_edge_site(self, ops, cuts):
    m = self.m_m2g_edge
    self._edge_site(
        ops, cuts, "m2g", "dec_edge",
        lambda: self._mlp_ln(self.e_m2g.n_rows, m, d=self.d_m2g, g0=self.pre_s_mesh, idx0=self.e_m2g.snd,
                             g1=self.pre_grid, idx1=self.e_m2g.rcv, edges=self.e_m2g, agg=self.agg_grid),
        lambda h: self._mlp_ln(h.e.n_rows, m, d=h.d, g0=self.pre_s_mesh, idx0=h.e.snd, g1=self.pre_grid,
                               idx1=h.e.rcv, edges=h.e, agg=self.agg_grid2),
        self.e_m2g, self.agg_grid, getattr(self, "agg_grid2", None))

  def _program_fused(self, batch):
    """GC_LAYOUT_HALF: the Linear layers that the reference applies to rows a launch has just
    produced are CHAINED onto that launch (gc_chain_stage: the rows are still in registers) --
      grid embedder      -> (h.W_s)            the encoder edge update's sender product
                                               (graphcast.py:561-598 + typed_graph_net.py:431-453)
      encoder grid nodes -> (h'.W_r)           the decoder edge update's receiver product (:641-678)
      encoder mesh nodes / processor node update i -> (h.W_s, h.W_r) of edge update i + 1
                                               (the last one: the decoder's sender product)
      decoder grid nodes -> swish(h.W1 + b1) -> .W2 + b2   the output MLP (deep_typed_graph_net.py:313-322)
    -- 36 launches, their row re-reads and the [N_grid, 512] write + read of the decoder latents
    disappear; the grid input is read in place (x[:, b, :448] as the first K chunks, a 32-column
    tail [x[:, b, 448:] | struct | 0] built by gc_prep_grid_input) instead of being copied."""
    ng, nm = self.n_grid, self.n_mesh
    ops, x_slots, y_slots = [], [], []
    cuts = []
    R, S, N = nat.CHAIN_ROWS, nat.CHAIN_SWISH, nat.CHAIN_NARROW
    rows = lambda mlp, blk, out: dict(w=_chained(mlp, blk), kind=R, out=out)
    k_full = (self.c_in // packing.K_CHUNK) * packing.K_CHUNK          # x columns read in place
    kt = self.kp - k_full                                              # tail: rest of x | struct | 0
    for b in range(batch):
      op = nat.Op()
      op.kind, op.tag, op.n = nat.OP_PREP, TAGS["prep"], ng
      op.batch, op.b, op.c_in, op.n_struct, op.kp = batch, b, self.c_in, self.n_struct, kt
      op.c0 = k_full
      op.node_struct, op.dst = nat.ptr(self.grid_struct), nat.ptr(self.xin)
      x_slots.append((len(ops), "prep"))
      ops.append(op)
      # ---- encoder (grid2mesh GNN) ----
      m = self.m_enc_grid
      if k_full > 0:
        x_slots.append((len(ops), "a0", b))
        src = dict(a0=self.xin, k0=k_full, lda0=batch * self.c_in, a1=self.xin, k1=kt, lda1=kt, rows_f32=True)
      else:                # fewer than 32 input channels: the tail [x | struct | 0] is the whole input
        src = dict(a0=self.xin, k0=kt, lda0=kt, rows_f32=True)
      src["check_range"] = True        # fed by external rows; the three node updates below read AGGREGATES (check_range())
      ops.append(self._op_mlp("enc_embed_grid", self._mlp_ln(
          ng, m, w1p=m.w1, b1=m.b1, out=self.h_grid, chain=[rows(self.m_g2m_edge, "s", self.pre_grid)],
          **src)))
      m = self.m_g2m_edge
      self._enc_edge_site(ops, cuts)
      m = self.m_g2m_mesh
      first = self.m_proc_edge[0]
      ops.append(self._op_mlp("enc_node_mesh", self._mlp_ln(
          nm, m, a0=self.agg_mesh, k0=D, w1p=m.w1["a"], d=self.d_enc_mesh, res=self.h_mesh0,
          out=self.h_mesh, chain=[rows(first, "s", self.pre_s_mesh), rows(first, "r", self.pre_r_mesh)],
          check_range=True)))     # (an AGGREGATE as layer-1 operand: a sum over up to 3,753 edges, not a LayerNorm output)
      m = self.m_g2m_grid
      ops.append(self._op_mlp("enc_node_grid", self._mlp_ln(
          ng, m, a0=self.h_grid, k0=D, w1p=m.w1, b1=m.b1, res=self.h_grid, out=self.h_grid2,
          chain=[rows(self.m_m2g_edge, "r", self.pre_grid)])))
      # (h'.W_r overwrites pre_grid: the encoder edge update that gathered h.W_s from it has run)
      # ---- processor (multi-mesh GNN) ----
      for i in range(self.num_steps):
        mn = self.m_proc_node[i]
        last = i == self.num_steps - 1
        self._proc_edge_site(ops, cuts, i)
        if last:
          chain = [rows(self.m_m2g_edge, "s", self.pre_s_mesh)]
        else:
          nxt = self.m_proc_edge[i + 1]
          chain = [rows(nxt, "s", self.pre_s_mesh), rows(nxt, "r", self.pre_r_mesh)]
        ops.append(self._op_mlp("proc_node", self._mlp_ln(
            nm, mn, a0=self.h_mesh, k0=D, a1=self.agg_mesh, k1=D, w1p=mn.w1, b1=mn.b1,
            res=self.h_mesh, out=self.h_mesh, chain=chain, check_range=True)))
      # ---- decoder (mesh2grid GNN) ----
      m = self.m_m2g_edge
      self._dec_edge_site(ops, cuts)
      m, mo = self.m_m2g_grid, self.m_out
      y_slots.append((len(ops), "chain", 1))
      ops.append(self._op_mlp("dec_node", self._mlp_ln(
          ng, m, a0=self.h_grid2, k0=D, a1=self.agg_grid, k1=D, w1p=m.w1, b1=m.b1,
          res=self.h_grid2, out=None, check_range=True,
          chain=[dict(w=_chained(mo), kind=S, b=mo.b1),
                 dict(w=mo.w2, kind=N, b=mo.b2, out_ptr=0, ldo=batch * self.c_out, n=self.c_out)])))
    return ops, x_slots, y_slots, cuts

  def bind(self, x: torch.Tensor, y: Optional[torch.Tensor] = None):
    """Validates x/y and returns (program array, y) with the x/y pointers patched in."""
    if x.dtype != torch.float32 or x.dim() != 3 or not x.is_contiguous() or x.device != self.dev:
      raise ValueError("x must be a contiguous float32 [N_grid, B, C_in] tensor on the engine's device")
    if x.shape[0] != self.n_grid or x.shape[2] != self.c_in:
      raise ValueError(f"x has shape {tuple(x.shape)}, expected [{self.n_grid}, B, {self.c_in}]")
    batch = x.shape[1]
    if y is None:
      y = torch.empty((self.n_grid, batch, self.c_out), dtype=torch.float32, device=self.dev)
    elif (y.shape != (self.n_grid, batch, self.c_out) or y.dtype != torch.float32
          or not y.is_contiguous() or y.device != self.dev):
      raise ValueError("y must be a contiguous float32 [N_grid, B, C_out] tensor on the engine's device")
    arr, x_slots, y_slots = self._program(batch)
    for slot in x_slots:
      if slot[1] == "prep":
        arr[slot[0]].x = x.data_ptr()
      else:                                      # the embed launch reads x[:, b, :k_full] in place
        arr[slot[0]].mlp.a0 = x.data_ptr() + 4 * slot[2] * self.c_in
    for b, slot in enumerate(y_slots):
      ptr = y.data_ptr() + 4 * b * self.c_out
      if slot[1] == "out":
        arr[slot[0]].mlp.out = ptr
      else:
        arr[slot[0]].mlp.chain[slot[2]].out = ptr
    return arr, y

  def _clear_tile_queue(self):
    """The two tile-queue words are left zero by every launch that COMPLETES; an aborted launch, or a caller who
    overlapped two launches of one engine on different streams, would leave them non-zero and every later launch
    would silently skip tiles.  One 8-byte memset at the head of each enqueued program (as gc_step_forward does)."""
    if self.tile_queue is not None:
      self.tile_queue.zero_()

  def forward(self, x: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    arr, y = self.bind(x, y)
    with torch.cuda.device(self.dev):       # launches go to the engine's device whatever is current
      self._clear_tile_queue()
      nat.check(self.lib.gc_run_program(arr, len(arr), self._stream_ptr()), "gc_run_program")
    return y

  __call__ = forward

  _range_pending = None

  def check_range(self, wait: bool = True):
    """Raises GcastRangeError if a step since the last call read an input value, or an AGGREGATE (the layer-1 operand
    of the encoder's mesh-node update, the processor's node updates and the decoder's grid-node update: a sum over
    up to 3,753 edges, which the reference up-casts to fp32 for this reason, graphcast.py:215), outside the exact range of the
    f16x3 arithmetic (|x| > 65504: the split halves saturate -- 5e-4 errors up to 1.3e5, garbage beyond -- where
    the reference's fp32 does not care; un-normalised geopotential is ~5e5).  SYNCHRONISES the launch stream:
    call it where the host waits for the step anyway (GraphCast.__call__ on host Datasets, DeviceRollout.run,
    bench.py do); ``wait=False`` never blocks (a device-resident Dataset rollout must not wait on the host
    once per step)."""
    if self.range_flag is None:
      return
    if not wait:
      # device-resident callers (torch-backed Datasets: nothing else makes the host wait): the word is copied to pinned
      # memory behind the step and tested at the NEXT call -- the launches never clear it, so an out-of-range step is
      # reported one call late at worst (and at the latest by the first blocking check: to_host, DeviceRollout.run)
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
    elif int(self.range_flag.item()) == 0:
      return
    self._range_pending = None
    self.range_flag.zero_()
    raise nat.GcastRangeError(
        f"an input value -- or a per-receiver sum of edge messages (a node update's aggregate operand) -- exceeds "
        f"{nat.F16X3_MAX:g} in magnitude: outside the exact range of the f16x3 arithmetic (precision='f16x3').  "
        "Normalise the inputs (normalization.InputsAndResiduals, as the reference's demo stack does) or run with "
        "precision='f32'.")

  def run_until(self, x: torch.Tensor, tag: str, y: Optional[torch.Tensor] = None):
    """Enqueues the step's launches up to (not including) the first launch tagged `tag` (batch
    element 0) and returns how many ran: the workspace then holds that stage boundary -- e.g.
    `run_until(x, "enc_node_mesh")` leaves the encoder's grid2mesh aggregate in `agg_mesh`.
    Verification hook (tests compare stage boundaries with the oracle); not on the product path."""
    arr, _ = self.bind(x, y)
    n = next(k for k in range(len(arr)) if arr[k].tag == TAGS[tag])
    self._clear_tile_queue()
    nat.check(self.lib.gc_run_program(arr, n, self._stream_ptr()), "gc_run_program")
    return n

  # ---------------------------------------------------------------- partitioned execution
  def segments(self, x: torch.Tensor, y: Optional[torch.Tensor] = None):
    """The step as launch segments separated by halo exchange points (partition.py).

    Returns ``(y, [(run, actions), ...])``: call ``run()`` (enqueues the segment's launches on the current
    stream), then perform ``actions`` in order -- ``("start", table)``: the launch that produced the owned
    rows of ``self.halo_table(table)`` is enqueued, the exchange of its halo suffix may begin (on another
    stream, behind an event); ``("wait", table)``: the next segment gathers from that suffix.  With split
    edge updates (``self.halo[table]``) the sender-local launch lies between the two; otherwise they come
    together.  18 exchanges per batch element: 1 encoder, 1 per processor step, 1 decoder."""
    bound, y = self.bind(x, y)
    # a private copy of the program: the closures below stay valid when the engine is bound to
    # other tensors before they have all run (interleaved partitioned steps, time_ops, ...)
    arr = (nat.Op * len(bound))()
    ctypes.memmove(arr, bound, ctypes.sizeof(bound))
    # ONE segment per cut (+ the tail), in program order, EMPTY segments included: every rank of a partitioned
    # step then has the same segment / action structure whatever its own edge sets decided in `split` (a rank
    # whose senders of a table are all local, or all remote, runs that edge update as one launch: its "start"
    # and "wait" cuts coincide and the segment between them is empty) -- partition.py walks the ranks' segment
    # lists in lockstep.
    cuts = self._cuts[x.shape[1]]
    assert all(a[0] <= b[0] for a, b in zip(cuts[:-1], cuts[1:])), "cuts must be recorded in program order"
    self._clear_tile_queue()
    segs = []
    lo = 0
    for hi, actions in [(c, [(kind, name)]) for c, name, kind in cuts] + [(len(arr), [])]:
      sub = ctypes.cast(ctypes.byref(arr, lo * ctypes.sizeof(nat.Op)), ctypes.POINTER(nat.Op))

      def run(sub=sub, n=hi - lo, keep=arr):        # (`keep`: the copy lives as long as the closure)
        if n > 0:
          with torch.cuda.device(self.dev):
            nat.check(self.lib.gc_run_program(sub, n, self._stream_ptr()), "gc_run_program")
      segs.append((run, actions))
      lo = hi
    return y, segs

  def halo_table(self, name: str) -> torch.Tensor:
    """The row table whose halo suffix the exchange `name` fills (owned prefix already valid)."""
    return self.pre_grid if name == "g2m" else self.pre_s_mesh

  def owned_rows(self, name: str) -> int:
    return self.n_grid if name == "g2m" else self.n_mesh

  def time_ops(self, x, iters=3):
    """Per-op mean milliseconds measured with HIP events on the launch stream."""
    arr, _ = self.bind(x)
    self._clear_tile_queue()
    ms = (ctypes.c_float * len(arr))()
    nat.check(self.lib.gc_time_program(arr, len(arr), iters, ms, self._stream_ptr()),
              "gc_time_program")
    return [(arr[k].tag, arr[k].kind, ms[k]) for k in range(len(arr))]
