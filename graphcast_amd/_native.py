"""ctypes binding of libgcast_hip.so (the C-ABI declared in include/gcast.h).

There is deliberately no fallback: if the library is missing or does not load
the import fails loudly -- the product path never computes on the CPU.
"""
import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

MODE_LINEAR, MODE_MLP_LN, MODE_MLP_OUT = 0, 1, 2
OP_ROWMLP, OP_FIXUP, OP_ZERO, OP_PREP, OP_ADD = 0, 1, 2, 3, 4
PREC_F32, PREC_F16X3, PREC_BF16 = 0, 1, 3          # (2: the bf16-GEMM-operand tier of rounds 1-4, retired; still the id of
PREC_BF16_IMAGE = 2                                # the plain bfloat16 weight image in gc_host_pack_weight)
PRECISIONS = {"f32": PREC_F32, "f16x3": PREC_F16X3, "bf16": PREC_BF16}
ROWS_F32 = 1                                      # GC_ROWS_F32 (gc_rowmlp_desc.flags)
W2_NATURAL = 2                                    # GC_W2_NATURAL
WG_ROWS_64, WG_ROWS_128 = 4, 8                    # GC_WG_ROWS_64 / GC_WG_ROWS_128 (GC_PREC_BF16: pin the rows per workgroup)
LAYOUT_CHUNKED, LAYOUT_HALF = 0, 2
LATENT = 512
TILE_MAP_XCD = 16                                 # GC_TILE_XCD
TILE_QUEUE_ANY = 128                              # GC_TILE_QUEUE_ANY: the dynamic tile queue whenever a launch has a second round
WG_HELPERS, WG_NO_HELPERS = 32, 64                # GC_WG_HELPERS / GC_WG_NO_HELPERS (eight-wave form of a GC_LAYOUT_HALF launch)
WG_WIDE = 256                                     # GC_WG_WIDE (eight MULTIPLYING waves per CU on one weight ring; round 6: segment-sum / one-pass launches too)
LATE_ADDENDS = 512                                # GC_LATE_ADDENDS (bf16 tier / the f16x3 wide form: gathered rows added when the hidden layer is formed)
WIDE_EDGES_DEFAULT = 3                            # GC_WIDE_EDGES_DEFAULT (gc_tuning.wide_edges of a process without GCAST_WIDE_EDGES)
TILE_ROWS = 64
K_CHUNK = 32
SCRATCH_SLOTS = 512                               # GC_SCRATCH_SLOTS: persistent workgroups of a GC_LAYOUT_HALF launch
SCRATCH_FLOATS = SCRATCH_SLOTS * TILE_ROWS * 256  # GC_SCRATCH_FLOATS: floats in gc_rowmlp_desc.scratch (32 MiB)

_fp = ctypes.c_void_p     # device pointers travel as integers


class ChainStage(ctypes.Structure):
  """struct gc_chain_stage."""
  _fields_ = [("wp", _fp), ("b", _fp), ("out", _fp), ("ldo", ctypes.c_int), ("n", ctypes.c_int),
              ("kind", ctypes.c_int), ("w_scale", ctypes.c_float)]


MAX_CHAIN = 2
CHAIN_LN, CHAIN_ROWS, CHAIN_SWISH, CHAIN_NARROW = 0, 1, 2, 3


class RowMlpDesc(ctypes.Structure):
  """struct gc_rowmlp_desc (include/gcast.h) -- field order must match exactly."""
  _fields_ = [
      ("mode", ctypes.c_int), ("prec", ctypes.c_int), ("n_rows", ctypes.c_int),
      ("layout", ctypes.c_int), ("w1_scale", ctypes.c_float), ("w2_scale", ctypes.c_float),
      ("a0", _fp), ("lda0", ctypes.c_int), ("k0", ctypes.c_int),
      ("a1", _fp), ("lda1", ctypes.c_int), ("k1", ctypes.c_int),
      ("w1p", _fp),
      ("d", _fp), ("ldd", ctypes.c_int),
      ("g0", _fp), ("idx0", _fp),
      ("g1", _fp), ("idx1", _fp),
      ("b1", _fp),
      ("w2p", _fp), ("b2", _fp), ("n2", ctypes.c_int),
      ("ln_scale", _fp), ("ln_offset", _fp),
      ("res", _fp), ("ldres", ctypes.c_int),
      ("out", _fp), ("ldo", ctypes.c_int),
      ("seg", _fp), ("tile_flags", _fp), ("agg", _fp), ("partial", _fp),
      ("scratch", _fp),
      ("n_chain", ctypes.c_int), ("chain", ChainStage * MAX_CHAIN),
      ("flags", ctypes.c_int),
      ("range_flag", _fp),
      ("tile_queue", _fp),
  ]


class Op(ctypes.Structure):
  """struct gc_op."""
  _fields_ = [
      ("kind", ctypes.c_int), ("tag", ctypes.c_int),
      ("mlp", RowMlpDesc),
      ("n", ctypes.c_int), ("i0", _fp), ("i1", _fp), ("i2", _fp),
      ("src", _fp), ("dst", _fp),
      ("batch", ctypes.c_int), ("b", ctypes.c_int), ("c_in", ctypes.c_int),
      ("n_struct", ctypes.c_int), ("kp", ctypes.c_int),
      ("x", _fp), ("node_struct", _fp),
      ("c0", ctypes.c_int),
  ]


class EdgeSet(ctypes.Structure):
  """struct gc_edge_set (host pointers)."""
  _fields_ = [("n_edges", ctypes.c_int), ("h_senders", _fp), ("h_receivers", _fp), ("h_feat", _fp),
              ("n_feat", ctypes.c_int)]


class ModelDesc(ctypes.Structure):
  """struct gc_model_desc."""
  _fields_ = [("n_grid", ctypes.c_int), ("n_mesh", ctypes.c_int), ("c_in", ctypes.c_int),
              ("c_out", ctypes.c_int), ("n_struct", ctypes.c_int), ("num_steps", ctypes.c_int),
              ("prec", ctypes.c_int), ("h_grid_node_feat", _fp), ("h_mesh_node_feat", _fp),
              ("g2m", EdgeSet), ("mesh", EdgeSet), ("m2g", EdgeSet), ("layout", ctypes.c_int),
              # spatially partitioned graphs: rows of the sender tables incl. their halo suffix (0 = none)
              ("n_grid_senders", ctypes.c_int), ("n_mesh_senders", ctypes.c_int), ("n_mesh_senders_dec", ctypes.c_int)]


class TensorDesc(ctypes.Structure):
  """struct gc_tensor_desc."""
  _fields_ = [("name", ctypes.c_char_p), ("h_data", _fp), ("rows", ctypes.c_int), ("cols", ctypes.c_int)]


class AdvanceDesc(ctypes.Structure):
  """struct gc_advance_desc."""
  _fields_ = [
      ("n_rows", ctypes.c_int), ("c_in", ctypes.c_int), ("c_out", ctypes.c_int),
      ("n_forc", ctypes.c_int),
      ("x", _fp), ("y", _fp), ("f_cur", _fp), ("f_next", _fp),
      ("src_x", _fp), ("ax", _fp), ("src_y", _fp), ("ay", _fp), ("src_f", _fp),
      ("x_next", _fp),
      ("p_src_x", _fp), ("p_ax", _fp), ("p_ay", _fp), ("p_b", _fp),
      ("pred", _fp),
  ]


class Tuning(ctypes.Structure):
  """struct gc_tuning (include/gcast.h): the library's ONE tuning surface -- speed-only A/B switches; the GCAST_*
  environment variables only initialise the process default."""
  _fields_ = [(name, ctypes.c_int) for name in (
      "grid_cap", "tile_map_xcd", "prio_set", "prio_gemm", "prio_other", "prio_stage", "helpers", "helpers_small",
      "helpers_edge", "helper_store", "helpers_min_rows", "wide", "wide_edges", "bf16_rows", "tile_queue", "fuse",
      "onepass", "split_tail", "bf16_stream", "wide_late", "split_edges")] + [("reserved", ctypes.c_int * 4)]

  def as_dict(self):
    return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved"}


EXPORTS = ("gc_get_tuning", "gc_set_tuning", "gc_plan_get_tuning", "gc_tuning_string", "gc_plan_create", "gc_plan_workspace_bytes", "gc_step_forward", "gc_plan_check_range", "gc_plan_destroy",
           "gc_plan_program", "gc_plan_tensor",
           "gc_host_pack_weight", "gc_host_pack_edges", "gc_host_pad_latent", "gc_advance_state", "gc_rowmlp", "gc_seg_fixup", "gc_zero_rows", "gc_seg_fixup_bf16", "gc_zero_rows_bf16", "gc_add_rows", "gc_prep_grid_input", "gc_prep_grid_tail",
           "gc_run_program", "gc_time_program", "gc_abi_sizeof", "gc_last_error", "gc_build_info")


# Build variants of the one source: "main" = the shipped library.  Further entries are A/B builds
# (GCAST_LIB_VARIANT=<name> selects one at load time); only "main" is built by default.
VARIANTS = {"main": ("libgcast_hip.so", "-DGC_PIPE=2")}


def library_path(variant=None):
  """The library to load: the in-tree product build, or -- A/B sessions only -- GCAST_LIB_PATH=<file>, a build of the
  same sources with other -D switches (e.g. ab_libs/libgcast_noweave.so = -DGC_H_WEAVE=0)."""
  if variant is None and os.environ.get("GCAST_LIB_PATH"):
    path = os.environ["GCAST_LIB_PATH"]
    return path if os.path.isabs(path) else os.path.join(os.path.dirname(_HERE), path)
  variant = variant or os.environ.get("GCAST_LIB_VARIANT", "main")
  return os.path.join(_CSRC, VARIANTS[variant][0])


# Register budget of the kernels that are built to run TWO workgroups per CU (csrc/rowmlp_half.inc,
# rowmlp_bf16.inc): hipcc's -Rpass-analysis=kernel-resource-usage remarks are parsed at build time and the
# build FAILS when one of them drops to one wave per SIMD or spills more than this many bytes per lane
# (a regression there costs the second workgroup or puts scratch traffic into the GEMM loops without
# changing any result, so no test would notice).  Today: LINEAR 0, MLP_OUT 0, one-pass 0, the two-pass
# MLP_LN 116 bytes (56 spilled VGPRs + 60 SGPRs parked in VGPR lanes, all outside the MFMA streams:
# address and descriptor values around the prologue / epilogue), the bf16 tier 260.
RESOURCE_LIMITS = {"rowmlp16h_kernel": dict(scratch=160, occupancy=2), "rowmlpbf_kernel": dict(scratch=320, occupancy=2),
                   # the eight-wave helper form (one 512-thread workgroup per CU = two waves per SIMD: the same 256-register budget)
                   "rowmlp16d_kernel": dict(scratch=160, occupancy=2),
                   # the wide form (eight multiplying waves, one workgroup per CU): the same budget again
                   # (round 6: + the late-addend instantiation of gc_tuning.wide_late)
                   "rowmlp16w_kernel": dict(scratch=200, occupancy=2)}


def check_resources(remarks, limits=None):
  """Parses hipcc's kernel-resource-usage remarks -> {kernel symbol: {scratch, occupancy, vgprs, ...}}; raises if
  a kernel named in `limits` exceeds its scratch budget or falls below its occupancy."""
  import re
  limits = RESOURCE_LIMITS if limits is None else limits
  usage, cur = {}, None
  for line in remarks.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
      cur = usage.setdefault(m.group(1), {})
      continue
    if cur is None:
      continue
    for key, pat in (("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occupancy", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("vgprs", r" VGPRs: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)")):
      m = re.search(pat, line)
      if m:
        cur[key] = int(m.group(1))
  bad = []
  # a gate that has nothing to check must not pass (another ROCm's remark format, remarks on stdout, a renamed
  # kernel): every kernel named in `limits` has to show up with both figures
  for name in limits:
    seen = [u for sym, u in usage.items() if name in sym and "scratch" in u and "occupancy" in u]
    if not seen:
      bad.append(f"{name}: no kernel-resource-usage remark found for it (hipcc -Rpass-analysis output format changed?)")
  for sym, u in usage.items():
    for name, lim in limits.items():
      if name in sym and u:
        if u.get("scratch", 0) > lim["scratch"] or u.get("occupancy", lim["occupancy"]) < lim["occupancy"]:
          bad.append(f"{sym}: scratch {u.get('scratch')} B/lane (limit {lim['scratch']}), occupancy "
                     f"{u.get('occupancy')} (needs {lim['occupancy']})")
  if bad:
    raise RuntimeError("register budget of a two-workgroups-per-CU kernel regressed:\n  " + "\n  ".join(bad))
  return usage


def source_files():
  """Every file the library is compiled from (the kernels' sources and the C-ABI header)."""
  return sorted([os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".inc"))]
                + [os.path.join(_INCLUDE, "gcast.h")])


def source_hash():
  """sha256 (16 hex digits) over the library's sources: compiled into the library (gc_build_info: ";src=...") and
  stamped into the counter summaries under profiles/ -- bench.py attaches a profile's numbers to its line only when
  the two agree (counters pasted from a profile of another build)."""
  import hashlib
  h = hashlib.sha256()
  for f in source_files():
    h.update(os.path.basename(f).encode() + b"\0")
    with open(f, "rb") as fh:
      h.update(fh.read())
  return h.hexdigest()[:16]


def loaded_source_hash():
  """The source hash the LOADED library was compiled from (None: a library from before round 4)."""
  info = lib().gc_build_info().decode()
  for part in info.split(";"):
    if part.startswith("src="):
      return part[4:]
  return None


def build(force=False, verbose=False):
  """Compiles csrc/gcast.hip for gfx950 with hipcc (all build variants) and checks the register budget of
  the hot kernels from the compiler's resource-usage remarks."""
  src = os.path.join(_CSRC, "gcast.hip")
  hdr = os.path.join(_INCLUDE, "gcast.h")
  deps = [src, hdr] + [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith(".inc")]
  newest = max(os.path.getmtime(f) for f in deps)
  src_hash = source_hash()
  for variant, (_, define) in VARIANTS.items():
    out = library_path(variant)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
      continue
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-inline-asm",
           "-Rpass-analysis=kernel-resource-usage", define, f'-DGC_SRC_HASH="{src_hash}"', "-I", _INCLUDE, "-shared",
           "-fPIC", src, "-o", out]
    if verbose:
      print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if res.returncode != 0:
      sys.stderr.write(res.stderr)
      raise subprocess.CalledProcessError(res.returncode, cmd)
    if verbose:      # compiler warnings of a successful build (the remarks themselves are summarised below)
      for line in res.stderr.splitlines():
        if "remark:" not in line and line.strip():
          print(line, file=sys.stderr)
    try:
      usage = check_resources(res.stderr)
    except RuntimeError:
      os.remove(out)
      raise
    if verbose:
      for sym, u in sorted(usage.items()):
        if any(k in sym for k in RESOURCE_LIMITS):
          print(f"  {sym}: {u}", file=sys.stderr)


_lib = None


def lib():
  """Loads the library once.  torch must be imported first so that the HIP runtime
  already in the process (torch's libamdhip64.so.7) is the one the kernels bind to."""
  global _lib
  if _lib is None:
    import torch  # noqa: F401  (loads libamdhip64 into the process)
    path = library_path()
    if not os.path.exists(path):
      raise RuntimeError(
          f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    l = ctypes.CDLL(path)
    l.gc_rowmlp.argtypes = [ctypes.POINTER(RowMlpDesc), ctypes.c_void_p]
    l.gc_seg_fixup.argtypes = [ctypes.c_int, _fp, _fp, _fp, _fp, _fp, ctypes.c_void_p]
    l.gc_zero_rows.argtypes = [ctypes.c_int, _fp, _fp, ctypes.c_void_p]
    l.gc_seg_fixup_bf16.argtypes = [ctypes.c_int, _fp, _fp, _fp, _fp, _fp, ctypes.c_void_p]
    l.gc_zero_rows_bf16.argtypes = [ctypes.c_int, _fp, _fp, ctypes.c_void_p]
    l.gc_seg_fixup_bf16.restype = l.gc_zero_rows_bf16.restype = ctypes.c_int
    l.gc_add_rows.argtypes = [ctypes.c_int, _fp, _fp, _fp, ctypes.c_void_p]
    l.gc_add_rows.restype = ctypes.c_int
    l.gc_prep_grid_input.argtypes = [ctypes.c_int] * 4 + [_fp, ctypes.c_int, _fp, ctypes.c_int, _fp,
                                                          ctypes.c_void_p]
    l.gc_prep_grid_tail.argtypes = [ctypes.c_int] * 5 + [_fp, ctypes.c_int, _fp, ctypes.c_int, _fp, ctypes.c_void_p]
    l.gc_prep_grid_tail.restype = ctypes.c_int
    l.gc_advance_state.argtypes = [ctypes.POINTER(AdvanceDesc), ctypes.c_void_p]
    l.gc_advance_state.restype = ctypes.c_int
    l.gc_run_program.argtypes = [ctypes.POINTER(Op), ctypes.c_int, ctypes.c_void_p]
    l.gc_time_program.argtypes = [ctypes.POINTER(Op), ctypes.c_int, ctypes.c_int,
                                  ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]
    for name in ("gc_rowmlp", "gc_seg_fixup", "gc_zero_rows", "gc_seg_fixup_bf16", "gc_zero_rows_bf16", "gc_add_rows", "gc_prep_grid_input",
                 "gc_run_program", "gc_time_program"):
      getattr(l, name).restype = ctypes.c_int
    l.gc_plan_create.argtypes = [ctypes.POINTER(ModelDesc), ctypes.POINTER(TensorDesc), ctypes.c_int,
                                 ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    l.gc_plan_create.restype = ctypes.c_int
    l.gc_plan_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]
    l.gc_plan_workspace_bytes.restype = ctypes.c_size_t
    l.gc_step_forward.argtypes = [ctypes.c_void_p, _fp, _fp, ctypes.c_int, _fp, ctypes.c_size_t, ctypes.c_void_p]
    l.gc_step_forward.restype = ctypes.c_int
    l.gc_plan_check_range.argtypes = [ctypes.c_void_p, _fp, ctypes.c_void_p]
    l.gc_plan_check_range.restype = ctypes.c_int
    l.gc_plan_destroy.argtypes = [ctypes.c_void_p]
    l.gc_plan_destroy.restype = None
    l.gc_plan_program.argtypes = [ctypes.c_void_p, _fp, _fp, ctypes.c_int, _fp, ctypes.c_size_t, ctypes.POINTER(Op),
                                  ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    l.gc_plan_program.restype = ctypes.c_int
    l.gc_plan_tensor.argtypes = [ctypes.c_void_p, _fp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p),
                                 ctypes.POINTER(ctypes.c_longlong), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    l.gc_plan_tensor.restype = ctypes.c_int
    l.gc_host_pack_weight.argtypes = [ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      _fp, ctypes.POINTER(ctypes.c_float)]
    l.gc_host_pack_weight.restype = ctypes.c_size_t
    l.gc_host_pack_edges.argtypes = [ctypes.c_int, _fp, _fp, ctypes.c_int] + [_fp] * 5 + [
        ctypes.POINTER(ctypes.c_int), _fp, ctypes.POINTER(ctypes.c_int)]
    l.gc_host_pack_edges.restype = ctypes.c_int
    l.gc_host_pad_latent.argtypes = [ctypes.POINTER(TensorDesc), ctypes.c_int, ctypes.c_int, _fp, ctypes.c_longlong,
                                     ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    l.gc_host_pad_latent.restype = ctypes.c_int
    l.gc_get_tuning.argtypes = [ctypes.POINTER(Tuning)]
    l.gc_set_tuning.argtypes = [ctypes.POINTER(Tuning)]
    l.gc_plan_get_tuning.argtypes = [ctypes.c_void_p, ctypes.POINTER(Tuning)]
    l.gc_get_tuning.restype = l.gc_set_tuning.restype = l.gc_plan_get_tuning.restype = ctypes.c_int
    l.gc_tuning_string.argtypes = [ctypes.POINTER(Tuning)]
    l.gc_tuning_string.restype = ctypes.c_char_p
    l.gc_last_error.restype = ctypes.c_char_p
    l.gc_abi_sizeof.argtypes = [ctypes.c_int]
    l.gc_abi_sizeof.restype = ctypes.c_size_t
    if (l.gc_abi_sizeof(0) != ctypes.sizeof(RowMlpDesc) or l.gc_abi_sizeof(1) != ctypes.sizeof(Op)
        or l.gc_abi_sizeof(2) != ctypes.sizeof(AdvanceDesc) or l.gc_abi_sizeof(3) != ctypes.sizeof(ModelDesc)
        or l.gc_abi_sizeof(4) != ctypes.sizeof(Tuning)):
      raise RuntimeError("ctypes struct layout does not match include/gcast.h "
                         f"({l.gc_abi_sizeof(0)}/{ctypes.sizeof(RowMlpDesc)}, "
                         f"{l.gc_abi_sizeof(1)}/{ctypes.sizeof(Op)}); rebuild the library")
    l.gc_build_info.restype = ctypes.c_char_p
    if b"PROFILING_BUILD" in l.gc_build_info():
      raise RuntimeError(f"{path} is a profiling build (GC_EXP / GC_TRACE): its results may be wrong; "
                         "rebuild with graphcast_amd._native.build(force=True)")
    _lib = l
  return _lib


class GcastError(RuntimeError):
  pass


class GcastRangeError(GcastError):
  """An input row held a value outside the exact range of the f16x3 arithmetic (|x| > F16X3_MAX)."""


F16X3_MAX = 65504.0        # GC_F16X3_MAX
EINVAL, ELAUNCH, ERANGE = -1, -2, -3


def check(rc, what):
  if rc != 0:
    raise GcastError(f"{what} failed ({rc}): {lib().gc_last_error().decode()}")


def get_tuning() -> Tuning:
  """The process default of the library's tuning (gc_get_tuning)."""
  t = Tuning()
  check(lib().gc_get_tuning(ctypes.byref(t)), "gc_get_tuning")
  return t


def set_tuning(t: Tuning = None, **fields) -> Tuning:
  """gc_set_tuning: replaces the process default by `t` (default: the current one) with `fields` overridden; returns
  the PREVIOUS tuning (hand it back to restore).  Plans created afterwards snapshot the new values."""
  prev = get_tuning()
  new = Tuning.from_buffer_copy(bytes(t if t is not None else prev))
  for k, v in fields.items():
    if k not in new.as_dict():
      raise KeyError(f"gc_tuning has no field {k!r}")
    setattr(new, k, int(v))
  check(lib().gc_set_tuning(ctypes.byref(new)), "gc_set_tuning")
  return prev


def tuning_string(t: Tuning = None) -> str:
  return lib().gc_tuning_string(ctypes.byref(t) if t is not None else None).decode()


def ptr(t):
  """Device pointer of a torch tensor (or None -> NULL)."""
  if t is None:
    return None
  return t.data_ptr()
