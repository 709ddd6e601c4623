"""The C-ABI shared library builds for gfx950, loads, and exports every symbol
that include/gcast.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from graphcast_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
  text = open(os.path.join(ROOT, "include", "gcast.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(gc_[a-z_0-9]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
  nat.build()
  return nat


def test_header_and_binding_agree():
  assert declared_symbols() == sorted(nat.EXPORTS)


@pytest.mark.parametrize("variant", sorted(nat.VARIANTS))
def test_library_exports_every_declared_symbol(built, variant):
  import torch  # noqa: F401  load torch's HIP runtime first, as the product does
  lib = ctypes.CDLL(built.library_path(variant))
  for name in declared_symbols():
    assert hasattr(lib, name), f"{name} missing from {variant} library"
  lib.gc_build_info.restype = ctypes.c_char_p
  info = lib.gc_build_info().decode()
  assert "gfx950" in info and "ring=4x16k" in info and "persistent" in info


def test_struct_layout_matches_header(built):
  lib = built.lib()       # lib() itself refuses to load on a mismatch
  assert lib.gc_abi_sizeof(0) == ctypes.sizeof(nat.RowMlpDesc)
  assert lib.gc_abi_sizeof(1) == ctypes.sizeof(nat.Op)
  assert lib.gc_abi_sizeof(2) == ctypes.sizeof(nat.AdvanceDesc)
  assert lib.gc_abi_sizeof(3) == ctypes.sizeof(nat.ModelDesc)      # (round 5: + the halo-table sizes)
  assert lib.gc_abi_sizeof(4) == ctypes.sizeof(nat.Tuning)         # (round 6: the one tuning surface)
  assert lib.gc_abi_sizeof(5) == 0


def test_tuning_is_one_struct_set_and_read_back(built):
  """include/gcast.h: gc_tuning -- every speed-only switch of the library in ONE struct.  The
  GCAST_* variables initialise the process default once; afterwards the struct is what the launch functions read:
  a host sets it, reads it back, and a value outside its range is refused without changing anything."""
  lib = built.lib()
  before = nat.get_tuning()
  d = before.as_dict()
  assert set(d) == {"grid_cap", "tile_map_xcd", "prio_set", "prio_gemm", "prio_other", "prio_stage", "helpers",
                    "helpers_small", "helpers_edge", "helper_store", "helpers_min_rows", "wide", "wide_edges", "bf16_rows",
                    "tile_queue", "fuse", "onepass", "split_tail", "bf16_stream", "wide_late", "split_edges"}
  if not any(k.startswith("GCAST_") for k in os.environ):      # the documented defaults of a process without overrides
    assert d == dict(grid_cap=512, tile_map_xcd=0, prio_set=0, prio_gemm=1, prio_other=0, prio_stage=0, helpers=-1,
                     helpers_small=1, helpers_edge=1, helper_store=2, helpers_min_rows=65536, wide=1,
                     wide_edges=nat.WIDE_EDGES_DEFAULT, bf16_rows=0, tile_queue=1, fuse=1, onepass=1, split_tail=0, bf16_stream=1, wide_late=1, split_edges=0)
  try:
    prev = nat.set_tuning(grid_cap=256, helpers_edge=2, wide_edges=3, prio_gemm=2)
    assert bytes(prev) == bytes(before)
    now = nat.get_tuning().as_dict()
    assert (now["grid_cap"], now["helpers_edge"], now["wide_edges"], now["prio_gemm"]) == (256, 2, 3, 2)
    assert {k: v for k, v in now.items() if k not in ("grid_cap", "helpers_edge", "wide_edges", "prio_gemm")} == \
           {k: v for k, v in d.items() if k not in ("grid_cap", "helpers_edge", "wide_edges", "prio_gemm")}
    s = nat.tuning_string()
    assert "grid_cap=256" in s and "helpers_edge=2" in s and "wide_edges=3" in s and "prio=2,0,0" in s
    # the environment is NOT consulted again: a variable set now changes nothing
    os.environ["GCAST_GRID_CAP"] = "17"
    try:
      assert nat.get_tuning().grid_cap == 256
    finally:
      del os.environ["GCAST_GRID_CAP"]
    bad = nat.get_tuning()
    bad.grid_cap = 0
    assert lib.gc_set_tuning(ctypes.byref(bad)) == nat.EINVAL and b"outside its range" in lib.gc_last_error()
    assert nat.get_tuning().grid_cap == 256
    with pytest.raises(KeyError):
      nat.set_tuning(no_such_field=1)
  finally:
    nat.set_tuning(before)
  assert bytes(nat.get_tuning()) == bytes(before)


def test_no_getenv_inside_a_launch_function():
  """The only getenv calls of the library sit in tuning_from_env() (csrc/gcast.hip), the default initialiser."""
  src = open(os.path.join(ROOT, "graphcast_amd", "csrc", "gcast.hip")).read()
  body = src[src.index("gc_tuning tuning_from_env() {"):]
  body = body[:body.index("\n}\n") + 3]
  for f in ("gcast.hip", "gcast_plan.inc", "rowmlp_half.inc", "rowmlp_bf16.inc"):
    text = open(os.path.join(ROOT, "graphcast_amd", "csrc", f)).read()
    if f == "gcast.hip":
      text = text.replace(body, "")
    code = "\n".join(line.split("//")[0] for line in text.splitlines())
    assert "getenv" not in code, f


def test_argument_validation_needs_no_gpu(built):
  lib = built.lib()
  d = nat.RowMlpDesc()
  d.mode, d.n_rows = nat.MODE_MLP_LN, 64
  assert lib.gc_rowmlp(ctypes.byref(d), None) == -1
  assert b"no layer-1 input" in lib.gc_last_error()
  d.k0 = 33
  assert lib.gc_rowmlp(ctypes.byref(d), None) == -1
  assert b"multiples of 32" in lib.gc_last_error()
  assert lib.gc_prep_grid_input(10, 1, 0, 471, None, 3, None, 474, None, None) == -1


def test_python_constants_mirror_the_header():
  """The ctypes side repeats the header's #defines by hand: every one of them is compared here."""
  text = open(os.path.join(ROOT, "include", "gcast.h")).read()
  defines = {m.group(1): m.group(2) for m in re.finditer(r"^#define (GC_[A-Z0-9_]+) \(?(-?[0-9.]+)f?\)?\s", text, flags=re.M)}
  mirror = dict(GC_LATENT=nat.LATENT, GC_TILE_ROWS=nat.TILE_ROWS, GC_K_CHUNK=nat.K_CHUNK, GC_SCRATCH_SLOTS=nat.SCRATCH_SLOTS,
                GC_EINVAL=nat.EINVAL, GC_ELAUNCH=nat.ELAUNCH, GC_ERANGE=nat.ERANGE, GC_F16X3_MAX=nat.F16X3_MAX,
                GC_ROWS_F32=nat.ROWS_F32, GC_W2_NATURAL=nat.W2_NATURAL, GC_WG_ROWS_64=nat.WG_ROWS_64,
                GC_WG_ROWS_128=nat.WG_ROWS_128, GC_TILE_XCD=nat.TILE_MAP_XCD, GC_WG_HELPERS=nat.WG_HELPERS,
                GC_WG_NO_HELPERS=nat.WG_NO_HELPERS, GC_TILE_QUEUE_ANY=nat.TILE_QUEUE_ANY, GC_MAX_CHAIN=nat.MAX_CHAIN,
                GC_WG_WIDE=nat.WG_WIDE, GC_LATE_ADDENDS=nat.LATE_ADDENDS)
  for name, value in mirror.items():
    assert name in defines, f"{name} not found in include/gcast.h"
    assert float(defines[name]) == float(value), (name, defines[name], value)
  flags = [mirror[k] for k in ("GC_ROWS_F32", "GC_W2_NATURAL", "GC_WG_ROWS_64", "GC_WG_ROWS_128", "GC_TILE_XCD",
                               "GC_WG_HELPERS", "GC_WG_NO_HELPERS", "GC_TILE_QUEUE_ANY", "GC_WG_WIDE", "GC_LATE_ADDENDS")]
  assert sorted(flags) == [1 << i for i in range(len(flags))], "gc_rowmlp_desc.flags bits must be distinct"


def test_tile_queue_arguments_are_validated(built):
  lib = built.lib()
  d = nat.RowMlpDesc()
  d.mode, d.n_rows, d.prec, d.layout = nat.MODE_MLP_LN, 64, nat.PREC_F16X3, nat.LAYOUT_HALF
  d.tile_queue = 0x1004                                  # (never dereferenced: validation comes first)
  assert lib.gc_rowmlp(ctypes.byref(d), None) == nat.EINVAL
  assert b"tile_queue must be an 8-byte aligned pair" in lib.gc_last_error()
  d.tile_queue, d.layout, d.prec = 0x1008, nat.LAYOUT_CHUNKED, nat.PREC_F32
  assert lib.gc_rowmlp(ctypes.byref(d), None) == nat.EINVAL
  assert b"tile_queue is a feature of the persistent kernels" in lib.gc_last_error()


def test_header_is_plain_c_and_the_c_host_example_links():
  """include/gcast.h must be consumable from C (the drop-in boundary is a C-ABI): the example host
  compiles as strict C99 and links against the built library (it needs a GPU to RUN; on the MI355X
  box it printed a finite checksum, see examples/plan_host.c)."""
  import shutil
  import subprocess
  import tempfile
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  gcc = shutil.which("gcc")
  if gcc is None:
    pytest.skip("no gcc")
  src = os.path.join(root, "examples", "plan_host.c")
  subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                  "-fsyntax-only", src], check=True)
  rocm_lib = "/opt/rocm/lib"
  if not os.path.exists(os.path.join(rocm_lib, "libamdhip64.so")):
    pytest.skip("no HIP runtime to link against")
  with tempfile.TemporaryDirectory() as tmp:
    subprocess.run([gcc, "-std=c99", "-I", os.path.join(root, "include"), src,
                    "-L", os.path.join(root, "graphcast_amd", "csrc"), "-lgcast_hip", "-L", rocm_lib, "-lamdhip64",
                    "-lm", "-o", os.path.join(tmp, "plan_host")], check=True)


def test_build_checks_the_register_budget_of_the_hot_kernels():
  """_native.build() parses hipcc's kernel-resource-usage remarks and refuses a library whose two-workgroups-per-CU
  kernels lost their occupancy or spill beyond their budget (nothing caught such a regression)."""
  remark = ("x.inc:1:1: remark: Function Name: _ZN12_GLOBAL__N_116rowmlp16h_kernelILi1ELi0EEEv14gc_rowmlp_desc [-R]\n"
            "x.inc:1:1: remark:     VGPRs: 256 [-R]\n"
            "x.inc:1:1: remark:     ScratchSize [bytes/lane]: {scratch} [-R]\n"
            "x.inc:1:1: remark:     Occupancy [waves/SIMD]: {occ} [-R]\n"
            "x.inc:1:1: remark:     SGPRs Spill: 60 [-R]\n"
            "x.inc:1:1: remark:     VGPRs Spill: 56 [-R]\n"
            "x.inc:1:1: remark: Function Name: some_other_kernel [-R]\n"
            "x.inc:1:1: remark:     ScratchSize [bytes/lane]: 4000 [-R]\n"
            "x.inc:1:1: remark: Function Name: _ZN12_GLOBAL__N_116rowmlp16d_kernelILi1ELi0EEEv14gc_rowmlp_desc [-R]\n"
            "x.inc:1:1: remark:     ScratchSize [bytes/lane]: 116 [-R]\n"
            "x.inc:1:1: remark:     Occupancy [waves/SIMD]: 2 [-R]\n"
            "x.inc:1:1: remark: Function Name: _ZN12_GLOBAL__N_116rowmlp16w_kernelILi1EEEv14gc_rowmlp_desc [-R]\n"
            "x.inc:1:1: remark:     ScratchSize [bytes/lane]: 112 [-R]\n"
            "x.inc:1:1: remark:     Occupancy [waves/SIMD]: 2 [-R]\n"
            "x.inc:1:1: remark: Function Name: _ZN12_GLOBAL__N_115rowmlpbf_kernelILb0ELi4EEEv14gc_rowmlp_desc [-R]\n"
            "x.inc:1:1: remark:     ScratchSize [bytes/lane]: 244 [-R]\n"
            "x.inc:1:1: remark:     Occupancy [waves/SIMD]: 2 [-R]\n")
  usage = nat.check_resources(remark.format(scratch=116, occ=2))
  sym = "_ZN12_GLOBAL__N_116rowmlp16h_kernelILi1ELi0EEEv14gc_rowmlp_desc"
  assert usage[sym] == dict(vgprs=256, scratch=116, occupancy=2, sgpr_spill=60, vgpr_spill=56)
  with pytest.raises(RuntimeError, match="register budget"):
    nat.check_resources(remark.format(scratch=348, occ=2))
  with pytest.raises(RuntimeError, match="occupancy 1"):
    nat.check_resources(remark.format(scratch=0, occ=1))
  # a gate with nothing to check must not pass: no remark for a kernel named in the limits
  with pytest.raises(RuntimeError, match="no kernel-resource-usage remark"):
    nat.check_resources("x.inc:1:1: remark: Function Name: some_other_kernel [-R]\n")
  with pytest.raises(RuntimeError, match="no kernel-resource-usage remark"):
    nat.check_resources("")


@pytest.mark.parametrize("name", ["rows_per_fragment", "gh_skeleton", "mfma_issue"])
def test_micro_benchmarks_cross_compile_for_gfx950(name, tmp_path):
  """scripts/ubench/*.hip -- the skeletons DESIGN.md section 9 prices kernel structures with before they are built --
  stay compilable for gfx950 (no GPU needed: hipcc cross-compiles)."""
  import shutil
  import subprocess
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  if not os.path.exists(hipcc):
    pytest.skip("no hipcc in this environment")
  src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "ubench", name + ".hip")
  r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-w", "-c", src, "-o", str(tmp_path / (name + ".o"))],
                     capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.parametrize("latent,hidden_layers", [(64, 1), (128, 2), (256, 1), (384, 1), (100, 2), (7, 1)])
def test_padded_parameters_of_a_narrow_latent_give_the_same_step(built, latent, hidden_layers):
  """gc_plan_create runs a model whose latent size L divides 512 on the 512-column kernels by re-shaping its PARAMETERS
  (csrc/gcast_plan.inc: pad_latent; zero-padded latent axes, LayerNorm-fed output columns replicated floor(512 / L) times,
  the mean column + a rescaling where L does not divide 512).
  The rule itself, checked without a GPU: the oracle (oracle/graphcast.py, ANY width) gives the same step on the tree
  gc_host_pad_latent hands back -- a latent-512 model -- as on the original one (weathernext1_graph/graphcast.py:123,
  deep_typed_graph_net.py:205-247)."""
  import numpy as np
  from graphcast_amd.engine import tensor_descs
  from oracle import graphcast as ogc
  from oracle import params as oparams
  lib = built.lib()
  c_in, c_out, steps = 11, 5, 2
  rng = np.random.default_rng(latent + hidden_layers)
  params = {}
  for stem, sizes, ln in oparams.module_specs(c_in, c_out, latent, steps):
    sizes = [sizes[0]] + [latent] * hidden_layers + [sizes[-1]]
    for k in range(len(sizes) - 1):
      params[f"{stem}_mlp/~/linear_{k}"] = {
          "w": (rng.standard_normal((sizes[k], sizes[k + 1])) / np.sqrt(sizes[k])).astype(np.float32),
          "b": (0.1 * rng.standard_normal(sizes[k + 1])).astype(np.float32)}
    if ln:
      params[f"{stem}_layer_norm"] = {"scale": (1 + 0.1 * rng.standard_normal(sizes[-1])).astype(np.float32),
                                      "offset": (0.1 * rng.standard_normal(sizes[-1])).astype(np.float32)}
  descs, keep = tensor_descs(params)
  padded = {}
  for i, t in enumerate(descs):
    rows, cols = ctypes.c_int(), ctypes.c_int()
    assert lib.gc_host_pad_latent(descs, len(descs), i, None, 0, ctypes.byref(rows), ctypes.byref(cols)) == 0, lib.gc_last_error()
    out = np.empty((rows.value, cols.value), np.float32)
    assert lib.gc_host_pad_latent(descs, len(descs), i, out.ctypes.data, out.size, None, None) == 0, lib.gc_last_error()
    module, leaf = t.name.decode().rsplit("/", 1)
    padded.setdefault(module, {})[leaf] = out if leaf == "w" else out[0]
  D = nat.LATENT
  for module, leaves in padded.items():           # every latent axis is 512 wide now; raw-feature rows / c_out columns stay
    for leaf, a in leaves.items():
      decoder_out = "decoder_" in module and module.endswith(f"linear_{hidden_layers}")
      assert a.shape[-1] == (c_out if decoder_out else D), (module, leaf, a.shape)
  res, mesh_size = 12.0, 2
  lat, lon = np.arange(-90, 90 + res / 2, res), np.arange(0, 360, res)
  graphs = ogc.build_graphs(lat, lon, mesh_size)
  x = rng.standard_normal((graphs["n_grid"], 2, c_in)).astype(np.float32)      # (module_specs adds the 3 structural features)
  want = ogc.forward(params, graphs, x, steps=steps, dtype=np.float64)
  got = ogc.forward(padded, graphs, x, steps=steps, dtype=np.float64)
  err = np.linalg.norm(got - want) / np.linalg.norm(want)
  # L dividing 512: pure copies and zeros -- the fp64 oracle cannot tell the trees apart.  Otherwise the Linear's columns
  # and the LayerNorm scale carry sqrt(512 / (R L)) and its inverse, each rounded to float32 once (6e-8 relative).
  assert err < (1e-12 if D % latent == 0 else 1e-6), err
  # and sizes the tile cannot hold are refused by name
  bad = {k: dict(v) for k, v in params.items()}
  stem = "grid2mesh_gnn/~_networks_builder/encoder_nodes_grid_nodes_mlp/~/linear_0"
  bad[stem]["w"] = np.zeros((bad[stem]["w"].shape[0], 640), np.float32)
  d2, keep2 = tensor_descs(bad)
  assert lib.gc_host_pad_latent(d2, len(d2), 0, None, 0, None, None) == nat.EINVAL
  assert b"1 .. 512" in lib.gc_last_error()
  del keep, keep2
