"""The kernels' inline-asm loads against their counted waits, checked on the ASSEMBLY of the sources in the tree.

scripts/asm_hazard_check.py models the vmcnt queue over every row-MLP kernel's assembly and reports any instruction
that reads or overwrites a register whose asm load may still be in flight.  Round 6 met that class of bug for the third
time -- a row request that outlived its loop made the bf16 tier's 0.25 deg step non-repeatable, DESIGN.md section 9.20 --
and every one of them was invisible to parity tests: a new instantiation of these kernels is a new register
allocation, and this is the check it has to pass before it sees a GPU.  (hipcc cross-compiles: no GPU needed.)
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
  if shutil.which("hipcc") is None:
    pytest.skip("hipcc not on PATH")
  out = str(tmp_path_factory.mktemp("asm") / "gcast.s")
  cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-inline-asm", "-S",
         "--cuda-device-only", "-DGC_PIPE=2", '-DGC_SRC_HASH="x"', "-I", os.path.join(ROOT, "include"),
         os.path.join(ROOT, "graphcast_amd", "csrc", "gcast.hip"), "-o", out]
  subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
  return out


def test_no_asm_load_is_read_or_overwritten_before_the_wait_that_retires_it(assembly):
  sys.path.insert(0, os.path.join(ROOT, "scripts"))
  import asm_hazard_check as chk
  kernels = {name: lines for name, lines in chk.functions(assembly).items() if "rowmlp" in name}
  # every form of the launch is there: four-wave, helper (three HST variants), wide (+ late addends), one-pass, bf16 (+ streamed)
  assert len(kernels) >= 28 and sum("rowmlpbf" in k for k in kernels) >= 6
  bad = {name: chk.check(lines)[:4] for name, lines in kernels.items()}
  bad = {k: v for k, v in bad.items() if v}
  assert not bad, bad


def test_the_checker_sees_the_bug_it_was_written_for(assembly, tmp_path):
  """The same sources without the landing wait (-DGC_BF_XR_LAND=0: the library that was not repeatable) must be reported --
  the layer-1 loop's last row request is overwritten by a weight-fragment read of layer 2."""
  sys.path.insert(0, os.path.join(ROOT, "scripts"))
  import asm_hazard_check as chk
  out = str(tmp_path / "gcast_no_wait.s")
  cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-Wno-inline-asm", "-S",
         "--cuda-device-only", "-DGC_PIPE=2", "-DGC_BF_XR_LAND=0", '-DGC_SRC_HASH="x"', "-I", os.path.join(ROOT, "include"),
         os.path.join(ROOT, "graphcast_amd", "csrc", "gcast.hip"), "-o", out]
  subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
  found = {name: chk.check(lines) for name, lines in chk.functions(out).items() if "rowmlpbf_kernelILb0ELi4ELi0E" in name}
  assert len(found) == 1
  hits = next(iter(found.values()))
  assert hits and any("OVERWRITES" in s for _, s, _, _ in hits), hits
