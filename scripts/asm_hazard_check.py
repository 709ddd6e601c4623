#!/usr/bin/env python
"""Static check of the kernels' inline-asm loads against the counted-wait protocol.

The row-MLP kernels request rows with `asm volatile("global_load_dwordx4 ...")` so that the compiler's own vmcnt
accounting does not drain the LDS-DMA ring, and consume them behind COUNTED `s_waitcnt vmcnt(N)`.  The compiler knows
neither: if its register allocation copies, spills or reuses such a destination register before the wait that retires
the load, the kernel reads (or clobbers) data that is not there yet -- timing-dependent, invisible to every parity test
(rounds 3-6 met this three times).  This script models the vmcnt queue over the ASSEMBLY of a build, in layout order
(basic blocks in layout order, each entered with the worst state of the branches seen to target it and of the
fall-through; loops twice): every vector-memory instruction enters the queue in issue order, an
`s_waitcnt vmcnt(N)` retires all but the youngest N, and any instruction that READS -- or OVERWRITES: the load would
land on top of the new value, which is what made the bf16 tier non-repeatable in round 6 -- a register whose asm load is
still in the queue is reported.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -DGC_PIPE=2 -DGC_SRC_HASH='"x"' -I include \\
          graphcast_amd/csrc/gcast.hip -o /tmp/gcast.s
    python scripts/asm_hazard_check.py /tmp/gcast.s [kernel-name-substring]       # exit 1 when something is reported
"""
import re
import sys

VM = ("global_load", "global_store", "global_atomic", "scratch_load", "scratch_store", "buffer_load", "buffer_store",
      "buffer_atomic", "flat_load", "flat_store", "flat_atomic")


def vregs(tok):
  out = []
  for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
    out += list(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else [int(m.group(3))]
  return out


def functions(path):
  cur, out = None, {}
  for line in open(path):
    m = re.match(r"^(_Z\w+):", line)
    if m:
      cur = m.group(1)
      out[cur] = []
    elif line.startswith(".Lfunc_end"):
      cur = None
    elif cur is not None:
      out[cur].append(line.rstrip("\n"))
  return out


def check(lines):
  queue = []            # outstanding vector-memory operations, oldest first: (line, destination registers of an ASM load or ())
  pending = {}          # register -> line of the asm load that will write it
  in_asm, found = False, []
  n = len(lines)
  snapshots = {}        # label -> states at the branches that target it
  after_jump = False    # the previous instruction was an unconditional branch: no fall-through into the next label
  for k, raw in enumerate(lines + lines):          # (twice: a request at the bottom of a loop is consumed at its top)
    s = raw.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
      # a block's entry state: of the states that can reach it (branches seen so far + the fall-through) the one with
      # the most loads in flight -- the arms of an if / else do not see each other's requests, a join sees the worst arm
      cands = list(snapshots.get(m.group(1), [])) + ([] if after_jump else [(queue, pending)])
      if cands:
        q, p = max(cands, key=lambda c: (len(c[1]), len(c[0])))
        queue, pending = list(q), dict(p)
      after_jump = False
      continue
    if s.startswith(";;#ASMSTART"):
      in_asm = True
      continue
    if s.startswith(";;#ASMEND"):
      in_asm = False
      continue
    if not s or s[0] in ";.":
      continue
    s = s.split(";")[0].strip()
    op = s.split()[0]
    ops = [t.strip() for t in s[len(op):].split(",")]
    if op == "s_branch" or op.startswith("s_cbranch"):
      snapshots.setdefault(ops[-1], []).append((list(queue), dict(pending)))
      if len(snapshots[ops[-1]]) > 8:
        snapshots[ops[-1]].pop(0)
      after_jump = op == "s_branch"
      continue
    after_jump = False
    m = re.search(r"vmcnt\((\d+)\)", s)
    if op == "s_waitcnt":
      if m:
        keep = int(m.group(1))
        while len(queue) > keep:
          _, regs = queue.pop(0)
          for r in regs:
            pending.pop(r, None)
      continue
    is_vm = op.startswith(VM)
    is_asm_load = in_asm and op.startswith("global_load") and "lds" not in op
    all_src = op.startswith(("global_store", "scratch_store", "buffer_store", "flat_store", "ds_write", "v_cmp", "v_cmpx"))
    srcs = ops if all_src else ops[1:]
    if not is_asm_load:
      for t in srcs:
        for r in vregs(t):
          if r in pending:
            found.append((k % n, s, r, pending[r] % n))
    if is_vm:
      dst = tuple(vregs(ops[0])) if is_asm_load else ()
      if is_asm_load:
        for r in dst:
          pending[r] = k
      elif not all_src:
        for r in vregs(ops[0]):                   # a compiler load into such a register while the asm load is in flight
          if r in pending:
            found.append((k % n, "OVERWRITES " + s, r, pending.pop(r) % n))
      queue.append((k, dst))
    elif not all_src and not in_asm and not op.startswith(("s_", "ds_write", "v_readfirstlane", "v_readlane")):
      for r in vregs(ops[0]):
        if r in pending:                          # overwritten while in flight: the load lands on top of the new value
          found.append((k % n, "OVERWRITES " + s, r, pending.pop(r) % n))
  seen, uniq = set(), []
  for f in found:
    if f[0] not in seen:
      seen.add(f[0])
      uniq.append(f)
  return uniq


def main():
  path = sys.argv[1]
  pat = sys.argv[2] if len(sys.argv) > 2 else "rowmlp"
  bad = 0
  for name, lines in functions(path).items():
    if pat in name:
      found = check(lines)
      print(f"{name}: {len(found)} use(s) of a register whose asm load may still be in flight")
      for line, s, r, at in found[:10]:
        print(f"    line {line}: v{r} (requested at line {at}): {s[:110]}")
      bad += len(found)
  sys.exit(1 if bad else 0)


if __name__ == "__main__":
  main()
