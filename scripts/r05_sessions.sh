#!/bin/bash
# The GPU-box sessions of round 5 in ONE file: `bash scripts/r05_sessions.sh <name>` runs one of them from the repository
# root; results land under gpurun_out/r05_<name>/, what is kept is copied to profiles/r05_<name>_*.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
NAME=${1:?session name}
OUT=gpurun_out/r05_$NAME; mkdir -p "$OUT"
export TMPDIR=/tmp
gate() { grep -q " passed" "$1" && ! grep -q "failed\|rror\|Timeout" "$1" || { echo "GATE: $2 failed"; tail -40 "$1"; exit 1; }; }
show() { python - "$1" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(j["ms_per_step"], 3), {k: round(v["ms"], 3) for k, v in (j.get("roofline") or {}).get("stages", {}).items()})
for k in ("rollout", "rollout_api"):
  if j.get(k): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in j[k].items() if a != "what"})
if (j.get("roofline") or {}).get("exchange"): print("exchange", j["roofline"]["exchange"])
PY
}
case "$NAME" in
  s1)
    # Round-5 session 1: the round's host-side changes on the GPU -- range flag on the aggregate-fed launches (engine +
    # plan), the fused loop behind rollout.chunked_prediction, the bench line's rollout_api, --mode partition's per-rank
    # roofline + exchange probe.
    timeout 900 python -m pytest tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_rollout_gpu.py tests/test_partition_gpu.py -m gpu -q -x --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log"
    gate "$OUT/pytest.log" "host-side changes"
    timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; show "$OUT/bench.json"
    timeout 600 python bench.py --mode partition --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_partition_n1.json" 2> "$OUT/bench_partition.err"; echo "partition rc=$?"; show "$OUT/bench_partition_n1.json"
    ;;
  s2)
    # Round-5 session 2: wave issue priority by phase (gc_rowmlp_desc.flags GC_PRIO / GCAST_PRIO="gemm,other,stage"):
    # same-session A/B of the whole step in both arithmetic tiers, two-workgroups-per-CU and helper-wave forms.
    bash scripts/session.sh bench-ab r05_s2 "GCAST_PRIO=0,0,0" "GCAST_PRIO=1,0,0" "GCAST_PRIO=2,0,0" "GCAST_PRIO=0,1,0" \
        "GCAST_PRIO=3,0,0" "GCAST_PRIO=0,0,0"
    bash scripts/session.sh bench-ab r05_s2h "GCAST_HELPERS=1 GCAST_PRIO=0,0,0" "GCAST_HELPERS=1 GCAST_PRIO=1,0,0" \
        "GCAST_HELPERS=1 GCAST_PRIO=3,0,0" "GCAST_HELPERS=1 GCAST_PRIO=0,0,1"
    bash scripts/session.sh bench-ab r05_s2b --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 --precision bf16 -- \
        "GCAST_PRIO=0,0,0" "GCAST_PRIO=1,0,0" "GCAST_PRIO=0,1,0" "GCAST_PRIO=0,0,0"
    timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check > "$OUT/bench_rollout_api.json" 2> "$OUT/bench_rollout_api.err"; echo "bench rc=$?"; show "$OUT/bench_rollout_api.json"
    ;;
  s3)
    # Round-5 session 3: where the host time of the fused rollout behind rollout.chunked_prediction_generator goes
    # (rollout_api.host_seconds), with the device-side upload in DeviceRollout._prepare and the sampled cross-check; the
    # shipped priority default against GCAST_PRIO=0,0,0; the emulated 8-way partition's per-rank time (baseline).
    timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; show "$OUT/bench.json"
    GCAST_PRIO=0,0,0 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 > "$OUT/bench_prio0.json" 2> "$OUT/bench_prio0.err"; echo "bench rc=$?"; show "$OUT/bench_prio0.json"
    timeout 600 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8.json" 2>&1 | tail -5
    ;;
  s4)
    # Round-5 session 4: after the prune (chunked f16x3 kernels + the bf16gemm tier gone) -- smoke + the GPU tests that
    # drive launches directly, then the bench line with rollout_api (channel tables from ONE grid point).
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
    timeout 1200 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_rollout_gpu.py tests/test_bf16_tier_gpu.py tests/test_native_abi.py tests/test_deepgnn_gpu.py tests/test_conditioned_gpu.py -m gpu -q -x --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log"
    gate "$OUT/pytest.log" "prune"
    timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; show "$OUT/bench.json"
    ;;
  s5)
    # Round-5 session 5: ONE program builder -- engine.StepEngine on gc_plan_program / gc_plan_tensor, partitioned graphs
    # through gc_model_desc's halo-table sizes, the Python recorder gone: smoke, the WHOLE GPU suite (without the 0.25 deg
    # 40-step fixture test while that fixture is being regenerated), bench + partition-mode bench + the emulated 8-way.
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.log"
    timeout 1500 python -m pytest tests -m gpu -q --timeout=900 --deselect tests/test_rollout40_fullsize_gpu.py > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest.log" | cut -c1-400
    grep -E "FULLSIZE_PARITY|ROLLOUT" "$OUT/pytest.log" | cut -c1-600
    timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; show "$OUT/bench.json"
    timeout 600 python bench.py --mode partition --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_partition_n1.json" 2> "$OUT/bench_partition.err"; echo "partition rc=$?"; show "$OUT/bench_partition_n1.json"
    timeout 600 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8.json" 2>&1 | tail -3 | cut -c1-600
    ;;
  s6)
    # Round-5 session 6: the driver's SCALE launch form at N = 1 (torch.distributed.run -> RCCL group of one rank): stdout
    # must carry the JSON line and NOTHING else (RCCL prints a banner to the C-level stdout); the same for --mode partition.
    timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-cross-check --rollout-steps 0 > "$OUT/bench_torchrun_n1.json" 2> "$OUT/bench_torchrun_n1.err"; echo "torchrun bench rc=$?"
    python - "$OUT/bench_torchrun_n1.json" <<'PY'
import json, sys
raw = open(sys.argv[1]).read()
lines = [l for l in raw.splitlines() if l.strip()]
print("stdout lines:", len(lines)); j = json.loads(raw); print("ONE JSON document on stdout:", j["n_gpus"], round(j["ms_per_step"], 3), j["config"]["parallelism"])
PY
    timeout 600 python bench.py --mode partition --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_partition_n1.json" 2> "$OUT/bench_partition.err"; echo "partition rc=$?"
    python - "$OUT/bench_partition_n1.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read()); print("partition: ONE JSON document on stdout:", round(j["ms_per_step"], 3), j["roofline"]["exchange"]["device_ms_per_step"], j["output_finite"])
PY
    grep -c "RCCL version" "$OUT/bench_partition.err" "$OUT/bench_torchrun_n1.err"
    ;;
  s7)
    # Round-5 session 7: the eight-wave form's staging waves take residual + store of the edge updates (HST): parity gate
    # (per-launch tests run both kernel forms against each other; the whole step with every launch in the eight-wave
    # form against the oracle), then same-session A/B of the step.
    GCAST_HELPERS=1 timeout 300 python -m pytest tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=120 2>&1 | tail -3 | tee "$OUT/pytest_rowmlp_helpers.log"
    gate "$OUT/pytest_rowmlp_helpers.log" "per-launch parity (GCAST_HELPERS=1)"
    timeout 300 python -m pytest tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=120 2>&1 | tail -3 | tee "$OUT/pytest_rowmlp.log"
    gate "$OUT/pytest_rowmlp.log" "per-launch parity"
    GCAST_HELPERS=1 timeout 400 python -m pytest tests/test_step_gpu.py tests/test_rollout_gpu.py -m gpu -q -x --timeout=200 2>&1 | tail -3 | tee "$OUT/pytest_step_helpers.log"
    gate "$OUT/pytest_step_helpers.log" "step parity (GCAST_HELPERS=1)"
    bash scripts/session.sh bench-ab r05_s7 "GCAST_HELPERS=0" "GCAST_HELPERS=1 GCAST_HELPER_STORE=0" "GCAST_HELPERS=1" "GCAST_HELPERS=0"
    ;;
  s8)
    # Round-5 session 8: upper bound of "the staging waves gather the next tile's addends" -- a profiling library whose
    # multiplying waves skip the gather (results wrong), processor-edge shape, eight-wave form with HST, against the
    # shipped library in both forms.
    GCAST_HELPERS=1 HALF_BUILDS="nogather:-DGC_H_NOGATHER=1" PROBE_SHAPES=proc_edge timeout 400 python -u scripts/half_probe.py --rounds 3 --iters 10 --out "$OUT/probe_helpers.json" 2>&1 | grep -v amdgpu.ids | cut -c1-900 | tail -3
    GCAST_HELPERS=0 PROBE_SHAPES=proc_edge timeout 300 python -u scripts/half_probe.py --rounds 3 --iters 10 --out "$OUT/probe_pair.json" 2>&1 | grep -v amdgpu.ids | cut -c1-500 | tail -2
    ;;
  s9)
    # Round-5 session 9: HST == 2 (the staging waves gather the next tile's addends): the processor-edge launch in both
    # kernel forms, bit for bit, under a short timeout (a barrier mismatch between the roles would hang the launch).
    timeout 240 python -m pytest tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=100 -k "processor_edge_update_in_both" 2>&1 | tail -15 | cut -c1-300 | tee "$OUT/pytest_proc_edge.log"
    ;;
  s10)
    # Round-5 session 10: HST == 2 on the whole step: parity gates with every launch in the eight-wave form, then the A/B.
    GCAST_HELPERS=1 timeout 300 python -m pytest tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=120 2>&1 | tail -3 | tee "$OUT/pytest_rowmlp_helpers.log"
    gate "$OUT/pytest_rowmlp_helpers.log" "per-launch parity (GCAST_HELPERS=1)"
    GCAST_HELPERS=1 timeout 400 python -m pytest tests/test_step_gpu.py tests/test_rollout_gpu.py tests/test_plan_gpu.py -m gpu -q -x --timeout=200 2>&1 | tail -3 | tee "$OUT/pytest_step_helpers.log"
    gate "$OUT/pytest_step_helpers.log" "step parity (GCAST_HELPERS=1)"
    bash scripts/session.sh bench-ab r05_s10 "GCAST_HELPERS=0" "GCAST_HELPERS=1 GCAST_HELPER_STORE=1" "GCAST_HELPERS=1" "GCAST_HELPERS=0" "GCAST_HELPERS=1"
    ;;
  s11)
    # Round-5 session 11: HST == 2 without its hand-over barrier, shipped by default on the processor edge update: parity,
    # then A/B against the four-wave form, and the wave priorities again now that the staging waves execute VALU work.
    timeout 300 python -m pytest tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=120 2>&1 | tail -3 | tee "$OUT/pytest_rowmlp.log"
    gate "$OUT/pytest_rowmlp.log" "per-launch parity"
    GCAST_HELPERS=1 timeout 300 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py -m gpu -q -x --timeout=120 2>&1 | tail -3 | tee "$OUT/pytest_helpers.log"
    gate "$OUT/pytest_helpers.log" "parity (GCAST_HELPERS=1)"
    bash scripts/session.sh bench-ab r05_s11 "GCAST_HELPERS_EDGE=0" "GCAST_HELPERS_EDGE=1" "GCAST_HELPERS_EDGE=1 GCAST_PRIO=3,0,0" "GCAST_HELPERS_EDGE=1 GCAST_PRIO=2,0,0" "GCAST_HELPERS_EDGE=0" "GCAST_HELPERS_EDGE=1"
    ;;
  s12)
    # Round-5 session 12: is the step power-bound?  Socket power / cap / shader clock polled while the step runs.
    which amd-smi rocm-smi
    timeout 300 python scripts/power_probe.py --seconds 6 --out "$OUT/power_probe.json" 2>&1 | grep -v amdgpu.ids | cut -c1-1500 | tail -4
    ;;
  s13)
    # Round-5 session 13: the shader clock per stage (SQ_WAVE_CYCLES / duration of the persistent launches) with the
    # processor edge update as two four-wave workgroups per CU and in the eight-wave HST == 2 form: what the faster launch
    # does to the one behind it.  (One --pmc pass each, kernel trace only: the node-safe combination.)
    for E in 0 1; do
      (cd /tmp && GCAST_HELPERS_EDGE=$E timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES --output-format csv -d "$OLDPWD/$OUT/pmc_edge$E" -o pmc -- \
          python "$OLDPWD/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-cross-check --rollout-steps 0 --op-timing-iters 1 > "$OLDPWD/$OUT/bench_edge$E.json" 2> "$OLDPWD/$OUT/bench_edge$E.err"); echo "pmc edge=$E rc=$?"
      python scripts/clock_by_stage.py "$OUT/pmc_edge$E" > "$OUT/clock_by_stage_edge$E.json" 2>> "$OUT/errors.txt"
      python -c "
import json; j=json.load(open('$OUT/clock_by_stage_edge$E.json')); print({k: (v['clock_ghz'], v['ms_per_launch_under_the_counter_pass']) for k, v in j.items()})"
    done
    find "$OUT" -type f -size +8M -delete
    ;;
  s14)
    # Round-5 session 14: the eight-wave HST forms where launches are SMALL (an 8-way rank's: 640 edge tiles, 80 node
    # tiles -- not power-bound): the emulated 8-way partition and the 1 deg step with GCAST_HELPERS_EDGE=0|1; and the
    # bf16 tier's power draw.
    for E in 0 1; do
      GCAST_HELPERS_EDGE=$E timeout 600 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8_edge$E.json" 2>&1 | grep -v amdgpu | tail -1 | cut -c1-330
    done
    bash scripts/session.sh bench-ab r05_s14 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --rollout-steps 0 --config 1deg_13L_M5 -- "GCAST_HELPERS_EDGE=0" "GCAST_HELPERS_EDGE=1" "GCAST_HELPERS_EDGE=0" "GCAST_HELPERS_EDGE=1"
    ;;
  s15)
    # Round-5 session 15 (after the round-end session on the library with the edge rule): (a) the driver's bench command
    # with the counter summaries of THIS library in profiles/current_* -- the line carries roofline.traffic / .pmc;
    # (b) the step / launch / plan / partition / rollout suites with the processor edge update pinned to each kernel
    # form (GCAST_HELPERS_EDGE=1: eight-wave HST form at every size; =0: the pair everywhere) -- the default rule mixes
    # them by size; (c) power / clock while the bf16 tier and the 1 deg step run: which of them are power-bound.
    timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; show "$OUT/bench.json"
    python - "$OUT/bench.json" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["roofline"]
print("traffic", r.get("traffic"), "frac", round(r["frac"], 4), "pmc", str(r.get("pmc"))[:200])
PY
    for E in 1 0; do
      GCAST_HELPERS_EDGE=$E timeout 900 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_partition_gpu.py tests/test_rollout40_gpu.py tests/test_fullsize_gpu.py \
          -m gpu -q -x --timeout=600 > "$OUT/pytest_edge$E.log" 2>&1; echo "pytest GCAST_HELPERS_EDGE=$E rc=$?"; tail -2 "$OUT/pytest_edge$E.log"
      gate "$OUT/pytest_edge$E.log" "suite with GCAST_HELPERS_EDGE=$E"
    done
    timeout 300 python scripts/power_probe.py --precision bf16 --out "$OUT/power_probe_bf16.json" | head -1 | cut -c1-400
    timeout 300 python scripts/power_probe.py --config 1deg_13L_M5 --out "$OUT/power_probe_1deg.json" | head -1 | cut -c1-400
    timeout 300 python scripts/power_probe.py --out "$OUT/power_probe_f16x3.json" | head -1 | cut -c1-400
    # (d) scripts/ubench/rows_per_fragment.hip: the same MFMA work per CU with half the weight stream / half the LDS
    # fragment reads / half the operand reads, at the power limit (what comes next)
    hipcc --offload-arch=gfx950 -O3 -w scripts/ubench/rows_per_fragment.hip -o /tmp/rows_per_fragment && \
      for V in 0 1 2 3 0 2 3; do timeout 60 /tmp/rows_per_fragment $V; done | tee "$OUT/rows_per_fragment.jsonl"
    ;;
  s16)
    # Round-5 session 16: the SHIPPED default (no switches) where the edge rule applies -- the 1 deg step with a kernel
    # trace (which kernel each stage's launches are), the emulated 8-way partition of the 0.25 deg step -- against
    # GCAST_HELPERS_EDGE=0, same session.
    bash scripts/session.sh bench-ab r05_s16 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --rollout-steps 0 --config 1deg_13L_M5 -- "" "GCAST_HELPERS_EDGE=0" "" "GCAST_HELPERS_EDGE=0"
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_1deg" -o trace -- \
        python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-cross-check --rollout-steps 0 --op-timing-iters 1 --config 1deg_13L_M5 > "$OLDPWD/$OUT/prof_1deg_bench.json" 2> "$OLDPWD/$OUT/prof_1deg.err"); echo "rocprof rc=$?"
    f=$(find "$OUT/prof_1deg" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cut -c1-240 "$f" | head -12 | tee "$OUT/kernel_stats_1deg_head.csv"; }
    for E in default 0; do
      if [ "$E" = default ]; then unset GCAST_HELPERS_EDGE; else export GCAST_HELPERS_EDGE=$E; fi
      timeout 600 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8_edge_$E.json" 2>&1 | grep -v amdgpu | tail -1 | cut -c1-330
    done
    unset GCAST_HELPERS_EDGE
    find "$OUT" -type f -size +8M -delete
    ;;
  s17)
    # Round-5 session 17: the WIDE form (rowmlp16w_kernel: eight multiplying waves per CU on one weight ring, GC_WG_WIDE /
    # GCAST_WIDE=1 for the launches the plan marks GC_WG_HELPERS) -- bit-identity per launch, then the step A/B.
    timeout 500 python -m pytest tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=120 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log" | cut -c1-300
    gate "$OUT/pytest.log" "per-launch suite with the wide form"
    bash scripts/session.sh bench-ab r05_s17 "" "GCAST_WIDE=1" "" "GCAST_WIDE=1"
    GCAST_WIDE=1 timeout 600 python -m pytest tests/test_step_gpu.py tests/test_plan_gpu.py -m gpu -q -x --timeout=300 > "$OUT/pytest_step_wide.log" 2>&1; echo "pytest step (GCAST_WIDE=1) rc=$?"; tail -2 "$OUT/pytest_step_wide.log" | cut -c1-300
    ;;
  s18)
    # Round-5 session 18: the wide form as the plan's DEFAULT for the big node-side launches -- gate (smoke, per-launch,
    # step and plan suites), the step against GCAST_WIDE=0, the 1 deg step with the row threshold lowered to its
    # 65,160-row grid launches, the emulated 8-way partition against GCAST_WIDE=0.
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -4
    timeout 600 python -m pytest tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_plan_gpu.py -m gpu -q -x --timeout=300 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log" | cut -c1-300
    gate "$OUT/pytest.log" "suites with the wide default"
    bash scripts/session.sh bench-ab r05_s18 "" "GCAST_WIDE=0" "" "GCAST_WIDE=0"
    bash scripts/session.sh bench-ab r05_s18_1deg --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --rollout-steps 0 --config 1deg_13L_M5 -- "" "GCAST_HELPERS_MIN_ROWS=32768" "" "GCAST_HELPERS_MIN_ROWS=32768" "GCAST_HELPERS_MIN_ROWS=32768 GCAST_WIDE=0"
    for E in default 0; do
      if [ "$E" = default ]; then unset GCAST_WIDE; else export GCAST_WIDE=$E; fi
      timeout 600 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8_wide_$E.json" 2>&1 | grep -v amdgpu | tail -1 | cut -c1-330
    done
    unset GCAST_WIDE
    ;;
  *) echo "unknown session $NAME"; exit 2;;
esac
