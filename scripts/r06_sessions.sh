#!/bin/bash
# The GPU-box sessions of round 6 in ONE file: `bash scripts/r06_sessions.sh <name>` runs one of them from the repository
# root; results land under gpurun_out/r06_<name>/, what is kept is copied to profiles/r06_<name>_*.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
NAME=${1:?session name}
OUT=gpurun_out/r06_$NAME; mkdir -p "$OUT"
export TMPDIR=/tmp
gate() { grep -q " passed" "$1" && ! grep -q "failed\|rror\|Timeout" "$1" || { echo "GATE: $2 failed"; tail -40 "$1"; exit 1; }; }
show() { python - "$1" <<'PY'
import json, sys
try:
  j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
  print("no bench line:", e); sys.exit(0)
print(round(j["ms_per_step"], 3), {k: round(v["ms"], 3) for k, v in (j.get("roofline") or {}).get("stages", {}).items()})
for k in ("rollout", "rollout_api"):
  if j.get(k): print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in j[k].items() if a != "what"})
print("tuning", j.get("tuning"))
PY
}
case "$NAME" in
  s1)
    # Round-6 session 1: the wide form widened -- ten of sixteen parked n-blocks in LDS (GC_W_PARK_NB; A/B library with
    # 0 = round 5's parking), segment-sum launches and the one-pass edge updates in the form (gc_tuning.wide_edges /
    # GCAST_WIDE_EDGES: bit 0 one-pass, bit 1 two-pass) -- bit-identity first, then same-session A/B of the whole step.
    timeout 1500 python -m pytest tests/test_native_abi.py tests/test_rowmlp_gpu.py tests/test_step_gpu.py -m gpu -q -x --timeout=900 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "wide form with segment-sum / one-pass / LDS parking"
    bash scripts/session.sh bench-ab r06_s1 "GCAST_WIDE_EDGES=0" "GCAST_WIDE_EDGES=1" "GCAST_WIDE_EDGES=2" "GCAST_WIDE_EDGES=3" \
        "GCAST_LIB_PATH=ab_libs/libgcast_wpark0.so" "GCAST_WIDE_EDGES=0" "GCAST_WIDE_EDGES=1" "GCAST_LIB_PATH=ab_libs/libgcast_wpark0.so"
    ;;
  s2)
    # Round-6 session 2: the host-side changes on the GPU -- the fused rollout as an opt-in (as_predictor_fn / fuse, first +
    # last chunk cross-checks on device-resident Datasets), pmap_devices (two engines on one GPU), per-plan tuning, the
    # plan against the reference-executed fixture; then the bench line with value_rollout / rollout_api, the
    # single-process launch form, and more same-session samples of gc_tuning.wide_edges.
    timeout 1500 python -m pytest tests/test_native_abi.py tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_rollout_gpu.py -m gpu -q -x --timeout=900 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "host-side changes"
    timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; show "$OUT/bench.json"
    timeout 900 python bench.py --gpus 2 --single-process --steps 10 --warmup 3 > "$OUT/bench_single_process_2.json" 2> "$OUT/bench_single_process_2.err"; echo "single-process rc=$?"; show "$OUT/bench_single_process_2.json"
    bash scripts/session.sh bench-ab r06_s2 "GCAST_WIDE_EDGES=0" "GCAST_WIDE_EDGES=1" "GCAST_WIDE_EDGES=3" "GCAST_WIDE_EDGES=0" "GCAST_WIDE_EDGES=1" "GCAST_WIDE_EDGES=3"
    ;;
  s3)
    # Round-6 session 3: gc_tuning.wide_edges = 3 as the shipped default (every edge update of >= 4096 tiles in the wide
    # form) -- the default against its parts and against the other switches that could interact with an all-wide step;
    # then the sizes the headline does not show: the 1 deg step, an 8-way rank (emulated), the bf16 tier.
    bash scripts/session.sh bench-ab r06_s3 "GCAST_WIDE_EDGES=3" "GCAST_WIDE_EDGES=0" "GCAST_WIDE_EDGES=2" "GCAST_WIDE_EDGES=3 GCAST_PRIO=0,0,0" \
        "GCAST_WIDE_EDGES=3 GCAST_TILE_QUEUE=0" "GCAST_WIDE_EDGES=3" "GCAST_WIDE_EDGES=0"
    timeout 600 python bench.py --config 1deg_13L_M5 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --rollout-steps 0 > "$OUT/bench_1deg.json" 2> "$OUT/bench_1deg.err"; echo "1deg rc=$?"; show "$OUT/bench_1deg.json"
    timeout 600 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8.json" 2>&1 | tail -2 | cut -c1-700
    timeout 600 python bench.py --precision bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; echo "bf16 rc=$?"; show "$OUT/bench_bf16.json"
    ;;
  s4)
    # Round-6 session 4: gc_tuning.split_tail -- the processor's node updates (641 tiles = 1.6 rounds of four-wave pairs) as
    # one full round of wide tiles + a helper-form tail: bit-identity (per launch, and the full-size step's properties),
    # then same-session A/B.
    timeout 1500 python -m pytest tests/test_native_abi.py tests/test_rowmlp_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout=900 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "split tail"
    bash scripts/session.sh bench-ab r06_s4 "GCAST_SPLIT_TAIL=1" "GCAST_SPLIT_TAIL=0" "GCAST_SPLIT_TAIL=1" "GCAST_SPLIT_TAIL=0" "GCAST_SPLIT_TAIL=1 GCAST_PRIO=0,0,0"
    ;;
  s5)
    # Round-6 session 5: the Bfloat16Cast tier's streamed edge updates (gc_tuning.bf16_stream: launches without a layer-1
    # GEMM form every K step's hidden pair on the fly) -- bit-identity with the unstreamed launch and the tier's oracle
    # tests first (run under a timeout of their own: new counted waits), then same-session A/B of the tier's step, plus
    # the 128-row workgroups on every launch as a reference point.
    timeout 900 python -m pytest tests/test_bf16_tier_gpu.py tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=300 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "bf16 stream"
    bash scripts/session.sh bench-ab r06_s5 --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 --precision bf16 -- \
        "GCAST_BF16_STREAM=1" "GCAST_BF16_STREAM=0" "GCAST_BF16_STREAM=1" "GCAST_BF16_STREAM=0" "GCAST_BF16_STREAM=1 GCAST_BF16_ROWS=128"
    timeout 900 python -m pytest tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_rollout_gpu.py tests/test_partition_gpu.py -m gpu -q -x --timeout=600 > "$OUT/pytest2.log" 2>&1; echo "pytest2 rc=$?"; tail -3 "$OUT/pytest2.log" | cut -c1-400
    ;;
  s6)
    # Round-6 session 6: the bf16 tier's LATE addends (GC_LATE_ADDENDS: the processor edge update adds its gathered rows
    # when the hidden layer is formed) -- the tier's oracle tests (per launch, whole step, rollout, partition), then
    # same-session A/B of gc_tuning.bf16_stream = 0 | 1 | 3.
    timeout 900 python -m pytest tests/test_bf16_tier_gpu.py -m gpu -q -x --timeout=300 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "bf16 late addends"
    grep "late addends" "$OUT/pytest.log" | cut -c1-200
    bash scripts/session.sh bench-ab r06_s6 --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 --precision bf16 -- \
        "GCAST_BF16_STREAM=3" "GCAST_BF16_STREAM=1" "GCAST_BF16_STREAM=0" "GCAST_BF16_STREAM=3" "GCAST_BF16_STREAM=1"
    timeout 900 python -m pytest tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_rollout_gpu.py tests/test_partition_gpu.py -m gpu -q -x --timeout=600 > "$OUT/pytest2.log" 2>&1; echo "pytest2 rc=$?"; tail -3 "$OUT/pytest2.log" | cut -c1-400
    ;;
  s7)
    # Round-6 session 7: late addends -- the bf16 tier's processor edge update (its own kernel instantiation) and, as an
    # experiment, the f16x3 wide form's (gc_tuning.wide_late; not bit-identical to the four-wave kernel): tests, then A/B.
    timeout 900 python -m pytest tests/test_native_abi.py tests/test_bf16_tier_gpu.py tests/test_rowmlp_gpu.py -m gpu -q -x --timeout=300 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "late addends"
    bash scripts/session.sh bench-ab r06_s7b --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 --precision bf16 -- \
        "GCAST_BF16_STREAM=3" "GCAST_BF16_STREAM=1" "GCAST_BF16_STREAM=0" "GCAST_BF16_STREAM=3" "GCAST_BF16_STREAM=1"
    bash scripts/session.sh bench-ab r06_s7 "GCAST_WIDE_LATE=0" "GCAST_WIDE_LATE=1" "GCAST_WIDE_LATE=0" "GCAST_WIDE_LATE=1"
    timeout 900 python -m pytest tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_rollout_gpu.py tests/test_partition_gpu.py -m gpu -q -x --timeout=600 > "$OUT/pytest2.log" 2>&1; echo "pytest2 rc=$?"; tail -3 "$OUT/pytest2.log" | cut -c1-400
    ;;
  s8)
    # Round-6 session 8: GC_LATE_ADDENDS as a property of the launch (four-wave and wide form: the same bits), gc_tuning.wide_late
    # = 1 as the shipped default; the bf16 tier without its LATE variant (measured slower, removed).
    timeout 1500 python -m pytest tests/test_native_abi.py tests/test_rowmlp_gpu.py tests/test_bf16_tier_gpu.py tests/test_step_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout=900 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "late addends as a launch property"
    grep "late addends" "$OUT/pytest.log" | cut -c1-200
    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -3 "$OUT/smoke.log"
    bash scripts/session.sh bench-ab r06_s8 "GCAST_WIDE_LATE=1" "GCAST_WIDE_LATE=0" "GCAST_WIDE_LATE=1" "GCAST_WIDE_LATE=0 GCAST_WIDE_EDGES=0"
    ;;
  s9)
    # Round-6 session 9: where a WIDE tile's time goes -- wave-0 phase traces (GC_H_TRACE profiling build) of the processor
    # edge update in the pair form, the wide form and the wide form with late addends; of the one-pass decoder edge update
    # and the grid node update in the wide form; and what the layer-1 row loads' memory latency costs in the wide form
    # (the same GEMM-only launch on streamed rows / on 64 cached rows).
    for F in 0 256 768; do
      HALF_TRACE=1 PROBE_FLAGS=$F PROBE_SHAPES=proc_edge timeout 300 python -u scripts/half_probe.py --out "$OUT/trace_proc_edge_flags$F.json" 2>&1 | grep htrace | cut -c1-900
    done
    HALF_TRACE=1 PROBE_FLAGS=256 PROBE_SHAPES=dec_edge_onepass,node_grid timeout 300 python -u scripts/half_probe.py --out "$OUT/trace_wide_onepass_node.json" 2>&1 | grep htrace | cut -c1-900
    PROBE_FLAGS=256 PROBE_SHAPES=gemm_only_mlp,gemm_only_cached_rows,node_grid timeout 300 python -u scripts/half_probe.py --rounds 2 --iters 10 --out "$OUT/probe_wide_rows.json" 2>&1 | grep -v amdgpu | cut -c1-500 | tail -4
    ;;
  s10)
    # Round-6 session 10: the epilogue of the late-addend kernels read from the disassembly -- no scratch reload (each one a
    # full vmcnt drain) between a segment-sum scan's runs, between layer 2's passes, inside LayerNorm; receiver ids prefetched
    # in the prologue; residual + store through the staged tile (full 1 KiB lines, scalar row bases); LDS barriers without
    # the global fence; LayerNorm's sums as packed f4 operations.  Tests first, then same-session A/B against the library of
    # commit e8bb181 (ab_libs/libgcast_r6head.so), then the wave-0 phase trace of the processor edge update.
    timeout 1500 python -m pytest tests/test_native_abi.py tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout=900 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "epilogue rewrite"
    bash scripts/session.sh bench-ab r06_s10 "" "GCAST_LIB_PATH=ab_libs/libgcast_r6head.so" "" "GCAST_LIB_PATH=ab_libs/libgcast_r6head.so"
    for F in 256 768; do
      HALF_TRACE=1 PROBE_FLAGS=$F PROBE_SHAPES=proc_edge timeout 300 python -u scripts/half_probe.py --out "$OUT/trace_proc_edge_flags$F.json" 2>&1 | grep htrace | cut -c1-900
    done
    ;;
  s11)
    # Round-6 session 11: the prologue of the two-pass launches -- both gather indices requested in front of the ring's first
    # pieces (they were two dependent round trips, each behind a full drain), b1 from LDS for launches without a chain, the
    # tile queue's atomic in front of the prologue's drain instead of at the top of the tile.  Tests, then A/B against the
    # library of session s10 (commit 2e20a7b, ab_libs/libgcast_s10.so).
    timeout 1500 python -m pytest tests/test_native_abi.py tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout=900 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "prologue rewrite"
    bash scripts/session.sh bench-ab r06_s11 "" "GCAST_LIB_PATH=ab_libs/libgcast_s10.so" "" "GCAST_LIB_PATH=ab_libs/libgcast_s10.so"
    ;;
  s12)
    # Round-6 session 12: no thread-id value of the top of the tile used behind layer 1 (bias addresses of layer 2's and the
    # chain's passes, chained stores, the row-owner epilogue: each was a scratch reload + ring drain).  Tests, A/B vs s11.
    timeout 1500 python -m pytest tests/test_native_abi.py tests/test_rowmlp_gpu.py tests/test_step_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout=900 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "fresh lane values behind layer 1"
    bash scripts/session.sh bench-ab r06_s12 "" "GCAST_LIB_PATH=ab_libs/libgcast_s11.so" "" "GCAST_LIB_PATH=ab_libs/libgcast_s11.so"
    ;;
  s13)
    # Round-6 session 13: the bf16 tier's epilogue after the f16x3 kernels' (ids / flags prefetched in the prologue, ballot
    # scan with scalar-base run stores, LDS barriers, packed LayerNorm sums, lane values formed where they are used).
    # The tier's tests, then A/B of its step against the library of session s12.
    timeout 1200 python -m pytest tests/test_bf16_tier_gpu.py tests/test_rowmlp_gpu.py tests/test_native_abi.py -m gpu -q -x --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "bf16 epilogue"
    bash scripts/session.sh bench-ab r06_s13 --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 --precision bf16 -- \
        "" "GCAST_LIB_PATH=ab_libs/libgcast_s12.so" "" "GCAST_LIB_PATH=ab_libs/libgcast_s12.so"
    timeout 900 python -m pytest tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_rollout_gpu.py tests/test_partition_gpu.py -m gpu -q -x --timeout=600 > "$OUT/pytest2.log" 2>&1; echo "pytest2 rc=$?"; tail -3 "$OUT/pytest2.log" | cut -c1-400
    ;;
  s14)
    # Round-6 session 14: bf16 tier -- the plain launch without a chain as an instantiation of its own (VAR 2: no spills),
    # both gather indices requested at once.  The tier's tests, A/B against the library of session s13, whole-step tests.
    timeout 1200 python -m pytest tests/test_bf16_tier_gpu.py tests/test_rowmlp_gpu.py tests/test_native_abi.py -m gpu -q -x --timeout=600 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "bf16 no-chain instantiation"
    bash scripts/session.sh bench-ab r06_s14 --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 --precision bf16 -- \
        "" "GCAST_LIB_PATH=ab_libs/libgcast_s13.so" "" "GCAST_LIB_PATH=ab_libs/libgcast_s13.so"
    timeout 900 python -m pytest tests/test_step_gpu.py tests/test_plan_gpu.py tests/test_rollout_gpu.py tests/test_partition_gpu.py -m gpu -q -x --timeout=600 > "$OUT/pytest2.log" 2>&1; echo "pytest2 rc=$?"; tail -3 "$OUT/pytest2.log" | cut -c1-400
    ;;
  s15)
    # Round-6 session 15: the small sizes on the library of session s14 -- the 1 deg step and an emulated 8-way rank, each
    # with the shipped form rules and with the processor edge update pinned away from the helper form (GCAST_HELPERS_EDGE=0:
    # four-wave pairs), and against the library of commit e8bb181.
    for E in "" "GCAST_HELPERS_EDGE=0" "GCAST_LIB_PATH=ab_libs/libgcast_r6head.so"; do
      tag=$(echo "${E:-default}" | tr ' =/' '__-')
      env $E timeout 600 python bench.py --config 1deg_13L_M5 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --rollout-steps 0 > "$OUT/bench_1deg_$tag.json" 2> "$OUT/bench_1deg_$tag.err"; echo "1deg [$E] rc=$?"; show "$OUT/bench_1deg_$tag.json"
      env $E timeout 600 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8_$tag.json" 2>&1 | tail -1 | cut -c1-600
    done
    ;;
  s16)
    # Round-6 session 16: gc_tuning.split_edges (the small graphs' processor edge update: full rounds of 256 wide tiles in the
    # wide form + a helper-form launch for a remainder of at most one tile per CU) and the emulated partition's exchange as
    # what a rank does (one packing + one landing index_select per rank and exchange, indices resident).  Tests, then the
    # 1 deg step and an emulated 8-way rank with the rule on / off, alternating; one headline-size line for reference.
    if [ "${SKIP_TESTS:-0}" != "1" ]; then
      timeout 1500 python -m pytest tests/test_native_abi.py tests/test_rowmlp_gpu.py tests/test_partition_gpu.py tests/test_step_gpu.py -m gpu -q -x --timeout=900 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
      gate "$OUT/pytest.log" "split_edges / LocalExchanger"
    fi
    for E in "" "GCAST_SPLIT_EDGES=0" "" "GCAST_SPLIT_EDGES=0"; do
      i=$((${i:-0} + 1)); tag=$(echo "${E:-default}" | tr ' =/' '__-')
      env $E timeout 600 python bench.py --config 1deg_13L_M5 --steps 20 --warmup 5 --no-cpu-baseline --no-cross-check --rollout-steps 0 > "$OUT/bench_1deg_${i}_$tag.json" 2> "$OUT/bench_1deg_${i}_$tag.err"; echo "1deg [$E] rc=$?"; show "$OUT/bench_1deg_${i}_$tag.json"
      env $E timeout 600 python scripts/partition_emulated_bench.py --parts 8 --out "$OUT/partition8_${i}_$tag.json" 2>&1 | tail -1 | cut -c1-900
    done
    timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 > "$OUT/bench_full.json" 2> "$OUT/bench_full.err"; echo "full rc=$?"; show "$OUT/bench_full.json"
    ;;
  s17)
    # Round-6 session 17: where the 1 deg step's time goes BETWEEN its kernels (session s16: a second launch per edge update
    # cost the step 0.3 ms where the launch itself lost 0.05) -- a kernel trace of the 1 deg step and of an emulated 8-way
    # step, gaps grouped by (previous kernel -> next kernel).
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/prof_1deg" -o trace -- \
        python "$OLDPWD/bench.py" --config 1deg_13L_M5 --steps 10 --warmup 2 --no-cpu-baseline --no-cross-check --rollout-steps 0 --op-timing-iters 1 > "$OLDPWD/$OUT/prof_1deg.json" 2> "$OLDPWD/$OUT/prof_1deg.err"); echo "rocprof rc=$?"
    python scripts/kernel_gaps.py "$OUT/prof_1deg" --steps 4 --skip-tail 3 > "$OUT/kernel_gaps_1deg.json"; head -c 3000 "$OUT/kernel_gaps_1deg.json"
    find "$OUT" -type f -size +8M -delete
    ;;
  s18)
    # Round-6 session 18: the shipped library once more on another box -- a repeatability stress (the same step N times, every
    # output bitwise against the first, other work beside every second repetition; both precisions, 1 deg and 0.25 deg) and the
    # driver's step count with the round-end session's stamped counters attached to the line (roofline.traffic / pmc).
    timeout 1200 python scripts/repeat_stress.py --reps 30 --out "$OUT/repeat_stress.json" 2>&1 | grep -v amdgpu.ids | tail -6; echo "stress rc=${PIPESTATUS[0]}"
    timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_counters_attached.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; show "$OUT/bench_counters_attached.json"
    python - "$OUT/bench_counters_attached.json" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print("traffic", r.get("traffic"), "frac", round(r["frac"], 4), "pmc", {k: v for k, v in (r.get("pmc") or {}).items() if k in ("mfma_busy_per_simd", "wave_waiting", "unavailable")})
PY
    ;;
  s19)
    # Round-6 session 19: session s18 found the bf16 tier NOT repeatable under the stress (4 of 120 runs at 1 deg, 5 of 30 at
    # 0.25 deg differ from the first; the f16x3 kernels 0 of 150).  Which library introduced it: the round's libraries in turn
    # (e8bb181 = before the second half, s12 = before the bf16 epilogue port, s13 = with it, s14 / in-tree = shipped), with
    # and without the side work.
    for L in "" ab_libs/libgcast_s13.so ab_libs/libgcast_s12.so ab_libs/libgcast_r6head.so; do
      tag=$(basename "${L:-in-tree}" .so)
      env ${L:+GCAST_LIB_PATH=$L} timeout 600 python scripts/repeat_stress.py --reps 40 --precisions bf16 --out "$OUT/stress_$tag.json" 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tail -3
    done
    timeout 600 python scripts/repeat_stress.py --reps 40 --precisions bf16 --no-side-work --out "$OUT/stress_in-tree_no_side_work.json" 2>&1 | grep -v amdgpu.ids | cut -c1-700 | tail -3
    ;;
  s20)
    # Round-6 session 20: which piece of the bf16 epilogue port (session s13) makes the tier non-repeatable at 0.25 deg --
    # libraries of the shipped sources with one piece taken back each: __syncthreads() where the port uses the LDS-only
    # barrier (GC_BF_SYNC=1), receiver ids / flags loaded between the segment-sum's barriers instead of in the prologue
    # (GC_BF_SEGPRE=0), both.
    for L in "" ${S20_LIBS:-ab_libs/libgcast_bf_sync.so ab_libs/libgcast_bf_segpre0.so ab_libs/libgcast_bf_both.so}; do
      tag=$(basename "${L:-in-tree}" .so)
      env ${L:+GCAST_LIB_PATH=$L} timeout 600 python scripts/repeat_stress.py --reps ${S20_REPS:-40} --precisions bf16 --configs 0.25deg_37L_M6 --no-side-work --out "$OUT/stress_$tag.json" 2>&1 | grep -v amdgpu.ids | cut -c1-400 | tail -2
    done
    ;;
  s21)
    # Round-6 session 21: the fix of the non-repeatable bf16 tier (the layer-1 loop's unconsumed row requests land before
    # their registers are anyone else's; the same guard for the eight-wave kernel's multiplying waves) -- the tier's and the
    # launch-level tests, the new repeatability test, then the stress at both sizes in both precisions.
    timeout 1800 python -m pytest tests/test_native_abi.py tests/test_rowmlp_gpu.py tests/test_bf16_tier_gpu.py tests/test_step_gpu.py tests/test_repeatability_gpu.py -m gpu -q -x --timeout=900 > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest.log" | cut -c1-400
    gate "$OUT/pytest.log" "landing guards"
    timeout 1200 python scripts/repeat_stress.py --reps 60 --out "$OUT/repeat_stress.json" 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -6; echo "stress rc=${PIPESTATUS[0]}"
    ;;
  s24|s26)
    # Round-6 sessions 24 / 26: ModelConfig.latent_size / hidden_layers other than 512 / 1 (csrc/gcast_plan.inc: pad_latent,
    # push_mlp) against the fp64 oracle in three precisions (s24: latent sizes dividing 512; s26: + 384, 320, 100, 200),
    # with the published model's step / plan tests beside them.
    timeout 1200 python -m pytest tests/test_general_sizes_gpu.py -q -m gpu -s 2>&1 | grep -v amdgpu.ids | grep -E "GENERAL|passed|failed" > "$OUT/general.log"; tail -40 "$OUT/general.log"
    timeout 900 python -m pytest tests/test_step_gpu.py tests/test_plan_gpu.py -q -m gpu 2>&1 | tail -3
    ;;
  s25|s28)
    # Round-6 sessions 25 / 28: the same sizes AT the headline graph -- f16x3 against the exact-fp32 kernel family, the bf16
    # tier against f16x3, bitwise repeatability, ms per step (s28: with the layer-1 K of narrow latents trimmed,
    # gc_plan.k_lat, and the rewritten input-tail kernel; + the prep tests and one bench line for the stage times).
    timeout 1400 python scripts/general_sizes_fullsize.py --cases "${CASES:-256x1,384x1,128x1,512x2,64x2,128x3}" --out "$OUT/general_sizes_fullsize.json" 2>&1 | grep -v amdgpu.ids | tail -8
    timeout 300 python -m pytest tests/test_rowmlp_gpu.py -q -m gpu -k prep 2>&1 | tail -2
    timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cross-check --rollout-steps 0 2>/dev/null | cut -c1-400
    ;;
  s27)
    # Round-6 session 27: the paths around the step on the library of final5 -- repeatability stress, emulated 8-way rank, 1 deg.
    timeout 1200 python scripts/repeat_stress.py --reps 40 --out "$OUT/repeat_stress.json" 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -6
    timeout 900 python scripts/partition_emulated_bench.py --parts 8 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-600
    timeout 600 python bench.py --config 1deg_13L_M5 --no-cpu-baseline --rollout-steps 0 2>/dev/null | cut -c1-300
    ;;
  *) echo "unknown session $NAME"; exit 2;;
esac
