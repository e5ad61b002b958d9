#!/bin/bash
# Run ON THE GPU BOX (via gpurun): every workload of DESIGN.md §5's table profiled at ONE build of
# the library — kernel trace + stats, the 3-counter VALU pass, and the FETCH_SIZE / WRITE_SIZE
# passes, each a separate rocprofv3 run of THE SAME `python3 bench.py ...` command (no env/bash hop
# after `--`; --pmc never together with a trace domain other than --kernel-trace).
#   tools/profile_all.sh <tag> [workload-name ...]     -> gpurun_out/prof_<tag>_<name>/summary.json
# Copy the summaries to profiles/<tag>_<name>.json and fold them into the tables bench.py reads with
# tools/update_profile_tables.py profiles/<tag>_*.json (entries carry the library's source hash).
set -u
TAG=${1:-r4a}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
declare -A ARGS=(
  [SoftPendulum-v0]=""
  [SoftPendulum3D-v0]="--env SoftPendulum3D-v0"
  [OctoArmSingle-v0]="--env OctoArmSingle-v0"
  [OctoArmSingle-v0_n100]="--env OctoArmSingle-v0 --n-elems 100"
  [OctoFlat-v0]="--env OctoFlat-v0"
  [OctoArmSingle-v0_taper]="--env OctoArmSingle-v0 --taper"
  [SoftArmTracking-v0]="--env SoftArmTracking-v0"
  [OctoArmPush-v1]="--env OctoArmPush-v1"
  [SoftPendulum-v0_libm]="--math-mode libm"
  [OctoArmPush-v0]="--env OctoArmPush-v0"
  [OctoArmPullWeight-v0]="--env OctoArmPullWeight-v0"
  [OctoCrawl-v0]="--env OctoCrawl-v0"
  [OctoArmTwo-v0]="--env OctoArmTwo-v0"
  [OctoReach-v0]="--env OctoReach-v0"
)
NAMES=${*:-SoftPendulum-v0 SoftPendulum3D-v0 OctoArmSingle-v0 OctoArmSingle-v0_n100 OctoArmSingle-v0_taper OctoFlat-v0 SoftArmTracking-v0 OctoArmPush-v1 SoftPendulum-v0_libm OctoArmPullWeight-v0 OctoCrawl-v0}
for NAME in $NAMES; do
  A="--no-cpu-baseline --no-secondary ${ARGS[$NAME]}"
  OUT=$ROOT/gpurun_out/prof_${TAG}_$NAME
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- \
      python3 "$ROOT/bench.py" $A > "$OUT/bench_trace.log" 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc3" -o pmc --output-format csv -- \
      python3 "$ROOT/bench.py" $A > "$OUT/bench_pmc3.log" 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o pmc --output-format csv -- \
      python3 "$ROOT/bench.py" $A > "$OUT/bench_pmc_fetch.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o pmc --output-format csv -- \
      python3 "$ROOT/bench.py" $A > "$OUT/bench_pmc_write.log" 2>&1
  ( cd "$ROOT" && python3 tools/summarize_profile.py "$OUT" $A > "$OUT/summary.json" 2> "$OUT/summary.err" )
  find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \; 2>/dev/null
  python3 - "$OUT/summary.json" "$NAME" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
p = d.get("pmc3", {})
print(sys.argv[2], "hash", d.get("library_source_hash"), "kernel ms (median window, trace)", d.get("step_kernel_timed_avg_ms"),
      "bench", d.get("bench_kernel_ms_avg_same_run"), "VALU/rod-substep", p.get("valu_instr_per_rod_substep"),
      "busy", p.get("valu_busy_frac"), "HBM MB", (d.get("hbm_bytes_per_launch") or 0) / 1e6)
PY
  find "$OUT" -name "*.csv" -size +2M -delete
  find "$OUT" -name "*.db" -delete
done
