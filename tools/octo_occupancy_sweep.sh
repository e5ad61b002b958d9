#!/bin/bash
# Run ON THE GPU BOX (via gpurun): OctoFlat-v0 at 1024 / 2048 / 4096 / 8192 envs on ONE GPU
# (BASELINE.json configs[4] is 8192 envs on 8 GPUs = 1024 per GPU): is the step kernel's VALU busy
# fraction at the 1024-env share limited by occupancy (2 waves / SIMD), i.e. does it rise with
# more resident envs?  Per point: a kernel-trace pass and the 3-counter VALU pass of the SAME
# `python3 bench.py ...` command (no env/bash hop after `--`).
#   tools/octo_occupancy_sweep.sh <tag> [envs ...]   -> gpurun_out/octosweep_<tag>_<envs>/summary.json
set -u
TAG=${1:-r5a}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
SIZES=${*:-1024 2048 4096 8192}
for N in $SIZES; do
  A="--no-cpu-baseline --no-secondary --env OctoFlat-v0 --envs-per-gpu $N --steps 5 --warmup 3 --windows 3"
  OUT=$ROOT/gpurun_out/octosweep_${TAG}_$N
  rm -rf "$OUT"; mkdir -p "$OUT"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- \
      python3 "$ROOT/bench.py" $A > "$OUT/bench_trace.log" 2>&1
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc3" -o pmc --output-format csv -- \
      python3 "$ROOT/bench.py" $A > "$OUT/bench_pmc3.log" 2>&1
  ( cd "$ROOT" && python3 tools/summarize_profile.py "$OUT" $A > "$OUT/summary.json" 2> "$OUT/summary.err" )
  python3 - "$OUT/summary.json" "$N" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
p = d.get("pmc3", {})
b = (d.get("bench_lines") or {}).get("bench_trace.log", {})
print("OctoFlat-v0 envs", sys.argv[2], "hash", d.get("library_source_hash"), "kernel ms", d.get("step_kernel_timed_avg_ms"),
      "env-steps/s", b.get("value"), "VALU/arm-substep", p.get("valu_instr_per_rod_substep"), "busy", p.get("valu_busy_frac"),
      "issue", p.get("valu_issue_frac_measured_cycles"))
PY
  find "$OUT" -name "*.csv" -size +2M -delete
  find "$OUT" -name "*.db" -delete
done
