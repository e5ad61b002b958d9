#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + HBM PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <tag> [bench args...]     outputs under gpurun_out/prof_<tag>/
# Each rocprofv3 run profiles `python3 bench.py` directly (no env/bash hop after `--`).
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# bench.py's own defaults (20 warm-up + 100 timed steps): the profile is of the SAME command
# --no-sustained: the sustained leg starts rocm-smi as a child; no child process under a profiler preload
ARGS="--no-cpu-baseline --no-sustained $*"

timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o trace --output-format csv -- \
    python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.log" 2>&1
# separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md)
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o pmc --output-format csv -- \
    python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o pmc --output-format csv -- \
    python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc_write.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace \
    -d "$OUT/pmc_sq" -o pmc --output-format csv -- \
    python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_pmc_sq.log" 2>&1

cd "$ROOT"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.json" 2> "$OUT/summary.err"
cat "$OUT/summary.json"
# keep only small files in gpurun_out
find "$OUT" -name "*.csv" -size +4M -delete
