#!/bin/bash
# A/B builds of the OctoFlat step kernel into variants/ (git-ignored; they travel to the GPU box), and — on
# the GPU box — their timing and VALU busy fraction on the configs[4] share (1024 envs):
#   tools/octo_ab.sh build                      (here: hipcc cross-compiles)
#   tools/octo_ab.sh run <tag>                  (on the box, via gpurun) -> gpurun_out/octo_ab_<tag>.txt
# Variants: r4 = the shipped kernel, basemask = joints and head step under the base-lane EXEC mask,
# diagN = SOFTROD_OCTO_DIAG=N (timing only: results are wrong by construction); round 6: r6 = the shipped kernel,
# thetaloop = round 5's large-angle tier of theta/sin(theta), waves3 = the kernel capped at 168 registers (what a ninth,
# coupling wave per workgroup would leave each arm wave: three waves on one SIMD).  Every run is bounded by
# `timeout`: a diagnostic build that breaks the rendezvous must not hold the box.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
declare -A FLAGS=(
  [r4]=""
  [basemask]="-DSOFTROD_OCTO_BASE_MASK=1"
  [diag4]="-DSOFTROD_OCTO_DIAG=4"
  [diag8]="-DSOFTROD_OCTO_DIAG=8"
  [diag12]="-DSOFTROD_OCTO_DIAG=12"
  [diag1]="-DSOFTROD_OCTO_DIAG=1"
  [diag2]="-DSOFTROD_OCTO_DIAG=2"
  [diag7]="-DSOFTROD_OCTO_DIAG=7"
  [nolds]="-DSOFTROD_OCTO_CONTACT_LDS=0"
  [diag7_nolds]="-DSOFTROD_OCTO_DIAG=7 -DSOFTROD_OCTO_CONTACT_LDS=0"
  [nomaskbase]="-DSOFTROD_OCTO_BASE_MASK=0"
  [maskjoints]="-DSOFTROD_OCTO_BASE_MASK=1"
  [maskhead]="-DSOFTROD_OCTO_BASE_MASK=2"
  [r6]=""
  [thetaloop]="-DSOFTROD_DIAG_THETA_LOOP"
  [waves3]="-DSOFTROD_OCTO_WAVES=3"
  [hoistlit]="-DSOFTROD_DIAG_HOIST_LITERALS"
)
NAMES=${OCTO_AB_VARIANTS:-r4 basemask diag4 diag8 diag12 diag1 diag2 diag7}
if [ "${1:-}" = build ]; then
  mkdir -p "$ROOT/variants"
  for n in $NAMES; do
    ( cd "$ROOT/gym_softrobot_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function \
        ${FLAGS[$n]} -DSOFTROD_SOURCE_HASH="\"octoab-$n\"" -shared -o "$ROOT/variants/libsoftrod_octo_$n.so" softrod_capi.hip ) &
    while [ "$(jobs -r | wc -l)" -ge 4 ]; do sleep 1; done
  done
  wait
  ls -la "$ROOT"/variants/libsoftrod_octo_*.so
  exit 0
fi
TAG=${2:-ab}
OUT=$ROOT/gpurun_out/octo_ab_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
A="--no-cpu-baseline --no-secondary --env OctoFlat-v0 --steps 5 --warmup 3 --windows 3 ${OCTO_AB_ARGS:-}"
for rep in 1 2; do
for n in $NAMES; do
  export SOFTROD_HIP_LIB=$ROOT/variants/libsoftrod_octo_$n.so
  timeout 150 python3 "$ROOT/bench.py" $A > "$OUT/bench_${n}_$rep.json" 2> "$OUT/bench_${n}_$rep.err"
  python3 - "$OUT/bench_${n}_$rep.json" "$n" "$rep" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("variant", sys.argv[2], "rep", sys.argv[3], "kernel_ms", round(d["roofline"]["kernel_ms_avg"], 4), "env-steps/s", round(d["value"]),
          "non-finite", d["config"]["non_finite_envs_at_end"])
except Exception as e:
    print("variant", sys.argv[2], "FAILED", e)
PY
done
done
for n in $NAMES; do
  export SOFTROD_HIP_LIB=$ROOT/variants/libsoftrod_octo_$n.so
  rm -rf "$OUT/pmc_$n"
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_$n" -o pmc --output-format csv -- \
      python3 "$ROOT/bench.py" $A > "$OUT/pmc_$n.log" 2>&1
  python3 - "$OUT/pmc_$n" "$n" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "octo_step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
if acc:
    k = min(len(v) for v in acc.values())
    m = {c: sum(v[-min(k, 15):]) / min(k, 15) for c, v in acc.items()}     # the measured batch's launches
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    print("variant", sys.argv[2], "VALU per arm-substep %.1f" % (m["SQ_INSTS_VALU"] / (1024 * 8 * 2857)),
          "busy %.3f" % (m["SQ_ACTIVE_INST_VALU"] * 4 / (cyc * 1024)), "Mcycles per launch %.2f" % (cyc / 1e6))
PY
  find "$OUT/pmc_$n" -name "*.csv" -size +2M -delete; find "$OUT/pmc_$n" -name "*.db" -delete
done
