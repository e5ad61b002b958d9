#!/bin/bash
# Run ON THE GPU BOX: SQ_INSTS_VALU per rod-substep of the step kernel for one workload.
# Usage: tools/pmc_valu_per_substep.sh <env-id> <envs> <steps> <amax> <substeps-per-step> <rods-per-env>
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu_$1
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT -o pmc --output-format csv -- \
    python3 $GRAFT_REPO_ROOT/tools/step_time_trace.py $1 $2 $3 $4 > $OUT/log.txt 2>&1
python3 - "$OUT" "$2" "$5" "$6" <<'PY'
import csv, glob, sys, collections
out, envs, nsub, rods = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
acc = collections.defaultdict(list)
for p in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "_step_" in r["Kernel_Name"] and "autoreset" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
units = envs * nsub * rods
print("VALU instructions per rod-substep: %.1f" % (m["SQ_INSTS_VALU"] / units),
      " VALU busy: %.1f %%" % (100 * m["SQ_ACTIVE_INST_VALU"] * 4 / (m["GRBM_GUI_ACTIVE"] / 8 * 1024)),
      " cycles per launch (per XCD): %.3f M" % (m["GRBM_GUI_ACTIVE"] / 8e6))
PY
