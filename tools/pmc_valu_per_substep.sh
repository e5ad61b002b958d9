#!/bin/bash
# Run ON THE GPU BOX: the 3-counter PMC pass (SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE)
# of one workload's step kernel -> gpurun_out/pmc_valu_<tag>/summary.json (copy it to
# profiles/<round>_pmc3_<tag>.json and fold it into profiles/valu_counts.json with
# tools/update_valu_counts.py).  Three counters only: the 8-counter pass of profile_gpu.sh
# perturbs the kernel (VALU busy 69 % there against 86 % here on the same binary).
# Usage: tools/pmc_valu_per_substep.sh <env-id> <envs> <steps> <amax> <substeps-per-step> <rods-per-env> [n_elems] [tag]
cd /tmp && export TMPDIR=/tmp
TAG=${8:-$1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_valu_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $OUT -o pmc --output-format csv -- \
    python3 $GRAFT_REPO_ROOT/tools/step_time_trace.py $1 $2 $3 $4 ${7:-0} > $OUT/log.txt 2>&1
python3 - "$OUT" "$1" "$2" "$5" "$6" "${7:-0}" <<'PY'
import csv, glob, json, sys, collections
out, env, envs, nsub, rods, nel = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
regs = {}
for p in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if "_step_" in k and "autoreset" not in k:
            acc[k.split("(")[0][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            regs[k.split("(")[0][:100]] = {x: r.get(x) for x in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size")}
units = envs * nsub * rods
doc = {"env": env, "envs": envs, "substeps_per_step": nsub, "rods_per_env": rods, "n_elems_arg": nel,
       "command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -- python3 tools/step_time_trace.py " + " ".join(sys.argv[2:4]),
       "kernels": {}}
tot_insts = 0.0
for k, m in acc.items():
    a = {c: sum(v) / len(v) for c, v in m.items()}
    a["launches"] = len(m["SQ_INSTS_VALU"])
    a["valu_busy_frac"] = a["SQ_ACTIVE_INST_VALU"] * 4 / (a["GRBM_GUI_ACTIVE"] / 8 * 1024)
    a["valu_issue_frac"] = a["SQ_INSTS_VALU"] * 4 / (a["GRBM_GUI_ACTIVE"] / 8 * 1024)
    a["cycles_per_launch_per_xcd"] = a["GRBM_GUI_ACTIVE"] / 8
    a["registers"] = regs[k]
    doc["kernels"][k] = a
    tot_insts += a["SQ_INSTS_VALU"]
# the workload's figure: all step kernels of one env.step together (the windowed arm runs two)
cyc = sum(a["GRBM_GUI_ACTIVE"] for a in doc["kernels"].values()) / 8
doc["valu_instr_per_rod_substep"] = tot_insts / units
doc["valu_busy_frac"] = sum(a["SQ_ACTIVE_INST_VALU"] for a in doc["kernels"].values()) * 4 / (cyc * 1024)
doc["valu_issue_frac_measured_cycles"] = tot_insts * 4 / (cyc * 1024)
json.dump(doc, open(out + "/summary.json", "w"), indent=1)
print(env, "VALU instructions per rod-substep: %.1f" % doc["valu_instr_per_rod_substep"],
      " VALU busy: %.1f %%" % (100 * doc["valu_busy_frac"]), " issue frac (measured cycles): %.3f" % doc["valu_issue_frac_measured_cycles"],
      " cycles per launch (per XCD): %.3f M" % (cyc / 1e6))
PY
find $OUT -name "*.csv" -size +2M -delete
