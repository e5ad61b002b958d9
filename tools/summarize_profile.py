#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory into one JSON document:
per-kernel time from `rocprofv3 --kernel-trace --stats`, and per-launch HBM bytes of the
step kernel from the FETCH_SIZE / WRITE_SIZE PMC passes.

Units (MI355X_MICROARCH.md §HBM, cdna_hip_programming.md §7): FETCH_SIZE and WRITE_SIZE
are in KiB; on gfx950 FETCH_SIZE under-reports a wide coalesced read stream by exactly 2x
(calibrated for 16 B/lane; this kernel loads 8 B/lane, uncalibrated) — both the raw and
the 2x-corrected read figure are reported and the corrected one is used for `traffic`.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KERNELS = ("softrod_step_", "softrod_octo_step_", "softrod_octo1w_step_")   # softrod_step_window_kernel matches the first


def is_step_kernel(name):
    return any(k in name for k in KERNELS)


def read_csv(path):
    with open(path, newline="") as f:
        return list(csv.DictReader(f))


def kernel_stats(root):
    out = []
    for p in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in read_csv(p):
            out.append({
                "name": r.get("Name", "")[:120],
                "calls": int(float(r.get("Calls", 0))),
                "total_ns": float(r.get("TotalDurationNs", 0)),
                "avg_ns": float(r.get("AverageNs", 0)),
                "min_ns": float(r.get("MinNs", 0)),
                "max_ns": float(r.get("MaxNs", 0)),
                "pct": float(r.get("Percentage", 0)),
            })
    out.sort(key=lambda r: -r["total_ns"])
    return out


def timed_launches(root):
    """Durations (ns) of the step kernel's launches in start order, from the kernel trace."""
    rows = []
    for p in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in read_csv(p):
            if is_step_kernel(r.get("Kernel_Name", "")):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows.sort()
    return [d for _, d in rows]


def counters(root, sub, last=None):
    """-> {counter_name: [per-dispatch values of the step kernel, in dispatch order]} and register info.
    last = n: only the last n step-kernel dispatches — the MEASURED batch's (warm-up + timed windows);
    bench.py's pre-heat steps a scratch batch with zero actions before them, and those launches take
    fewer of the longer series tiers, so they must not dilute the per-launch averages."""
    rows = defaultdict(list)
    regs = {}
    for p in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in read_csv(p):
            if not is_step_kernel(r.get("Kernel_Name", "")):
                continue
            rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
            regs = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                                          "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size")}
    vals = {}
    for k, v in rows.items():
        v.sort()
        vals[k] = [x for _, x in (v[-last:] if last else v)]
    return vals, regs


def measured_dispatches(root):
    """Step-kernel dispatches of the measured batch in one bench.py run: (warmup + windows x steps) x
    kernels per env.step, from the bench line of the trace pass (all passes run the same command)."""
    for name in ("bench_trace.log", "bench_pmc3.log", "bench_pmc_fetch.log", "bench_pmc_write.log"):
        p = os.path.join(root, name)
        if os.path.exists(p):
            for line in open(p):
                if line.startswith("{"):
                    try:
                        b = json.loads(line)
                        per = 2 if "window" in b["roofline"]["kernel"] else 1
                        return (int(b["warmup"]) + int(b["steps"]) * int((b.get("windows") or {}).get("count", 1))) * per
                    except Exception:  # noqa: BLE001
                        pass
    return None


def main(root):
    doc = {"dir": os.path.basename(root.rstrip("/"))}
    ks = kernel_stats(root)
    doc["kernel_stats"] = ks[:8]
    step = [k for k in ks if is_step_kernel(k["name"])]
    if step:
        doc["step_kernel_avg_ms"] = step[0]["avg_ns"] / 1e6
        doc["step_kernel_calls"] = step[0]["calls"]
    last = measured_dispatches(root)
    doc["counter_dispatches_used"] = last      # the measured batch's launches only (not the pre-heat's)
    fetch, regs = counters(root, "pmc_fetch", last)
    write, _ = counters(root, "pmc_write", last)
    sq, _ = counters(root, "pmc_sq", last)
    pmc3, regs3 = counters(root, "pmc3", last)
    doc["registers"] = regs or regs3
    if fetch.get("FETCH_SIZE"):
        f = fetch["FETCH_SIZE"]
        doc["FETCH_SIZE_KiB_per_launch_raw"] = sum(f) / len(f)
    if write.get("WRITE_SIZE"):
        w = write["WRITE_SIZE"]
        doc["WRITE_SIZE_KiB_per_launch_raw"] = sum(w) / len(w)
    if "FETCH_SIZE_KiB_per_launch_raw" in doc and "WRITE_SIZE_KiB_per_launch_raw" in doc:
        rd = doc["FETCH_SIZE_KiB_per_launch_raw"] * 1024.0
        wr = doc["WRITE_SIZE_KiB_per_launch_raw"] * 1024.0
        doc["hbm_bytes_per_launch_raw"] = rd + wr
        doc["hbm_bytes_per_launch"] = 2.0 * rd + wr  # gfx950 FETCH_SIZE x2 correction
    if sq:
        doc["sq_per_launch"] = {k: sum(v) / len(v) for k, v in sq.items()}
    bench = {}
    for name in ("bench_trace.log", "bench_pmc_fetch.log", "bench_pmc_write.log", "bench_pmc3.log"):
        p = os.path.join(root, name)
        if os.path.exists(p):
            for line in open(p):
                if line.startswith("{"):
                    try:
                        b = json.loads(line)
                        bench[name] = b
                        doc.setdefault("bench_lines", {})[name] = {
                            "value": b["value"], "ms_per_step": b["ms_per_step"],
                            "kernel_ms_avg": b["roofline"]["kernel_ms_avg"],
                            "frac": b["roofline"]["frac"],
                            "library_source_hash": b["config"].get("library_source_hash"),
                            "preheat_launches": (b.get("preheat") or {}).get("launches"),
                            "windows": (b.get("windows") or {}).get("count"),
                        }
                    except Exception:  # noqa: BLE001
                        pass
    hashes = {v["config"].get("library_source_hash") for v in bench.values()}
    doc["library_source_hash"] = hashes.pop() if len(hashes) == 1 else None    # all passes on ONE build, or none
    any_line = next(iter(bench.values()), None)
    if any_line:
        doc["step_kernels_per_env_step"] = 2 if "window" in any_line["roofline"]["kernel"] else 1
        doc["workload"] = any_line["config"]["workload"]
        doc["command"] = "rocprofv3 <pass> -- python3 bench.py " + " ".join(sys.argv[2:])
    # the 3-counter pass (SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE on the same bench.py command): VALU
    # wave-instructions per rod-substep over ALL step-kernel launches of one env.step (the windowed arm runs
    # two kernels per step: their sums), and the busy / issue fractions from the profiler's own cycle counter
    if pmc3.get("SQ_INSTS_VALU") and any_line:
        units = any_line["roofline"]["rod_substeps_per_launch"]
        per_step_kernels = 2 if "window" in any_line["roofline"]["kernel"] else 1
        n = len(pmc3["SQ_INSTS_VALU"]) / per_step_kernels          # env.steps profiled
        insts = sum(pmc3["SQ_INSTS_VALU"]) / n
        active = sum(pmc3["SQ_ACTIVE_INST_VALU"]) / n
        cyc = sum(pmc3["GRBM_GUI_ACTIVE"]) / n / 8.0               # per XCD (the counter sums the 8 XCDs)
        doc["pmc3"] = {
            "env_steps_profiled": n, "SQ_INSTS_VALU_per_env_step": insts, "SQ_ACTIVE_INST_VALU_per_env_step": active,
            "cycles_per_env_step_per_xcd": cyc,
            "valu_instr_per_rod_substep": insts / units,
            "valu_busy_frac": active * 4 / (cyc * 1024),
            "valu_issue_frac_measured_cycles": insts * 4 / (cyc * 1024),
        }
    # the launches bench.py times are the LAST windows x steps ones of the trace (pre-heat and warm-up come
    # first); the window bench.py reports is windows.median_index: its average is the number to hold
    # against roofline.kernel_ms_avg of the same command
    dur = timed_launches(root)
    b = bench.get("bench_trace.log")
    if dur and b:
        per = 2 if "window" in b["roofline"]["kernel"] else 1
        if per == 2:        # window kernel + epilogue kernel of one env.step: add them up
            dur = [dur[i] + dur[i + 1] for i in range(0, len(dur) - 1, 2)]
        steps, R = int(b["steps"]), int((b.get("windows") or {}).get("count", 1))
        m = int((b.get("windows") or {}).get("median_index", 0))
        if len(dur) >= steps * R:
            timed = dur[-steps * R:]
            wins = [sum(timed[w * steps:(w + 1) * steps]) / steps / 1e6 for w in range(R)]
            doc["step_kernel_timed_launches"] = steps * R
            doc["step_kernel_window_avg_ms"] = wins
            doc["step_kernel_timed_avg_ms"] = wins[m]
            doc["step_kernel_untimed_avg_ms"] = (sum(dur[:-steps * R]) / max(1, len(dur) - steps * R)) / 1e6
            doc["bench_kernel_ms_avg_same_run"] = b["roofline"]["kernel_ms_avg"]
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
