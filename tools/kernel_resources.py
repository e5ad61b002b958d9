#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table from `make -C gym_softrobot_amd/csrc resources`
(hipcc -Rpass-analysis=kernel-resource-usage).  Usage: tools/kernel_resources.py [remarks.txt]"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def remarks() -> str:
    if len(sys.argv) > 1:
        return Path(sys.argv[1]).read_text()
    p = subprocess.run(["make", "-C", str(ROOT / "gym_softrobot_amd" / "csrc"), "resources"],
                       capture_output=True, text=True)
    return p.stdout + p.stderr


def main() -> None:
    blocks = re.split(r"remark: [^\n]*Function Name: ", remarks())[1:]
    keys = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("spill", r"VGPRs Spill"), ("sgpr-spill", r"SGPRs Spill"), ("scratch B/lane", r"ScratchSize \[bytes/lane\]"),
            ("waves/SIMD", r"Occupancy \[waves/SIMD\]"), ("LDS B", r"LDS Size \[bytes/block\]")]
    for b in blocks:
        name = b.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(.*", "", dem).replace("softrod::", "").replace("void ", "")
        vals = []
        for label, k in keys:
            m = re.search(k + r": (\d+)", b)
            vals.append(f"{label} {m.group(1) if m else '?':>5}")
        print(f"{dem[:70]:70s} " + "  ".join(vals))


if __name__ == "__main__":
    main()
