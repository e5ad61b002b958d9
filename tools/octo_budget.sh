#!/bin/bash
# Instruction budget of the OctoFlat wave-substep (VERDICT r4 "next" #3), on the CPU: the hot-path VALU
# count (tools/hot_path_isa.py) of softrod_octo_step_kernel<zup,2,4> with one block compiled out at a time
# (SOFTROD_OCTO_DIAG bits, softrod_octo.hpp).  hipcc cross-compiles; no GPU needed.
#   tools/octo_budget.sh > profiles/r5_octo_budget.txt
set -eu
cd "$(dirname "$0")/../gym_softrobot_amd/csrc"
K=octo_step_kernelILj1073743625ELi2ELi4E
for D in 0 1 2 4 8 3 6 7; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -DSOFTROD_OCTO_DIAG=$D -o /tmp/octo_diag_$D.s softrod_capi.hip 2>/dev/null
  printf "SOFTROD_OCTO_DIAG=%s  " $D
  python3 ../../tools/hot_path_isa.py /tmp/octo_diag_$D.s $K
done
printf "OctoArmSingle-v0 (one 50-element contact arm per wave, the same rod physics)  "
python3 ../../tools/hot_path_isa.py /tmp/octo_diag_0.s fast_kernelILj1073742601ELi3ELi1ELb0E
