set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2a_gputests.log
python bench.py > gpurun_out/r2a_bench.json 2> gpurun_out/r2a_bench.err
tools/pmc_valu_per_substep.sh SoftPendulum-v0 4096 40 22 400 1
tools/pmc_valu_per_substep.sh SoftPendulum3D-v0 4096 20 1 400 1
tools/pmc_valu_per_substep.sh OctoArmSingle-v0 4096 12 6 714 1
tools/pmc_valu_per_substep.sh OctoArmSingle-v0 4096 8 6 714 1 100 OctoArmSingle-v0_n100
tools/pmc_valu_per_substep.sh OctoFlat-v0 1024 4 22 2857 8
tools/pmc_valu_per_substep.sh SoftArmTracking-v0 4096 40 1 50 1
cat gpurun_out/r2a_gputests.log; cat gpurun_out/r2a_bench.json
