set -x
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/r2c_gputests.log
bash tools/profile_gpu.sh r2c > gpurun_out/r2c_profile.log 2>&1
cat gpurun_out/r2c_gputests.log; tail -60 gpurun_out/r2c_profile.log
