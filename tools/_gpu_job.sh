set -x
python -m pytest tests/test_gpu_reference_fixtures.py -m gpu -q 2>&1 | tail -40 > gpurun_out/r2b_reffix.log
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_reference_fixtures.py 2>&1 | tail -25 > gpurun_out/r2b_gputests.log
cat gpurun_out/r2b_reffix.log gpurun_out/r2b_gputests.log
