python -m pytest tests/test_gpu_taper_suckers.py -m gpu -q -x 2>&1 | tail -30
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_taper_suckers.py 2>&1 | tail -8
