python -m pytest tests/test_gpu_full_size.py tests/test_gpu_episode_parity.py -m gpu -q -x 2>&1 | tail -15
