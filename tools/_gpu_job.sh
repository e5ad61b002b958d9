python -m pytest tests -m gpu -q -x 2>&1 | tail -5
tools/pmc_valu_per_substep.sh OctoFlat-v0 1024 4 22 2857 8
python bench.py --env OctoFlat-v0 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_avg'])"
