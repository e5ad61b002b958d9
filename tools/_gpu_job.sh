python bench.py > gpurun_out/r2d_bench.json 2>gpurun_out/r2d_bench.err
python bench.py --actions zero --no-cpu-baseline > gpurun_out/r2d_zero.json 2>>gpurun_out/r2d_bench.err
: > gpurun_out/r2d_widened.jsonl
python bench.py --env SoftPendulum3D-v0 --no-cpu-baseline >> gpurun_out/r2d_widened.jsonl 2>>gpurun_out/r2d_bench.err
python bench.py --env OctoArmSingle-v0 --no-cpu-baseline >> gpurun_out/r2d_widened.jsonl 2>>gpurun_out/r2d_bench.err
python bench.py --env OctoArmSingle-v0 --n-elems 100 --steps 40 --warmup 5 --no-cpu-baseline >> gpurun_out/r2d_widened.jsonl 2>>gpurun_out/r2d_bench.err
python bench.py --env OctoFlat-v0 --steps 10 --warmup 2 --no-cpu-baseline >> gpurun_out/r2d_widened.jsonl 2>>gpurun_out/r2d_bench.err
python bench.py --env SoftArmTracking-v0 --no-cpu-baseline >> gpurun_out/r2d_widened.jsonl 2>>gpurun_out/r2d_bench.err
python bench.py --steps 300 --warmup 20 --no-cpu-baseline >> gpurun_out/r2d_widened.jsonl 2>>gpurun_out/r2d_bench.err
python - <<'PY'
import json
for f in ('gpurun_out/r2d_bench.json','gpurun_out/r2d_zero.json','gpurun_out/r2d_widened.jsonl'):
    for l in open(f):
        d=json.loads(l); print(d['config']['workload'][:60], d['config']['actions'], '%.4g'%d['value'], 'ms/step %.4f'%d['ms_per_step'], 'kernel %.4f'%d['roofline']['kernel_ms_avg'], 'frac', d['roofline']['frac'])
PY
tail -3 gpurun_out/r2d_bench.err
