set -x
make -C oracle fma >/dev/null 2>&1
timeout 1500 python tools/episode_parity.py > gpurun_out/parity_episode.json 2> gpurun_out/parity_episode.err
tail -5 gpurun_out/parity_episode.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_episode.json'))
for k,v in d['scenarios'].items():
    print(k, v['steps_within_1e-5'], {a: '%.1e'%b for a,b in v['max'].items()}, v['flags_equal_all_steps'])
PY
