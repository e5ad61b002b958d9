#!/bin/bash
# Run ON THE GPU BOX: what each reformulation of the planar SoftPendulum substep costs in parity
# horizon.  The stabilised inverted pendulum (tools/episode_parity.py --pd-horizons: 8 envs, 126
# steps, a PD law on the oracle's observations) is run on the shipped library (fast and libm
# kernels, plus the control = the oracle built with FMA contraction) and on each diagnostic build
# of `make -C gym_softrobot_amd/csrc diag` (variants/libsoftrod_diag_<X>.so: ONE reformulation
# undone).  Output: gpurun_out/fastmath_cost.jsonl, one JSON object per library.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out
OUT=gpurun_out/fastmath_cost.jsonl
: > $OUT
python3 tools/episode_parity.py --pd-horizons >> $OUT 2> gpurun_out/fastmath_cost.err
for lib in variants/libsoftrod_diag_*.so; do
  [ -f "$lib" ] || continue
  SOFTROD_HIP_LIB=$ROOT/$lib python3 tools/episode_parity.py --pd-horizons >> $OUT 2>> gpurun_out/fastmath_cost.err
done
python3 - <<'PY'
import json
for l in open("gpurun_out/fastmath_cost.jsonl"):
    d = json.loads(l)
    print(d["library"].split("/")[-1], "fast", d["fast"]["steps_within_1e-5"], "libm", d["libm"]["steps_within_1e-5"],
          "control", d["control"]["steps_within_1e-5"])
PY
