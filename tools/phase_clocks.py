"""Diagnostic: where inside one launch of the step kernel the time goes.  Needs the library built
with -DSOFTROD_PHASE_CLOCKS (lane 0 of every wave stamps the 100 MHz wall clock at the phase
boundaries; the stamps add s_waitcnt's, so this build is for looking, not for timing):

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSOFTROD_PHASE_CLOCKS -shared \\
        -o variants/libsoftrod_phase.so gym_softrobot_amd/csrc/softrod_capi.hip
  SOFTROD_HIP_LIB=variants/libsoftrod_phase.so python tools/phase_clocks.py [envs]
"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import gym_softrobot_amd as gsa  # noqa: E402
from gym_softrobot_amd import _capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = gsa.make_vec("SoftPendulum-v0", n, device=0)
env.reset(seed=0)
acts = torch.from_numpy(np.random.default_rng(1).uniform(-22, 22, (60, n, 1)).astype(np.float32)).cuda()
for t in range(60):
    env.step(acts[t])
torch.cuda.synchronize()
lib = _capi.load_library()
out = np.zeros((n, 8), np.uint64)
rc = lib.softrod_debug_phase_clocks(C.c_void_p(out.ctypes.data), C.c_int(n))
assert rc == 0, rc
t = out[:, :6].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0            # 100 MHz
names = ["entry", "state loaded", "loop begins", "loop ends", "state stored", "epilogue done"]
print(f"{n} rods; microseconds since the first wave's entry (min / median / max over waves)")
for i, nm in enumerate(names):
    print(f"  {nm:14s} {us[:, i].min():8.2f} {np.median(us[:, i]):8.2f} {us[:, i].max():8.2f}")
d = np.diff(us, axis=1)
print("phase lengths per wave, microseconds (min / median / max)")
for i, nm in enumerate(["load", "prologue", "loop", "store", "epilogue"]):
    print(f"  {nm:14s} {d[:, i].min():8.2f} {np.median(d[:, i]):8.2f} {d[:, i].max():8.2f}")
order = np.argsort(us[:, 0])
print("entry time of waves by start order: ", " ".join(f"{us[order[k], 0]:.1f}" for k in np.linspace(0, n - 1, 9).astype(int)))
