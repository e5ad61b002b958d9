"""Per-env.step kernel time of the bench workload (diagnostic): shows how the wave-uniform
range-reduction loops of the fast kernel respond to the motion getting more violent."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import gym_softrobot_amd as gsa  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "SoftPendulum-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
T = int(sys.argv[3]) if len(sys.argv) > 3 else 60
amax = float(sys.argv[4]) if len(sys.argv) > 4 else 22.0
n_elems = int(sys.argv[5]) if len(sys.argv) > 5 else 0
fps = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0     # SoftPendulum: substeps per step = 1e4 / fps
kw = {"n_elems": n_elems} if n_elems else {}
if fps:
    kw.update(recording_fps=fps, final_time=1e9)
env = gsa.make_vec(env_id, n, device=0, **kw)
env.reset(seed=0)
adim = env.action_dim
acts = torch.from_numpy(np.random.default_rng(1).uniform(-amax, amax, (T, n, adim)).astype(np.float32)).cuda()
env.backend.set_timing(T)
for t in range(T):
    obs, rew, term, trunc, _ = env.step(acts[t])
torch.cuda.synchronize()
kt = env.backend.kernel_times_ms()
print(" ".join(f"{x:.3f}" for x in kt))
print("mean", kt.mean(), "min", kt.min(), "max", kt.max(), "terminated", int(term.sum()))
