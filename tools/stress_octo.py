"""Stress the four-envs-per-workgroup OctoFlat kernel (LDS flag rendezvous): many launches of
different batch sizes / substep counts / auto-reset, under an external timeout."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import gym_softrobot_amd as gsa

t0 = time.time()
for n, fps, steps, auto in ((1024, 5, 6, False), (1023, 50, 120, "device"), (5, 500, 400, True), (1, 100, 100, False),
                            (2048, 20, 40, "device"), (7, 5, 4, False), (4096, 200, 100, False)):
    env = gsa.make_vec("OctoFlat-v0", n, recording_fps=fps, autoreset=auto)
    env.reset(seed=0)
    acts = torch.from_numpy(np.random.default_rng(0).uniform(-22, 22, (steps, n, 24)).astype(np.float32)).cuda()
    for t in range(steps):
        out = env.step(acts[t])
    torch.cuda.synchronize()
    print(n, fps, steps, auto, "ok", round(time.time() - t0, 1), flush=True)
    env.close()
print("stress ok")
