import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import gym_softrobot_amd as gsa
def run(n, fps):
    env = gsa.make_vec("SoftPendulum-v0", n, recording_fps=fps, final_time=1e9)
    env.reset(seed=0)
    acts = torch.zeros((n, 1), device="cuda")
    steps = int(40000 / env.cfg.n_substeps) + 20
    for t in range(steps): env.step(acts)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(steps): env.step(acts)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    nsub = env.cfg.n_substeps
    env.close()
    return nsub, ms
for n in (256, 1024, 3072, 4096):
    for fps in (100, 50, 25, 12.5, 6.25):
        nsub, ms = run(n, fps)
        print(n, nsub, round(ms, 4), flush=True)
