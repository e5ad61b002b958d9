"""`python -m gym_softrobot_amd [--env ID] [--num-envs N] [--steps K]` — registry listing and
a random-action rollout on the GPU (the counterpart of gym_softrobot/debug/registry.py:16-24
and debug/make.py:6-23)."""
from __future__ import annotations

import argparse
import time

import numpy as np


def main() -> None:
    import gym_softrobot_amd as gsa

    ap = argparse.ArgumentParser(prog="python -m gym_softrobot_amd")
    ap.add_argument("--env", default=None, help="env id; omit to list the registry")
    ap.add_argument("--num-envs", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    if args.env is None:
        for i, name in enumerate(gsa.registered()):
            label = gsa.parity_label(name)
            print(f"{i:3d}  {name}" + (f"    [{label}]" if label else ""))
        return
    import torch

    env = gsa.make_vec(args.env, args.num_envs, autoreset=True)
    obs, _ = env.reset(seed=args.seed)
    rng = np.random.default_rng(args.seed)
    lo, hi = env.action_low, env.action_high
    t0 = time.perf_counter()
    for k in range(args.steps):
        a = rng.uniform(lo, hi, (args.num_envs, env.action_dim)).astype(np.float32)
        if getattr(env, "mode", None) == 0:          # OctoArmPush-v0: Discrete(2)
            a = np.round(a)
        obs, rew, term, trunc, info = env.step(a)
        print(f"step {k + 1:4d}  time {info['time'][0]:.3f}  reward[0] {float(rew[0]):+.5f}  "
              f"terminated {int(term.sum())}  truncated {int(trunc.sum())}")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{args.num_envs * args.steps / dt:.1f} env-steps/s including printing")
    env.close()


if __name__ == "__main__":
    main()
