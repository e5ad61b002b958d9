#!/usr/bin/env python3
"""Soak run of the device-side auto-reset path (N2): many episodes back to back with a policy in the loop and no
host synchronisation except the queue top-ups — the way an RL job runs for hours.  Checks what a long job rests on:
no reset record ever missing (underflow 0), every env restarts (episodes end by truncation on step 126 or by a NaN
termination, and the next step returns a finite reset observation), the step rate does not decay, the host
bookkeeping (staged draws, consumed counters) stays consistent with the device's.

    python tools/soak.py [--env SoftPendulum-v0] [--envs 4096] [--steps 60000]      (on the MI355X box)
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import bench  # noqa: E402
import gym_softrobot_amd as gsa  # noqa: E402

AMAX = {"SoftPendulum-v0": 22.0, "SoftPendulum3D-v0": 1.0, "OctoArmSingle-v0": 6.0, "OctoFlat-v0": 22.0}
UNIT_BOX = ("OctoArmPush-v1", "OctoArmPullWeight-v0", "OctoCrawl-v0", "OctoArmTwo-v0", "OctoReach-v0")     # Box(0, 1) actions


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="SoftPendulum-v0")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=60000)
    ap.add_argument("--check-every", type=int, default=2000)
    a = ap.parse_args()
    env = gsa.make_vec(a.env, a.envs, device=0, autoreset="device")
    obs, _ = env.reset(seed=0)
    if a.env in UNIT_BOX:          # the muscle envs: the MLP's tanh output mapped onto [0, 1]
        raw = bench.make_policy(torch, env.backend.device, env.obs_dim, env.action_dim, 1.0)

        def policy(o):
            return 0.5 + 0.5 * raw(o)
    else:
        policy = bench.make_policy(torch, env.backend.device, env.obs_dim, env.action_dim, AMAX[a.env])
    dev = env.backend.device
    ended = torch.zeros(a.envs, dtype=torch.int64, device=dev)       # episodes ended per env (device-side count)
    bad_after_reset = torch.zeros((), dtype=torch.int64, device=dev)
    prev_done = torch.zeros(a.envs, dtype=torch.bool, device=dev)
    rates, t0, tlast = [], time.perf_counter(), time.perf_counter()
    out = {"env": a.env, "envs": a.envs, "steps": a.steps, "checks": []}
    for t in range(a.steps):
        obs, rew, term, trunc, _ = env.step(policy(obs))
        done = term | trunc
        # the step AFTER an episode end returns the reset observation: finite, flags clear (NEXT_STEP)
        bad_after_reset += (prev_done & (done | ~torch.isfinite(obs).all(dim=1))).sum()
        ended += done
        prev_done = done.clone()
        if (t + 1) % a.check_every == 0:
            torch.cuda.synchronize()
            now = time.perf_counter()
            consumed, underflow = env.backend.queue_status()
            rec = {"step": t + 1, "ms_per_step": (now - tlast) / a.check_every * 1e3, "underflow": int(underflow),
                   "episodes_ended": int(ended.sum().item()), "resets_consumed": int(consumed.sum()),
                   "bad_after_reset": int(bad_after_reset.item()),
                   "terminated_now": int(term.sum().item()), "non_finite_obs_now": int((~torch.isfinite(obs).all(dim=1)).sum().item())}
            out["checks"].append(rec)
            rates.append(rec["ms_per_step"])
            tlast = time.perf_counter()
    torch.cuda.synchronize()
    consumed, underflow = env.backend.queue_status()
    ended_h = ended.cpu().numpy()
    pend = int(prev_done.sum().item())               # episodes that ended on the very last step: not restarted yet
    out["summary"] = {
        "seconds": time.perf_counter() - t0, "underflow": int(underflow),
        "episodes_ended": int(ended_h.sum()), "resets_consumed": int(consumed.sum()), "ended_on_last_step": pend,
        "consistent": bool(int(consumed.sum()) == int(ended_h.sum()) - pend and (np.asarray(consumed) == ended_h - prev_done.cpu().numpy()).all()),
        "bad_after_reset": int(bad_after_reset.item()),
        "ms_per_step_first_last": [rates[0], rates[-1]] if rates else None,
        "rate_drift": (rates[-1] / np.median(rates) - 1.0) if rates else None,
        "every_env_restarted_at_least": int(ended_h.min()),
    }
    env.close()
    print(json.dumps(out))
    ok = (out["summary"]["underflow"] == 0 and out["summary"]["consistent"] and out["summary"]["bad_after_reset"] == 0)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
