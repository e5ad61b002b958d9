"""PPO on SoftPendulum3D-v0 with the whole loop on one MI355X — the counterpart of the
reference's only training example (examples/soft_pendulum_3d/train_ppo.py, Stable-Baselines3
on ONE env).  Here `num_envs` pendulums step in one kernel launch, observations, rewards and
flags stay on the device (zero-copy views of the stepper's buffers), finished episodes restart
on the device (`autoreset="device"`, Gymnasium NEXT_STEP semantics), and a small torch policy
acts on the same GPU: no host round trip inside the rollout.

    python examples/soft_pendulum_3d_ppo.py --num-envs 1024 --updates 30

Plain PPO (clipped surrogate, GAE); nothing here is tuned — it is a usage example.
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_softrobot_amd as gsa  # noqa: E402


class ActorCritic(nn.Module):
    def __init__(self, obs_dim, act_dim, hidden=64):
        super().__init__()
        self.pi = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                                nn.Linear(hidden, act_dim))
        self.v = nn.Sequential(nn.Linear(obs_dim, hidden), nn.Tanh(), nn.Linear(hidden, hidden), nn.Tanh(),
                               nn.Linear(hidden, 1))
        self.log_std = nn.Parameter(torch.full((act_dim,), -0.5))

    def dist(self, obs):
        return torch.distributions.Normal(self.pi(obs), self.log_std.exp())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=1024)
    ap.add_argument("--horizon", type=int, default=32, help="env.steps per env per update")
    ap.add_argument("--updates", type=int, default=30)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--minibatches", type=int, default=4)
    ap.add_argument("--lr", type=float, default=3e-4)
    ap.add_argument("--gamma", type=float, default=0.99)
    ap.add_argument("--lam", type=float, default=0.95)
    ap.add_argument("--clip", type=float, default=0.2)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()

    torch.manual_seed(args.seed)
    dev = torch.device("cuda", 0)
    env = gsa.make_vec("SoftPendulum3D-v0", args.num_envs, device=0, autoreset="device")
    N, T = args.num_envs, args.horizon
    od, ad = env.obs_dim, env.action_dim
    lo, hi = env.action_low, env.action_high
    net = ActorCritic(od, ad).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr)

    obs_buf = torch.empty((T, N, od), device=dev)
    act_buf = torch.empty((T, N, ad), device=dev)
    logp_buf = torch.empty((T, N), device=dev)
    rew_buf = torch.empty((T, N), device=dev)
    done_buf = torch.empty((T, N), device=dev)
    val_buf = torch.empty((T + 1, N), device=dev)

    obs, _ = env.reset(seed=args.seed)
    obs = obs.clone()
    for update in range(args.updates):
        t0 = time.perf_counter()
        with torch.no_grad():
            for t in range(T):
                d = net.dist(obs)
                a = d.sample()
                obs_buf[t], act_buf[t], logp_buf[t] = obs, a, d.log_prob(a).sum(-1)
                val_buf[t] = net.v(obs).squeeze(-1)
                nobs, rew, term, trunc, _ = env.step(a.clamp(lo, hi))
                rew_buf[t] = rew.float()
                done_buf[t] = (term | trunc).float()
                obs = nobs.clone()                    # the env's buffers are overwritten by the next step
            val_buf[T] = net.v(obs).squeeze(-1)
            adv = torch.zeros((T, N), device=dev)
            last = torch.zeros(N, device=dev)
            for t in reversed(range(T)):              # GAE; an env that finished restarts on its next step
                nonterminal = 1.0 - done_buf[t]
                delta = rew_buf[t] + args.gamma * val_buf[t + 1] * nonterminal - val_buf[t]
                last = delta + args.gamma * args.lam * nonterminal * last
                adv[t] = last
            ret = adv + val_buf[:T]
        torch.cuda.synchronize()
        t_roll = time.perf_counter() - t0             # (the first update also pays torch's one-time set-up)

        b_obs, b_act = obs_buf.reshape(T * N, od), act_buf.reshape(T * N, ad)
        b_logp, b_adv, b_ret = logp_buf.reshape(-1), adv.reshape(-1), ret.reshape(-1)
        b_adv = (b_adv - b_adv.mean()) / (b_adv.std() + 1e-8)
        for _ in range(args.epochs):
            perm = torch.randperm(T * N, device=dev)
            for idx in perm.chunk(args.minibatches):
                d = net.dist(b_obs[idx])
                ratio = (d.log_prob(b_act[idx]).sum(-1) - b_logp[idx]).exp()
                pg = -torch.min(ratio * b_adv[idx], ratio.clamp(1 - args.clip, 1 + args.clip) * b_adv[idx]).mean()
                vloss = 0.5 * (net.v(b_obs[idx]).squeeze(-1) - b_ret[idx]).pow(2).mean()
                loss = pg + 0.5 * vloss - 0.0 * d.entropy().sum(-1).mean()
                opt.zero_grad(set_to_none=True)
                loss.backward()
                nn.utils.clip_grad_norm_(net.parameters(), 0.5)
                opt.step()
        steps = (update + 1) * T * N
        print(f"update {update + 1:3d}  env-steps {steps:9d}  mean reward/step {rew_buf.mean().item():+.4f}  "
              f"mean tilt {obs_buf[..., 8].mean().item():.4f}  rollout {T * N / t_roll:,.0f} env-steps/s (policy included)")
    env.close()


if __name__ == "__main__":
    main()
