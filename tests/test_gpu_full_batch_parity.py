"""EVERY env of the BASELINE-size batches against the oracle (VERDICT r5 "next" #2), not a handful of spot envs:

    configs[1]        SoftPendulum-v0, 4096 envs x 50 elements, 3 env.steps            (soft_pendulum.py:176-251)
    configs[2]        OctoArmSingle-style reach, 4096 envs x 100 elements, 1 env.step  (octopus/arm_single_env.py:237-316)
    configs[4] share  OctoFlat-v0, 1024 envs x 8 arms x 10 elements, 1 env.step        (octopus/flat_env.py:315-408)
    N3 (no config)    OctoArmPush-v1, 1024 envs x 40 elements tapered + muscles, 1 step (octopus/arm_push_env.py:276-347; parity
                      unpinned underneath: the restated COOMM law)

The reference's own shape for this is tests/envs/test_determinism.py:46-54 (reset(seed), sampled actions, three
steps, arrays compared) — here on a population, HIP against the C oracle stepping the SAME envs with OpenMP over
rods (oracle_env_step_batch / _arm_batch / oracle_octo_env_step_batch: the driver of bench.py's cpu_baseline leg;
1-7 s per env.step of a whole batch on the GPU box's 16 cores).  rtol 1e-5 (north_star), with the absolute floors
the spot tests use.  The worst env of each batch and where it sits — workgroup, wave, XCD (workgroups are dealt
round-robin over the 8 XCDs) — goes into gpurun_out/full_batch_parity.json: a mapping bug (a wave that reads its
neighbour's row, an XCD-specific cache effect) would show up as a position pattern, not as noise.

ILL-CONDITIONED ENVS.  The plane-contact law is discontinuous (static / kinetic branches at slip_velocity_tol =
1e-8), and in a handful of envs of a batch of thousands one env.step amplifies a perturbation of 1e-14 to 1e-7 in the
ORACLE ITSELF (measured: env 2505 of configs[2]: x0 (1 + 1e-14) moves the end state by 1.1e-7, env 0 by 1e-14).  No
second evaluation can hold 1e-5 of a near-zero coordinate there.  The assertion is therefore: every env is within the
band, OR the oracle re-run from a state perturbed by 1e-14 / 1e-13 leaves its own trajectory by at least a tenth of
what HIP does (so HIP's deviation is the env's conditioning, not a kernel's arithmetic), and such envs are < 1 % of
the batch.  Observations and rewards of configs[1] and [2] need no such clause (all envs within the band).
OctoFlat beyond its first whole step is the ensemble tests' regime."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5
ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


def _report(name, got, want, atol, envs_per_wg=1, waves_per_env=1, extra=None):
    """Per-env worst deviation relative to the tolerance band (<= 1 passes); the record of the worst env."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    band = atol + RTOL * np.abs(want)
    ratio = (np.abs(got - want) / band).reshape(got.shape[0], -1)
    per_env = ratio.max(axis=1)
    i = int(per_env.argmax())
    rec = {"batch": name, "envs": int(got.shape[0]), "rtol": RTOL, "atol": atol,
           "worst_env": i, "worst_over_band": float(per_env[i]), "worst_entry": int(ratio[i].argmax()),
           "worst_abs_diff": float(np.abs(got - want).reshape(got.shape[0], -1)[i].max()),
           "median_env_over_band": float(np.median(per_env)), "envs_over_half_band": int((per_env > 0.5).sum()),
           "position_of_worst": {"workgroup": i // envs_per_wg, "slot_in_workgroup": i % envs_per_wg,
                                 "waves_per_env": waves_per_env, "xcd": (i // envs_per_wg) % 8},
           # worst env per XCD: a position pattern would show here
           "worst_over_band_by_xcd": [float(per_env[np.arange(got.shape[0]) // envs_per_wg % 8 == x].max()) for x in range(8)]}
    rec["over_band_envs"] = [int(k) for k in np.nonzero(per_env > 1.0)[0]]
    rec["abs_dev_of_over_band_envs"] = {int(k): float(np.abs(got - want).reshape(got.shape[0], -1)[k].max())
                                        for k in rec["over_band_envs"]}
    rec.update(extra or {})
    out = ROOT / "gpurun_out"
    out.mkdir(exist_ok=True)
    f = out / "full_batch_parity.json"
    doc = json.loads(f.read_text()) if f.exists() else {}
    doc[name] = rec
    f.write_text(json.dumps(doc, indent=1) + "\n")
    return rec


def _explained_by_conditioning(rec, per_env_dev, rerun_perturbed, n, limit_frac=0.01):
    """Every env over the band must be ill-conditioned in the oracle itself.  per_env_dev: HIP's absolute deviation
    per env (for the over-band envs); rerun_perturbed(i, eps) -> the oracle's own absolute deviation of env i when its
    initial positions are scaled by (1 + eps).  Records the evidence; returns the list of unexplained envs."""
    over = rec["over_band_envs"]
    assert len(over) <= limit_frac * n, (len(over), rec["batch"])
    unexplained, evidence = [], []
    for i in over:
        own = max(rerun_perturbed(i, 1e-14), rerun_perturbed(i, 1e-13))
        evidence.append({"env": i, "hip_abs_dev": per_env_dev[i], "oracle_abs_dev_under_1e-14_perturbation": own})
        if own < 0.1 * per_env_dev[i]:
            unexplained.append(i)
    rec["ill_conditioned_evidence"] = evidence
    f = ROOT / "gpurun_out" / "full_batch_parity.json"
    doc = json.loads(f.read_text())
    doc[rec["batch"]] = rec
    f.write_text(json.dumps(doc, indent=1) + "\n")
    return unexplained


def test_config2_every_one_of_4096_pendulums_three_steps(torch_gpu, hip_lib, oracle_built):
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd.seeding import initial_angle, np_random

    n, T = 4096, 3
    env = gsa.make_vec("SoftPendulum-v0", n)
    env.reset(seed=0)
    ref = oracle_built.OracleBatch(env.cfg, n, omp=True)
    ref.reset([initial_angle(np_random(i)[0]) for i in range(n)])
    acts = np.random.default_rng(1).uniform(-22, 22, (T, n)).astype(np.float32)
    for t in range(T):
        o, r, te, tr, _ = env.step(acts[t])
        o2, r2, te2, tr2 = ref.env_step(acts[t])
        torch_gpu.cuda.synchronize()
        ro = _report(f"configs[1] SoftPendulum-v0 4096x50 step {t + 1} obs", o.cpu().numpy(), o2, 1e-7)
        rr = _report(f"configs[1] SoftPendulum-v0 4096x50 step {t + 1} reward", r.cpu().numpy(), r2, 1e-9)
        assert ro["worst_over_band"] <= 1.0 and rr["worst_over_band"] <= 1.0, (ro, rr)
        np.testing.assert_array_equal(te.cpu().numpy(), te2)
        np.testing.assert_array_equal(tr.cpu().numpy(), tr2)
    st = env.backend.state_numpy()
    x = np.stack([r_.get("x") for r_ in ref.rods])
    rx = _report("configs[1] SoftPendulum-v0 4096x50 step 3 positions", st["x"], x, 1e-9)
    assert rx["worst_over_band"] <= 1.0, rx
    env.close()


def test_config3_every_one_of_4096_arms_of_100_elements(torch_gpu, hip_lib, oracle_built):
    import gym_softrobot_amd as gsa

    n = 4096
    env = gsa.make_vec("OctoArmSingle-v0", n, n_elems=100)
    assert "window" in env.backend.kernel_tier()
    env.reset(seed=0)
    ref = oracle_built.OracleArmBatch(gsa._capi.arm_single_config(1, n_elems=100), n, omp=True)
    ref.reset()
    acts = np.random.default_rng(2).uniform(-6, 6, (n, 7)).astype(np.float32)
    o, r, te, tr, _ = env.step(acts)
    o2, r2, te2, tr2 = ref.env_step(acts)
    torch_gpu.cuda.synchronize()
    ro = _report("configs[2] OctoArmSingle-v0 4096x100 step 1 obs", o.cpu().numpy(), o2, 2e-6, envs_per_wg=4, waves_per_env=2)
    rr = _report("configs[2] OctoArmSingle-v0 4096x100 step 1 reward", r.cpu().numpy(), r2, 1e-7, envs_per_wg=4, waves_per_env=2)
    st = env.backend.state_numpy()
    rx = _report("configs[2] OctoArmSingle-v0 4096x100 step 1 positions", st["x"], np.stack([q.get("x") for q in ref.rods]),
                 1e-8, envs_per_wg=4, waves_per_env=2)
    assert ro["worst_over_band"] <= 1.0 and rr["worst_over_band"] <= 1.0, (ro, rr)
    np.testing.assert_array_equal(te.cpu().numpy(), te2)
    np.testing.assert_array_equal(tr.cpu().numpy(), tr2)
    # node positions (near-zero transverse coordinates against an absolute floor of 1e-8): within the band, or the env
    # is ill-conditioned in the oracle itself (module docstring)
    cfg1 = gsa._capi.arm_single_config(1, n_elems=100)

    def rerun(i, eps):
        q = oracle_built.OracleRod(cfg1)
        q.reset_arm()
        x = q.get("x")
        x[0, 1:] *= 1.0 + eps
        q.set("x", x)
        q.env_step_arm(acts[i])
        return float(np.abs(q.get("x") - ref.rods[i].get("x")).max())

    assert _explained_by_conditioning(rx, rx["abs_dev_of_over_band_envs"], rerun, n) == [], rx
    env.close()


def test_config5_share_every_one_of_1024_octoflat_envs(torch_gpu, hip_lib, oracle_built):
    import gym_softrobot_amd as gsa

    n = 1024
    env = gsa.make_vec("OctoFlat-v0", n, numpy_output=True)
    assert env.backend.kernel_tier() == "softrod_octo_step_kernel<zup,2 waves,4 envs/wg>"
    env.reset(seed=0)
    ref = oracle_built.OracleOctoBatch(gsa._capi.octo_flat_config(1), n, omp=True)
    ref.reset(env.targets)
    acts = np.random.default_rng(3).uniform(-22, 22, (n, 24)).astype(np.float32)
    o, r, te, tr, _ = env.step(acts)
    o2, r2, te2, tr2 = ref.env_step(acts)
    ro = _report("configs[4] share OctoFlat-v0 1024x8x10 step 1 obs", o, o2, 2e-6, envs_per_wg=4, waves_per_env=2)
    rr = _report("configs[4] share OctoFlat-v0 1024x8x10 step 1 reward", r, r2, 1e-6, envs_per_wg=4, waves_per_env=2,
                 extra={"crossing_counts_equal": None})
    np.testing.assert_array_equal(te, te2)
    np.testing.assert_array_equal(tr, tr2)
    assert rr["worst_over_band"] <= 1.0, rr
    # 2857 substeps of eight arms on the frictional plane joined by stiff springs: a few envs of a thousand are
    # ill-conditioned already in their first whole step (the oracle's FMA build against itself leaves the band in
    # about 1 % of them)
    cfg1 = gsa._capi.octo_flat_config(1)

    def rerun(i, eps):
        q = oracle_built.OracleOcto(cfg1)
        q.reset(env.targets[i])
        for a in range(q.n_arm):
            x = q.arm(a).get("x")
            x[:2] *= 1.0 + eps
            q.arm(a).set("x", x)
        ob, _, _, _ = q.env_step(acts[i])
        flat = np.concatenate([ob["individual"].ravel(), ob["shared"]]).astype(np.float64)
        return float(np.abs(flat - o2[i].astype(np.float64)).max())

    assert _explained_by_conditioning(ro, ro["abs_dev_of_over_band_envs"], rerun, n) == [], ro
    assert ro["worst_over_band"] <= 100.0, ro          # still the same trajectory: 1e-3 of the entry at worst


def test_muscle_arm_every_env_of_a_thousand(torch_gpu, hip_lib, oracle_built):
    """N3 at batch scale: 1024 OctoArmPush-v1 envs (the `secondary` bench entry's workload, a quarter of its batch), one
    env.step each under the bench's own actions, every env against the oracle's ArmPush env.  PARITY UNPINNED underneath
    (the restated muscle law); what is held is HIP == oracle at rtol 1e-5 for every env, with the worst env's position."""
    import gym_softrobot_amd as gsa

    n = 1024
    env = gsa.make_vec("OctoArmPush-v1", n)
    env.reset(seed=0)
    cfg1 = gsa._capi.arm_push_config(1, mode="continuous")
    radii = gsa._capi.arm_push_radii(40)
    layers = gsa._capi.es_muscle_layers(radii, 0.012)
    rods = []
    for _ in range(n):
        r = oracle_built.OracleRod(cfg1)
        r.set_radius_profile(radii)
        r.set_muscle_layers(*layers)
        r.reset_push()
        rods.append(r)
    acts = np.random.default_rng(1).uniform(0.0, 1.0, (n, 2)).astype(np.float32)
    o, r_, te, tr, _ = env.step(acts)
    ref = [q.env_step_push(acts[i]) for i, q in enumerate(rods)]
    torch_gpu.cuda.synchronize()
    o2 = np.stack([x[0] for x in ref])
    r2 = np.array([x[1] for x in ref])
    ro = _report("N3 OctoArmPush-v1 1024x40 step 1 obs", o.cpu().numpy(), o2, 2e-7)
    rr = _report("N3 OctoArmPush-v1 1024x40 step 1 reward", r_.cpu().numpy(), r2, 1e-9)
    assert ro["worst_over_band"] <= 1.0 and rr["worst_over_band"] <= 1.0, (ro, rr)
    np.testing.assert_array_equal(te.cpu().numpy(), np.array([x[2] for x in ref]))
    np.testing.assert_array_equal(tr.cpu().numpy(), np.array([x[3] for x in ref]))
    env.close()
