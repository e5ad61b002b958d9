"""HIP path vs the oracle for the on-disk part of SURVEY.md §8(f) N3: a TAPERED rod
(softrod_set_radius_profile: CosseratRod.straight_rod(base_radius=<array>), arm_push_env.py:160-179)
and ControllableFixConstraint suckers (controllable_constraint.py:24-69) whose ratios the caller
changes between steps (arm_two_env.py:228).  Both math modes; through the C-ABI.  The reference
class itself pins the oracle's constraint in tests/test_taper_and_suckers.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


def _radii(n, base=0.012, tip=0.001):
    r = np.linspace(base, tip, n + 1)                  # arm_push_env.py:163-165
    return (r[:-1] + r[1:]) / 2


def _cfg(n_envs, n, math_mode, features, **kw):
    from gym_softrobot_amd import _capi

    cfg = _capi.softpendulum_config(n_envs, n_elems=n, math_mode=math_mode)
    cfg.env_kind = _capi.ENV_NONE
    cfg.features = features
    cfg.base_length, cfg.density, cfg.youngs_modulus, cfg.shear_modulus = 0.2, 700.0, 1e4, 1e4 / 1.5
    cfg.damping_constant = 0.05 * 2 * 1e2               # arm_push_env.py:166,183
    cfg.dt = 1e-4
    cfg.gravity[0], cfg.gravity[1], cfg.gravity[2] = 0.0, 0.0, -9.81
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def _compare(be, rods, atol=1e-9):
    st = be.state_numpy()
    for i, r in enumerate(rods):
        for name in ("x", "v", "w", "Q"):
            np.testing.assert_allclose(st[name][i], r.get(name), rtol=RTOL, atol=atol, err_msg=f"{name} env {i}")


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_tapered_arm_with_suckers_matches_oracle(torch_gpu, hip_lib, oracle_built, math_mode):
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n, N = 40, 3
    feats = _capi.FEAT_GRAVITY | _capi.FEAT_ANALYTICAL_DAMPER | _capi.FEAT_SUCKER_CONSTRAINT
    cfg = _cfg(N, n, math_mode, feats, n_suckers=2, sucker_reduction_ratio=1.0)
    cfg.sucker_index[0], cfg.sucker_index[1] = 0, 25
    be = HipRodBackend(cfg, 0)
    be.set_radius_profile(_radii(n))
    start, direction, normal = np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])
    be.reset_straight(start, direction, normal)
    with pytest.raises(_capi.SoftrodError):
        be.set_radius_profile(_radii(n))                # only before the first reset
    rods = []
    for i in range(N):
        c1 = cfg.copy()
        c1.n_envs = 1
        r = oracle_built.OracleRod(c1)
        r.set_radius_profile(_radii(n))
        r.reset_straight(start, direction, normal)
        rods.append(r)
    st = be.state()
    # phase 1: both suckers hold fully in env 0; env 1 has the second one at 0.3; env 2 has none
    ratios = np.array([[1.0, 1.0, 0.0], [1.0, 0.3, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])     # [sucker][env]
    st["sucker_ratio"][:] = torch_gpu.from_numpy(ratios).to(st["sucker_ratio"].device)
    for i, r in enumerate(rods):
        r.set_sucker_ratio(ratios[:2, i])
    be.substeps(None, 250)
    for r in rods:
        r.substeps(0.0, 250)
    torch_gpu.cuda.synchronize()
    _compare(be, rods)
    x = be.state_numpy()["x"]
    assert abs(x[0, :, 0]).max() < 1e-12 and x[2, 2, 0] < -1e-4      # node 0 held in env 0, falling in env 2
    # phase 2: the base sucker lets go in env 0, the other weakens (arm_two_env.py:228)
    ratios[:, 0] = [0.0, 0.9, 0.0, 0.0]
    st["sucker_ratio"][:] = torch_gpu.from_numpy(ratios).to(st["sucker_ratio"].device)
    rods[0].set_sucker_ratio(ratios[:2, 0])
    be.substeps(None, 250)
    for r in rods:
        r.substeps(0.0, 250)
    torch_gpu.cuda.synchronize()
    _compare(be, rods)
    be.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_tapered_arm_on_the_plane_matches_oracle(torch_gpu, hip_lib, oracle_built, math_mode):
    """Per-element contact radius: the thick base of a tapered arm lies a hair above the plane,
    its thin tip well above it; 60 ms of falling, landing and sliding."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n = 40
    feats = _capi.FEAT_GRAVITY | _capi.FEAT_ANALYTICAL_DAMPER | _capi.FEAT_PLANE_CONTACT_ANISO
    arm = _capi.arm_single_config(1)
    cfg = _cfg(2, n, math_mode, feats, contact_k=arm.contact_k, contact_nu=arm.contact_nu,
               slip_velocity_tol=arm.slip_velocity_tol, surface_tol=arm.surface_tol)
    for i in range(3):
        cfg.kinetic_mu[i], cfg.static_mu[i] = arm.kinetic_mu[i], arm.static_mu[i]
        cfg.plane_normal[i] = [0.0, 0.0, 1.0][i]
        cfg.plane_origin[i] = [0.0, 0.0, -0.012][i]
    be = HipRodBackend(cfg, 0)
    be.set_radius_profile(_radii(n))
    start, direction, normal = np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 0.0, 1.0])
    be.reset_straight(start, direction, normal)
    c1 = cfg.copy()
    c1.n_envs = 1
    rod = oracle_built.OracleRod(c1)
    rod.set_radius_profile(_radii(n))
    rod.reset_straight(start, direction, normal)
    be.substeps(None, 600)
    rod.substeps(0.0, 600)
    torch_gpu.cuda.synchronize()
    _compare(be, [rod, rod], atol=1e-8)
    z = be.state_numpy()["x"][0, 2]
    assert abs(z[0]) < 2e-3          # the plane holds the base (free fall would be at -0.0177 by now)
    be.close()


def test_snapshot_of_a_tapered_handle_is_refused_by_other_tapers(torch_gpu, hip_lib):
    """The radius profile holds masses, stiffnesses and damping per lane OUTSIDE softrod_config:
    the snapshot fingerprint carries a digest of it, so a tapered batch does not
    restore into a uniform handle or one with another taper, and does into the same taper."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n = 40
    feats = _capi.FEAT_GRAVITY | _capi.FEAT_ANALYTICAL_DAMPER
    start, direction, normal = np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])

    def handle(radii):
        be = HipRodBackend(_cfg(2, n, 1, feats), 0)
        if radii is not None:
            be.set_radius_profile(radii)
        be.reset_straight(start, direction, normal)
        return be

    tapered, same, other, uniform = handle(_radii(n)), handle(_radii(n)), handle(_radii(n, tip=0.002)), handle(None)
    tapered.substeps(None, 50)
    snap = tapered.snapshot()
    same.restore(snap)
    same.substeps(None, 20)
    tapered.substeps(None, 20)
    torch_gpu.cuda.synchronize()
    np.testing.assert_array_equal(same.state_numpy()["x"], tapered.state_numpy()["x"])
    for be in (other, uniform):
        with pytest.raises(ValueError, match="radius profile"):
            be.restore(snap)
    for be in (tapered, same, other, uniform):
        be.close()


def test_tapered_arm_single_env_matches_oracle(torch_gpu, hip_lib, oracle_built):
    """VecArmSingleEnv(radius_profile=...): the OctoArmSingle feature set (gravity, plane contact with
    anisotropic friction, damper, rest-curvature actuation, reward/observation epilogue) on a rod tapered
    12:1 like arm_push_env.py:160-165 — the tapered instantiation `bench.py --taper` measures — against
    the oracle stepping the same env, three env.steps of 714 substeps."""
    import gym_softrobot_amd as gsa
    from tests.oracle_backend import OracleBackend

    n, N = 50, 2
    r0 = gsa._capi.arm_single_config(1).base_radius
    edge = np.linspace(r0, r0 / 12.0, n + 1)
    prof = (edge[:-1] + edge[1:]) / 2
    env = gsa.make_vec("OctoArmSingle-v0", N, n_elems=n, radius_profile=prof)
    ref = gsa.make_vec("OctoArmSingle-v0", N, n_elems=n, radius_profile=prof,
                       backend=OracleBackend(gsa._capi.arm_single_config(N, n_elems=n)))
    env.reset(seed=0)
    ref.reset(seed=0)
    acts = np.random.default_rng(5).uniform(-6, 6, (3, N, 7)).astype(np.float32)
    for t in range(3):
        o, r, te, tr, _ = env.step(acts[t])
        o2, r2, te2, tr2, _ = ref.step(acts[t])
        torch_gpu.cuda.synchronize()
        np.testing.assert_allclose(o.cpu().numpy(), np.asarray(o2), rtol=RTOL, atol=2e-6, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r.cpu().numpy(), np.asarray(r2), rtol=RTOL, atol=1e-8, err_msg=f"reward step {t}")
        np.testing.assert_array_equal(te.cpu().numpy().astype(bool), np.asarray(te2).astype(bool))
        np.testing.assert_array_equal(tr.cpu().numpy().astype(bool), np.asarray(tr2).astype(bool))
    st = env.backend.state_numpy()
    for i in range(N):
        for name in ("x", "v", "w", "Q"):
            np.testing.assert_allclose(st[name][i], ref.backend.rods[i].get(name), rtol=RTOL, atol=1e-8,
                                       err_msg=f"{name} env {i}")
    env.close()
    ref.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_known_answer_rolling_without_slipping_on_the_gpu(torch_gpu, hip_lib, math_mode):
    """The static rolling-friction branch as a physics known answer on the HIP kernels themselves (no
    oracle involved): an unactuated arm on the plane pushed across its axis by a body force m gy rolls
    without slipping at 2 gy / 3 and spins at v / r; pushed along its axis below mu_s g it stays put."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n_sub, dt, r = 500, 7e-5, 0.35 * 0.02
    for gx, gy in ((0.0, 0.5), (1.0, 0.0)):
        cfg = _capi.arm_single_config(3, n_elems=20, math_mode=math_mode)
        cfg.damping_constant = 0.0
        cfg.gravity[0], cfg.gravity[1] = gx, gy
        be = HipRodBackend(cfg, 0)
        be.reset_straight(np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 0.0, 1.0]))
        be.substeps(None, n_sub)
        torch_gpu.cuda.synchronize()
        st = be.state_numpy()
        for e in range(3):
            if gy:
                vy = st["v"][e, 1]
                np.testing.assert_allclose(vy, 2.0 / 3.0 * gy * n_sub * dt, rtol=1e-6)
                np.testing.assert_allclose(st["w"][e, 2] * r, -vy[:-1], rtol=1e-6)
            else:
                assert np.abs(st["v"][e, 0]).max() < 1e-12
        be.close()
