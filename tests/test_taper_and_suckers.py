"""SURVEY.md §8(f) N3, the part whose source IS on disk: ControllableFixConstraint
(gym_softrobot/envs/octopus/controllable_constraint.py:24-69, pinned by executing the reference's
class: tests/golden/ref_sucker.npz) and the tapered CosseratRod.straight_rod call of the muscle-arm
envs (octopus/arm_push_env.py:160-179).  COOMM's muscle force model is not on disk and is not
restated.  CPU: the oracle; the HIP path is held to the oracle in tests/test_gpu_taper_suckers.py."""
import json
from pathlib import Path

import numpy as np

GOLD = Path(__file__).parent / "golden"


def _bare_cfg(n_elems, **kw):
    from gym_softrobot_amd import _capi

    cfg = _capi.softpendulum_config(1, n_elems=n_elems)
    cfg.env_kind = _capi.ENV_NONE
    cfg.features = 0
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def test_sucker_constraint_against_the_reference_class(oracle_built):
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_sucker.npz")
    rec = json.loads((GOLD / "ref_build_records.json").read_text())["ControllableFixConstraint"]
    assert rec == {"default_reduction_ratio": 1.0, "default_flag": True}
    assert (~z["op_flag"]).any() and (z["op_ratio"] == 0.3).any()
    for i in range(len(z["op_index"])):
        cfg = _bare_cfg(20, features=_capi.FEAT_SUCKER_CONSTRAINT, n_suckers=1)
        cfg.sucker_index[0] = int(z["op_index"][i])
        cfg.sucker_reduction_ratio = float(z["op_ratio"][i])
        rod = oracle_built.OracleRod(cfg)
        rod.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
        if not z["op_flag"][i]:
            rod.set_sucker_ratio([0.0])              # controller.turn_off(): the constraint does nothing
        for name, key in (("x", "x_in"), ("v", "v_in"), ("Q", "Q_in"), ("w", "w_in")):
            rod.set(name, z["op_" + key][i])
        rod.constrain_probe()
        for name, key in (("x", "x_out"), ("v", "v_out"), ("Q", "Q_out"), ("w", "w_out")):
            np.testing.assert_array_equal(rod.get(name), z["op_" + key][i], err_msg=name)


def _arm_push_radii(n_elem, radius_base=0.012, radius_tip=0.001):
    radius = np.linspace(radius_base, radius_tip, n_elem + 1)        # arm_push_env.py:163-165
    return (radius[:-1] + radius[1:]) / 2


def test_tapered_allocation(oracle_built):
    """straight_rod(base_radius=radius_mean) of ArmPushEnv._build: masses, inertias and stiffnesses
    follow the per-element radius (the allocation formulas are PyElastica's, recalled)."""
    n = 40
    r = _arm_push_radii(n)
    cfg = _bare_cfg(n, base_length=0.2, density=700.0, youngs_modulus=1e4, shear_modulus=1e4 / 1.5)
    rod = oracle_built.OracleRod(cfg)
    rod.set_radius_profile(r)
    rod.reset_straight([0, 0, 0], [1, 0, 0], [0, 1, 0])
    ell = 0.2 / n
    vol = np.pi * r ** 2 * ell
    mass = np.zeros(n + 1)
    mass[:-1] += 0.5 * 700.0 * vol
    mass[1:] += 0.5 * 700.0 * vol
    np.testing.assert_allclose(rod.get("mass"), mass, rtol=1e-13)
    A = np.pi * r ** 2
    I1 = A * A / (4 * np.pi)
    np.testing.assert_allclose(rod.get("J")[0], I1 * 700.0 * ell, rtol=1e-12)
    np.testing.assert_allclose(rod.get("J")[2], 2 * I1 * 700.0 * ell, rtol=1e-12)
    np.testing.assert_allclose(rod.get("shear")[2], 1e4 * A, rtol=1e-13)
    np.testing.assert_allclose(rod.get("shear")[0], cfg.alpha_c * (1e4 / 1.5) * A, rtol=1e-13)
    np.testing.assert_allclose(rod.get("bend")[0], 0.5 * 1e4 * (I1[1:] + I1[:-1]), rtol=1e-12)   # Voronoi average
    np.testing.assert_allclose(rod.get("radius"), r, rtol=1e-11)      # sqrt(V / (pi (l + 1e-14)))


def test_tapered_cantilever_known_answer(oracle_built):
    """Static tip deflection of a clamped TAPERED rod under a small transverse tip force against
    the exact small-deflection answer of the discrete chain: bending rotations F (L - s_k) D / B_k
    at the Voronoi vertices plus the shear of every element, F l / (alpha_c G A_k)."""
    from gym_softrobot_amd import _capi

    n, L, F, E = 20, 1.0, 0.02, 1e6
    r = np.linspace(0.06, 0.03, n)
    cfg = _bare_cfg(n, base_length=L, density=1000.0, youngs_modulus=E, shear_modulus=E / 3.0, dt=2e-4,
                    damping_constant=0.8, features=_capi.FEAT_FIXED_BC | _capi.FEAT_TIP_FORCE | _capi.FEAT_ANALYTICAL_DAMPER)
    cfg.tip_force[1] = F
    rod = oracle_built.OracleRod(cfg)
    rod.set_radius_profile(r)
    rod.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    for _ in range(40):
        rod.substeps(0.0, 2500)
    tip = rod.get("x")[1, -1]
    ell = L / n
    A = np.pi * r ** 2
    EI = E * A * A / (4 * np.pi)
    B = 0.5 * (EI[1:] + EI[:-1])
    s = ell * np.arange(1, n)                      # Voronoi vertices
    bending = np.sum(F * (L - s) * ell / B * (L - s))
    shear = np.sum(F * ell / (cfg.alpha_c * (E / 3.0) * A))
    assert abs(np.max(np.abs(rod.get("v")))) < 1e-9          # at rest
    np.testing.assert_allclose(tip, bending + shear, rtol=2e-4)
    # the same load on the uniform rod of the mean radius deflects differently: the profile is in use
    cfg2 = cfg.copy()
    cfg2.base_radius = float(r.mean())
    u = oracle_built.OracleRod(cfg2)
    u.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    for _ in range(40):
        u.substeps(0.0, 2500)
    assert abs(u.get("x")[1, -1] - tip) > 0.05 * abs(tip)
