"""Run the reference's OWN env code (gym_softrobot/envs/**, utils/**) in this container, where
neither gymnasium nor pyelastica nor numba nor coomm can be imported.

What this module does — and does not — stand in for:

  * `gym_softrobot` and its sub-packages are registered as EMPTY package objects whose __path__
    points at the real directories under /root/reference, so `from gym_softrobot.envs.octopus.build
    import build_arm` executes the reference's real file while the package __init__ files (which
    pull in every env, coomm, gymnasium's registry) are not run.  Nothing is copied; bytecode
    writing is switched off so /root/reference stays untouched.
  * `gymnasium`: `Env` (reset(seed) -> self.np_random = Generator(PCG64(SeedSequence(seed))), what
    gymnasium.utils.seeding.np_random does), `spaces.Box` / `spaces.Dict` holding their arguments.
  * `numba.njit`: the identity decorator (it does not change what a function computes).
  * `elastica`: RECORDING stand-ins with no physics.  `CosseratRod.straight_rod(...)`, `Cylinder`,
    `Plane` return objects that remember their arguments and expose zero-filled arrays of the
    shapes PyElastica uses; the simulator mixins record every `append / constrain / add_forcing_to
    / dampen / connect / detect_contact_between / collect_diagnostics` call in order;
    `finalize()` instantiates the operator classes that the REFERENCE defines (subclasses of
    ConstraintBase / NoForces / FreeJoint written in the reference's files) the way PyElastica's
    mixins hand them their arguments (constraint: fixed positions/directors of the constrained
    indices, then the keywords) and leaves PyElastica's own operators (GravityForces,
    AnalyticalLinearDamper, ...) as records of their keywords.  `PositionVerlet().step` is a
    SCRIPTED stepper: it performs no integration; the fixture generator tells it what state and
    time the loop ends in (taken from this repo's oracle or synthetic), so that everything the
    reference does AROUND the stepper — set_action, the NaN check, rewards, truncation, get_state —
    runs as the reference wrote it on a state we know.
  * Three PyElastica one-liners the reference's epilogues call are restated here and therefore
    remain RECALLED (flagged in the fixtures' README): `_isnan_check(a)` = np.isnan(a).any(),
    `RodBase.compute_position_center_of_mass()` = sum_j m_j x_j / sum_j m_j (einsum form), and the
    argument order in which the Constraints mixin calls the constraint class.

TEST-FIXTURE TOOLING: used by tools/make_env_golden.py inside the build container only; nothing
here travels into the product path, and the GPU box never needs /root/reference.
"""
from __future__ import annotations

import importlib
import sys
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
PKG = REF / "gym_softrobot"


# ---------------------------------------------------------------------------------------------
# gymnasium
# ---------------------------------------------------------------------------------------------
def _np_random(seed=None):
    seq = np.random.SeedSequence(seed)
    return np.random.Generator(np.random.PCG64(seq)), seq.entropy


class _Env:
    metadata = {}
    render_mode = None
    _np_random = None

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random, self._np_random_seed = _np_random(seed)

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random, self._np_random_seed = _np_random()
        return self._np_random

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape)

    def contains(self, x):
        x = np.asarray(x)
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                    and np.all(x >= self.low) and np.all(x <= self.high))


class _Dict(dict):
    def __init__(self, spaces):
        super().__init__(spaces)


class _Discrete:
    def __init__(self, n, seed=None, start=0):
        self.n, self.start = int(n), int(start)
        self.shape, self.dtype = (), np.dtype(np.int64)


def _install_gymnasium():
    g = types.ModuleType("gymnasium")
    sp = types.ModuleType("gymnasium.spaces")
    sp.Box, sp.Dict, sp.Discrete = _Box, _Dict, _Discrete
    g.Env, g.spaces = _Env, sp
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = lambda *a, **k: None
    envs = types.ModuleType("gymnasium.envs")
    envs.registration = reg
    g.envs = envs
    for name, mod in (("gymnasium", g), ("gymnasium.spaces", sp), ("gymnasium.envs", envs),
                      ("gymnasium.envs.registration", reg)):
        sys.modules[name] = mod


# ---------------------------------------------------------------------------------------------
# elastica: recording stand-ins
# ---------------------------------------------------------------------------------------------
class FakeRod:
    """Arrays of a CosseratRod in PyElastica's shapes; filled by the fixture generator."""

    def __init__(self, n_elems, **recorded):
        n = int(n_elems)
        self.n_elems = n
        self.recorded = recorded
        self.position_collection = np.zeros((3, n + 1))
        self.velocity_collection = np.zeros((3, n + 1))
        self.acceleration_collection = np.zeros((3, n + 1))
        self.director_collection = np.zeros((3, 3, n))
        self.omega_collection = np.zeros((3, n))
        self.alpha_collection = np.zeros((3, n))
        self.tangents = np.zeros((3, n))
        self.kappa = np.zeros((3, n - 1))
        self.rest_kappa = np.zeros((3, n - 1))
        self.sigma = np.zeros((3, n))
        self.rest_sigma = np.zeros((3, n))
        self.external_forces = np.zeros((3, n + 1))
        self.external_torques = np.zeros((3, n))
        self.mass = np.zeros(n + 1)
        self.lengths = np.zeros(n)
        self.rest_lengths = np.zeros(n)
        self.radius = np.zeros(n)

    # elastica/rod/cosserat_rod.py compute_position_center_of_mass (RECALLED; see module docstring)
    def compute_position_center_of_mass(self):
        mass_times_position = np.einsum("j,ij->ij", self.mass, self.position_collection)
        sum_mass_times_position = np.einsum("ij->i", mass_times_position)
        return sum_mass_times_position / self.mass.sum()


class CosseratRod:
    @staticmethod
    def straight_rod(n_elements, start, direction, normal, base_length, base_radius, density, **kw):
        return FakeRod(n_elements, kind="CosseratRod.straight_rod", n_elements=int(n_elements),
                       start=np.array(start, dtype=np.float64), direction=np.array(direction, dtype=np.float64),
                       normal=np.array(normal, dtype=np.float64), base_length=base_length,
                       base_radius=base_radius, density=density, **kw)


class FakeRigidBody:
    def __init__(self, **recorded):
        self.recorded = recorded
        self.position_collection = np.zeros((3, 1))
        self.velocity_collection = np.zeros((3, 1))
        self.acceleration_collection = np.zeros((3, 1))
        self.director_collection = np.zeros((3, 3, 1))
        self.omega_collection = np.zeros((3, 1))
        self.alpha_collection = np.zeros((3, 1))
        self.external_forces = np.zeros((3, 1))
        self.external_torques = np.zeros((3, 1))


class Cylinder(FakeRigidBody):
    def __init__(self, start, direction, normal, base_length, base_radius, density):
        super().__init__(kind="Cylinder", start=np.array(start, dtype=np.float64),
                         direction=np.array(direction, dtype=np.float64), normal=np.array(normal, dtype=np.float64),
                         base_length=base_length, base_radius=base_radius, density=density)


class Plane:
    def __init__(self, plane_origin, plane_normal):
        self.recorded = dict(kind="Plane", plane_origin=np.array(plane_origin, dtype=np.float64),
                             plane_normal=np.array(plane_normal, dtype=np.float64))


class _Recorded:
    """A PyElastica operator that is only recorded (its arithmetic is not on disk)."""

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs


def _recorded(name):
    return type(name, (_Recorded,), {})


class ConstraintBase:
    def __init__(self, *args, **kwargs):
        self._system = kwargs.get("_system")
        self.constrained_position_idx = np.array(kwargs.get("constrained_position_idx", []), dtype=int)
        self.constrained_director_idx = np.array(kwargs.get("constrained_director_idx", []), dtype=int)


class FreeBC(ConstraintBase):
    pass


class NoForces:
    def __init__(self):
        pass

    def apply_forces(self, system, time=0.0):
        pass

    def apply_torques(self, system, time=0.0):
        pass


class FreeJoint:
    def __init__(self, k, nu):
        self.k, self.nu = k, nu


class CallBackBaseClass:
    def __init__(self):
        pass


class _Using:
    def __init__(self, sim, kind, targets):
        self.sim, self.kind, self.targets = sim, kind, targets

    def using(self, cls, *args, **kwargs):
        self.sim._ops.append({"kind": self.kind, "targets": self.targets, "cls": cls, "args": args, "kwargs": kwargs})
        return self


class BaseSystemCollection:
    def __init__(self):
        self._systems = []
        self._ops = []
        self._instances = []
        self._script = None
        self._calls = 0
        self.finalized = False

    def append(self, system):
        self._systems.append(system)
        self._ops.append({"kind": "append", "targets": (system,), "cls": type(system), "args": (), "kwargs": {}})

    def _sys(self, s):                       # build_octopus passes an index to dampen()
        return self._systems[s] if isinstance(s, (int, np.integer)) else s

    def finalize(self):
        """Instantiate the reference-defined operator classes as PyElastica's mixins do."""
        for op in self._ops:
            cls, kw = op["cls"], dict(op["kwargs"])
            inst = None
            if op["kind"] == "constrain" and issubclass(cls, ConstraintBase):
                rod = op["targets"][0]
                pos_idx = kw.get("constrained_position_idx", ())
                dir_idx = kw.get("constrained_director_idx", ())
                positions = [rod.position_collection[..., i].copy() for i in pos_idx]
                directors = [rod.director_collection[..., i].copy() for i in dir_idx]
                inst = cls(*positions, *directors, *op["args"], _system=rod, **kw)
            elif op["kind"] == "forcing" and issubclass(cls, NoForces):
                inst = cls(*op["args"], **kw)
            elif op["kind"] == "connect" and issubclass(cls, FreeJoint):
                inst = cls(*op["args"], **kw)
            elif op["kind"] in ("forcing", "damping", "contact", "constrain", "connect", "callback"):
                inst = _Recorded(*op["args"], **kw) if not issubclass(cls, CallBackBaseClass) else None
            op["instance"] = inst
        self.finalized = True

    def order(self):
        """Registration order as readable strings (what fixes the operator order of a substep)."""
        out = []
        for op in self._ops:
            tg = ",".join(str(self._systems.index(t)) if t in self._systems else "?" for t in op["targets"])
            out.append(f"{op['kind']}:{op['cls'].__name__}[{tg}]")
        return out


class Constraints:
    def constrain(self, system):
        return _Using(self, "constrain", (self._sys(system),))


class Forcing:
    def add_forcing_to(self, system):
        return _Using(self, "forcing", (self._sys(system),))


class Damping:
    def dampen(self, system):
        return _Using(self, "damping", (self._sys(system),))


class Connections:
    def connect(self, first_rod, second_rod, first_connect_idx=0, second_connect_idx=-1):
        u = _Using(self, "connect", (self._sys(first_rod), self._sys(second_rod)))
        u.indices = (first_connect_idx, second_connect_idx)
        self._last_connect_idx = (first_connect_idx, second_connect_idx)
        return u


class Contact:
    def detect_contact_between(self, a, b):
        return _Using(self, "contact", (self._sys(a), self._sys(b)))


class CallBacks:
    def collect_diagnostics(self, system):
        return _Using(self, "callback", (self._sys(system),))


class PositionVerlet:
    """SCRIPTED: `sim._script(call_index, time, dt) -> time'` decides what the loop ends in."""

    def step(self, sim, time, dt):
        sim._calls += 1
        if sim._script is None:
            raise RuntimeError("refshim: no script installed for the stepper")
        return sim._script(sim._calls, time, dt)


def _isnan_check(array):          # elastica/_calculus.py (RECALLED): njit np.isnan(array).any()
    return bool(np.isnan(array).any())


def _install_elastica():
    el = types.ModuleType("elastica")
    for cls in (BaseSystemCollection, Constraints, Forcing, Damping, Connections, Contact, CallBacks,
                PositionVerlet, ConstraintBase, FreeBC, NoForces, FreeJoint, CosseratRod, Cylinder, Plane):
        setattr(el, cls.__name__, cls)
    for name in ("GravityForces", "AnalyticalLinearDamper", "LaplaceDissipationFilter",
                 "RodPlaneContactWithAnisotropicFriction", "OneEndFixedBC", "Sphere", "MuscleTorques"):
        setattr(el, name, _recorded(name))
    calc = types.ModuleType("elastica._calculus")
    calc._isnan_check = _isnan_check
    cb = types.ModuleType("elastica.callback_functions")
    cb.CallBackBaseClass = CallBackBaseClass
    rod = types.ModuleType("elastica.rod")
    rod.RodBase = FakeRod
    el._calculus, el.callback_functions, el.rod = calc, cb, rod
    el.CallBackBaseClass = CallBackBaseClass
    for name, mod in (("elastica", el), ("elastica._calculus", calc), ("elastica.callback_functions", cb),
                      ("elastica.rod", rod)):
        sys.modules[name] = mod


def _install_misc():
    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    sys.modules["numba"] = nb
    # coomm: imported at module level by octopus/build.py, used only by the muscle envs
    for name, attr in (("coomm", None), ("coomm.actuations", None), ("coomm.actuations.muscles", None),
                       ("coomm.actuations.muscles.longitudinal_muscle", "LongitudinalMuscle"),
                       ("coomm.actuations.muscles.transverse_muscle", "TransverseMuscle")):
        m = types.ModuleType(name)
        if attr:
            setattr(m, attr, type(attr, (), {}))
        sys.modules[name] = m


def _stub_package(name, path):
    m = types.ModuleType(name)
    m.__path__ = [str(path)]
    m.__package__ = name
    sys.modules[name] = m
    return m


def install():
    """Install every stand-in; afterwards `importlib.import_module("gym_softrobot.envs....")` runs
    the reference's real files."""
    sys.dont_write_bytecode = True
    _install_gymnasium()
    _install_elastica()
    _install_misc()
    root = _stub_package("gym_softrobot", PKG)
    cfg = importlib.import_module("gym_softrobot.config")          # real file: RendererType enum
    root.RENDERER_CONFIG = cfg.RendererType.POVRAY                   # gym_softrobot/__init__.py:83
    _stub_package("gym_softrobot.envs", PKG / "envs")
    _stub_package("gym_softrobot.envs.octopus", PKG / "envs" / "octopus")
    _stub_package("gym_softrobot.utils", PKG / "utils")
    _stub_package("gym_softrobot.utils.custom_elastica", PKG / "utils" / "custom_elastica")
    rnd = _stub_package("gym_softrobot.utils.render", PKG / "utils" / "render")
    pp = types.ModuleType("gym_softrobot.utils.render.post_processing")
    pp.plot_video = lambda *a, **k: None
    br = types.ModuleType("gym_softrobot.utils.render.base_renderer")
    br.BaseRenderer = type("BaseRenderer", (), {})
    br.BaseElasticaRendererSession = type("BaseElasticaRendererSession", (), {})
    rnd.post_processing, rnd.base_renderer = pp, br
    sys.modules[pp.__name__] = pp
    sys.modules[br.__name__] = br


def load(module: str):
    return importlib.import_module(module)
