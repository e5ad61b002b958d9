"""gym_softrobot_amd — MI355X-native batched Cosserat-rod stepper behind the
Gymnasium surface of gym-softrobot's SoftPendulum-v0."""
from . import _capi
from .envs import SoftPendulumEnv, VecSoftPendulumEnv
from .registration import make, register, registered

__version__ = "0.1.0"

# gym_softrobot/__init__.py:74-76
register(id="SoftPendulum-v0", entry_point=SoftPendulumEnv)


def make_vec(id: str, num_envs: int, **kwargs):  # noqa: A002
    """N parallel envs on one GPU (the batched form of `make`)."""
    if id != "SoftPendulum-v0":
        raise KeyError(f"no batched implementation registered for {id!r}")
    return VecSoftPendulumEnv(num_envs, **kwargs)


__all__ = [
    "SoftPendulumEnv", "VecSoftPendulumEnv", "make", "make_vec", "register", "registered", "_capi",
]
