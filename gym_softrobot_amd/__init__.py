"""gym_softrobot_amd — MI355X-native batched Cosserat-rod stepper behind the
Gymnasium surface of gym-softrobot's SoftPendulum-v0 / SoftPendulum3D-v0."""
from . import _capi
from .envs import (
    ArmPullWeightEnv,
    ArmPushEnv,
    ArmSingleEnv,
    ArmTwoEnv,
    CrawlEnv,
    FlatEnv,
    ReachEnv,
    SoftArmTrackingEnv,
    SoftPendulum3DEnv,
    SoftPendulumEnv,
    VecArmPullWeightEnv,
    VecArmPushEnv,
    VecArmSingleEnv,
    VecArmTwoEnv,
    VecCrawlEnv,
    VecOctoFlatEnv,
    VecReachEnv,
    VecSoftArmTrackingEnv,
    VecSoftPendulum3DEnv,
    VecSoftPendulumEnv,
)
from .registration import make, parity_label, register, registered

__version__ = "0.2.0"
VERSION = __version__                       # gym_softrobot/version.py

from .render import RendererType  # noqa: E402

# gym_softrobot/__init__.py:83 sets POVRAY; the matplotlib session is the one provided here
RENDERER_CONFIG = RendererType.MATPLOTLIB

# the batched form of every id: (class, the registration's kwargs)
_VEC = {
    "SoftPendulum-v0": (VecSoftPendulumEnv, {}),
    "SoftPendulum3D-v0": (VecSoftPendulum3DEnv, {}),
    "OctoArmSingle-v0": (VecArmSingleEnv, {}),
    "OctoFlat-v0": (VecOctoFlatEnv, {}),
    "OctoFlatLite-v0": (VecOctoFlatEnv, dict(n_arm=1, n_action=8)),   # gym_softrobot/__init__.py:11-15
    "SoftArmTracking-v0": (VecSoftArmTrackingEnv, {}),
    "OctoArmPush-v0": (VecArmPushEnv, {}),                            # gym_softrobot/__init__.py:37-46
    "OctoArmPush-v1": (VecArmPushEnv, dict(mode="continuous")),
    "OctoArmPullWeight-v0": (VecArmPullWeightEnv, dict(mode="continuous")),   # gym_softrobot/__init__.py:48-52
    "OctoCrawl-v0": (VecCrawlEnv, {}),                                # gym_softrobot/__init__.py:17-25
    "OctoReach-v0": (VecReachEnv, {}),
    "OctoArmTwo-v0": (VecArmTwoEnv, {}),                              # gym_softrobot/__init__.py:32-35
}

# gym_softrobot/__init__.py:27-30,74-80
register(id="SoftPendulum-v0", entry_point=SoftPendulumEnv, vector_entry_point=VecSoftPendulumEnv)
register(id="SoftPendulum3D-v0", entry_point=SoftPendulum3DEnv, vector_entry_point=VecSoftPendulum3DEnv)
register(id="OctoArmSingle-v0", entry_point=ArmSingleEnv, vector_entry_point=VecArmSingleEnv)
# gym_softrobot/__init__.py:6-15
register(id="OctoFlat-v0", entry_point=FlatEnv, vector_entry_point=VecOctoFlatEnv)
register(id="OctoFlatLite-v0", entry_point=FlatEnv, kwargs=dict(n_arm=1, n_action=8), vector_entry_point=VecOctoFlatEnv)
# gym_softrobot/__init__.py:60-63
register(id="SoftArmTracking-v0", entry_point=SoftArmTrackingEnv, vector_entry_point=VecSoftArmTrackingEnv)
# gym_softrobot/__init__.py:37-46 — the COOMM muscle arm: registered WITH its caveat (envs/arm_push.py)
from .envs.arm_push import PARITY_LABEL as _UNPINNED  # noqa: E402

register(id="OctoArmPush-v0", entry_point=ArmPushEnv, vector_entry_point=VecArmPushEnv, label=_UNPINNED)
register(id="OctoArmPush-v1", entry_point=ArmPushEnv, kwargs=dict(mode="continuous"), vector_entry_point=VecArmPushEnv,
         label=_UNPINNED)
# gym_softrobot/__init__.py:48-52
register(id="OctoArmPullWeight-v0", entry_point=ArmPullWeightEnv, kwargs=dict(mode="continuous"),
         vector_entry_point=VecArmPullWeightEnv, label=_UNPINNED)
# gym_softrobot/__init__.py:17-25,32-35 — the muscle octopus (octopus/build_muscle_octopus.py): the same caveat
register(id="OctoCrawl-v0", entry_point=CrawlEnv, vector_entry_point=VecCrawlEnv, label=_UNPINNED)
register(id="OctoReach-v0", entry_point=ReachEnv, vector_entry_point=VecReachEnv, label=_UNPINNED)
register(id="OctoArmTwo-v0", entry_point=ArmTwoEnv, vector_entry_point=VecArmTwoEnv, label=_UNPINNED)


def make_vec(id: str, num_envs: int, **kwargs):  # noqa: A002
    """N parallel envs on one GPU (the batched form of `make`)."""
    if id not in _VEC:
        raise KeyError(f"no batched implementation registered for {id!r}; have {sorted(_VEC)}")
    cls, kw = _VEC[id]
    return cls(num_envs, **{**kw, **kwargs})


__all__ = [
    "SoftPendulumEnv", "VecSoftPendulumEnv", "SoftPendulum3DEnv", "VecSoftPendulum3DEnv",
    "ArmSingleEnv", "VecArmSingleEnv", "FlatEnv", "VecOctoFlatEnv", "SoftArmTrackingEnv", "VecSoftArmTrackingEnv", "ArmPushEnv", "VecArmPushEnv", "ArmPullWeightEnv", "VecArmPullWeightEnv", "CrawlEnv", "VecCrawlEnv", "ArmTwoEnv", "VecArmTwoEnv", "ReachEnv", "VecReachEnv", "parity_label", "make", "make_vec", "register", "registered", "_capi",
]
