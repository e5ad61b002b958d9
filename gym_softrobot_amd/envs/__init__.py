from .arm_push import ArmPullWeightEnv, ArmPushEnv, VecArmPullWeightEnv, VecArmPushEnv
from .arm_single import ArmSingleEnv, VecArmSingleEnv
from .muscle_octopus import ArmTwoEnv, CrawlEnv, ReachEnv, VecArmTwoEnv, VecCrawlEnv, VecReachEnv
from .octo_flat import FlatEnv, VecOctoFlatEnv
from .soft_arm import SoftArmTrackingEnv, VecSoftArmTrackingEnv
from .soft_pendulum import SoftPendulumEnv, VecSoftPendulumEnv
from .soft_pendulum_3d import SoftPendulum3DEnv, VecSoftPendulum3DEnv

__all__ = [
    "SoftPendulumEnv", "VecSoftPendulumEnv", "SoftPendulum3DEnv", "VecSoftPendulum3DEnv",
    "ArmSingleEnv", "VecArmSingleEnv", "FlatEnv", "VecOctoFlatEnv",
    "SoftArmTrackingEnv", "VecSoftArmTrackingEnv", "ArmPushEnv", "VecArmPushEnv", "ArmPullWeightEnv", "VecArmPullWeightEnv",
    "CrawlEnv", "VecCrawlEnv", "ArmTwoEnv", "VecArmTwoEnv", "ReachEnv", "VecReachEnv",
]
