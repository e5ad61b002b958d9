from .soft_pendulum import SoftPendulumEnv, VecSoftPendulumEnv

__all__ = ["SoftPendulumEnv", "VecSoftPendulumEnv"]
