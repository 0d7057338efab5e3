"""EnvRegistry and GymnasiumRegistration (Gymnasium/EnvRegistry.cs,
GymnasiumRegistration.cs) for the environments of the device path."""
from . import envs as _envs


class EnvRegistry:
    """EnvRegistry.cs:11-35.  Register does not append to `training_debug.log`
    as the reference does on every call (EnvRegistry.cs:16)."""

    _registry = {}

    @classmethod
    def Register(cls, id, factory):
        cls._registry[id] = factory

    @classmethod
    def Make(cls, id, expected_type=None):
        """EnvRegistry.cs:19-32: unknown id -> ArgumentException (ValueError here);
        a type mismatch -> InvalidCastException (TypeError here)."""
        if id not in cls._registry:
            raise ValueError("No environment registered with id '%s'" % id)
        env = cls._registry[id]()
        if expected_type is not None and not isinstance(env, expected_type):
            raise TypeError("Registered environment '%s' does not match requested types." % id)
        return env

    @classmethod
    def List(cls):
        return list(cls._registry.keys())


def RegisterAll():
    """GymnasiumRegistration.cs:8-30, restricted to the ids with a device kernel.  The
    Box2D-backed and the Atari/MuJoCo stub ids of the reference are
    not registered: Make() on them fails as for any unknown id."""
    for cls in (_envs.CartPole, _envs.MountainCar, _envs.MountainCarContinuous, _envs.Acrobot, _envs.Pendulum,
                _envs.FrozenLake, _envs.Taxi, _envs.CliffWalking, _envs.Blackjack, _envs.Pong, _envs.Breakout, _envs.SpaceInvaders):
        EnvRegistry.Register(cls.ID, cls)
