"""MI355X-native batched environment engine behind the API surface of the C#
Gymnasium port (Env / VectorEnv / EnvRegistry / Spaces)."""
from . import spaces
from .env import (ActionWrapper, Env, EnvWrapper, ObservationWrapper, RecordEpisodeStatistics, RewardWrapper,
                  TimeLimit)
from .envs import (Acrobot, Blackjack, Breakout, CartPole, CliffWalking, DummyEnv, FrozenLake, MountainCar,
                   MountainCarContinuous, Pendulum, Pong, SpaceInvaders, TableEnv, Taxi)
from .registry import EnvRegistry, RegisterAll
from .spaces import Box, Dict, Discrete, MultiBinary, MultiDiscrete, Tuple
from .vector_env import VectorEnv

__all__ = ["spaces", "Env", "EnvWrapper", "ObservationWrapper", "ActionWrapper", "RewardWrapper",
           "RecordEpisodeStatistics", "TimeLimit", "CartPole", "MountainCar", "MountainCarContinuous", "Pendulum",
           "Acrobot", "FrozenLake", "Taxi", "TableEnv", "CliffWalking", "Blackjack", "Pong", "Breakout", "SpaceInvaders", "DummyEnv", "EnvRegistry", "RegisterAll",
           "Box", "Discrete", "MultiBinary", "MultiDiscrete", "Tuple", "Dict", "VectorEnv"]
