"""Env and the wrapper bases, mirroring Gymnasium/Env.cs and EnvWrapper.cs.

Method and property names keep the reference's spelling (Reset, Step,
ActionSpace, ...) so code written against the C# library reads the same.
"""


class Env:
    """Env<TState,TAction>, Env.cs:7-52"""

    def Reset(self):
        raise NotImplementedError

    def Step(self, action):
        """-> (state, reward: float, done: bool, info: dict), Env.cs:18"""
        raise NotImplementedError

    def Render(self, mode="human"):
        raise NotImplementedError

    def Close(self):
        raise NotImplementedError

    def Seed(self, seed=None):
        """Env.cs:33-36: an empty virtual in the reference"""

    @property
    def Spec(self):
        return None

    @property
    def Metadata(self):
        return None

    @property
    def RewardRange(self):
        return None


class EnvWrapper(Env):
    """EnvWrapper.cs:8-21"""

    def __init__(self, env):
        self.Env = env

    def Reset(self):
        return self.Env.Reset()

    def Step(self, action):
        return self.Env.Step(action)

    def Render(self, mode="human"):
        return self.Env.Render(mode)

    def Close(self):
        return self.Env.Close()

    # what agents read through duck typing falls through to the wrapped env
    def __getattr__(self, name):
        if name in ("ActionSpace", "ObservationSpace"):
            return getattr(self.Env, name)
        raise AttributeError(name)


class RecordEpisodeStatistics(EnvWrapper):
    """EnvWrapper.cs:23-58"""

    def __init__(self, env):
        super().__init__(env)
        self.EpisodeRewards = []
        self.EpisodeLengths = []
        self._currentReward = 0.0
        self._currentLength = 0

    def Reset(self):
        if self._currentLength > 0:
            self.EpisodeRewards.append(self._currentReward)
            self.EpisodeLengths.append(self._currentLength)
        self._currentReward = 0.0
        self._currentLength = 0
        return super().Reset()

    def Step(self, action):
        state, reward, done, info = super().Step(action)
        self._currentReward += reward
        self._currentLength += 1
        if done:
            self.EpisodeRewards.append(self._currentReward)
            self.EpisodeLengths.append(self._currentLength)
            self._currentReward = 0.0
            self._currentLength = 0
        return state, reward, done, info


class TimeLimit(EnvWrapper):
    """Wrappers.TimeLimit.cs:6-34"""

    def __init__(self, env, maxSteps):
        super().__init__(env)
        self._maxSteps = int(maxSteps)
        self._steps = 0

    def Reset(self):
        self._steps = 0
        return super().Reset()

    def Step(self, action):
        state, reward, done, info = super().Step(action)
        self._steps += 1
        if self._steps >= self._maxSteps:
            done = True
            info["TimeLimit.truncated"] = True
        return state, reward, done, info


class ObservationWrapper(EnvWrapper):
    """EnvWrapper.cs:60-70"""

    def Reset(self):
        return self.TransformObservation(super().Reset())

    def Step(self, action):
        state, reward, done, info = super().Step(action)
        return self.TransformObservation(state), reward, done, info

    def TransformObservation(self, observation):
        return observation


class ActionWrapper(EnvWrapper):
    """EnvWrapper.cs:72-81"""

    def Step(self, action):
        return super().Step(self.TransformAction(action))

    def TransformAction(self, action):
        return action


class RewardWrapper(EnvWrapper):
    """EnvWrapper.cs:83-93"""

    def Step(self, action):
        state, reward, done, info = super().Step(action)
        return state, self.TransformReward(reward), done, info

    def TransformReward(self, reward):
        return reward
