"""VectorEnv (Gymnasium/VectorEnv.cs:10-35) as a device-resident batch.

The reference loops over a List<Env> on one thread.  Here the list only
describes the batch: all envs must be the same device-backed type, and the
TimeLimit / RecordEpisodeStatistics wrappers around them are fused into the
step kernel.  State lives as SoA rows in HBM; Step moves only actions in and
(reward, done, observation) out, or nothing at all on the zero-copy path.
"""
import numpy as np

from . import _lib as L
from .env import EnvWrapper, RecordEpisodeStatistics, TimeLimit
from .envs import DeviceEnv, default_context


def _unwrap(env):
    """-> (device env, time_limit, record_stats); RecordEpisodeStatistics may only sit
    outside TimeLimit, the order in which the kernel applies them"""
    time_limit, record = 0, False
    seen_time_limit = False
    while isinstance(env, EnvWrapper):
        if isinstance(env, RecordEpisodeStatistics):
            if record or seen_time_limit:
                raise NotImplementedError("fused wrappers support RecordEpisodeStatistics(TimeLimit(env)) only")
            record = True
        elif isinstance(env, TimeLimit):
            if seen_time_limit:
                raise NotImplementedError("nested TimeLimit wrappers are not fused")
            seen_time_limit = True
            time_limit = env._maxSteps
        else:
            raise NotImplementedError("wrapper %s cannot be fused into the device step" % type(env).__name__)
        env = env.Env
    if not isinstance(env, DeviceEnv):
        raise TypeError("VectorEnv needs device-backed envs, got %s" % type(env).__name__)
    return env, time_limit, record


class VectorEnv:
    """VectorEnv<TState,TAction>"""

    def __init__(self, envs, autoreset=False, env_index_base=0):
        if len(envs) == 0:
            raise ValueError("VectorEnv needs at least one env")
        base, time_limit, record = _unwrap(envs[0])
        for e in envs[1:]:
            b, t, r = _unwrap(e)
            if type(b) is not type(base) or t != time_limit or r != record or b.batch_config() != base.batch_config():
                raise ValueError("all envs of a VectorEnv must share one type and configuration")
        self._proto = base
        self._wrappers = envs if record else None
        cfg = base.batch_config()
        flags = cfg["flags"] | (L.RECORD_STATS if record else 0) | (L.AUTORESET if autoreset else 0)
        self._make(cfg, len(envs), flags, time_limit, env_index_base)

    @classmethod
    def Make(cls, env_type, num_envs, seed=0, autoreset=False, record_stats=False, time_limit=0, device=0,
             mode=L.MODE_EXACT, env_index_base=0, **env_kwargs):
        """Batch of `num_envs` envs without building num_envs Python objects first."""
        self = cls.__new__(cls)
        self._proto = env_type(seed=seed, device=device, mode=mode, **env_kwargs)
        self._wrappers = None
        cfg = self._proto.batch_config()
        flags = cfg["flags"] | (L.RECORD_STATS if record_stats else 0) | (L.AUTORESET if autoreset else 0)
        self._make(cfg, num_envs, flags, time_limit, env_index_base)
        return self

    def _make(self, cfg, n, flags, time_limit, env_index_base):
        from .engine import DeviceBatch

        self._time_limit = time_limit
        self._batch = DeviceBatch(default_context(cfg["device"]), cfg["env_id"], n, seed=cfg["seed"], flags=flags,
                                  time_limit=time_limit, mode=cfg["mode"], map_rows=cfg["map_rows"],
                                  env_index_base=env_index_base, **cfg.get("extra", {}))
        self.ActionSpace = self._proto.ActionSpace
        self.ObservationSpace = self._proto.ObservationSpace

    @property
    def NumEnvs(self):  # VectorEnv.cs:13
        return self._batch.n

    @property
    def Batch(self):
        """the device batch: zero-copy views, rollouts, statistics"""
        return self._batch

    def _states(self, obs):
        return [self._proto._state_of(obs, i) for i in range(self._batch.n)]

    def Reset(self):
        """VectorEnv.cs:20-26 -> list of TState"""
        self._batch.reset()
        if self._wrappers is not None:
            for w in self._record_wrappers():
                # EnvWrapper.cs:34-38, on the host-side mirror of the per-env lists
                if w._currentLength > 0:
                    w.EpisodeRewards.append(w._currentReward)
                    w.EpisodeLengths.append(w._currentLength)
                w._currentReward, w._currentLength = 0.0, 0
        return self._states(self._batch.get_obs())

    def _record_wrappers(self):
        out = []
        for e in self._wrappers:
            while not isinstance(e, RecordEpisodeStatistics):
                e = e.Env
            out.append(e)
        return out

    def Step(self, actions):
        """VectorEnv.cs:28-34 -> list of (state, reward, done, info)"""
        obs, reward, done, trunc = self.StepArrays(actions, want_obs=True)
        states = self._states(obs)
        out = []
        for i in range(self._batch.n):
            info = self._proto._info(self._batch, i) if self._proto.ENV_ID in (L.ENV_PONG, L.ENV_BREAKOUT, L.ENV_SPACEINVADERS) else {}
            if trunc is not None and trunc[i]:
                info["TimeLimit.truncated"] = True  # Wrappers.TimeLimit.cs:30
            out.append((states[i], float(reward[i]), bool(done[i]), info))
        return out

    def StepArrays(self, actions, want_obs=True):
        """Same step, results as arrays: (obs, reward[N], done[N], truncated[N] or None)."""
        try:
            res = self._batch.step_host(actions, want_obs=want_obs, want_trunc=self._time_limit > 0)
            if self._wrappers is not None:
                # keep the caller's RecordEpisodeStatistics objects in step with the device
                # counters (EnvWrapper.cs:46-55); the aggregate lives on the device (Batch.stats())
                _, reward, done, _ = res
                for w, r, d in zip(self._record_wrappers(), reward, done):
                    w._currentReward += float(r)
                    w._currentLength += 1
                    if d:
                        w.EpisodeRewards.append(w._currentReward)
                        w.EpisodeLengths.append(w._currentLength)
                        w._currentReward, w._currentLength = 0.0, 0
            return res
        except L.GymxError as e:
            if e.code == L.GX_ERR_INVALID and self._proto.ENV_ID == L.ENV_ACROBOT:
                raise IndexError(str(e))  # Envs.Acrobot.cs:51
            raise

    def Close(self):
        self._batch.close()
