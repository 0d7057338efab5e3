"""The environments of the device path as Env objects (Gymnasium/Envs.*.cs).

Every instance is backed by a device-resident batch of one env: Reset/Step
launch the same kernels VectorEnv uses, there is no host implementation of any
dynamics.  Constructing an env needs no GPU; the first Reset()/Step() does.
"""
import numpy as np

from . import _lib as L
from .env import Env
from .spaces import Box, Discrete

_contexts = {}


def default_context(device=0):
    from .engine import Context

    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


class DeviceEnv(Env):
    """Base of the device-backed envs; subclasses set ENV_ID, spaces and the state shape."""

    ENV_ID = None
    ID = None
    _instances = 0  # process-wide count of envs constructed without a seed

    def __init__(self, seed=None, device=0, mode=L.MODE_EXACT, flags=0, map_rows=None):
        # The reference gives every env its own unseeded Random (Envs.CartPole.cs:19), so two envs
        # built side by side differ.  Here an env built without a seed takes the next stream of
        # seed 0 (its index keys the counter RNG); an explicit seed is deterministic: stream 0 of it.
        self._stream = 0
        if seed is None:
            seed = 0
            self._stream = DeviceEnv._instances
            DeviceEnv._instances += 1
        self._seed = seed
        self._device = device
        self._mode = mode
        self._flags = flags
        self._map_rows = map_rows
        self._batch = None

    # -- engine plumbing -------------------------------------------------
    def _ensure(self):
        if self._batch is None:
            from .engine import DeviceBatch

            self._batch = DeviceBatch(default_context(self._device), self.ENV_ID, 1, seed=self._seed, mode=self._mode,
                                      flags=self._flags, map_rows=self._map_rows, env_index_base=self._stream,
                                      **self._batch_extra())
        return self._batch

    def _batch_extra(self):
        """further DeviceBatch arguments of this env type (the caller-supplied table)"""
        return {}

    def batch_config(self):
        """what VectorEnv needs to build one batch for a list of these envs"""
        return dict(env_id=self.ENV_ID, seed=self._seed, mode=self._mode, flags=self._flags, map_rows=self._map_rows,
                    device=self._device, extra=self._batch_extra())

    def _state_of(self, obs, index=0):
        """observation arrays of a batch -> the reference's TState for env `index`"""
        raise NotImplementedError

    def _info(self, batch, index=0):
        return {}

    # -- Env ---------------------------------------------------------------
    def Seed(self, seed=None):
        """Extension: the reference's Seed is empty (Env.cs:33-36); here it re-keys the counter RNG."""
        if seed is not None:
            self._seed = int(seed)
            self._stream = 0
            if self._batch is not None:
                self._batch.close()
                self._batch = None

    def Reset(self):
        b = self._ensure()
        b.reset()
        return self._state_of(b.get_obs())

    def Step(self, action):
        b = self._ensure()
        obs, reward, done, _ = b.step_host(np.array([action], b.action_dtype), want_obs=True)
        return self._state_of(obs), float(reward[0]), bool(done[0]), self._info(b)

    def Render(self, mode="human"):
        print("State: %s" % (self._state_of(self._ensure().get_obs()),))

    def Close(self):
        if self._batch is not None:
            self._batch.close()
            self._batch = None


class _TupleStateEnv(DeviceEnv):
    def _state_of(self, obs, index=0):
        return tuple(float(x) for x in obs[:, index])


class _ArrayStateEnv(DeviceEnv):
    def _state_of(self, obs, index=0):
        # a fresh array per call: the reference hands out its internal array (Envs.Pendulum.cs:59)
        return np.array(obs[:, index], obs.dtype)  # float32; float64 in the FP64 mode


class _IntStateEnv(DeviceEnv):
    def _state_of(self, obs, index=0):
        return int(obs[index])


class CartPole(_TupleStateEnv):
    """Envs.CartPole.cs:10-83: Env<(float,float,float,float), int>"""
    ENV_ID, ID = L.ENV_CARTPOLE, "CartPole-v1"

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Discrete(2)
        self.ObservationSpace = Box([-2.4, -3.0, -0.2095, -3.0], [2.4, 3.0, 0.2095, 3.0])


class MountainCar(_TupleStateEnv):
    """Envs.MountainCar.cs:10-58: Env<(float,float), int>"""
    ENV_ID, ID = L.ENV_MOUNTAINCAR, "MountainCar-v0"

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Discrete(3)
        self.ObservationSpace = Box([-1.2, -0.07], [0.6, 0.07])


class MountainCarContinuous(_TupleStateEnv):
    """Envs.MountainCarContinuous.cs:10-57: Env<(float,float), float>"""
    ENV_ID, ID = L.ENV_MOUNTAINCAR_CONT, "MountainCarContinuous-v0"

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Box([-1.0], [1.0])
        self.ObservationSpace = Box([-1.2, -0.07], [0.6, 0.07])


class Pendulum(_ArrayStateEnv):
    """Envs.Pendulum.cs:10-68: Env<float[], float>"""
    ENV_ID, ID = L.ENV_PENDULUM, "Pendulum-v1"

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Box([-2.0], [2.0])
        self.ObservationSpace = Box([-1.0, -1.0, -8.0], [1.0, 1.0, 8.0])


class Acrobot(_ArrayStateEnv):
    """Envs.Acrobot.cs:10-99: Env<float[], int>"""
    ENV_ID, ID = L.ENV_ACROBOT, "Acrobot-v1"

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Discrete(3)
        self.ObservationSpace = Box([-1, -1, -1, -1, -6, -6], [1, 1, 1, 1, 6, 6])

    def Step(self, action):
        try:
            return super().Step(action)
        except L.GymxError as e:
            if e.code == L.GX_ERR_INVALID:
                raise IndexError(str(e))  # IndexOutOfRangeException, Envs.Acrobot.cs:51
            raise


class FrozenLake(_IntStateEnv):
    """Envs.FrozenLake.cs:10-88: Env<int,int>; `map` as the constructor argument (:23)"""
    ENV_ID, ID = L.ENV_FROZENLAKE, "FrozenLake-v1"

    def __init__(self, map=None, **kw):
        super().__init__(map_rows=list(map) if map is not None else None, **kw)
        rows = list(map) if map is not None else ["SFFF", "FHFH", "FFFH", "HFFG"]
        self.ActionSpace = Discrete(4)
        self.ObservationSpace = Discrete(len(rows) * len(rows[0]))


class TableEnv(_IntStateEnv):
    """Env<int,int> stepped from a transition table the caller supplies: what FrozenLake's `map` argument
    (Envs.FrozenLake.cs:23) generalises to.  `table[s][a]` (or `[s][a][o]` with three equally likely outcomes) holds
    (next_state, reward, done) records of engine.TRANSITION_DTYPE; Reset() sets `start_state` (-1: a uniform draw over
    the states, as Envs.Taxi.cs:20); `step_limit` is the env's own `steps >= limit` term of done (0: none).  An action
    outside Discrete(n_actions) raises IndexError, as an out-of-range index does in the reference."""
    ENV_ID, ID = L.ENV_TABLE, None

    def __init__(self, table, start_state=0, step_limit=0, **kw):
        from .engine import TRANSITION_DTYPE

        super().__init__(**kw)
        t = np.ascontiguousarray(table, dtype=TRANSITION_DTYPE)
        if t.ndim == 2:
            t = t[:, :, None]
        if t.ndim != 3:
            raise ValueError("table must have shape [n_states, n_actions] or [n_states, n_actions, n_outcomes]")
        self._table, self._start_state, self._step_limit = t, int(start_state), int(step_limit)
        self.ActionSpace = Discrete(t.shape[1])
        self.ObservationSpace = Discrete(t.shape[0])

    def _batch_extra(self):
        return dict(table=_TableKey(self._table), n_outcomes=self._table.shape[2], start_state=self._start_state,
                    step_limit=self._step_limit)

    def Step(self, action):
        try:
            return super().Step(action)
        except L.GymxError as e:
            if e.code == L.GX_ERR_INVALID:
                raise IndexError(str(e))
            raise


class _TableKey:
    """a table that compares by content, so that envs built from equal tables share a VectorEnv"""

    def __init__(self, array):
        self.array = array

    def __eq__(self, other):
        return isinstance(other, _TableKey) and self.array.shape == other.array.shape and self.array.tobytes() == other.array.tobytes()

    def __array__(self, dtype=None, copy=None):
        return self.array


class Taxi(_IntStateEnv):
    """Envs.Taxi.cs:10-41 (a stub in the reference: the state never changes)"""
    ENV_ID, ID = L.ENV_TAXI, "Taxi-v3"

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Discrete(6)
        self.ObservationSpace = Discrete(500)


class CliffWalking(_IntStateEnv):
    """Envs.CliffWalking.cs:10-77"""
    ENV_ID, ID = L.ENV_CLIFFWALKING, "CliffWalking-v0"

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Discrete(4)
        self.ObservationSpace = Discrete(48)


class Blackjack(DeviceEnv):
    """Envs.Blackjack.cs:10-48: Env<(int,int,bool), int>.  The reference's Step is a ten-step stub; `full=True`
    plays upstream Blackjack-v1 (infinite deck, natural = False, sab = False), an extension."""
    ENV_ID, ID = L.ENV_BLACKJACK, "Blackjack-v1"

    def __init__(self, full=False, **kw):
        kw["flags"] = kw.get("flags", 0) | (L.BLACKJACK_FULL if full else 0)
        super().__init__(**kw)
        self.ActionSpace = Discrete(2)  # 0: stick, 1: hit
        self.ObservationSpace = Discrete(32 * 11 * 2)

    def _state_of(self, obs, index=0):
        # the device keeps the observation as the index of Discrete(32*11*2)
        i = int(obs[index])
        return (i // 22, (i // 2) % 11, bool(i & 1))


class Pong(DeviceEnv):
    """Envs.Pong.cs:12-266: Env<byte[], int>; the observation is the 210x160x3 frame"""
    ENV_ID, ID = L.ENV_PONG, "Pong-v4"
    FRAME = 210 * 160 * 3

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Discrete(4)
        self.ObservationSpace = Box(np.zeros(self.FRAME, np.float32), np.full(self.FRAME, 255.0, np.float32))

    @property
    def Spec(self):
        return {"id": "Pong-v4", "max_episode_steps": 1000}  # Envs.Pong.cs:43

    @property
    def Metadata(self):
        return {"render_modes": ["human", "rgb_array"]}

    @property
    def RewardRange(self):
        return (-1, 1)

    def _state_of(self, obs, index=0):
        return np.array(obs[index], np.uint8)

    def _info(self, batch, index=0):
        reals, ints = batch.get_state()  # Envs.Pong.cs:148-154
        return {"left_score": int(ints[0, index]), "right_score": int(ints[1, index]),
                "ball_x": float(reals[0, index]), "ball_y": float(reals[1, index])}


class Breakout(DeviceEnv):
    """Envs.Breakout.cs:12-360: Env<byte[], int>; the observation is the 210x160x3 frame"""
    ENV_ID, ID = L.ENV_BREAKOUT, "Breakout-v4"
    FRAME = 210 * 160 * 3

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Discrete(4)  # 0 NOOP, 1 FIRE, 2 RIGHT, 3 LEFT
        self.ObservationSpace = Box(np.zeros(self.FRAME, np.float32), np.full(self.FRAME, 255.0, np.float32))

    @property
    def Spec(self):
        return {"id": "Breakout-v4", "max_episode_steps": 10000}  # Envs.Breakout.cs:50

    @property
    def Metadata(self):
        return {"render_modes": ["human", "rgb_array"]}

    @property
    def RewardRange(self):
        return (0, 7)

    def _state_of(self, obs, index=0):
        return np.array(obs[index], np.uint8)

    def _info(self, batch, index=0):
        reals, ints = batch.get_state()  # Envs.Breakout.cs:203-210
        words = ints[3:7, index].astype(np.int64) & 0xFFFFFFFF
        return {"score": int(ints[0, index]), "lives": int(ints[1, index]), "ball_x": float(reals[0, index]),
                "ball_y": float(reals[1, index]), "bricks_remaining": sum(bin(int(w)).count("1") for w in words)}


class SpaceInvaders(DeviceEnv):
    """Envs.SpaceInvaders.cs:12-541: Env<byte[], int>; the observation is the 210x160x3 frame"""
    ENV_ID, ID = L.ENV_SPACEINVADERS, "SpaceInvaders-v4"
    FRAME = 210 * 160 * 3

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Discrete(6)  # 0 NOOP, 1 FIRE, 2 RIGHT, 3 LEFT, 4 RIGHTFIRE, 5 LEFTFIRE
        self.ObservationSpace = Box(np.zeros(self.FRAME, np.float32), np.full(self.FRAME, 255.0, np.float32))

    @property
    def Spec(self):
        return {"id": "SpaceInvaders-v4", "max_episode_steps": 10000}  # Envs.SpaceInvaders.cs:53

    @property
    def Metadata(self):
        return {"render_modes": ["human", "rgb_array"]}

    @property
    def RewardRange(self):
        return (0, 200)

    def _state_of(self, obs, index=0):
        return np.array(obs[index], np.uint8)

    def _info(self, batch, index=0):
        reals, ints = batch.get_state()  # Envs.SpaceInvaders.cs:181-187
        bits = (int(ints[6, index]) & 0xFFFFFFFF) | ((int(ints[7, index]) & 0xFFFFFFFF) << 32)
        return {"score": int(ints[0, index]), "lives": int(ints[1, index]), "aliens_remaining": bin(bits).count("1"),
                "player_x": float(reals[0, index])}


class DummyEnv(_IntStateEnv):
    """DummyEnv.cs:10-38"""
    ENV_ID, ID = L.ENV_DUMMY, "Dummy"

    def __init__(self, **kw):
        super().__init__(**kw)
        self.ActionSpace = Discrete(3)
        self.ObservationSpace = Discrete(20)
