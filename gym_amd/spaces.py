"""Box, Discrete and the composite spaces, with the members agents read through duck typing
(Gymnasium/Spaces.*.cs).

Host-side metadata only.  Sample() uses a numpy generator per space (the
reference owns an unseeded System.Random per space, Spaces.Box.cs:14); batched
action sampling on the device goes through the engine's counter RNG instead.
"""
import numpy as np


class Space:
    """Spaces.Space.cs:8-13"""

    @property
    def Shape(self):
        raise NotImplementedError

    def Sample(self):
        raise NotImplementedError

    def Contains(self, x):
        raise NotImplementedError


class Box(Space):
    """Spaces.Box.cs:8-44"""

    def __init__(self, low, high, seed=None):
        low = np.asarray(low, np.float32).reshape(-1)
        high = np.asarray(high, np.float32).reshape(-1)
        if low.shape != high.shape:
            raise ValueError("Low and high must have the same length.")  # ArgumentException, Spaces.Box.cs:18-19
        self.Low = low
        self.High = high
        self.Dimension = int(low.shape[0])
        self._rng = np.random.default_rng(seed)

    @property
    def Shape(self):
        return [self.Dimension]

    def Sample(self):
        # (float)(NextDouble() * (High[i] - Low[i]) + Low[i]): float range, double arithmetic, one rounding
        u = self._rng.random(self.Dimension)
        rng32 = (self.High - self.Low).astype(np.float32)
        return (u * rng32.astype(np.float64) + self.Low.astype(np.float64)).astype(np.float32)

    def Contains(self, x):
        x = np.asarray(x, np.float32).reshape(-1)
        if x.shape[0] != self.Dimension:
            return False
        return bool(np.all((x >= self.Low) & (x <= self.High)))


class Discrete(Space):
    """Spaces.Discrete.cs:8-20"""

    def __init__(self, n, seed=None):
        self.N = int(n)
        self._rng = np.random.default_rng(seed)

    @property
    def Shape(self):
        return [1]

    def Sample(self):
        return int(self._rng.integers(0, self.N))

    def Contains(self, x):
        return 0 <= int(x) < self.N


class MultiBinary(Space):
    """Spaces.MultiBinary.cs:8-34"""

    def __init__(self, n, seed=None):
        self.N = int(n)
        self._rng = np.random.default_rng(seed)

    @property
    def Shape(self):
        return [self.N]

    def Sample(self):
        return [bool(b) for b in self._rng.integers(0, 2, self.N)]

    def Contains(self, x):
        return len(x) == self.N  # the reference checks the length only (Spaces.MultiBinary.cs:30-33)


class MultiDiscrete(Space):
    """Spaces.MultiDiscrete.cs:8-39"""

    def __init__(self, nvec, seed=None):
        self.Nvec = [int(n) for n in nvec]
        self._rng = np.random.default_rng(seed)

    @property
    def Shape(self):
        return [len(self.Nvec)]

    def Sample(self):
        return [int(self._rng.integers(0, n)) for n in self.Nvec]

    def Contains(self, x):
        return len(x) == len(self.Nvec) and all(0 <= int(v) < n for v, n in zip(x, self.Nvec))


class Tuple(Space):
    """Spaces.Tuple.cs:8-39"""

    def __init__(self, spaces):
        self.Spaces = list(spaces)

    @property
    def Shape(self):
        return [len(self.Spaces)]  # the number of elements (Spaces.Tuple.cs:11)

    def Sample(self):
        return [s.Sample() for s in self.Spaces]

    def Contains(self, x):
        return len(x) == len(self.Spaces) and all(s.Contains(v) for s, v in zip(self.Spaces, x))


class Dict(Space):
    """Spaces.Dict.cs:8-39"""

    def __init__(self, spaces):
        self.Spaces = dict(spaces)

    @property
    def Shape(self):
        return [len(self.Spaces)]  # the number of keys (Spaces.Dict.cs:11)

    def Sample(self):
        return {k: s.Sample() for k, s in self.Spaces.items()}

    def Contains(self, x):
        return all(k in x and s.Contains(x[k]) for k, s in self.Spaces.items())
