"""Box and Discrete, with the members agents read through duck typing
(Gymnasium/Spaces.Box.cs, Spaces.Discrete.cs, Spaces.Space.cs).

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
