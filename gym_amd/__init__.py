"""MI355X-native batched environment engine behind the API surface of the C#
Gymnasium port (Env / VectorEnv / EnvRegistry / Spaces)."""
from . import _lib  # noqa: F401
