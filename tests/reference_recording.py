"""Reader, writer and replayer of the recordings csharp/ReplayRecorder.cs makes from the REFERENCE
environments (file layout: see that file's header).  A recording holds outputs of the reference
itself -- the one thing that can pin the oracle and the HIP path to it -- so the replay is
teacher-forced: every step starts from the recorded pre-step state, takes the recorded actions and
must give the recorded reward, done flag and post-step state.  The reference's unseeded
System.Random cannot be reproduced, so components a draw inside Step sets are excluded on the steps
where that draw happens (Pong: ball velocity after a point, Envs.Pong.cs:159-165; Breakout: ballX and
ballVelX when FIRE releases the ball, Envs.Breakout.cs:97-103) and checked for range instead.

Bars: integer rows, rewards of the tabular and frame envs, done flags, frames and the binary64 game
state of Pong / Breakout bit-exact; binary32 continuous control within 1e-5 relative per step (the
number of elements that differ at all is reported).
"""
import struct

import numpy as np

MAGIC = b"GXR1"
FRAME_BYTES = 210 * 160 * 3
PONG, BREAKOUT = 8, 10
REL_TOL = 1e-5


class Recording:
    def __init__(self, env_id, n, n_real, real_is_f64, n_int, action_is_float, frame_every, init, actions, rewards, dones,
                 images, frames):
        self.env_id, self.n, self.n_real, self.real_is_f64, self.n_int = env_id, n, n_real, real_is_f64, n_int
        self.action_is_float, self.frame_every = action_is_float, frame_every
        self.init, self.actions, self.rewards, self.dones, self.images, self.frames = init, actions, rewards, dones, images, frames

    @property
    def steps(self):
        return len(self.actions)

    @property
    def image_bytes(self):
        return self.n * (self.n_real * (8 if self.real_is_f64 else 4) + self.n_int * 4)

    def split(self, image):
        rdt = np.float64 if self.real_is_f64 else np.float32
        nb = self.n * self.n_real * np.dtype(rdt).itemsize
        return image[:nb].view(rdt).reshape(self.n_real, self.n), image[nb:].view(np.int32).reshape(self.n_int, self.n)


def read(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != MAGIC:
        raise ValueError("%s is not a GXR1 recording" % path)
    env_id, n, steps, n_real, real_is_f64, n_int, action_is_float, frame_every = struct.unpack_from("<8i", data, 4)
    rec = Recording(env_id, n, n_real, bool(real_is_f64), n_int, bool(action_is_float), frame_every, None, [], [], [], [], {})
    pos = 36
    ib = rec.image_bytes

    def take(nbytes, dtype):
        nonlocal pos
        if pos + nbytes > len(data):
            raise ValueError("%s is truncated" % path)
        a = np.frombuffer(data, dtype, nbytes // np.dtype(dtype).itemsize, pos).copy()
        pos += nbytes
        return a

    rec.init = take(ib, np.uint8)
    for t in range(steps):
        rec.actions.append(take(4 * n, np.float32 if action_is_float else np.int32))
        rec.rewards.append(take(8 * n, np.float64))
        rec.dones.append(take(n, np.uint8))
        rec.images.append(take(ib, np.uint8))
        if frame_every > 0 and (t + 1) % frame_every == 0:
            rec.frames[t] = take(FRAME_BYTES, np.uint8)
    if pos != len(data):
        raise ValueError("%s has %d trailing bytes" % (path, len(data) - pos))
    return rec


def write(path, rec):
    """the same layout ReplayRecorder.cs writes (used to test this module without a .NET machine)"""
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<8i", rec.env_id, rec.n, rec.steps, rec.n_real, int(rec.real_is_f64), rec.n_int,
                            int(rec.action_is_float), rec.frame_every))
        f.write(rec.init.tobytes())
        for t in range(rec.steps):
            f.write(rec.actions[t].tobytes())
            f.write(rec.rewards[t].astype(np.float64).tobytes())
            f.write(rec.dones[t].astype(np.uint8).tobytes())
            f.write(rec.images[t].tobytes())
            if t in rec.frames:
                f.write(rec.frames[t].tobytes())


def record_from(engine_step, get_state_raw, get_obs, env_id, n, n_real, real_is_f64, n_int, action_is_float, actions, frame_every=0):
    """build a Recording by driving any engine (the oracle stands in for the reference in the self-test)"""
    rec = Recording(env_id, n, n_real, real_is_f64, n_int, action_is_float, frame_every, get_state_raw().copy(), [], [], [], [], {})
    for t, a in enumerate(actions):
        rew, done = engine_step(a)
        rec.actions.append(np.array(a))
        rec.rewards.append(np.array(rew, np.float64))
        rec.dones.append(np.array(done, np.uint8))
        rec.images.append(get_state_raw().copy())
        if frame_every > 0 and (t + 1) % frame_every == 0:
            rec.frames[t] = get_obs()[0].copy()
    return rec


def draw_for_speed(vel):
    """A value u of `_random.NextDouble()` with (u - 0.5) * 2.0 * 3.0 == vel bit for bit (Envs.Pong.cs:164,
    Envs.Breakout.cs:100; both ball speeds are 3): (u[n], found[n]).  The expression is monotone in u, so such a value
    lies within a few units in the last place of vel / 6 + 0.5."""
    vel = np.asarray(vel, np.float64)
    u = np.zeros_like(vel)
    found = np.zeros(vel.shape, bool)
    lo = vel / 6.0 + 0.5
    hi = lo.copy()
    for k in range(17):
        for cand in ((lo,) if k == 0 else (lo, hi)):
            ok = ~found & (cand >= 0.0) & (cand < 1.0) & (bits64((cand - 0.5) * 2.0 * 3.0) == bits64(vel))
            u[ok] = cand[ok]
            found |= ok
        lo = np.nextafter(lo, -np.inf)
        hi = np.nextafter(hi, np.inf)
    return u, found


def pong_draws_from(vel_x, vel_y):
    """Values of `_random.NextDouble()` that Envs.Pong.cs:163-164 turns into a recorded ball velocity: (u[2][n], found[n]).
    The direction needs any value on the right side of 0.5."""
    u1, found = draw_for_speed(vel_y)
    return np.stack([np.where(np.asarray(vel_x) > 0, 0.75, 0.25), u1]), found


def bits64(a):
    return np.ascontiguousarray(a, np.float64).view(np.uint64)


def replay(rec, set_state_raw, step, get_state_raw, get_obs=None, set_rng_draws=None):
    """Teacher-forced replay through an engine given as three callables.  Returns a dict of counts;
    raises AssertionError where a bar is missed.  With set_rng_draws (an engine that takes the values of Pong's ball
    reset and Breakout's release of the ball from its caller) what those draws set is checked too: the values are
    recovered from the recorded post-step state and handed over before the step."""
    exact_reals = rec.real_is_f64  # Pong / Breakout game state: plain binary64 arithmetic, no libm
    report = {"steps": rec.steps, "real_elements": 0, "real_elements_differing": 0, "max_rel": 0.0, "unpinned_events": 0,
              "frames": 0}
    pre = rec.init
    for t in range(rec.steps):
        set_state_raw(pre)
        want_r, want_i = rec.split(rec.images[t])
        pre_r, pre_i = rec.split(pre)
        recovered = None
        if rec.env_id == PONG and set_rng_draws is not None and (rec.rewards[t] != 0).any():
            u, recovered = pong_draws_from(want_r[2], want_r[3])
            set_rng_draws(u)  # envs that score no point on this step ignore theirs
        if rec.env_id == BREAKOUT and set_rng_draws is not None and ((rec.actions[t] == 1) & (pre_i[2] == 1)).any():
            # the release leaves ballVelX as drawn (the stuck ball sits over the paddle, three pixels cannot reach a wall)
            u, recovered = draw_for_speed(want_r[2])
            set_rng_draws(u)
        rew, done = step(rec.actions[t])
        got_r, got_i = rec.split(np.ascontiguousarray(get_state_raw()))
        assert np.array_equal(np.asarray(done, np.uint8), rec.dones[t]), "done differs at step %d" % t
        assert np.array_equal(got_i, want_i), "integer state differs at step %d" % t
        mask = np.ones((rec.n_real, rec.n), bool)  # components the reference's unseeded Random did not set
        if rec.env_id == PONG:
            scored = rec.rewards[t] != 0
            if recovered is not None:
                report["replayed_draws"] = report.get("replayed_draws", 0) + int((scored & recovered).sum())
                scored = scored & ~recovered  # what is left is a velocity no value of NextDouble() gives
            mask[2:4, scored] = False
            assert np.isin(np.abs(want_r[2, scored]), [3.0]).all() and (np.abs(want_r[3, scored]) <= 3.0).all()
            report["unpinned_events"] += int(scored.sum())
        if rec.env_id == BREAKOUT:
            fired = (rec.actions[t] == 1) & (pre_i[2] == 1)
            if recovered is not None:
                report["replayed_draws"] = report.get("replayed_draws", 0) + int((fired & recovered).sum())
                fired = fired & ~recovered
            mask[0, fired] = False
            mask[2, fired] = False
            assert (np.abs(want_r[2, fired]) <= 3.0).all()
            report["unpinned_events"] += int(fired.sum())
        rew = np.asarray(rew, np.float64)
        if rec.n_real == 0 or exact_reals:
            assert np.array_equal(rew, rec.rewards[t]), "reward differs at step %d" % t
        else:
            assert np.allclose(rew, rec.rewards[t], rtol=REL_TOL, atol=1e-6), "reward differs at step %d" % t
        if rec.n_real:
            g, w = got_r[mask], want_r[mask]
            differ = g.view(np.uint64 if rec.real_is_f64 else np.uint32) != w.view(np.uint64 if rec.real_is_f64 else np.uint32)
            report["real_elements"] += int(mask.sum())
            report["real_elements_differing"] += int(differ.sum())
            if differ.any():
                assert not exact_reals, "binary64 game state differs at step %d" % t
                rel = np.abs(g[differ].astype(np.float64) - w[differ]) / np.maximum(np.abs(w[differ].astype(np.float64)), 0.1)
                report["max_rel"] = max(report["max_rel"], float(rel.max()))
                assert rel.max() <= REL_TOL, "state relative error %g at step %d" % (rel.max(), t)
        if t in rec.frames and get_obs is not None:
            set_state_raw(rec.images[t])  # the frame shows the recorded post-step state
            assert np.array_equal(np.asarray(get_obs()[0]).reshape(-1), rec.frames[t]), "frame differs at step %d" % t
            report["frames"] += 1
        pre = rec.images[t]
    return report
