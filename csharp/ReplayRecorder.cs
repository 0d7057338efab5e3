// Records a run of the REFERENCE environments (the CPU classes of Gymnasium/Envs.*.cs) into a
// plain binary file that the replay harness of the GPU engine reads
// (tests/reference_recording.py, tests/test_reference_recording.py): initial states, the actions
// taken, and every step's reward, done flag and post-step state.  The reference keeps its state
// in private fields and has no getters, so the recorder reads them by reflection, in the order
// of the engine's SoA state image (include/gymx.h, gx_vec_get_state).
//
// This is the only route to pinning the engine against outputs of the reference itself: run it
// once on a machine with a .NET SDK, e.g.
//     ReplayRecorder.Record("CartPole-v1", numEnvs: 256, steps: 64, seed: 1, "tests/golden/csharp/cartpole.gxr");
// and commit the files; the tests pick up whatever tests/golden/csharp/*.gxr holds.
// Not compiled in the build container (no .NET SDK).
//
// File layout, little-endian:
//   char[4] "GXR1"
//   int32   env_id (gx_env_id), num_envs, steps, n_real, real_is_f64, n_int, action_is_float, frame_every
//   image                                   -- states after Reset()
//   steps x { action[num_envs] (int32 or float), reward[num_envs] (float64), done[num_envs] (uint8), image,
//             frame of env 0 (100800 bytes) when frame_every > 0 and (t + 1) % frame_every == 0 }
//   image = n_real rows of num_envs reals (float32, or float64 when real_is_f64), then n_int rows of int32.
// Draws of the reference's unseeded System.Random cannot be reproduced by another engine; the
// harness therefore starts every step from the recorded pre-step state.  What a draw inside Step
// sets (Pong's ball velocity after a point, Breakout's after FIRE) it replays as well: it recovers
// values of NextDouble() that give the recorded velocity bit for bit and hands them to the engine
// (gx_vec_set_rng_draws) before the step, so the draws themselves need not be recorded.
using System;
using System.Collections.Generic;
using System.IO;
using System.Reflection;
using Gymnasium.Envs;

namespace Gymnasium.Gpu;

public static class ReplayRecorder
{
    private const BindingFlags Private = BindingFlags.Instance | BindingFlags.NonPublic;

    private sealed record Layout(GxEnvId Id, int NReal, bool RealIsF64, int NInt, bool ActionIsFloat, int ActionN,
                                 float ActionLow, float ActionHigh, Func<object> Make, Action<object, double[], int[]> Read);

    private static T Field<T>(object env, string name)
    {
        FieldInfo f = env.GetType().GetField(name, Private)
            ?? throw new MissingFieldException(env.GetType().Name, name);
        return (T)f.GetValue(env)!;
    }

    private static void Tuple2(object env, double[] r) { var s = Field<(float, float)>(env, "_state"); r[0] = s.Item1; r[1] = s.Item2; }
    private static void Array(object env, double[] r) { float[] s = Field<float[]>(env, "_state"); for (int k = 0; k < s.Length; k++) r[k] = s[k]; }

    private static readonly Dictionary<string, Layout> Layouts = new()
    {
        // Envs.CartPole.cs:18-20
        ["CartPole-v1"] = new(GxEnvId.CartPole, 4, false, 1, false, 2, 0, 0, () => new CartPole(), (e, r, i) =>
        {
            var s = Field<(float, float, float, float)>(e, "_state");
            r[0] = s.Item1; r[1] = s.Item2; r[2] = s.Item3; r[3] = s.Item4; i[0] = Field<int>(e, "_steps");
        }),
        ["MountainCar-v0"] = new(GxEnvId.MountainCar, 2, false, 1, false, 3, 0, 0, () => new MountainCar(),
            (e, r, i) => { Tuple2(e, r); i[0] = Field<int>(e, "_steps"); }),                                   // Envs.MountainCar.cs:18-20
        ["MountainCarContinuous-v0"] = new(GxEnvId.MountainCarContinuous, 2, false, 1, true, 0, -1f, 1f, () => new MountainCarContinuous(),
            (e, r, i) => { Tuple2(e, r); i[0] = Field<int>(e, "_steps"); }),                                   // Envs.MountainCarContinuous.cs:15-17
        ["Pendulum-v1"] = new(GxEnvId.Pendulum, 3, false, 1, true, 0, -2f, 2f, () => new Pendulum(),
            (e, r, i) => { Array(e, r); i[0] = Field<int>(e, "_steps"); }),                                    // Envs.Pendulum.cs:15-17
        ["Acrobot-v1"] = new(GxEnvId.Acrobot, 6, false, 1, false, 3, 0, 0, () => new Acrobot(),
            (e, r, i) => { Array(e, r); i[0] = Field<int>(e, "_steps"); }),                                    // Envs.Acrobot.cs:18-20
        ["FrozenLake-v1"] = new(GxEnvId.FrozenLake, 0, false, 2, false, 4, 0, 0, () => new FrozenLake(),
            (e, r, i) => { i[0] = Field<int>(e, "_state"); i[1] = Field<int>(e, "_steps"); }),                 // Envs.FrozenLake.cs:15-17
        ["Taxi-v3"] = new(GxEnvId.Taxi, 0, false, 2, false, 6, 0, 0, () => new Taxi(),
            (e, r, i) => { i[0] = Field<int>(e, "_state"); i[1] = Field<int>(e, "_steps"); }),                 // Envs.Taxi.cs:14-16
        ["CliffWalking-v0"] = new(GxEnvId.CliffWalking, 0, false, 2, false, 4, 0, 0, () => new CliffWalking(),
            (e, r, i) => { i[0] = Field<int>(e, "_state"); i[1] = Field<int>(e, "_steps"); }),                 // Envs.CliffWalking.cs:14,20
        ["Blackjack-v1"] = new(GxEnvId.Blackjack, 0, false, 3, false, 2, 0, 0, () => new Blackjack(), (e, r, i) =>
        {                                                                                                      // Envs.Blackjack.cs:14-18
            var s = Field<(int, int, bool)>(e, "_state");
            i[0] = (s.Item1 * 11 + s.Item2) * 2 + (s.Item3 ? 1 : 0); i[1] = 0; i[2] = Field<int>(e, "_steps");
        }),
        ["Pong-v4"] = new(GxEnvId.Pong, 6, true, 3, false, 4, 0, 0, () => new Pong(), (e, r, i) =>
        {                                                                                                      // Envs.Pong.cs:27-33
            r[0] = Field<double>(e, "_ballX"); r[1] = Field<double>(e, "_ballY");
            r[2] = Field<double>(e, "_ballVelX"); r[3] = Field<double>(e, "_ballVelY");
            r[4] = Field<double>(e, "_leftPaddleY"); r[5] = Field<double>(e, "_rightPaddleY");
            i[0] = Field<int>(e, "_leftScore"); i[1] = Field<int>(e, "_rightScore"); i[2] = Field<int>(e, "_steps");
        }),
        ["Breakout-v4"] = new(GxEnvId.Breakout, 5, true, 8, false, 4, 0, 0, () => new Breakout(), (e, r, i) =>
        {                                                                                                      // Envs.Breakout.cs:32-41
            r[0] = Field<double>(e, "_ballX"); r[1] = Field<double>(e, "_ballY");
            r[2] = Field<double>(e, "_ballVelX"); r[3] = Field<double>(e, "_ballVelY"); r[4] = Field<double>(e, "_paddleX");
            i[0] = Field<int>(e, "_score"); i[1] = Field<int>(e, "_lives"); i[2] = Field<bool>(e, "_ballStuck") ? 1 : 0;
            bool[,] bricks = Field<bool[,]>(e, "_bricks");   // [row, col], 6 x 18; bit row*18+col of the four words
            uint[] w = new uint[4];
            for (int row = 0; row < 6; row++)
                for (int col = 0; col < 18; col++)
                    if (bricks[row, col]) { int b = row * 18 + col; w[b >> 5] |= 1u << (b & 31); }
            for (int k = 0; k < 4; k++) i[3 + k] = unchecked((int)w[k]);
            i[7] = Field<int>(e, "_steps");
        }),
        // SpaceInvaders keeps its bullets in unbounded lists and fires on a timer with a random shooter; it has no
        // recorder layout (the engine's parity for it rests on the restatements only).
    };

    public static IEnumerable<string> SupportedIds => Layouts.Keys;

    /// <summary>Steps `numEnvs` fresh reference envs for `steps` steps with actions drawn from `seed` and writes the file.</summary>
    public static void Record(string id, int numEnvs, int steps, int seed, string path, int frameEvery = 0)
    {
        Layout L = Layouts[id];
        bool frames = frameEvery > 0 && (L.Id == GxEnvId.Pong || L.Id == GxEnvId.Breakout);
        var envs = new List<object>(numEnvs);
        for (int k = 0; k < numEnvs; k++) envs.Add(L.Make());
        var rng = new Random(seed);
        using var w = new BinaryWriter(File.Create(path));   // BinaryWriter is little-endian on every platform
        w.Write(new[] { (byte)'G', (byte)'X', (byte)'R', (byte)'1' });
        foreach (int v in new[] { (int)L.Id, numEnvs, steps, L.NReal, L.RealIsF64 ? 1 : 0, L.NInt, L.ActionIsFloat ? 1 : 0, frames ? frameEvery : 0 })
            w.Write(v);
        foreach (dynamic e in envs) e.Reset();
        WriteImage(w, L, envs);
        var reward = new double[numEnvs];
        var done = new byte[numEnvs];
        for (int t = 0; t < steps; t++)
        {
            byte[]? frame0 = null;
            for (int k = 0; k < numEnvs; k++)
            {
                dynamic e = envs[k];
                if (L.ActionIsFloat)
                {
                    // beyond the Box on purpose (a factor 1.25): the envs clamp, and the clamp is part of what is recorded
                    float a = (float)((rng.NextDouble() * 2.0 - 1.0) * 1.25 * L.ActionHigh);
                    w.Write(a);
                    var res = e.Step(a);
                    reward[k] = res.Item2; done[k] = (byte)(res.Item3 ? 1 : 0);
                }
                else
                {
                    int a = rng.Next(L.ActionN);
                    w.Write(a);
                    var res = e.Step(a);
                    reward[k] = res.Item2; done[k] = (byte)(res.Item3 ? 1 : 0);
                    if (k == 0 && frames) frame0 = ((byte[])res.Item1).AsSpan().ToArray();   // `_screen` is reused: copy it now
                }
            }
            foreach (double x in reward) w.Write(x);
            w.Write(done);
            WriteImage(w, L, envs);
            if (frames && (t + 1) % frameEvery == 0) w.Write(frame0!);
        }
    }

    private static void WriteImage(BinaryWriter w, Layout L, List<object> envs)
    {
        int n = envs.Count;
        var reals = new double[Math.Max(L.NReal, 1) * n];
        var ints = new int[L.NInt * n];
        var r = new double[16];
        var i = new int[12];
        for (int k = 0; k < n; k++)
        {
            L.Read(envs[k], r, i);
            for (int c = 0; c < L.NReal; c++) reals[c * n + k] = r[c];
            for (int c = 0; c < L.NInt; c++) ints[c * n + k] = i[c];
        }
        for (int c = 0; c < L.NReal * n; c++)
        {
            if (L.RealIsF64) w.Write(reals[c]); else w.Write((float)reals[c]);   // the float states were widened exactly: narrowing gives them back
        }
        foreach (int v in ints) w.Write(v);
    }
}
