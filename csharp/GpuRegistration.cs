// Device-backed Env<TState,TAction> shells and their registration through the unchanged
// EnvRegistry.Register (EnvRegistry.cs:13-17), next to GymnasiumRegistration.RegisterAll
// (GymnasiumRegistration.cs:8-30).  A single-env object is a batch of one, so
// MainWindowViewModel's episode loop (MainWindowViewModel.cs:265-312) and the baseline agents,
// which only use Reset / Step / ActionSpace / ObservationSpace through `dynamic` and test the
// spaces with `is Discrete` / `is Box` (PPOAgent.cs:249-339), run on it unmodified: every shell
// declares the same two properties with the same types and bounds as the class it stands in for.
// Not compiled in the build container.
using System;
using System.Collections.Generic;
using System.Linq;
using Gymnasium.Spaces;

namespace Gymnasium.Gpu;

/// <summary>One env of the reference = a device batch of one.</summary>
public abstract class GpuEnv<TState, TAction, TObs> : Env<TState, TAction>
    where TAction : unmanaged where TObs : unmanaged
{
    private readonly Func<ulong, ulong, GpuVectorEnv<TState, TAction, TObs>> _make;   // (seed, env index base) -> batch of one
    private GpuVectorEnv<TState, TAction, TObs>? _vec;   // created by the first Reset/Step: constructing an env needs no GPU
    private readonly List<TAction> _one = new() { default };
    // An env built without a seed takes the next stream of seed 0 (its index keys the counter RNG), so that envs
    // constructed side by side differ, as the reference's do with their unseeded `new Random()` (Envs.CartPole.cs:19).
    private ulong _seed = 0;
    private ulong _stream = Streams.Next();

    protected GpuEnv(Func<ulong, ulong, GpuVectorEnv<TState, TAction, TObs>> make) { _make = make; }
    protected GpuVectorEnv<TState, TAction, TObs> Vec => _vec ??= _make(_seed, _stream);

    /// <summary>Extension: the reference's Seed is an empty virtual that no env overrides (Env.cs:33-36).  Here a
    /// seed re-keys the counter RNG -- stream 0 of that seed, deterministic -- and the batch is rebuilt by the next
    /// Reset/Step; Seed(null) changes nothing, like the reference's.</summary>
    public override void Seed(int? seed = null)
    {
        if (seed is null) return;
        _seed = unchecked((ulong)(long)seed.Value);
        _stream = 0;
        _vec?.Dispose();
        _vec = null;
    }

    public override TState Reset() => Vec.Reset()[0];
    public override (TState state, double reward, bool done, IDictionary<string, object> info) Step(TAction action)
    {
        _one[0] = action;
        return Vec.Step(_one)[0];
    }
    public override void Render(string mode = "human") { }
    public override void Close() { _vec?.Dispose(); _vec = null; }
}

// Consecutive streams of seed 0 for envs built without a seed (see GpuEnv).
internal static class Streams
{
    private static long _next = -1;
    internal static ulong Next() => (ulong)System.Threading.Interlocked.Increment(ref _next);
}

public sealed class GpuCartPole : GpuEnv<(float, float, float, float), int, float>
{
    public Discrete ActionSpace { get; } = new(2);                                      // Envs.CartPole.cs:12
    public Box ObservationSpace { get; } = new(new[] { -2.4f, -3.0f, -0.2095f, -3.0f }, new[] { 2.4f, 3.0f, 0.2095f, 3.0f }); // :13-16
    public GpuCartPole() : base((seed, stream) => new(GxEnvId.CartPole, 1, (o, N, i) => (o[i], o[N + i], o[2 * N + i], o[3 * N + i]), seed: seed, envIndexBase: stream)) { }
}

public sealed class GpuMountainCar : GpuEnv<(float, float), int, float>
{
    public Discrete ActionSpace { get; } = new(3);                                      // Envs.MountainCar.cs:12
    public Box ObservationSpace { get; } = new(new[] { -1.2f, -0.07f }, new[] { 0.6f, 0.07f });   // :13-16
    public GpuMountainCar() : base((seed, stream) => new(GxEnvId.MountainCar, 1, (o, N, i) => (o[i], o[N + i]), seed: seed, envIndexBase: stream)) { }
}

public sealed class GpuMountainCarContinuous : GpuEnv<(float, float), float, float>
{
    public Box ActionSpace { get; } = new(new[] { -1.0f }, new[] { 1.0f });             // Envs.MountainCarContinuous.cs:12
    public Box ObservationSpace { get; } = new(new[] { -1.2f, -0.07f }, new[] { 0.6f, 0.07f });   // :13
    public GpuMountainCarContinuous() : base((seed, stream) => new(GxEnvId.MountainCarContinuous, 1, (o, N, i) => (o[i], o[N + i]), seed: seed, envIndexBase: stream)) { }
}

public sealed class GpuPendulum : GpuEnv<float[], float, float>
{
    public Box ActionSpace { get; } = new(new[] { -2.0f }, new[] { 2.0f });             // Envs.Pendulum.cs:12
    public Box ObservationSpace { get; } = new(new[] { -1.0f, -1.0f, -8.0f }, new[] { 1.0f, 1.0f, 8.0f }); // :13
    public GpuPendulum() : base((seed, stream) => new(GxEnvId.Pendulum, 1, (o, N, i) => new[] { o[i], o[N + i], o[2 * N + i] }, seed: seed, envIndexBase: stream)) { }
}

public sealed class GpuAcrobot : GpuEnv<float[], int, float>
{
    public Discrete ActionSpace { get; } = new(3);                                      // Envs.Acrobot.cs:12
    public Box ObservationSpace { get; } = new(new[] { -1f, -1f, -1f, -1f, -6f, -6f }, new[] { 1f, 1f, 1f, 1f, 6f, 6f }); // :13-16
    public GpuAcrobot() : base((seed, stream) => new(GxEnvId.Acrobot, 1,
        (o, N, i) => new[] { o[i], o[N + i], o[2 * N + i], o[3 * N + i], o[4 * N + i], o[5 * N + i] }, seed: seed, envIndexBase: stream)) { }
    // torque[action] throws IndexOutOfRangeException (Envs.Acrobot.cs:45,51); the library reports GX_ERR_INVALID
    public override (float[] state, double reward, bool done, IDictionary<string, object> info) Step(int action)
    {
        try { return base.Step(action); }
        catch (ArgumentException e) { throw new IndexOutOfRangeException(e.Message); }
    }
}

public sealed class GpuFrozenLake : GpuEnv<int, int, int>
{
    public Discrete ActionSpace { get; } = new(4);                                      // Envs.FrozenLake.cs:12
    public Discrete ObservationSpace { get; }                                           // :33
    public GpuFrozenLake(string[]? map = null)                                          // the constructor argument of :23
        : base((seed, stream) => new(GxEnvId.FrozenLake, 1, (o, N, i) => o[i], seed: seed, envIndexBase: stream, mapRows: map is null ? null : string.Join('/', map)))
    {
        ObservationSpace = new(map is null ? 16 : map.Length * map[0].Length);
    }
}

/// <summary>Env&lt;int,int&gt; stepped from a transition table the caller supplies (gx_desc.table): what FrozenLake's
/// `map` argument (Envs.FrozenLake.cs:23) generalises to.  An action outside Discrete(NActions) throws
/// IndexOutOfRangeException, as an out-of-range index does in the reference.</summary>
public sealed class GpuTableEnv : GpuEnv<int, int, int>
{
    public Discrete ActionSpace { get; }
    public Discrete ObservationSpace { get; }
    public GpuTableEnv(GxTable table)
        : base((seed, stream) => new(GxEnvId.Table, 1, (o, N, i) => o[i], seed: seed, envIndexBase: stream, table: table))
    {
        ActionSpace = new(table.NActions);
        ObservationSpace = new(table.NStates);
    }
    public override (int state, double reward, bool done, IDictionary<string, object> info) Step(int action)
    {
        try { return base.Step(action); }
        catch (ArgumentException e) { throw new IndexOutOfRangeException(e.Message); }
    }
}

public sealed class GpuTaxi : GpuEnv<int, int, int>
{
    public Discrete ActionSpace { get; } = new(6);                                      // Envs.Taxi.cs:12
    public Discrete ObservationSpace { get; } = new(500);                               // :13
    public GpuTaxi() : base((seed, stream) => new(GxEnvId.Taxi, 1, (o, N, i) => o[i], seed: seed, envIndexBase: stream)) { }
}

public sealed class GpuCliffWalking : GpuEnv<int, int, int>
{
    public Discrete ActionSpace { get; } = new(4);                                      // Envs.CliffWalking.cs:12
    public Discrete ObservationSpace { get; } = new(48);                                // :13
    public GpuCliffWalking() : base((seed, stream) => new(GxEnvId.CliffWalking, 1, (o, N, i) => o[i], seed: seed, envIndexBase: stream)) { }
}

public sealed class GpuBlackjack : GpuEnv<(int, int, bool), int, int>
{
    public Discrete ActionSpace { get; } = new(2);                                      // Envs.Blackjack.cs:12
    public Discrete ObservationSpace { get; } = new(32 * 11 * 2);                       // :13
    /// <param name="upstreamRules">false: the reference's ten-step stub (Envs.Blackjack.cs:34-39); true: upstream Blackjack-v1</param>
    public GpuBlackjack(bool upstreamRules = false)
        : base((seed, stream) => new(GxEnvId.Blackjack, 1, (o, N, i) => (o[i] / 22, (o[i] / 2) % 11, (o[i] & 1) != 0),
                         flags: upstreamRules ? GxFlags.BlackjackFull : GxFlags.None, seed: seed, envIndexBase: stream)) { }
}

/// <summary>The three frame games: byte[] states of 210*160*3, info dictionaries as the reference fills them.</summary>
public abstract class GpuFrameEnv : GpuEnv<byte[], int, byte>
{
    public Discrete ActionSpace { get; }
    public Box ObservationSpace { get; } = new(new float[GpuEnvs.FrameBytes],                                    // Envs.Pong.cs:39-41
                                               Enumerable.Repeat(255.0f, GpuEnvs.FrameBytes).ToArray());
    public override object? Metadata => new { render_modes = new[] { "human", "rgb_array" } };                  // Envs.Pong.cs:44
    protected GpuFrameEnv(GxEnvId id, int actions)
        : base((seed, stream) => new(id, 1, (o, N, i) => o.AsSpan(i * GpuEnvs.FrameBytes, GpuEnvs.FrameBytes).ToArray(), seed: seed, envIndexBase: stream))
    {
        ActionSpace = new(actions);
    }
}

public sealed class GpuPong : GpuFrameEnv
{
    public GpuPong() : base(GxEnvId.Pong, 4) { }                                                                // Envs.Pong.cs:36
    public override object? Spec => new { id = "Pong-v4", max_episode_steps = 1000 };                           // :43
    public override (double, double)? RewardRange => (-1, 1);                                                   // :45
}

public sealed class GpuBreakout : GpuFrameEnv
{
    public GpuBreakout() : base(GxEnvId.Breakout, 4) { }                                                        // Envs.Breakout.cs:44
    public override object? Spec => new { id = "Breakout-v4", max_episode_steps = 10000 };                      // :50
    public override (double, double)? RewardRange => (0, 7);                                                    // :52
}

public sealed class GpuSpaceInvaders : GpuFrameEnv
{
    public GpuSpaceInvaders() : base(GxEnvId.SpaceInvaders, 6) { }                                              // Envs.SpaceInvaders.cs:47
    public override object? Spec => new { id = "SpaceInvaders-v4", max_episode_steps = 10000 };                 // :55
    public override (double, double)? RewardRange => (0, 200);                                                  // :57
}

public static class GpuRegistration
{
    /// <summary>The ids of GymnasiumRegistration.cs:10-28 that have a device kernel, registered as "&lt;id&gt;-gpu";
    /// with <paramref name="replace"/> also under the reference's own ids, so that callers of
    /// EnvRegistry.Make("CartPole-v1") get the device-backed env without a change.  The Box2D-backed ids
    /// (LunarLander, BipedalWalker, CarRacing) and the Atari/MuJoCo stubs have no device form and keep their
    /// reference registration.</summary>
    public static void RegisterAll(bool replace = false)
    {
        var factories = new (string id, Func<object> make)[]
        {
            ("CartPole-v1", () => new GpuCartPole()),
            ("MountainCar-v0", () => new GpuMountainCar()),
            ("MountainCarContinuous-v0", () => new GpuMountainCarContinuous()),
            ("Acrobot-v1", () => new GpuAcrobot()),
            ("Pendulum-v1", () => new GpuPendulum()),
            ("FrozenLake-v1", () => new GpuFrozenLake()),
            ("Taxi-v3", () => new GpuTaxi()),
            ("Blackjack-v1", () => new GpuBlackjack()),
            ("CliffWalking-v0", () => new GpuCliffWalking()),
            ("Pong-v4", () => new GpuPong()),
            ("Breakout-v4", () => new GpuBreakout()),
            ("SpaceInvaders-v4", () => new GpuSpaceInvaders()),
        };
        foreach (var (id, make) in factories)
        {
            EnvRegistry.Register(id + "-gpu", make);
            if (replace) EnvRegistry.Register(id, make);
        }
    }
}
