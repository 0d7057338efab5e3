// Device-resident replacement of VectorEnv<TState,TAction> (VectorEnv.cs:10-35).
// Same public members -- NumEnvs, Reset(), Step(List<TAction>) with the same
// element tuple -- plus the array and zero-copy forms that avoid building a list
// of boxed tuples.  TObs is the element type of the observation the device
// hands back: float (classic control), int (FrozenLake, Taxi, CliffWalking,
// Blackjack), byte (Pong, Breakout, SpaceInvaders frames).
// Not compiled in the build container (no .NET SDK).
using System;
using System.Buffers;
using System.Collections;
using System.Collections.Generic;
using System.Runtime.InteropServices;

namespace Gymnasium.Gpu;

/// <summary>(observation array as the ABI returns it, NumEnvs, env index) -> the reference's TState.</summary>
public delegate TState StateReader<TObs, TState>(TObs[] obs, int numEnvs, int index);

public sealed unsafe class GpuVectorEnv<TState, TAction, TObs> : IDisposable
    where TAction : unmanaged where TObs : unmanaged
{
    private readonly ContextHandle _ctx;
    private readonly VecHandle _vec = new();
    private readonly GxInfo _info;
    private readonly StateReader<TObs, TState> _stateOf;
    // Pinned host buffers: allocated on the pinned object heap, so their addresses never move;
    // the MemoryHandles keep that explicit until Dispose.  Copies run on the library's side stream
    // (or through its staging buffer for small batches) and are complete when gx_vec_wait returns.
    private readonly TAction[] _actions;
    private readonly TObs[] _obs;
    private readonly byte[] _done, _truncated, _rewardBytes, _image;
    private MemoryHandle _hActions, _hObs, _hDone, _hTruncated, _hReward, _hImage;
    private readonly int _timeLimit;
    private bool _imageFresh;

    public int NumEnvs => (int)_info.NumEnvs; // VectorEnv.cs:13
    public GxInfo Info => _info;
    public GxEnvId EnvId => (GxEnvId)_info.EnvId;

    public GpuVectorEnv(GxEnvId env, int numEnvs, StateReader<TObs, TState> stateOf, ulong seed = 0,
                        GxFlags flags = GxFlags.None, int timeLimit = 0, GxMode mode = GxMode.Exact,
                        int device = 0, ulong envIndexBase = 0, string? mapRows = null, GxTable? table = null)
    {
        _ctx = ContextHandle.Create(device);
        _stateOf = stateOf;
        _timeLimit = timeLimit;
        byte[]? map = mapRows is null ? null : System.Text.Encoding.UTF8.GetBytes(mapRows + "\0");
        fixed (byte* pMap = map)
        fixed (GxTransition* pTable = table?.Entries)   // copied by gx_vec_create; not kept
        {
            var d = new GxDesc
            {
                StructSize = (uint)sizeof(GxDesc), EnvId = (int)env, NumEnvs = (ulong)numEnvs, Seed = seed,
                EnvIndexBase = envIndexBase, Mode = (int)mode, Flags = (uint)flags, TimeLimit = timeLimit, MapRows = pMap,
            };
            if (table is not null)
            {
                d.Table = pTable; d.TableLen = (ulong)table.Entries.Length;
                d.NStates = table.NStates; d.NActions = table.NActions; d.NOutcomes = table.NOutcomes;
                d.StartState = table.StartState; d.StepLimit = table.StepLimit;
            }
            Native.Check(Native.gx_vec_create(_ctx.DangerousGetHandle(), &d, out IntPtr v));
            _vec.Set(v, _ctx);
        }
        Native.Check(Native.gx_vec_info(_vec.DangerousGetHandle(), out _info));
        int obsElem = _info.ObsDType switch { (int)GxDType.F64 => 8, (int)GxDType.U8 => 1, _ => 4 };
        if (obsElem != sizeof(TObs))
            throw new NotSupportedException($"observation elements of this batch are {obsElem} bytes, TObs is {sizeof(TObs)}");
        _actions = GC.AllocateUninitializedArray<TAction>(numEnvs, pinned: true);
        _obs = GC.AllocateUninitializedArray<TObs>(checked((int)(_info.ObsBytes / (ulong)sizeof(TObs))), pinned: true);
        _done = GC.AllocateUninitializedArray<byte>(numEnvs, pinned: true);
        _truncated = GC.AllocateUninitializedArray<byte>(numEnvs, pinned: true);
        _rewardBytes = GC.AllocateUninitializedArray<byte>(numEnvs * (_info.RewardDType == (int)GxDType.F64 ? 8 : 4), pinned: true);
        _image = GC.AllocateUninitializedArray<byte>(HasFrameInfo ? checked((int)_info.StateBytes) : 0, pinned: true);
        _hActions = _actions.AsMemory().Pin(); _hObs = _obs.AsMemory().Pin(); _hDone = _done.AsMemory().Pin();
        _hTruncated = _truncated.AsMemory().Pin(); _hReward = _rewardBytes.AsMemory().Pin(); _hImage = _image.AsMemory().Pin();
    }

    // VectorEnv.Reset(), VectorEnv.cs:20-26
    public List<TState> Reset()
    {
        Native.Check(Native.gx_vec_reset(_vec.DangerousGetHandle()));
        Native.Check(Native.gx_vec_get_obs(_vec.DangerousGetHandle(), _hObs.Pointer, (nuint)_info.ObsBytes));
        _imageFresh = false;
        var list = new List<TState>(NumEnvs);
        for (int i = 0; i < NumEnvs; i++) list.Add(_stateOf(_obs, NumEnvs, i));
        return list;
    }

    /// <summary>VectorEnv.Step(List&lt;TAction&gt;), VectorEnv.cs:28-34.  The result has the reference's element tuple
    /// and list shape (Count, indexer, foreach); an element is built from the pinned arrays when it is first
    /// read, so a caller that looks at three envs of a million pays for three.  The view is valid until the next
    /// Step/Reset; ToList() gives the reference's exact return type.</summary>
    public StepResults Step(List<TAction> actions)
    {
        if (actions.Count != NumEnvs) throw new ArgumentException($"Step needs {NumEnvs} actions, got {actions.Count}");
        CollectionsMarshal.AsSpan(actions).CopyTo(_actions);
        StepArrays(wantObs: true);
        return new StepResults(this);
    }

    // Array form: fill Actions, call, read Rewards / Dones / Observations.  No allocation per step.
    public Span<TAction> Actions => _actions;
    public ReadOnlySpan<byte> Dones => _done;
    public ReadOnlySpan<byte> Truncated => _truncated;
    public ReadOnlySpan<TObs> Observations => _obs;
    public double RewardAt(int i) => _info.RewardDType == (int)GxDType.F64
        ? MemoryMarshal.Cast<byte, double>(_rewardBytes)[i]
        : MemoryMarshal.Cast<byte, float>(_rewardBytes)[i];

    public void StepArrays(bool wantObs)
    {
        IntPtr v = _vec.DangerousGetHandle();
        Native.Check(Native.gx_vec_step_async(v, _hActions.Pointer, wantObs ? _hObs.Pointer : null, _hReward.Pointer,
                                              (byte*)_hDone.Pointer, _timeLimit > 0 ? (byte*)_hTruncated.Pointer : null));
        Native.Check(Native.gx_vec_wait(v));
        _imageFresh = false;
    }

    // Zero-copy path for agents that already live on the GPU.
    public GxView DeviceView() { Native.Check(Native.gx_vec_device_view(_vec.DangerousGetHandle(), out GxView w)); return w; }
    public void StepDevice(IntPtr dActions, IntPtr stream = default) => Native.Check(Native.gx_vec_step_device(_vec.DangerousGetHandle(), dActions, stream));
    public void Rollout(int kSteps, GxPolicy policy = GxPolicy.Random, IntPtr dActions = default, IntPtr stream = default)
        => Native.Check(Native.gx_vec_rollout(_vec.DangerousGetHandle(), kSteps, (int)policy, dActions, stream));
    /// <summary>K fused steps that also write every step's reward / done / truncated into caller-owned device rows
    /// [kSteps][stride] (GxRolloutOut.All), or none at all (GxRolloutOut.None).</summary>
    public void Rollout(int kSteps, GxRolloutOut output, IntPtr dRewardRows, IntPtr dDoneRows, IntPtr dTruncatedRows = default,
                        GxPolicy policy = GxPolicy.Random, IntPtr dActions = default, IntPtr stream = default)
        => Native.Check(Native.gx_vec_rollout_out(_vec.DangerousGetHandle(), kSteps, (int)policy, dActions, (int)output,
                                                  dRewardRows, dDoneRows, dTruncatedRows, stream));
    public GxStats Stats() { Native.Check(Native.gx_vec_stats(_vec.DangerousGetHandle(), out GxStats s)); return s; }

    /// <summary>The SoA state image (gx_vec_get_state): n_real rows of NumEnvs reals, then n_int rows of int32.</summary>
    public byte[] GetState()
    {
        var image = new byte[checked((int)_info.StateBytes)];
        fixed (byte* p = image) Native.Check(Native.gx_vec_get_state(_vec.DangerousGetHandle(), p, (nuint)image.Length));
        return image;
    }
    public void SetState(ReadOnlySpan<byte> image)
    {
        fixed (byte* p = image) Native.Check(Native.gx_vec_set_state(_vec.DangerousGetHandle(), p, (nuint)image.Length));
        _imageFresh = false;
    }

    /// <summary>The <c>_random.NextDouble()</c> values of each env's next draw inside Step, for the next Reset / Step only:
    /// Pong <c>u[2][NumEnvs]</c> (ball reset, Envs.Pong.cs:163-164), Breakout <c>u[1][NumEnvs]</c> (release of the ball,
    /// Envs.Breakout.cs:100).  An empty span withdraws values not used yet.</summary>
    public void SetRngDraws(ReadOnlySpan<double> u)
    {
        fixed (double* p = u) Native.Check(Native.gx_vec_set_rng_draws(_vec.DangerousGetHandle(), u.IsEmpty ? null : p, (nuint)u.Length));
    }

    // ---- info dictionaries ---------------------------------------------------
    private bool HasFrameInfo => EnvId is GxEnvId.Pong or GxEnvId.Breakout or GxEnvId.SpaceInvaders;

    // The frame envs report game facts in `info` (Envs.Pong.cs:148-154, Envs.Breakout.cs:203-210,
    // Envs.SpaceInvaders.cs:181-187); they are read from the state image, fetched once per step and only
    // when some element's info is looked at.
    private IDictionary<string, object> InfoAt(int i)
    {
        var info = new Dictionary<string, object>();
        if (_timeLimit > 0 && _truncated[i] != 0) info["TimeLimit.truncated"] = true; // Wrappers.TimeLimit.cs:30
        if (!HasFrameInfo) return info;
        if (!_imageFresh)
        {
            Native.Check(Native.gx_vec_get_state(_vec.DangerousGetHandle(), _hImage.Pointer, (nuint)_image.Length));
            _imageFresh = true;
        }
        int n = NumEnvs;
        ReadOnlySpan<double> re = MemoryMarshal.Cast<byte, double>(_image.AsSpan(0, _info.NReal * n * 8));
        ReadOnlySpan<int> it = MemoryMarshal.Cast<byte, int>(_image.AsSpan(_info.NReal * n * 8));
        switch (EnvId)
        {
            case GxEnvId.Pong:
                info["left_score"] = it[i]; info["right_score"] = it[n + i];
                info["ball_x"] = re[i]; info["ball_y"] = re[n + i];
                break;
            case GxEnvId.Breakout:
                int bricks = 0;
                for (int w = 3; w < 7; w++) bricks += System.Numerics.BitOperations.PopCount((uint)it[w * n + i]);
                info["score"] = it[i]; info["lives"] = it[n + i];
                info["ball_x"] = re[i]; info["ball_y"] = re[n + i]; info["bricks_remaining"] = bricks;
                break;
            case GxEnvId.SpaceInvaders:
                int aliens = System.Numerics.BitOperations.PopCount((uint)it[6 * n + i])
                           + System.Numerics.BitOperations.PopCount((uint)it[7 * n + i] & 0x7FFFFFu);
                info["score"] = it[i]; info["lives"] = it[n + i];
                info["aliens_remaining"] = aliens; info["player_x"] = re[i];
                break;
        }
        return info;
    }

    /// <summary>Lazy view of one step's results in the shape of the reference's list.</summary>
    public sealed class StepResults : IReadOnlyList<(TState state, double reward, bool done, IDictionary<string, object> info)>
    {
        private readonly GpuVectorEnv<TState, TAction, TObs> _owner;
        internal StepResults(GpuVectorEnv<TState, TAction, TObs> owner) { _owner = owner; }
        public int Count => _owner.NumEnvs;
        public (TState state, double reward, bool done, IDictionary<string, object> info) this[int i]
        {
            get
            {
                if ((uint)i >= (uint)Count) throw new ArgumentOutOfRangeException(nameof(i));
                return (_owner._stateOf(_owner._obs, Count, i), _owner.RewardAt(i), _owner._done[i] != 0, _owner.InfoAt(i));
            }
        }
        public IEnumerator<(TState state, double reward, bool done, IDictionary<string, object> info)> GetEnumerator()
        {
            for (int i = 0; i < Count; i++) yield return this[i];
        }
        IEnumerator IEnumerable.GetEnumerator() => GetEnumerator();
        /// <summary>The reference's exact return type, VectorEnv.cs:28.</summary>
        public List<(TState state, double reward, bool done, IDictionary<string, object> info)> ToList()
        {
            var list = new List<(TState, double, bool, IDictionary<string, object>)>(Count);
            for (int i = 0; i < Count; i++) list.Add(this[i]);
            return list;
        }
    }

    public void Dispose()
    {
        _hActions.Dispose(); _hObs.Dispose(); _hDone.Dispose(); _hTruncated.Dispose(); _hReward.Dispose(); _hImage.Dispose();
        _vec.Dispose();   // releases its reference on the context after the batch is destroyed
        _ctx.Dispose();
    }
}

/// <summary>Factories with the reference's state types (Envs.CartPole.cs:10, Envs.Pendulum.cs:10, Envs.FrozenLake.cs:10,
/// Envs.Pong.cs:12, ...), one per id of GymnasiumRegistration.cs:10-28 that has a device kernel.</summary>
public static class GpuEnvs
{
    public const int FrameBytes = 210 * 160 * 3;

    public static GpuVectorEnv<(float, float, float, float), int, float> CartPole(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.CartPole, n, (o, N, i) => (o[i], o[N + i], o[2 * N + i], o[3 * N + i]), seed, flags, timeLimit);
    public static GpuVectorEnv<(float, float), int, float> MountainCar(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.MountainCar, n, (o, N, i) => (o[i], o[N + i]), seed, flags, timeLimit);
    public static GpuVectorEnv<(float, float), float, float> MountainCarContinuous(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.MountainCarContinuous, n, (o, N, i) => (o[i], o[N + i]), seed, flags, timeLimit);
    public static GpuVectorEnv<float[], float, float> Pendulum(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.Pendulum, n, (o, N, i) => new[] { o[i], o[N + i], o[2 * N + i] }, seed, flags, timeLimit);
    public static GpuVectorEnv<float[], int, float> Acrobot(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.Acrobot, n, (o, N, i) => new[] { o[i], o[N + i], o[2 * N + i], o[3 * N + i], o[4 * N + i], o[5 * N + i] }, seed, flags, timeLimit);

    // int states: the observation is int32[NumEnvs]
    public static GpuVectorEnv<int, int, int> FrozenLake(int n, string[]? map = null, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.FrozenLake, n, (o, N, i) => o[i], seed, flags, timeLimit, mapRows: map is null ? null : string.Join('/', map));
    public static GpuVectorEnv<int, int, int> Taxi(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.Taxi, n, (o, N, i) => o[i], seed, flags, timeLimit);
    public static GpuVectorEnv<int, int, int> CliffWalking(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.CliffWalking, n, (o, N, i) => o[i], seed, flags, timeLimit);
    // a table-driven env whose transitions the caller supplies (what FrozenLake's map argument generalises to)
    public static GpuVectorEnv<int, int, int> Table(int n, GxTable table, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.Table, n, (o, N, i) => o[i], seed, flags, timeLimit, table: table);
    // Envs.Blackjack.cs:10: (playerSum, dealerCard, usableAce); the device holds the index of Discrete(32*11*2)
    public static GpuVectorEnv<(int, int, bool), int, int> Blackjack(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.Blackjack, n, (o, N, i) => (o[i] / 22, (o[i] / 2) % 11, (o[i] & 1) != 0), seed, flags, timeLimit);

    // byte[] states: the observation is uint8[NumEnvs][100800]; every state is a fresh array, as the agents keep
    // references to earlier frames (the reference hands out its one internal `_screen`, Envs.Pong.cs:155)
    private static byte[] Frame(byte[] o, int N, int i) => o.AsSpan(i * FrameBytes, FrameBytes).ToArray();
    public static GpuVectorEnv<byte[], int, byte> Pong(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.Pong, n, Frame, seed, flags, timeLimit);
    public static GpuVectorEnv<byte[], int, byte> Breakout(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.Breakout, n, Frame, seed, flags, timeLimit);
    public static GpuVectorEnv<byte[], int, byte> SpaceInvaders(int n, ulong seed = 0, GxFlags flags = GxFlags.None, int timeLimit = 0)
        => new(GxEnvId.SpaceInvaders, n, Frame, seed, flags, timeLimit);
}
