// Device-resident replacement of VectorEnv<TState,TAction> (VectorEnv.cs:10-35).
// Same public members -- NumEnvs, Reset(), Step(List<TAction>) with the same
// element types -- plus the array and zero-copy forms that avoid building a
// List of boxed tuples.  Not compiled in the build container (no .NET SDK).
using System;
using System.Buffers;
using System.Collections.Generic;
using System.Runtime.InteropServices;

namespace Gymnasium.Gpu;

public sealed unsafe class GpuVectorEnv<TState, TAction> : IDisposable where TAction : unmanaged
{
    private readonly ContextHandle _ctx;
    private readonly VecHandle _vec = new();
    private readonly GxInfo _info;
    private readonly Func<float[], int, int, TState> _stateOf; // (obs SoA [dim][N], N, i) -> TState
    // pinned host buffers (Memory<T>.Pin() keeps them fixed until disposed); copies run on the
    // library's side stream and are complete when gx_vec_wait returns
    private readonly Memory<TAction> _actions;
    private readonly Memory<float> _obs;
    private readonly Memory<byte> _done, _truncated, _rewardBytes;
    private MemoryHandle _hActions, _hObs, _hDone, _hTruncated, _hReward;
    private readonly int _timeLimit;

    public int NumEnvs => (int)_info.NumEnvs; // VectorEnv.cs:13
    public GxInfo Info => _info;

    public GpuVectorEnv(GxEnvId env, int numEnvs, Func<float[], int, int, TState> stateOf, ulong seed = 0,
                        GxFlags flags = GxFlags.None, int timeLimit = 0, GxMode mode = GxMode.Exact,
                        int device = 0, ulong envIndexBase = 0, string? mapRows = null)
    {
        // the reference's TState types are float-valued (Env.cs:7); binary64 batches are driven through the
        // raw entry points (gx_vec_step_device / gx_vec_device_view) by callers that hold doubles
        if (mode == GxMode.Fp64 && env <= GxEnvId.Acrobot)
            throw new NotSupportedException("GxMode.Fp64 observations are double; this facade materialises the reference's float states");
        _ctx = ContextHandle.Create(device);
        _stateOf = stateOf;
        _timeLimit = timeLimit;
        byte[]? map = mapRows is null ? null : System.Text.Encoding.UTF8.GetBytes(mapRows + "\0");
        fixed (byte* pMap = map)
        {
            var d = new GxDesc
            {
                StructSize = (uint)sizeof(GxDesc), EnvId = (int)env, NumEnvs = (ulong)numEnvs, Seed = seed,
                EnvIndexBase = envIndexBase, Mode = (int)mode, Flags = (uint)flags, TimeLimit = timeLimit, MapRows = pMap,
            };
            bool added = false;
            _ctx.DangerousAddRef(ref added);
            try
            {
                Native.Check(Native.gx_vec_create(_ctx.DangerousGetHandle(), &d, out IntPtr v));
                _vec.Set(v);
            }
            finally { if (added) _ctx.DangerousRelease(); }
        }
        Native.Check(Native.gx_vec_info(_vec.DangerousGetHandle(), out _info));
        _actions = GC.AllocateUninitializedArray<TAction>(numEnvs, pinned: true);
        _obs = GC.AllocateUninitializedArray<float>(checked((int)(_info.ObsBytes / 4)), pinned: true);
        _done = GC.AllocateUninitializedArray<byte>(numEnvs, pinned: true);
        _truncated = GC.AllocateUninitializedArray<byte>(numEnvs, pinned: true);
        _rewardBytes = GC.AllocateUninitializedArray<byte>(numEnvs * (_info.RewardDType == (int)GxDType.F64 ? 8 : 4), pinned: true);
        _hActions = _actions.Pin(); _hObs = _obs.Pin(); _hDone = _done.Pin(); _hTruncated = _truncated.Pin(); _hReward = _rewardBytes.Pin();
    }

    // VectorEnv.Reset(), VectorEnv.cs:20-26
    public List<TState> Reset()
    {
        Native.Check(Native.gx_vec_reset(_vec.DangerousGetHandle()));
        Native.Check(Native.gx_vec_get_obs(_vec.DangerousGetHandle(), _hObs.Pointer, (nuint)_info.ObsBytes));
        return Materialise();
    }

    // VectorEnv.Step(List<TAction>), VectorEnv.cs:28-34: same element tuple, built lazily from the arrays
    public List<(TState state, double reward, bool done, IDictionary<string, object> info)> Step(List<TAction> actions)
    {
        CollectionsMarshal.AsSpan(actions).CopyTo(_actions.Span);
        StepArrays(wantObs: true);
        float[] obs = ObsArray();
        var results = new List<(TState, double, bool, IDictionary<string, object>)>(NumEnvs);
        for (int i = 0; i < NumEnvs; i++)
        {
            var info = new Dictionary<string, object>();
            if (_timeLimit > 0 && _truncated.Span[i] != 0) info["TimeLimit.truncated"] = true; // Wrappers.TimeLimit.cs:30
            results.Add((_stateOf(obs, NumEnvs, i), RewardAt(i), _done.Span[i] != 0, info));
        }
        return results;
    }

    // Array form: fill Actions, call, read Rewards / Dones / Observations.  No allocation per step.
    public Span<TAction> Actions => _actions.Span;
    public ReadOnlySpan<byte> Dones => _done.Span;
    public ReadOnlySpan<float> Observations => _obs.Span;
    public double RewardAt(int i) => _info.RewardDType == (int)GxDType.F64
        ? MemoryMarshal.Cast<byte, double>(_rewardBytes.Span)[i]
        : MemoryMarshal.Cast<byte, float>(_rewardBytes.Span)[i];

    public void StepArrays(bool wantObs)
    {
        IntPtr v = _vec.DangerousGetHandle();
        Native.Check(Native.gx_vec_step_async(v, _hActions.Pointer, wantObs ? _hObs.Pointer : null, _hReward.Pointer,
                                              (byte*)_hDone.Pointer, _timeLimit > 0 ? (byte*)_hTruncated.Pointer : null));
        Native.Check(Native.gx_vec_wait(v));
    }

    // Zero-copy path for agents that already live on the GPU.
    public GxView DeviceView() { Native.Check(Native.gx_vec_device_view(_vec.DangerousGetHandle(), out GxView w)); return w; }
    public void StepDevice(IntPtr dActions, IntPtr stream = default) => Native.Check(Native.gx_vec_step_device(_vec.DangerousGetHandle(), dActions, stream));
    public void Rollout(int kSteps, GxPolicy policy = GxPolicy.Random, IntPtr dActions = default, IntPtr stream = default)
        => Native.Check(Native.gx_vec_rollout(_vec.DangerousGetHandle(), kSteps, (int)policy, dActions, stream));
    public GxStats Stats() { Native.Check(Native.gx_vec_stats(_vec.DangerousGetHandle(), out GxStats s)); return s; }

    private float[] ObsArray() => MemoryMarshal.TryGetArray<float>(_obs, out var seg) ? seg.Array! : _obs.ToArray();
    private List<TState> Materialise()
    {
        float[] obs = ObsArray();
        var list = new List<TState>(NumEnvs);
        for (int i = 0; i < NumEnvs; i++) list.Add(_stateOf(obs, NumEnvs, i));
        return list;
    }

    public void Dispose()
    {
        _hActions.Dispose(); _hObs.Dispose(); _hDone.Dispose(); _hTruncated.Dispose(); _hReward.Dispose();
        _vec.Dispose();
        _ctx.Dispose();
    }
}

// Factories with the reference's state types (Envs.CartPole.cs:10, Envs.Pendulum.cs:10, ...).
public static class GpuEnvs
{
    public static GpuVectorEnv<(float, float, float, float), int> CartPole(int n, ulong seed = 0, GxFlags flags = GxFlags.None)
        => new(GxEnvId.CartPole, n, (o, N, i) => (o[i], o[N + i], o[2 * N + i], o[3 * N + i]), seed, flags);
    public static GpuVectorEnv<(float, float), int> MountainCar(int n, ulong seed = 0, GxFlags flags = GxFlags.None)
        => new(GxEnvId.MountainCar, n, (o, N, i) => (o[i], o[N + i]), seed, flags);
    public static GpuVectorEnv<(float, float), float> MountainCarContinuous(int n, ulong seed = 0, GxFlags flags = GxFlags.None)
        => new(GxEnvId.MountainCarContinuous, n, (o, N, i) => (o[i], o[N + i]), seed, flags);
    public static GpuVectorEnv<float[], float> Pendulum(int n, ulong seed = 0, GxFlags flags = GxFlags.None)
        => new(GxEnvId.Pendulum, n, (o, N, i) => new[] { o[i], o[N + i], o[2 * N + i] }, seed, flags);
    public static GpuVectorEnv<float[], int> Acrobot(int n, ulong seed = 0, GxFlags flags = GxFlags.None)
        => new(GxEnvId.Acrobot, n, (o, N, i) => new[] { o[i], o[N + i], o[2 * N + i], o[3 * N + i], o[4 * N + i], o[5 * N + i] }, seed, flags);
}
