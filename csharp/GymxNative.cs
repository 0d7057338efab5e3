// P/Invoke surface of libgymx.so (include/gymx.h).  Sources only: the build
// container has no .NET SDK, so these files are written against the header and
// the C# language reference and are not compiled here.
//
// A maintainer adds this folder to Gymnasium.csproj (net8.0, AllowUnsafeBlocks)
// next to VectorEnv.cs; nothing else in the library changes.
using System;
using System.Runtime.InteropServices;

namespace Gymnasium.Gpu;

public enum GxStatus { Ok = 0, Invalid = 1, Hip = 2, Oom = 3, Unsupported = 4, Rccl = 5 }

public enum GxEnvId
{
    CartPole = 0, MountainCar = 1, MountainCarContinuous = 2, Pendulum = 3, Acrobot = 4,
    FrozenLake = 5, Taxi = 6, CliffWalking = 7, Pong = 8, Dummy = 9, Breakout = 10, SpaceInvaders = 11, Blackjack = 12,
    Table = 13
}

public enum GxMode { Exact = 0, Fast32 = 1, Fp64 = 2 }

[Flags]
public enum GxFlags : uint { None = 0, AutoReset = 1, RecordStats = 2, Slippery = 4, TaxiFull = 8, AcrobotRk4 = 16, BlackjackFull = 64 }

public enum GxPolicy { DeviceActions = 0, Random = 1 }

public enum GxRolloutOut { Last = 0, All = 1, None = 2 }

public enum GxDType { F32 = 0, F64 = 1, I32 = 2, U8 = 3 }

[StructLayout(LayoutKind.Sequential)]
public unsafe struct GxDesc
{
    public uint StructSize;
    public int EnvId;
    public ulong NumEnvs;
    public ulong Seed;
    public ulong EnvIndexBase;
    public int Mode;
    public uint Flags;
    public int TimeLimit;
    public int Reserved;
    public byte* MapRows; // UTF-8, rows joined by '/', or null
    // GxEnvId.Table only
    public GxTransition* Table; // [NStates][NActions][NOutcomes]
    public ulong TableLen;
    public int NStates, NActions, NOutcomes, StartState, StepLimit, Reserved2;
}

// gx_transition: one entry of a caller-supplied table
[StructLayout(LayoutKind.Sequential)]
public struct GxTransition
{
    public int NextState;
    public float Reward;
    public int Done;
}

/// <summary>A caller-supplied transition table for GxEnvId.Table: Entries[(s * NActions + a) * NOutcomes + o].</summary>
public sealed class GxTable
{
    public GxTransition[] Entries = Array.Empty<GxTransition>();
    public int NStates, NActions;
    public int NOutcomes = 1;   // 1, or 3 equally likely outcomes drawn per step
    public int StartState = 0;  // Reset(): this state; -1: a uniform draw over the states (Envs.Taxi.cs:20)
    public int StepLimit = 0;   // the env's own `steps >= limit` term of done (Envs.FrozenLake.cs:65); 0: none
}

[StructLayout(LayoutKind.Sequential)]
public struct GxInfo
{
    public ulong NumEnvs;
    public int EnvId, NReal, RealDType, NInt, ObsDim, ObsDType, ActionDType, ActionN;
    public float ActionLow, ActionHigh;
    public int RewardDType, ObsIsState;
    public ulong StateBytes, ObsBytes;
}

[StructLayout(LayoutKind.Sequential)]
public struct GxView
{
    public ulong NumEnvs, Stride;
    public IntPtr StateReal, StateInt, Obs, Reward, Done, Truncated;
}

[StructLayout(LayoutKind.Sequential)]
public struct GxStats
{
    public long Episodes;
    public double SumReturn;
    public long SumLength;
    public double SumReturnSq, MinReturn, MaxReturn;
}

// gx_rows: caller-owned device rows for gx_rows_step, the entry point that allocates nothing
[StructLayout(LayoutKind.Sequential)]
public struct GxRows
{
    public uint StructSize;
    public int EnvId;
    public ulong NumEnvs, Stride, Seed, EnvIndexBase, Tick;
    public int Mode, StateBounded;
    public IntPtr StateReal, Steps, EpReturn, Actions, Reward, Done, Stats, Error;
}

internal static unsafe partial class Native
{
    private const string Lib = "gymx"; // resolves libgymx.so on Linux

    [LibraryImport(Lib)] internal static partial uint gx_version();
    [LibraryImport(Lib)] internal static partial IntPtr gx_last_error();
    [LibraryImport(Lib)] internal static partial int gx_device_count(out int count);
    [LibraryImport(Lib)] internal static partial int gx_ctx_create(int device, out IntPtr ctx);
    [LibraryImport(Lib)] internal static partial int gx_ctx_destroy(IntPtr ctx);
    [LibraryImport(Lib)] internal static partial int gx_ctx_synchronize(IntPtr ctx);
    [LibraryImport(Lib)] internal static partial int gx_ctx_stream(IntPtr ctx, out IntPtr stream);
    [LibraryImport(Lib)] internal static partial int gx_vec_create(IntPtr ctx, GxDesc* desc, out IntPtr vec);
    [LibraryImport(Lib)] internal static partial int gx_vec_destroy(IntPtr vec);
    [LibraryImport(Lib)] internal static partial int gx_vec_info(IntPtr vec, out GxInfo info);
    [LibraryImport(Lib)] internal static partial int gx_vec_reset(IntPtr vec);
    [LibraryImport(Lib)] internal static partial int gx_vec_get_state(IntPtr vec, void* hostSoa, nuint bytes);
    [LibraryImport(Lib)] internal static partial int gx_vec_set_state(IntPtr vec, void* hostSoa, nuint bytes);
    [LibraryImport(Lib)] internal static partial int gx_vec_set_rng_draws(IntPtr vec, double* u, nuint count);
    [LibraryImport(Lib)] internal static partial int gx_vec_get_obs(IntPtr vec, void* hostObs, nuint bytes);
    [LibraryImport(Lib)] internal static partial int gx_vec_step_async(IntPtr vec, void* hActions, void* hObs, void* hReward, byte* hDone, byte* hTruncated);
    [LibraryImport(Lib)] internal static partial int gx_vec_wait(IntPtr vec);
    [LibraryImport(Lib)] internal static partial int gx_vec_step_device(IntPtr vec, IntPtr dActions, IntPtr stream);
    [LibraryImport(Lib)] internal static partial int gx_vec_step_random(IntPtr vec, IntPtr stream);
    [LibraryImport(Lib)] internal static partial int gx_vec_device_view(IntPtr vec, out GxView view);
    [LibraryImport(Lib)] internal static partial int gx_vec_rollout(IntPtr vec, int kSteps, int policy, IntPtr dActions, IntPtr stream);
    [LibraryImport(Lib)] internal static partial int gx_vec_rollout_out(IntPtr vec, int kSteps, int policy, IntPtr dActions, int @out, IntPtr dRewardRows, IntPtr dDoneRows, IntPtr dTruncatedRows, IntPtr stream);
    [LibraryImport(Lib)] internal static partial int gx_rows_step(IntPtr ctx, GxRows* rows, int kSteps, IntPtr stream);
    [LibraryImport(Lib)] internal static partial int gx_vec_sample_actions(IntPtr vec, ulong tick, IntPtr dActions, IntPtr stream);
    [LibraryImport(Lib)] internal static partial int gx_vec_tick(IntPtr vec, out ulong tick);
    [LibraryImport(Lib)] internal static partial int gx_vec_stats(IntPtr vec, out GxStats stats);
    [LibraryImport(Lib)] internal static partial int gx_comm_unique_id(byte* id128);
    [LibraryImport(Lib)] internal static partial int gx_comm_init(IntPtr ctx, byte* id128, int rank, int world, out IntPtr comm);
    [LibraryImport(Lib)] internal static partial int gx_comm_destroy(IntPtr comm);
    [LibraryImport(Lib)] internal static partial int gx_stats_allreduce(IntPtr comm, IntPtr vec, out GxStats stats);

    // The reference reports errors as exceptions (EnvRegistry.cs:22,31; Spaces.Box.cs:19;
    // Envs.Acrobot.cs:51): status codes are turned back into the matching exception types.
    internal static void Check(int status)
    {
        if (status == 0) return;
        string msg = Marshal.PtrToStringUTF8(gx_last_error()) ?? "gymx error";
        throw (GxStatus)status switch
        {
            GxStatus.Invalid => new ArgumentException(msg),
            GxStatus.Oom => new OutOfMemoryException(msg),
            GxStatus.Unsupported => new NotSupportedException(msg),
            _ => new InvalidOperationException(msg),
        };
    }
}

internal sealed class ContextHandle : SafeHandle
{
    public ContextHandle() : base(IntPtr.Zero, true) { }
    public override bool IsInvalid => handle == IntPtr.Zero;
    protected override bool ReleaseHandle() => Native.gx_ctx_destroy(handle) == 0;
    public static ContextHandle Create(int device)
    {
        Native.Check(Native.gx_ctx_create(device, out IntPtr p));
        var h = new ContextHandle();
        h.SetHandle(p);
        return h;
    }
}

// A batch handle keeps its context alive: gx_vec_destroy synchronises the context's streams, and
// gx_ctx_destroy refuses while batches exist, so the reference taken here is released only after
// the batch is gone -- whatever order the finalizers run in.
internal sealed class VecHandle : SafeHandle
{
    private ContextHandle? _owner;
    private bool _ownerRef;
    public VecHandle() : base(IntPtr.Zero, true) { }
    public override bool IsInvalid => handle == IntPtr.Zero;
    protected override bool ReleaseHandle()
    {
        bool ok = Native.gx_vec_destroy(handle) == 0;
        if (_ownerRef) { _owner!.DangerousRelease(); _ownerRef = false; }
        return ok;
    }
    internal void Set(IntPtr p, ContextHandle owner)
    {
        _owner = owner;
        owner.DangerousAddRef(ref _ownerRef);
        SetHandle(p);
    }
}
