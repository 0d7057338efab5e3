// Registers device-backed ids through the unchanged EnvRegistry.Register
// (EnvRegistry.cs:13-17), next to GymnasiumRegistration.RegisterAll
// (GymnasiumRegistration.cs:8-30).  A single-env object is a batch of one, so
// MainWindowViewModel's episode loop (MainWindowViewModel.cs:265-312) and the
// baseline agents, which only use Reset / Step / ActionSpace / ObservationSpace
// through `dynamic`, run on it unmodified.  Not compiled in the build container.
using System.Collections.Generic;
using Gymnasium.Spaces;

namespace Gymnasium.Gpu;

public sealed class GpuCartPole : Env<(float, float, float, float), int>
{
    public Discrete ActionSpace { get; } = new(2);                       // Envs.CartPole.cs:12
    public Box ObservationSpace { get; } = new(                          // Envs.CartPole.cs:13-16
        new float[] { -2.4f, -3.0f, -0.2095f, -3.0f }, new float[] { 2.4f, 3.0f, 0.2095f, 3.0f });
    private readonly GpuVectorEnv<(float, float, float, float), int> _vec = GpuEnvs.CartPole(1);
    private readonly List<int> _one = new() { 0 };

    public override (float, float, float, float) Reset() => _vec.Reset()[0];
    public override ((float, float, float, float) state, double reward, bool done, IDictionary<string, object> info) Step(int action)
    {
        _one[0] = action;
        return _vec.Step(_one)[0];
    }
    public override void Render(string mode = "human") { }
    public override void Close() => _vec.Dispose();
}

public static class GpuRegistration
{
    public static void RegisterAll()
    {
        EnvRegistry.Register("CartPole-v1-gpu", () => new GpuCartPole());
        // the other ids follow the same pattern with GpuEnvs.MountainCar / Pendulum / Acrobot ...
    }
}
