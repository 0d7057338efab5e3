import torch, time
n = 65536 * 100800
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n // 4, dtype=torch.int32, device="cuda")
for name, fn in [("uint8 zero_", lambda: x.zero_()), ("int32 fill_", lambda: y.fill_(7)), ("hipMemsetAsync", lambda: torch.cuda.cudart().cudaMemset(x.data_ptr(), 0, n) if hasattr(torch.cuda.cudart(), "cudaMemset") else x.zero_())]:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%s: %.3f ms, %.0f GB/s" % (name, ms, n / ms / 1e6))
