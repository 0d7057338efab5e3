"""Pure-write rate of this GPU: hipMemset-style fill of a buffer the size of 65 536 frames, and a
device-to-device copy, for comparison with the frame raster kernels."""
import torch

n = 65536 * 100800
x = torch.empty(n, dtype=torch.uint8, device="cuda")
y = torch.empty(n, dtype=torch.uint8, device="cuda")


def rate(fn, bytes_moved, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, bytes_moved / (ms * 1e-3) / 1e9


print("fill   %d bytes: %.3f ms, %.0f GB/s written" % ((n,) + rate(lambda: x.zero_(), n)))
print("fill32 %d bytes: %.3f ms, %.0f GB/s written" % ((n,) + rate(lambda: x.view(torch.int32).fill_(7), n)))
print("copy   %d bytes: %.3f ms, %.0f GB/s read+written" % ((n,) + rate(lambda: y.copy_(x), 2 * n)))
