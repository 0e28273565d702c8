"""Read-only, write-only and copy bandwidth of plain torch kernels on this GPU (context for the streaming kernels' numbers).
    python tools/hbm_probe.py"""
import torch

n = 1258 * 1024 * 1024 // 4   # 1.26 GB of fp32, the size of the 32-channel fine tensor of the U-Net decoder
a = torch.empty(n, device="cuda")
b = torch.empty(n, device="cuda")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gb = n * 4 / 1e9
t = timeit(lambda: a.fill_(1.0))
print("write-only (fill_)      %.3f ms  %.2f TB/s" % (t, gb / t))
t = timeit(lambda: a.sum())
print("read-only  (sum)        %.3f ms  %.2f TB/s" % (t, gb / t))
t = timeit(lambda: b.copy_(a))
print("copy       (read+write) %.3f ms  %.2f TB/s of traffic" % (t, 2 * gb / t))
t = timeit(lambda: torch.add(a, 1.0, out=b))
print("add scalar (read+write) %.3f ms  %.2f TB/s of traffic" % (t, 2 * gb / t))
