"""Micro-benchmark of one Conv3d geometry through the C ABI (forward, dgrad, wgrad), for kernel tuning and for
rocprofv3 --pmc runs:   python tools/conv_bench.py CIN COUT D H W [N] [reps] [passes=fwd,dgrad,wgrad] [f32|bf16]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import _lib, ops  # noqa: E402


def main():
    if "--lib" in sys.argv:   # a tuning build (python -m mri_epilepsy_diagnosis_amd.build --variant NAME -D...), tools only
        i = sys.argv.index("--lib")
        _lib.LIB_PATH = os.path.abspath(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    ci, co, d, h, w = (int(a) for a in sys.argv[1:6])
    n = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
    passes = sys.argv[8].split(",") if len(sys.argv) > 8 else ["fwd", "dgrad", "wgrad"]
    dt = torch.bfloat16 if (len(sys.argv) > 9 and sys.argv[9] == "bf16") else torch.float32
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(n, ci, d, h, w, device=dev, generator=g).contiguous(memory_format=torch.channels_last_3d)
    dy = torch.randn(n, co, d, h, w, device=dev, generator=g).contiguous(memory_format=torch.channels_last_3d)
    x, dy = x.to(dt), dy.to(dt)
    wt = torch.randn(co, ci, 3, 3, 3, device=dev, generator=g) * 0.1
    b = torch.randn(co, device=dev, generator=g)
    geom = ops._conv_geom(x.shape, wt.shape, (1, 1, 1), (1, 1, 1), (1, 1, 1), dtype=ops._dt(x))
    flops = 2.0 * n * co * ci * 27 * d * h * w
    fns = {"fwd": lambda: ops._conv_fwd(geom, x, wt, b), "dgrad": lambda: ops._conv_dgrad(geom, dy, wt, None, x),
           "wgrad": lambda: ops._conv_wgrad(geom, x, dy, wt, True)}
    for p in passes:
        fn = fns[p]
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-5s %d->%d @%dx%dx%d n%d %s: %.3f ms  %.1f TFLOP/s" % (p, ci, co, d, h, w, n, str(dt)[6:], ms, flops / ms / 1e9),
              flush=True)


if __name__ == "__main__":
    main()
