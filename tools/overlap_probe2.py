"""Do the data gradient and the weight gradient of one conv layer — independent kernels — finish sooner on two streams than back to
back?  bf16 kernels sit at 30-45 % MFMA busy (latency / fabric bound), fp32 at 80 %.   python tools/overlap_probe2.py [f32|bf16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops  # noqa: E402

dt = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream()
for ci, co, sp in ((48, 16, (160, 192, 160)), (16, 16, (160, 192, 160)), (96, 32, (80, 96, 80))):
    x = torch.randn(2, ci, *sp, device=dev, generator=g).to(dt).contiguous(memory_format=torch.channels_last_3d)
    dy = torch.randn(2, co, *sp, device=dev, generator=g).to(dt).contiguous(memory_format=torch.channels_last_3d)
    wt = torch.randn(co, ci, 3, 3, 3, device=dev, generator=g) * 0.1
    geom = ops._conv_geom(x.shape, wt.shape, (1, 1, 1), (1, 1, 1), (1, 1, 1), dtype=ops._dt(x))

    def seq():
        ops._conv_dgrad(geom, dy, wt, None, x)
        ops._conv_wgrad(geom, x, dy, wt, True)

    def par():
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(side):
            side.wait_event(ev)
            ops._conv_wgrad(geom, x, dy, wt, True)
            done = torch.cuda.Event()
            done.record()
        ops._conv_dgrad(geom, dy, wt, None, x)
        torch.cuda.current_stream().wait_event(done)

    def timed(fn, reps=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    with torch.cuda.stream(side):
        ops._conv_wgrad(geom, x, dy, wt, True)
    torch.cuda.synchronize()
    print("%s %d->%d @%s: dgrad + wgrad back to back %.3f ms, on two streams %.3f ms" % (str(dt)[6:], ci, co, "x".join(map(str, sp)), timed(seq), timed(par)), flush=True)
