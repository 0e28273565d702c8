"""Can an HBM-bound kernel run under an MFMA-bound one on a second stream?  Times the 48->16 weight gradient (or forward) and
a plain streaming kernel separately and concurrently.   python tools/overlap_probe.py [wgrad|fwd]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "wgrad"
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(2, 48, 160, 192, 160, device=dev, generator=g).contiguous(memory_format=torch.channels_last_3d)
dy = torch.randn(2, 16, 160, 192, 160, device=dev, generator=g).contiguous(memory_format=torch.channels_last_3d)
wt = torch.randn(16, 48, 3, 3, 3, device=dev, generator=g) * 0.1
b = torch.randn(16, device=dev, generator=g)
geom = ops._conv_geom(x.shape, wt.shape, (1, 1, 1), (1, 1, 1), (1, 1, 1), dtype=ops._dt(x))
a1 = torch.empty(629 * 1024 * 1024 // 4, device=dev)
a2 = torch.empty_like(a1)
side = torch.cuda.Stream()


def conv():
    if which == "wgrad":
        ops._conv_wgrad(geom, x, dy, wt, True)
    else:
        ops._conv_fwd(geom, x, wt, b)


def stream_kernel():
    torch.add(a1, 1.0, out=a2)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def both():
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        conv()
        done = torch.cuda.Event()
        done.record()
    for _ in range(6):
        stream_kernel()
    torch.cuda.current_stream().wait_event(done)


with torch.cuda.stream(side):
    conv()          # the side stream gets its own workspace
torch.cuda.synchronize()
t_conv = timed(conv)
t_str = timed(lambda: [stream_kernel() for _ in range(6)])
t_both = timed(both)
print("%s 48->16: %.3f ms; 6 x add over 629 MB: %.3f ms; both on two streams: %.3f ms (serial would be %.3f)"
      % (which, t_conv, t_str, t_both, t_conv + t_str))
