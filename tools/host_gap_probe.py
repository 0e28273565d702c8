"""Where does the HOST spend its time between and inside the C-ABI calls of an eager step?

VERDICT r2 weak #6: the eager Modified3DUNet table shows `conv3d_fwd 1x1x1 64->64 @20x24x20` at 7.8 ms per step although the
kernel takes microseconds (the whole step replays in 15.5 ms as a hipGraph).  A bracket is two events on the launch stream;
when the device is AHEAD of the host an event pair measures host time, not kernel time — so this probe stamps the host clock
at every bracket's entry and exit and prints, per step, the brackets with a long host-side body and the long host-side gaps
in front of a bracket, next to the device time of the same bracket.

    python tools/host_gap_probe.py [m3d|cfg3ae|cfg5] [steps]
"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "m3d"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)

log = []          # (tag, t_enter, t_exit, e0, e1)
_enter, _exit = ops._timed.__enter__, ops._timed.__exit__
state = {"on": False}


class Probe(ops._timed):
    __slots__ = ("t0", "ev0")

    def __enter__(self):
        if not state["on"]:
            return _enter(self)
        self.tag = self.tag_fn()
        self.t0 = time.perf_counter()
        self.ev0 = torch.cuda.Event(enable_timing=True)
        self.ev0.record()
        self.e0 = None

    def __exit__(self, *exc):
        if not state["on"]:
            return _exit(self, *exc)
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        log.append((self.tag, self.t0, time.perf_counter(), self.ev0, ev1))
        return False


ops._timed = Probe

if which == "m3d":
    from mri_epilepsy_diagnosis_amd.segmentation.models.modified_3dunet import Modified3DUNet
    torch.manual_seed(0)
    m = Modified3DUNet(1, 2, 8).to(dev)
    opt = torch.optim.AdamW(m.parameters())
    x = torch.randn(1, 1, 160, 192, 160, device=dev, generator=g)
    t = (torch.rand(1, 1, 160, 192, 160, device=dev, generator=g) < 0.1).float()

    def step():
        opt.zero_grad()
        ops.softmax_dice_loss(m(x), t).backward()
        opt.step()
else:
    raise SystemExit("unknown configuration " + which)

for _ in range(4):
    step()
torch.cuda.synchronize()
gc_events = []
gc.callbacks.append(lambda phase, info: gc_events.append((phase, info.get("generation"), time.perf_counter())))
if len(sys.argv) > 3 and sys.argv[3] == "nosync":
    # the model_bench regime: steps issued back to back, the host free to run ahead of the device, every event kept alive
    state["on"] = True
    t_all0 = time.perf_counter()
    marks = []
    for s in range(steps):
        marks.append((len(log), time.perf_counter()))
        step()
    t_host = time.perf_counter() - t_all0
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t_all0
    state["on"] = False
    print("== %d steps back to back: host issue %.2f ms/step, wall %.2f ms/step, %d brackets" % (steps, t_host * 1e3 / steps, t_wall * 1e3 / steps, len(log)))
    rows = []
    prev_exit = t_all0
    for i, (tag, t0, t1, e0, e1) in enumerate(log):
        stp = max(k for k, (n0, _) in enumerate(marks) if n0 <= i)
        rows.append((e0.elapsed_time(e1), (t0 - prev_exit) * 1e3, (t1 - t0) * 1e3, stp, i - marks[stp][0], tag))
        prev_exit = t1
    print("   brackets by device time:")
    for devms, gap, body, stp, idx, tag in sorted(rows, key=lambda r: -r[0])[:8]:
        print("   step %d bracket #%3d  device %8.3f ms  host gap before %7.3f ms  host body %7.3f ms  %s" % (stp, idx, devms, gap, body, tag))
    print("   brackets by host time (gap before + body):")
    for devms, gap, body, stp, idx, tag in sorted(rows, key=lambda r: -(r[1] + r[2]))[:8]:
        print("   step %d bracket #%3d  device %8.3f ms  host gap before %7.3f ms  host body %7.3f ms  %s" % (stp, idx, devms, gap, body, tag))
    sys.exit(0)
for s in range(steps):
    log.clear()
    gc_events.clear()
    state["on"] = True
    t_step0 = time.perf_counter()
    step()
    t_host = time.perf_counter() - t_step0
    torch.cuda.synchronize()
    t_wall = time.perf_counter() - t_step0
    state["on"] = False
    print("== step %d: host issue %.2f ms, wall %.2f ms, %d brackets, %d gc phases" % (s, t_host * 1e3, t_wall * 1e3, len(log), len(gc_events)))
    prev_exit = t_step0
    rows = []
    for tag, t0, t1, e0, e1 in log:
        rows.append((tag, (t0 - prev_exit) * 1e3, (t1 - t0) * 1e3, e0.elapsed_time(e1), (t0 - t_step0) * 1e3))
        prev_exit = t1
    for tag, gap, body, devms, at in sorted(rows, key=lambda r: -max(r[1], r[2], r[3]))[:10]:
        print("   at %7.2f ms  host gap before %6.3f ms  host body %6.3f ms  device bracket %6.3f ms  %s" % (at, gap, body, devms, tag))
    for ph, gen, tt in gc_events:
        if ph == "stop":
            print("   gc gen %s ended at %.2f ms" % (gen, (tt - t_step0) * 1e3))
