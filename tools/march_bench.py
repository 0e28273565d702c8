"""A/B of the d-marching forward / data-gradient kernel (csrc/conv_march.hip) against what the dispatcher picks, on the
3x3x3 layers of unet.UNet(c0=8) at 2 x 160x192x160 (BASELINE configs[1] / [3]) through the C ABI.

    python tools/march_bench.py [--lib SO] [--mode auto|march] [--dtype bf16|f32] [--reps R] [--layers NAME,NAME...]

--mode auto : mri3d_conv3d_{fwd,dgrad}[_cat]  (the dispatcher: marching kernel where its plan takes the layer)
--mode march: mri3d_conv3d_{fwd,dgrad}_march   (the marching kernel by name; 'unsupported' where it cannot compute the layer)
--lib SO    : a tuning build, e.g. `python -m mri_epilepsy_diagnosis_amd.build --variant nomarch -DMRI3D_NO_MARCH` = the round-2
              dispatcher (tiled kernel everywhere)
Prints per layer and pass: ms, TFLOP/s, algorithmic GB/s (input + output once)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import _lib, ops  # noqa: E402

LAYERS = {   # name: (ci, co, (d, h, w), split or 0)
    "enc0.conv2": (8, 16, (160, 192, 160), 0),
    "dec1.conv2": (16, 16, (160, 192, 160), 0),
    "dec1.conv1": (48, 16, (160, 192, 160), 16),
    "enc1.conv1": (16, 16, (80, 96, 80), 0),
    "enc1.conv2": (16, 32, (80, 96, 80), 0),
    "dec0.conv1": (96, 32, (80, 96, 80), 32),
    "dec0.conv2": (32, 32, (80, 96, 80), 0),
    "bottom.conv1": (32, 32, (40, 48, 40), 0),
    "bottom.conv2": (32, 64, (40, 48, 40), 0),
}


def main():
    argv = sys.argv[1:]

    def opt(name, dflt):
        if name in argv:
            i = argv.index(name)
            v = argv[i + 1]
            del argv[i:i + 2]
            return v
        return dflt
    lib = opt("--lib", None)
    if lib:
        _lib.LIB_PATH = os.path.abspath(lib)
    mode = opt("--mode", "auto")
    dt = torch.bfloat16 if opt("--dtype", "bf16") == "bf16" else torch.float32
    reps = int(opt("--reps", "20"))
    names = opt("--layers", ",".join(LAYERS)).split(",")
    n = int(opt("--batch", "2"))
    warm = float(opt("--warm", "1.5"))   # seconds of back-to-back launches before the first measurement (the chip's clock settles)
    L = _lib.lib()
    dev = torch.device("cuda")
    CL = torch.channels_last_3d
    P, st = ops._ptr, ops._stream
    esz = 2 if dt == torch.bfloat16 else 4
    print("# lib %s  mode %s  dtype %s  batch %d  reps %d" % (os.path.basename(_lib.LIB_PATH), mode, str(dt)[6:], n, reps))
    for name in names:
        ci, co, (d, h, w), split = LAYERS[name]
        gen = torch.Generator(device=dev).manual_seed(0)
        c1 = split if split else ci
        xa = torch.randn(n, c1, d, h, w, device=dev, generator=gen).to(dt).contiguous(memory_format=CL)
        xb = torch.randn(n, ci - c1, d, h, w, device=dev, generator=gen).to(dt).contiguous(memory_format=CL) if split else None
        dy = torch.randn(n, co, d, h, w, device=dev, generator=gen).to(dt).contiguous(memory_format=CL)
        wt = torch.randn(co, ci, 3, 3, 3, device=dev, generator=gen) * 0.1
        b = torch.randn(co, device=dev, generator=gen)
        g = ops._conv_geom((n, ci, d, h, w), wt.shape, (1, 1, 1), (1, 1, 1), (1, 1, 1), x_ld=c1, y_ld=co, dtype=ops._dt(xa))
        y = torch.empty_like(dy)
        dxa = torch.empty_like(xa)
        dxb = torch.empty_like(xb) if split else None
        ws = {ps: ops._workspace(L.mri3d_conv3d_workspace_bytes(ctypes.byref(g), ps), dev) for ps in (ops.PASS_FWD, ops.PASS_DGRAD)}
        c2 = ci - c1

        def fwd():
            wsf = ws[ops.PASS_FWD]
            if mode == "march":
                return L.mri3d_conv3d_fwd_march(ctypes.byref(g), P(xa), P(xb), c1, c2, P(wt), P(b), P(y), None, P(wsf), wsf.numel(), st())
            if split:
                return L.mri3d_conv3d_fwd_cat(ctypes.byref(g), P(xa), P(xb), c1, c2, P(wt), P(b), P(y), None, P(wsf), wsf.numel(), st())
            return L.mri3d_conv3d_fwd(ctypes.byref(g), P(xa), P(wt), P(b), P(y), P(wsf), wsf.numel(), st())

        def dgrad():
            wsd = ws[ops.PASS_DGRAD]
            if mode == "march":
                return L.mri3d_conv3d_dgrad_march(ctypes.byref(g), P(dy), P(wt), P(dxa), P(dxb), c1, c2, P(wsd), wsd.numel(), st())
            if split:
                return L.mri3d_conv3d_dgrad_cat(ctypes.byref(g), P(dy), P(wt), P(dxa), P(dxb), c1, c2, P(wsd), wsd.numel(), st())
            return L.mri3d_conv3d_dgrad(ctypes.byref(g), P(dy), P(wt), None, P(dxa), P(wsd), wsd.numel(), st())

        flops = 2.0 * n * co * ci * 27 * d * h * w
        nbytes = float(n * d * h * w * (ci + co) * esz)
        if warm > 0 and fwd() == 0:
            import time
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < warm:
                for _ in range(50):
                    fwd()
                torch.cuda.synchronize()
            warm = 0.25   # later layers: a short refresher
        for pname, fn in (("fwd", fwd), ("dgrad", dgrad)):
            rc = fn()
            if rc != 0:
                print("%-12s %-5s %3d->%-3d @%dx%dx%d: unsupported (%s)" % (name, pname, ci, co, d, h, w, L.mri3d_last_error().decode()[:60]), flush=True)
                continue
            fn()
            torch.cuda.synchronize()
            stamps = getattr(L, "mri3d_debug_march_stamps", None)   # -DMRI3D_EXPERIMENT_STAMPS builds only
            if stamps is not None:
                stamps(None, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print("%-12s %-5s %3d->%-3d @%dx%dx%d: %7.3f ms  %7.1f TFLOP/s  %7.1f GB/s" % (name, pname, ci, co, d, h, w, ms, flops / ms / 1e9, nbytes / ms / 1e6),
                  flush=True)
            if stamps is not None:   # wave 0 of workgroup 0: shader clocks per item in retire (store + rotate) / DMA wait / the 15 groups
                buf = (ctypes.c_ulonglong * 8)()
                stamps(buf, 0)
                items = max(1, buf[3])
                print("      stamps/item (%d items per launch): retire %.0f  dma-wait %.0f  groups %.0f  | whole march %.0f clk/item, clock %.2f GHz"
                      % (items // reps, buf[0] / items, buf[1] / items, buf[2] / items, buf[6] / items, buf[6] / max(1, buf[7]) * 0.1), flush=True)
                spans = getattr(L, "mri3d_debug_march_spans", None)
                if spans is not None:   # when do the workgroups of the last launch start and end their march (100 MHz ticks)?
                    import numpy as np
                    sb = (ctypes.c_ulonglong * 4096)()
                    spans(sb)
                    a = np.array(sb[:], dtype=np.int64).reshape(1024, 4)
                    a = a[a[:, 2] > 0]
                    t0 = a[:, 0].min()
                    ent, st_, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0
                    print("      %d workgroups: kernel entry (us) median %.1f max %.1f | march start median %.1f max %.1f | march end min %.1f p10 %.1f median %.1f p90 %.1f max %.1f | march length min %.1f median %.1f max %.1f"
                          % (len(a), np.median(ent), ent.max(), np.median(st_), st_.max(), en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max(),
                             (en - st_).min(), np.median(en - st_), (en - st_).max()), flush=True)
        del xa, xb, dy, y, dxa, dxb


if __name__ == "__main__":
    main()
