"""Micro-benchmark of one Conv3d geometry through the C ABI (forward, dgrad, wgrad), for kernel tuning and for
rocprofv3 --pmc runs:   python tools/conv_bench.py [--lib SO] [--cat SPLIT] CIN COUT D H W [N] [reps] [passes=fwd,dgrad,wgrad] [f32|bf16]
--cat SPLIT: the convolution over cat((xa, xb)) read from two dense tensors of SPLIT and CIN - SPLIT channels (mri3d_conv3d_*_cat:
the U-Net decoder's first convolution)."""
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
    split = 0
    if "--cat" in sys.argv:
        i = sys.argv.index("--cat")
        split = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    ci, co, d, h, w = (int(a) for a in sys.argv[1:6])
    n = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10   # >= 2 s of back-to-back launches for a settled clock: reps ~ 2000
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
    if split:
        L = _lib.lib()
        CL = torch.channels_last_3d
        xa, xb = x[:, :split].contiguous(memory_format=CL), x[:, split:].contiguous(memory_format=CL)
        del x
        gf = ops._conv_geom((n, ci, d, h, w), wt.shape, (1, 1, 1), (1, 1, 1), (1, 1, 1), x_ld=split, dtype=ops._dt(xa))
        for ps, cb in ((ops.PASS_FWD, ci - split), (ops.PASS_DGRAD, ci - split), (ops.PASS_WGRAD, ci - split)):
            assert L.mri3d_conv3d_cat_supported(ctypes.byref(gf), split, cb, ps), "split geometry not served"
        y = torch.empty((n, co, d, h, w), device=dev, dtype=dt).contiguous(memory_format=CL)
        dxa, dxb, dw, db = torch.empty_like(xa), torch.empty_like(xb), torch.empty_like(wt), torch.empty_like(b)
        ws = {ps: ops._workspace(L.mri3d_conv3d_workspace_bytes(ctypes.byref(gf), ps), dev) for ps in (ops.PASS_FWD, ops.PASS_DGRAD, ops.PASS_WGRAD)}
        P, st = ops._ptr, ops._stream

        def chk(rc):
            assert rc == 0, L.mri3d_last_error().decode()
        fns = {"fwd": lambda: chk(L.mri3d_conv3d_fwd_cat(ctypes.byref(gf), P(xa), P(xb), split, ci - split, P(wt), P(b), P(y), None,
                                                         P(ws[ops.PASS_FWD]), ws[ops.PASS_FWD].numel(), st())),
               "dgrad": lambda: chk(L.mri3d_conv3d_dgrad_cat(ctypes.byref(gf), P(dy), P(wt), P(dxa), P(dxb), split, ci - split,
                                                             P(ws[ops.PASS_DGRAD]), ws[ops.PASS_DGRAD].numel(), st())),
               "wgrad": lambda: chk(L.mri3d_conv3d_wgrad_cat(ctypes.byref(gf), P(xa), P(xb), split, ci - split, P(dy), P(dw), P(db),
                                                             P(ws[ops.PASS_WGRAD]), ws[ops.PASS_WGRAD].numel(), st()))}
    for p in passes:
        fn = fns[p]
        fn(); fn()
        torch.cuda.synchronize()
        stamps = getattr(_lib.lib(), "mri3d_debug_stamps", None) if p != "wgrad" else None   # -DMRI3D_EXPERIMENT_STAMPS builds only
        if stamps is not None:
            stamps(None, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%-5s %s->%d @%dx%dx%d n%d %s: %.3f ms  %.1f TFLOP/s" % (p, ("%d+%d" % (split, ci - split)) if split else str(ci), co, d, h, w, n, str(dt)[6:], ms, flops / ms / 1e9),
              flush=True)
        if stamps is not None:   # wave 0 of workgroup 0: clocks per work item in the chunk loop / epilogue / DMA wait / barrier
            buf = (ctypes.c_ulonglong * 8)()
            stamps(buf, 0)
            items = max(1, buf[5])
            print("      stamps/item (%d items): mfma %.0f  epilogue %.0f  dma-wait %.0f  barrier %.0f  total %.0f clk"
                  % (items // reps, buf[0] / items, buf[1] / items, buf[2] / items, buf[3] / items, buf[4] / items), flush=True)
            spans = getattr(_lib.lib(), "mri3d_debug_block_spans", None)
            if spans is not None:   # when do the workgroups of the last launch enter and leave their tile loop (100 MHz ticks)?
                import numpy as np
                sb = (ctypes.c_ulonglong * 2048)()
                spans(sb)
                a = np.array(sb[:], dtype=np.int64).reshape(1024, 2)[:512]
                t0 = a[:, 0].min()
                st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
                print("      workgroup loop entry (us after the first): median %.1f max %.1f;  exit: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f"
                      % (np.median(st), st.max(), en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max()))
                print("      exit by XCD (median us): " + " ".join("%.1f" % np.median(en[k::8]) for k in range(8))
                      + ";  loop length by XCD: " + " ".join("%.1f" % np.median((en - st)[k::8]) for k in range(8)), flush=True)
            print("      in-kernel clock %.2f GHz (s_memtime / s_memrealtime x 100 MHz, MI355X_MICROARCH.md DVFS item 6); MFMA-pipe cycles per"
                  " CU-cycle at that clock = TFLOP/s / (%.1f x clock/2.4)" % (buf[6] / max(1, buf[7]) * 0.1, 157.3 if dt == torch.float32 else 2516.6), flush=True)


if __name__ == "__main__":
    main()
