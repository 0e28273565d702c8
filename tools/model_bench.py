"""Throughput of the other BASELINE.json configurations (parity-test cases, not bench lines) on one MI355X:
  cfg3  classification encoder (93_6_4 kwargs) + head, batch 4 x 160x192x160, CE step;   cfg3ae  the full AE, MSE step
  cfg5  CNN(32^3 patches), batch 512 (stand-in for the 2-D detection net)
  m3d   Modified3DUNet(1,2,8), batch 1 x 160x192x160, soft-Dice step
  cfg4  unet.UNet(c0=8) under the bf16 autocast region, batch 2 x 160x192x160 per GPU (configs[3]'s per-GPU share);
        cfg2 = the same step in fp32 for the side-by-side
  patch16  the reference's patch training (pretraining_3d_unet.ipynb cells 24-25): U-Net c0=16 (c8: c0=8) on batches of
        16 random 64^3 windows cut from HBM-resident subjects by segmentation/patches.py
Prints ms/step, units/s and the per-operator device-time table (top N)."""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mri_epilepsy_diagnosis_amd import ops  # noqa: E402
from mri_epilepsy_diagnosis_amd.classification.models import AE_model, cnn_model  # noqa: E402
from mri_epilepsy_diagnosis_amd.segmentation.models.modified_3dunet import Modified3DUNet  # noqa: E402
from util import AE_KWARGS_93_6_4, CLF_KWARGS  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which == "all":
    # one fresh process per configuration (this parent never touches the GPU): run back to back in ONE process, the
    # caching allocator state left by the previous configuration showed up as multi-millisecond host stalls inside
    # later steps (full-AE step 23 ms in sequence vs 10.3 ms on its own)
    import subprocess
    rc = 0
    for cfg in ("cfg3", "cfg3_graph", "cfg3ae", "cfg3ae_graph", "cfg5", "m3d", "m3d_graph", "patch16", "patch16_c8", "cfg2", "cfg4"):
        rc |= subprocess.run([sys.executable, os.path.abspath(__file__), cfg]).returncode
    sys.exit(rc)
dev = torch.device("cuda")
TOP = int(os.environ.get("TOP", "12"))


def run(name, units, step, steps=8, warmup=4):
    for _ in range(warmup):
        step()
    # two bracketed steps that are thrown away: the first few hundred timing events of a process make the HIP runtime grow its
    # signal pool ONCE, a 55-77 ms host stall inside whatever call happens to be running (tools/host_gap_probe.py) — measured as
    # a 7-9 ms-per-step bracket on a microsecond kernel when it fell into the timed steps
    ops.set_timer(ops.KernelTimer())
    for _ in range(2):
        step()
    ops.set_timer(None)
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ops.set_timer(None)
    print("== %s: %.2f ms/step, %.1f units/s" % (name, dt * 1e3, units / dt), flush=True)
    agg = timer.summary()
    tot = sum(a["ms"] for a in agg.values())
    for tag, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:TOP]:
        w = a["work"] or {}
        print("   %8.3f ms/step %5.1f%% %7.2f TF/s %8.1f GB/s x%-3d %s" % (
            a["ms"] / steps, 100 * a["ms"] / tot, w.get("flops", 0) * a["calls"] / (a["ms"] / 1e3) / 1e12,
            w.get("bytes", 0) * a["calls"] / (a["ms"] / 1e3) / 1e9, a["calls"] // steps, tag))


g = torch.Generator(device=dev).manual_seed(0)
if which == "cfg3":
    torch.manual_seed(0)
    enc = AE_model.AE(**AE_KWARGS_93_6_4).enc.to(dev)
    clf = AE_model.Classificator(**dict(CLF_KWARGS, conv_pad=1, l_in=64 * 2 * 3 * 2)).to(dev)
    opt = torch.optim.Adam(list(enc.parameters()) + list(clf.parameters()), lr=7e-4, weight_decay=1e-4)
    x = torch.randn(4, 1, 160, 192, 160, device=dev, generator=g)
    y = torch.randint(0, 2, (4,), device=dev, generator=g)

    def step3():
        opt.zero_grad()
        lat, _ = enc(x)
        F.cross_entropy(clf(lat), y).backward()
        opt.step()
    run("cfg3 encoder+clf CE step, 4 x 160x192x160", 4, step3)
if which == "cfg3ae":
    torch.manual_seed(0)
    x = torch.randn(4, 1, 160, 192, 160, device=dev, generator=g)
    ae = AE_model.AE(**AE_KWARGS_93_6_4).to(dev)
    opt2 = torch.optim.Adam(ae.parameters(), lr=1e-3)

    def step3b():
        opt2.zero_grad()
        F.mse_loss(ae(x), x).backward()
        opt2.step()
    run("cfg3 full AE MSE step, 4 x 160x192x160", 4, step3b)
if which in ("cfg3_graph", "cfg3ae_graph"):
    # the same two steps with forward+backward replayed as one hipGraph (parallel.CapturedStep) and the fused flat Adam:
    # ~200 small launches per step make the eager numbers above a measurement of the host, not of the kernels
    from mri_epilepsy_diagnosis_amd import parallel
    torch.manual_seed(0)
    x = torch.randn(4, 1, 160, 192, 160, device=dev, generator=g)
    if which == "cfg3_graph":
        enc = AE_model.AE(**AE_KWARGS_93_6_4).enc.to(dev)
        clf = AE_model.Classificator(**dict(CLF_KWARGS, conv_pad=1, l_in=64 * 2 * 3 * 2)).to(dev)
        mod = torch.nn.ModuleList([enc, clf])
        y = torch.randint(0, 2, (4,), device=dev, generator=g)
        loss_fn = lambda: F.cross_entropy(clf(enc(x)[0]), y)  # noqa: E731
        name, lr, wd = "cfg3 encoder+clf CE step", 7e-4, 1e-4
    else:
        mod = AE_model.AE(**AE_KWARGS_93_6_4).to(dev)
        loss_fn = lambda: F.mse_loss(mod(x), x)  # noqa: E731
        name, lr, wd = "cfg3 full AE MSE step", 1e-3, 0.0
    flat = parallel.FlatParams(mod)
    fopt = parallel.FlatAdam(flat, lr=lr, weight_decay=wd, decoupled=False)
    cap = parallel.CapturedStep(flat, loss_fn).capture()

    def step3g():
        cap.run()
        fopt.step(flat.all_reduce())
    run("%s, hipGraph fwd+bwd, 4 x 160x192x160" % name, 4, step3g)
if which == "cfg5":
    torch.manual_seed(0)
    net = torch.nn.Sequential(cnn_model.CNN(input_shape=(32, 32, 32), n_filters=16, n_blocks=3), torch.nn.Linear(128, 2)).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-5, weight_decay=0.01)
    x5 = torch.randn(512, 1, 32, 32, 32, device=dev, generator=g)
    y5 = torch.randint(0, 2, (512,), device=dev, generator=g)

    def step5():
        opt.zero_grad()
        F.cross_entropy(net(x5), y5).backward()
        opt.step()
    run("cfg5 CNN 32^3 patches, batch 512", 512, step5)
if which == "m3d":
    torch.manual_seed(0)
    m = Modified3DUNet(1, 2, 8).to(dev)
    opt = torch.optim.AdamW(m.parameters())
    x6 = torch.randn(1, 1, 160, 192, 160, device=dev, generator=g)
    t6 = (torch.rand(1, 1, 160, 192, 160, device=dev, generator=g) < 0.1).float()

    def step6():
        opt.zero_grad()
        ops.softmax_dice_loss(m(x6), t6).backward()
        opt.step()
    run("Modified3DUNet(1,2,8) dice step, 1 x 160x192x160", 1, step6)
if which == "m3d_graph":
    # the same step with forward+backward replayed as one hipGraph (parallel.CapturedStep) and the fused flat AdamW
    from mri_epilepsy_diagnosis_amd import parallel
    torch.manual_seed(0)
    m = Modified3DUNet(1, 2, 8).to(dev)
    flat = parallel.FlatParams(m)
    fopt = parallel.FlatAdam(flat, lr=1e-3, weight_decay=0.01, decoupled=True)
    x6 = torch.randn(1, 1, 160, 192, 160, device=dev, generator=g)
    t6 = (torch.rand(1, 1, 160, 192, 160, device=dev, generator=g) < 0.1).float()
    cap = parallel.CapturedStep(flat, lambda: ops.softmax_dice_loss(m(x6), t6)).capture()

    def step6g():
        cap.run()
        fopt.step(flat.all_reduce())
    run("Modified3DUNet(1,2,8) dice step, hipGraph fwd+bwd, 1 x 160x192x160", 1, step6g)
for tag, c0 in (("patch16", 16), ("patch16_c8", 8)):
    # patch training of pretraining_3d_unet.ipynb cells 24-25: get_model_and_optimizer's U-Net (c0=16), batches of 16 random
    # 64^3 windows from the HBM-resident queue (segmentation/patches.py), soft-Dice step; the queue pop + 2 extract launches
    # are inside the timed step
    if which != tag:
        continue
    from mri_epilepsy_diagnosis_amd import parallel
    from mri_epilepsy_diagnosis_amd.segmentation import patches as PT
    from mri_epilepsy_diagnosis_amd.unet import UNet
    torch.manual_seed(0)
    net = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=c0,
               normalization="batch", upsampling_type="linear", padding=True, activation="PReLU").to(dev)
    flat = parallel.FlatParams(net)
    fopt = parallel.FlatAdam(flat, lr=1e-3, weight_decay=0.01, decoupled=True)
    subjects = [{PT.MRI: {PT.DATA: torch.randn(1, 160, 192, 160, device=dev, generator=g)},
                 PT.LABEL: {PT.DATA: (torch.rand(1, 160, 192, 160, device=dev, generator=g) < 0.1).float()}}
                for _ in range(30)]
    queue = PT.Queue(subjects, max_length=240, samples_per_volume=8, patch_size=64, seed=0)
    state = {"it": queue.batches(16)}

    def step8():
        try:
            b = next(state["it"])
        except StopIteration:
            state["it"] = queue.batches(16)
            b = next(state["it"])
        flat.zero_grad()
        ops.softmax_dice_loss(net(b[PT.MRI][PT.DATA]), b[PT.LABEL][PT.DATA]).backward()
        fopt.step(flat.all_reduce())
    run("%s unet.UNet(c0=%d) fp32 dice step on queue batches of 16 x 64^3 windows" % (tag, c0), 16, step8)
for tag, use_bf16 in (("cfg2", False), ("cfg4", True)):
    if which != tag:
        continue
    from mri_epilepsy_diagnosis_amd import parallel
    from mri_epilepsy_diagnosis_amd.unet import UNet
    torch.manual_seed(0)
    net = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=8,
               normalization="batch", upsampling_type="linear", padding=True, activation="PReLU").to(dev)
    flat = parallel.FlatParams(net)
    fopt = parallel.FlatAdam(flat, lr=1e-3, weight_decay=0.01, decoupled=True)
    x7 = torch.randn(2, 1, 160, 192, 160, device=dev, generator=g)
    t7 = (torch.rand(2, 1, 160, 192, 160, device=dev, generator=g) < 0.1).float()

    def step7():
        flat.zero_grad()
        with ops.autocast(enabled=use_bf16):
            loss = ops.softmax_dice_loss(net(x7), t7)
        loss.backward()
        fopt.step(flat.all_reduce())
    run("%s unet.UNet(c0=8) %s dice step, 2 x 160x192x160" % (tag, "bf16 autocast" if use_bf16 else "fp32"), 2, step7)
