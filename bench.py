#!/usr/bin/env python3
"""bench.py — MRI volumes/sec (fwd + bwd + optimizer step) of the 3-D U-Net at 160x192x160 on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype f32|bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): `unet.UNet`(c0=8, 3 encoding blocks,
BatchNorm + PReLU, trilinear upsampling) on a batch of 2 synthetic z-normalised-T1-like fp32 volumes of
1x160x192x160 per GPU, soft-Dice loss against a Bernoulli(0.1) mask, AdamW.  One step = forward + softmax-Dice +
backward + (flat-gradient RCCL all-reduce when N > 1) + fused AdamW on every rank (weak scaling: 2 volumes per GPU).
Inputs are generated on the device before the timed region.
`--dtype bf16` (not the default, not the headline) runs the same step inside the bf16 storage region — BASELINE configs[3]'s
per-GPU share (2 volumes per GPU, batch 16 over 8 GPUs): bf16 activations, fp32 accumulate / parameters / loss.

Prints ONE JSON line on rank 0 (see the driver contract), including
  roofline     — the dominant kernel (largest share of step time, found over the last two warm-up steps where every operator
                 is bracketed) measured live with stream events inside the timed region: algorithmic FLOP (or bytes) per launch /
                 average launch time against the gfx950 peak;
  cpu_baseline — the CPU oracle (PyTorch restatement of the reference model, kind "port") timed on this box's host
                 cores on a bounded sample (batch-1 volumes of the same size), rank 0, N == 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver of this pool only supports dmabuf IPC: without this RCCL's multi-process setup fails with
# `hipIpcGetMemHandle: invalid argument` (normally already exported by the environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

PEAK_F32_TFLOPS = 157.3  # MI355X dense fp32 (vector == fp32-input MFMA), MI355X_MICROARCH.md
PEAK_BF16_TFLOPS = 2516.0  # MI355X dense bf16 MFMA
PEAK_HBM_GBS = 8000.0    # HBM3E spec
SHAPE = (160, 192, 160)
PER_GPU_BATCH = 2
C0 = 8


def build_model(device):
    import torch
    from mri_epilepsy_diagnosis_amd.unet import UNet
    torch.manual_seed(0)
    return UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=C0,
                normalization="batch", upsampling_type="linear", padding=True, activation="PReLU").to(device)


def cpu_baseline(max_seconds=25.0):
    """Oracle (CPU restatement of the reference model) fwd+bwd+AdamW on batch-1 volumes of the bench size."""
    import torch
    from oracle import losses, unet_recon
    torch.manual_seed(0)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))  # the GPU box's CPU share for one GPU is 16 cores; more threads only oversubscribe
    torch.set_num_threads(threads)
    m = unet_recon.UNetRecon(out_channels_first_layer=C0)
    opt = torch.optim.AdamW(m.parameters())
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(1, 1, *SHAPE, generator=g)
    t = (torch.rand(1, 1, *SHAPE, generator=g) < 0.1).float()
    times = []
    t_start = time.perf_counter()
    for it in range(3):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = losses.softmax_dice_loss(m(x), t)
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if it > 0:
            times.append(dt)
        if time.perf_counter() - t_start > max_seconds and times:
            break
    if not times:
        times = [dt]
    sec = sum(times) / len(times)
    return {"value": 1.0 / sec, "unit": "volumes/s", "cores": threads, "kind": "port",
            "sample": "%d timed fwd+bwd+AdamW iterations (after 1 warm-up) of the CPU oracle UNetRecon(c0=%d) on 1 volume "
                      "of 1x%dx%dx%d fp32, torch %s, %d threads" % (len(times), C0, *SHAPE, torch.__version__, threads)}


# classification/train_ENC_CLF.ipynb cell 17 ("93_6_4" checkpoints): the constructor arguments BASELINE configs[2] names
AE_KWARGS_93_6_4 = dict(
    c_in=1, is_skip=False, deapth=3, c_base=8, inc_size=2, reduce_size=False,
    down_block_kwargs=dict(conv_k=6, conv_pad=2, conv_s=2, maxpool_k=2, maxpool_s=2, batch_norm=True, act="l_relu"),
    up_block_kwargs=dict(up="upsample", scale=4, scale_mode="nearest", conv_k=3, conv_pad=1, conv_s=1, batch_norm=False,
                         act="l_relu"))
CLF_KWARGS = dict(c_in=32, c_out=64, conv_k=3, conv_s=1, conv_pad=1, l_in=64 * 2 * 3 * 2, l_out=32, batch_norm=True, act="relu",
                  p_drop=0.5, n_class=2)   # conv_pad=1 / l_in=768: the shipped pad-0 head only fits 192^3


def secondary_benchmarks(device, steps=10, warmup=3):
    """The other BASELINE.json configurations on this GPU, measured AFTER the headline's timed region (never inside it): each one
    a training step (zero_grad + forward + loss + backward replayed as one hipGraph, then the fused flat Adam), `warmup` untimed
    and `steps` timed replays between two device synchronisations.  `roofline_ms` is SURVEY §8d / BASELINE.md §4's per-unit
    roofline time x the units of one step; `frac` = roofline_ms / ms_per_step.  Returns {name: {...}}; a configuration that
    fails is reported with its error instead of a number (the headline line is never lost to a secondary)."""
    import torch
    import torch.nn.functional as F
    from mri_epilepsy_diagnosis_amd import ops, parallel
    from mri_epilepsy_diagnosis_amd.classification.models import AE_model, cnn_model
    from mri_epilepsy_diagnosis_amd.segmentation.models.modified_3dunet import Modified3DUNet

    def randn(*shape, seed=0):
        return torch.randn(*shape, device=device, generator=torch.Generator(device=device).manual_seed(seed))

    def cfg4():
        m = build_model(device).train()
        x = randn(PER_GPU_BATCH, 1, *SHAPE, seed=1234)
        t = (torch.rand(PER_GPU_BATCH, 1, *SHAPE, device=device) < 0.1).float()

        def loss():
            with ops.autocast():
                return ops.softmax_dice_loss(m(x), t)
        return m, loss, dict(lr=1e-3, weight_decay=1e-2, decoupled=True), PER_GPU_BATCH, "volumes/s", 2 * 1.98, \
            "cfg4 per-GPU share: unet.UNet(c0=8) bf16 region, batch 2 x 1x160x192x160, soft-Dice + AdamW (configs[3])"

    def cfg3_ae():
        m = AE_model.AE(**AE_KWARGS_93_6_4).to(device).train()
        x = randn(4, 1, *SHAPE, seed=3)
        return m, (lambda: F.mse_loss(m(x), x)), dict(lr=1e-3, weight_decay=0.0, decoupled=False), 4, "volumes/s", 4 * 0.277, \
            "cfg3 full autoencoder AE(**93_6_4) MSE step, batch 4 x 1x160x192x160 fp32 (configs[2])"

    def cfg3_enc():
        enc = AE_model.AE(**AE_KWARGS_93_6_4).enc.to(device)
        clf = AE_model.Classificator(**CLF_KWARGS).to(device)
        m = torch.nn.ModuleList([enc, clf]).train()
        x = randn(4, 1, *SHAPE, seed=3)
        y = torch.randint(0, 2, (4,), device=device)
        return m, (lambda: F.cross_entropy(clf(enc(x)[0]), y)), dict(lr=7e-4, weight_decay=1e-4, decoupled=False), 4, \
            "volumes/s", 4 * 0.105, "cfg3 encoder + classifier head CE step, batch 4 x 1x160x192x160 fp32 (configs[2])"

    def cfg5():
        m = torch.nn.Sequential(cnn_model.CNN(input_shape=(32, 32, 32), n_filters=16, n_blocks=3),
                                torch.nn.Linear(128, 2)).to(device).train()
        x = randn(512, 1, 32, 32, 32, seed=5)
        y = torch.randint(0, 2, (512,), device=device)
        return m, (lambda: F.cross_entropy(m(x), y)), dict(lr=1e-5, weight_decay=0.01, decoupled=False), 512, "patches/s", \
            512 * 0.0286, "cfg5 stand-in: CNN(32^3, 16 filters, 3 blocks) + Linear CE step, batch 512 fp32 (configs[4])"

    def m3d():
        m = Modified3DUNet(1, 2, 8).to(device).train()
        x = randn(1, 1, *SHAPE, seed=6)
        t = (torch.rand(1, 1, *SHAPE, device=device) < 0.1).float()
        return m, (lambda: ops.softmax_dice_loss(m(x), t)), dict(lr=1e-3, weight_decay=1e-2, decoupled=True), 1, "volumes/s", \
            7.37, "Modified3DUNet(1,2,8) soft-Dice + AdamW step, batch 1 x 1x160x192x160 fp32 (SURVEY a10)"

    out = {}
    for name, make in (("cfg4_bf16", cfg4), ("cfg3_autoencoder", cfg3_ae), ("cfg3_encoder_head", cfg3_enc), ("cfg5_patch_cnn", cfg5),
                       ("modified3dunet", m3d)):
        cap = None
        try:
            torch.manual_seed(0)
            model, loss_fn, okw, units, unit, roof_ms, what = make()
            flat = parallel.FlatParams(model)
            opt = parallel.FlatAdam(flat, **okw)
            cap = parallel.CapturedStep(flat, loss_fn).capture()

            def step():
                cap.run()
                opt.step(flat.all_reduce())
            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            lv = cap.loss.item()
            out[name] = {"workload": what, "ms_per_step": round(ms, 3), "value": round(units / ms * 1e3, 2), "unit": unit,
                         "steps": steps, "warmup": warmup, "roofline_ms": round(roof_ms, 3), "frac": round(roof_ms / ms, 4),
                         "launch": "hipGraph(fwd+bwd) + AdamW", "final_loss": round(lv, 6)}
            del model, loss_fn, flat, opt, step
        except Exception as e:  # noqa: BLE001 — reported in the line, the headline survives
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.synchronize()
        finally:
            if cap is not None:
                cap.release()
            cap = None
            import gc
            gc.collect()
            torch.cuda.empty_cache()
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` (N > 1) without a torchrun environment: this process becomes a pure launcher.  It has not
    imported torch and never touches the GPU (no os.exec of a GPU process either): it starts
    `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` as a CHILD, one fresh process per rank,
    relays rank 0's single JSON line, and exits non-zero if any rank failed or no line came back."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:     # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MRI3D_BENCH_LAUNCHED_BY_PARENT="1")
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, text=True)
    lines = []
    for ln in proc.stdout:
        if ln.startswith("{"):
            lines.append(ln.strip())
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if rc != 0:
        raise SystemExit("bench.py: the %d-rank run failed (torch.distributed.run exit code %d)" % (n, rc))
    if len(lines) != 1:
        raise SystemExit("bench.py: expected ONE JSON line from rank 0 of the %d-rank run, got %d" % (n, len(lines)))
    out = json.loads(lines[0])
    if out.get("n_gpus") != n or out.get("ranks_seen") != n:
        raise SystemExit("bench.py: asked for %d ranks, the run saw n_gpus=%r ranks_seen=%r"
                         % (n, out.get("n_gpus"), out.get("ranks_seen")))
    print(lines[0], flush=True)


def rank_plan(rank, local_rank, rehearsal=False):
    """What a rank of the N-rank run uses: its device (cuda:LOCAL_RANK — one process per GPU; cuda:0 for every rank in the
    one-GPU rehearsal) and the seed of its synthetic volumes (1234 + rank: every rank trains on different data).
    The single place both main() and --launch-check read it from, so the CPU suite checks the mapping the real run uses."""
    return {"rank": int(rank), "local_rank": int(local_rank), "device_index": 0 if rehearsal else int(local_rank),
            "data_seed": 1234 + int(rank)}


def launch_check(args):
    """--launch-check: the N-rank plumbing only (rendezvous, barrier, MAX all-reduce, one JSON line from rank 0), no model
    and no kernels — what the CPU suite can drive without a GPU (gloo).  Its line carries "launch_check": true and no value."""
    import torch
    import torch.distributed as dist
    from mri_epilepsy_diagnosis_amd import parallel
    backend = "nccl" if (torch.cuda.is_available() and os.environ.get("MRI3D_BENCH_ONE_GPU_REHEARSAL") != "1") else "gloo"
    rank, local, world = parallel.init_from_env(backend)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dev = torch.device("cuda", local) if backend == "nccl" else torch.device("cpu")
    seen = torch.ones(1, device=dev, dtype=torch.float64)
    tt = torch.tensor([float(rank + 1)], device=dev, dtype=torch.float64)
    plan = rank_plan(rank, local, os.environ.get("MRI3D_BENCH_ONE_GPU_REHEARSAL") == "1")
    plans = [plan]
    if world > 1:
        parallel.barrier(local if backend == "nccl" else None)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        mine = torch.tensor([plan["rank"], plan["local_rank"], plan["device_index"], plan["data_seed"]], device=dev, dtype=torch.int64)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        plans = [dict(zip(("rank", "local_rank", "device_index", "data_seed"), (int(v) for v in g.tolist()))) for g in got]
    if rank == 0:
        print(json.dumps({"launch_check": True, "metric": "MRI volumes/sec (fwd+bwd) 3D U-Net @160x192x160", "value": None,
                          "n_gpus": world, "ranks_seen": dist.get_world_size() if world > 1 else 1,
                          "ranks_counted_by_all_reduce": int(seen.item()), "max_over_ranks": tt.item(), "rank_plans": plans,
                          "backend": backend if world > 1 else None}), flush=True)
    if world > 1:
        parallel.barrier(local if backend == "nccl" else None)
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` object (the other BASELINE configurations, measured after the timed region)")
    ap.add_argument("--dtype", choices=("f32", "bf16"), default="f32")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="replay forward+backward as one captured hipGraph.  auto = off for f32 (GPU-bound, and the "
                         "per-kernel events of the roofline object need eager launches inside the timed region), on for "
                         "bf16 (host-bound when eager)")
    ap.add_argument("--launch-check", action="store_true",
                    help="exercise only the N-rank launch / rendezvous / JSON-line plumbing (no model, no kernels)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # N > 1 and no torchrun environment: become the launcher BEFORE torch is imported or any GPU call is made.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` (self-launching) or as "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, env_world))
    if args.launch_check:
        return launch_check(args)

    import torch
    from mri_epilepsy_diagnosis_amd import _lib, ops, parallel
    if not os.path.exists(_lib.LIB_PATH):
        raise SystemExit("libmri3d_hip.so is missing — run `python __graft_entry__.py` first (no fallback path exists)")
    # MRI3D_BENCH_ONE_GPU_REHEARSAL=1: every rank uses cuda:0 and the collectives go through gloo, so that the N > 1 code path
    # (launcher, barriers, max-over-ranks timing, the rank-0 JSON line) can be rehearsed on a one-GPU box; the value it prints
    # is meaningless (the ranks time-slice one device).  Never set by the driver: real runs use one GPU per rank over RCCL.
    rehearsal = os.environ.get("MRI3D_BENCH_ONE_GPU_REHEARSAL") == "1"
    # device_count() does not initialise the GPU: a rank never grabs a device that does not exist
    if not rehearsal and torch.cuda.device_count() < args.gpus:
        raise SystemExit("--gpus %d but only %d ROCm device(s) are visible" % (args.gpus, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device")

    backend = "gloo" if rehearsal else "nccl"
    rank, local, world = parallel.init_from_env(backend)
    assert world == args.gpus
    ranks_seen = torch.distributed.get_world_size() if world > 1 else 1
    if ranks_seen != args.gpus:
        raise SystemExit("--gpus %d but the process group has %d ranks" % (args.gpus, ranks_seen))
    plan = rank_plan(rank, local, rehearsal)
    device = torch.device("cuda", plan["device_index"])
    torch.cuda.set_device(device)

    model = build_model(device)
    flat = parallel.FlatParams(model)
    opt = parallel.FlatAdam(flat, lr=1e-3, weight_decay=1e-2, decoupled=True)  # torch.optim.AdamW defaults
    g = torch.Generator(device=device).manual_seed(plan["data_seed"])
    x = torch.randn(PER_GPU_BATCH, 1, *SHAPE, device=device, generator=g)
    t = (torch.rand(PER_GPU_BATCH, 1, *SHAPE, device=device, generator=g) < 0.1).float()
    model.train()

    bf16 = args.dtype == "bf16"
    peak_tflops = PEAK_BF16_TFLOPS if bf16 else PEAK_F32_TFLOPS

    def loss_fn():
        with ops.autocast(enabled=bf16):
            return ops.softmax_dice_loss(model(x), t)

    # forward + backward as ONE hipGraph launch (same kernels, same work); all-reduce and AdamW stay eager
    cap = parallel.CapturedStep(flat, loss_fn)
    graphed = False
    if args.graph == "on" or (args.graph == "auto" and bf16):
        try:
            cap.capture()
            graphed = True
        except Exception as e:  # noqa: BLE001 — capture support is a property of the runtime, not of the workload
            sys.stderr.write("hipGraph capture unavailable (%s: %s); running eagerly\n" % (type(e).__name__, e))
            cap.graph = None
            torch.cuda.synchronize()

    def step():
        loss = cap.run()
        opt.step(flat.all_reduce())
        return loss

    def barrier():
        parallel.barrier(None if rehearsal else local)
        torch.cuda.synchronize()

    # Eager launches (default, f32).  The per-operator table comes from the last (up to two) WARM-UP steps, where every C-ABI call
    # is bracketed by two events on the launch stream; inside the TIMED region only the dominant operator found there (the
    # `roofline` kernel) is bracketed — 2 events per step instead of ~600, whose timestamp packets between the kernels cost the
    # step 1 % (31.1 against 30.8 ms).  With no warm-up step every operator is bracketed in the timed region, as before.
    # Graph replay cannot be bracketed from the host: there the largest layer is timed right after the timed region.
    table_steps = 0 if graphed else min(2, args.warmup)
    # two more warm-up steps run bracketed with a timer that is thrown away: the first few hundred timing events of a process make
    # the HIP runtime grow its signal pool once — a 55-77 ms host stall inside whatever call is running (tools/host_gap_probe.py)
    # that must not land in the table the dominant operator is chosen from
    discard_steps = 0 if graphed else min(2, args.warmup - table_steps)
    for _ in range(args.warmup - table_steps - discard_steps):
        step()
    if discard_steps:
        ops.set_timer(ops.KernelTimer())
        for _ in range(discard_steps):
            step()
        ops.set_timer(None)
    table = None
    if table_steps and rank == 0:
        torch.cuda.synchronize()
        wt = ops.KernelTimer()
        ops.set_timer(wt)
        for _ in range(table_steps):
            step()
        ops.set_timer(None)
        table = wt.summary()
    else:
        for _ in range(table_steps):
            step()
    barrier()
    only = None
    if table:
        only = {max(table.items(), key=lambda kv: kv[1]["ms"])[0]}
    timer = ops.KernelTimer(only=only) if rank == 0 else None
    if not graphed:
        ops.set_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ops.set_timer(None)
    probe_steps = args.steps
    if graphed and rank == 0:
        # kernels inside a replayed graph cannot be bracketed from the host, and an eager re-run is host-bound at bf16
        # speeds: time the three passes of the model's largest conv layer (decoder 48->16 at full resolution, 42 % of the
        # step's FLOPs) back to back between two events on the launch stream instead.
        conv = model.decoder.decoding_blocks[-1].conv1.conv_layer
        act_dt = torch.bfloat16 if bf16 else torch.float32
        c_skip = conv.in_channels // 3        # cat((skip, upsampled)): C + 2C channels, read from two dense tensors (ops.conv3d_cat)
        CL = torch.channels_last_3d
        xs_ = torch.randn(PER_GPU_BATCH, c_skip, *SHAPE, device=device).to(act_dt).contiguous(memory_format=CL).requires_grad_(True)
        xu_ = torch.randn(PER_GPU_BATCH, conv.in_channels - c_skip, *SHAPE, device=device).to(act_dt).contiguous(memory_format=CL).requires_grad_(True)
        dys_ = torch.randn(PER_GPU_BATCH, conv.out_channels, *SHAPE, device=device).to(act_dt).contiguous(memory_format=CL)
        wq_, bq_ = conv.weight.detach().clone().requires_grad_(True), conv.bias.detach().clone().requires_grad_(True)

        def layer_passes():
            ops.conv3d_cat(xs_, xu_, wq_, bq_, padding=1).backward(dys_)

        probe_steps = 5
        layer_passes()
        torch.cuda.synchronize()
        ops.set_timer(timer)
        for _ in range(probe_steps):
            layer_passes()
        ops.set_timer(None)
        torch.cuda.synchronize()
        del xs_, xu_, dys_
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = tt.item()
    final_loss = loss.item()

    if rank == 0:
        agg = timer.summary()
        dom_tag, dom = max(agg.items(), key=lambda kv: kv[1]["ms"])   # (timed region: the one bracketed operator, or all of them)
        if table:   # the table and the operator's share of a step come from the warm-up steps
            total_ms = sum(a["ms"] for a in table.values())
            dom_share = table[dom_tag]["ms"] / total_ms
            agg, probe_steps = table, table_steps
        else:
            total_ms = sum(a["ms"] for a in agg.values())
            dom_share = dom["ms"] / total_ms
        avg_s = dom["ms"] / dom["calls"] / 1e3
        work = dom["work"] or {"flops": 0.0, "bytes": 0.0}
        ai = work["flops"] / max(work["bytes"], 1.0)
        if ai > peak_tflops * 1e12 / (PEAK_HBM_GBS * 1e9):
            achieved, peak, unit, bound = work["flops"] / avg_s / 1e12, peak_tflops, "TFLOP/s", "mfma"
        else:
            achieved, peak, unit, bound = work["bytes"] / avg_s / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
        # HBM-side traffic per launch of the dominant kernel.  PMC counters cannot be read from inside this process, so this is
        # NOT a measurement of this run: it is the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE record (separate passes,
        # tools/pmc_conv.sh, round 3) of the same kernel on the same geometry, with FETCH_SIZE x 2 as calibrated on this library's
        # access shapes (profiles/r02_fetch_size_calibration.txt).  `traffic_source` says so in the line itself; null if the
        # dominant operator has no record.
        traffic, traffic_source = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "r03_hbm_traffic_48_16.json")) as f:
                rec = json.load(f).get(dom_tag.replace(" +bn-stats", ""))
            if rec:
                traffic = round(rec["traffic_bytes"])
                traffic_source = "profiles/r03_hbm_traffic_48_16.json (rocprofv3 --pmc, separate run of this kernel and geometry; 2 x FETCH_SIZE + WRITE_SIZE)"
        except (OSError, ValueError):
            pass
        roofline = {"bound": bound, "achieved": round(achieved, 3), "peak": peak, "unit": unit,
                    "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_source, "kernel": dom_tag,
                    "avg_launch_ms": round(avg_s * 1e3, 4), "launches_timed": dom["calls"],
                    "share_of_timed_ops": round(dom_share, 4)}
        top = sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("MRI3D_BENCH_TOP", "12"))]
        sys.stderr.write("per-operator device time over %d %s (events on the launch stream):\n"
                         % (probe_steps, "back-to-back launches of the largest conv layer" if graphed else
                            ("warm-up steps; the timed steps bracket only the first line's operator" if table else "timed steps")))
        for tag, a in top:
            w = a["work"] or {}
            tf = (w.get("flops", 0) * a["calls"] / (a["ms"] / 1e3) / 1e12) if a["ms"] > 0 else 0
            gb = (w.get("bytes", 0) * a["calls"] / (a["ms"] / 1e3) / 1e9) if a["ms"] > 0 else 0
            sys.stderr.write("  %8.2f ms %5.1f%%  %7.2f TF/s %8.1f GB/s  x%-4d %s\n"
                             % (a["ms"], 100 * a["ms"] / total_ms, tf, gb, a["calls"], tag))
        if graphed:
            sys.stderr.write("  wall %.2f ms/step (hipGraph replay)\n" % (elapsed * 1e3 / args.steps))
        else:
            sys.stderr.write("  bracketed operators: %.2f ms/step of kernels%s; timed wall %.2f ms/step (eager)\n"
                             % (total_ms / probe_steps, " (warm-up steps, every operator bracketed)" if table else "",
                                elapsed * 1e3 / args.steps))
        out = {
            "metric": "MRI volumes/sec (fwd+bwd) 3D U-Net @160x192x160",
            "value": round(world * PER_GPU_BATCH * args.steps / elapsed, 4),
            "unit": "volumes/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "backend": ("rccl(nccl)" if backend == "nccl" else backend) if world > 1 else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": "unet.UNet(c0=8, 3 enc blocks, BN+PReLU, trilinear) fwd+softmax-Dice+bwd+AdamW, "
                                   + ("batch 2 x 1x160x192x160 per GPU, bf16 activations / fp32 accumulate+master weights "
                                      "(BASELINE configs[3] per-GPU share)" if bf16 else
                                      "batch 2 x 1x160x192x160 fp32 per GPU (BASELINE configs[1])"),
                       "global_batch": world * PER_GPU_BATCH, "volume": list(SHAPE),
                       "parallelism": "dp%d" % world, "final_loss": round(final_loss, 6),
                       "launch": "hipGraph(fwd+bwd) + all-reduce + AdamW" if graphed else "eager"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and not bf16:
            out["cpu_baseline"] = cpu_baseline()
        if world == 1 and not args.no_secondary and not bf16:
            # the timed region is over and `value` is final: the other configurations run now, on the same device, one by one
            del x, t
            out["secondary"] = secondary_benchmarks(device)
        print(json.dumps(out), flush=True)
    if world > 1:
        parallel.barrier(None if rehearsal else local)
        torch.distributed.destroy_process_group()
    # Explicit teardown while the HIP runtime is alive, then an ordinary exit (round 2 left through os._exit here): the captured
    # graph and its private pool, the event pairs of the operator timers and the scratch buffers go now, not in whatever order
    # interpreter finalisation finds them.
    del timer, table
    cap.release()
    parallel.release_captured_graphs()
    ops.release_workspaces()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
