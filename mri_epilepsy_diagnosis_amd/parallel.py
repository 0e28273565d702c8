"""Data-parallel plumbing for one-process-per-GPU training over RCCL/xGMI (torch.distributed backend "nccl").

The reference has no distributed code at all; volumes are independent, so the path shards as pure
data parallelism: every rank runs the full model on its own volumes and the only exchange is ONE all-reduce(sum) of
a flat gradient buffer per step (0.99 MB for the c0=8 U-Net: latency-bound, so one collective, no bucketing), with
the 1/world_size averaging folded into the fused optimizer step (`FlatAdam`).

`FlatParams` re-homes every parameter (and its .grad) as a view into one contiguous fp32 buffer, so the collective
and the optimizer each touch a single allocation.  BatchNorm statistics stay local to each rank by default (the
per-GPU batch is what the reference's single-GPU batch was); this is stated in DESIGN.md.
"""
import atexit
import ctypes
import gc
import weakref

import torch
import torch.distributed as dist

from . import _lib, ops

# Every object that owns a captured hipGraph (StepCache entries, CapturedStep) is registered here so that the graphs — and the
# private allocator pools their activations live in — can be destroyed EXPLICITLY while the HIP runtime is alive
# (`release_captured_graphs`, also run as an atexit hook).  A StepCache hangs on its model and refers back to it: such a cycle
# is only collected by the cyclic GC, which for a model that lives until the end of the program means interpreter finalisation
# — after which the order of graph-exec destruction against the runtime's own static teardown is nobody's to choose
# ("teardown").
_graph_owners = weakref.WeakSet()


def release_captured_graphs():
    """Destroy every captured hipGraph of this process now (StepCache entries and CapturedStep graphs), on a quiet device.
    The owners stay usable: a StepCache re-captures on its next run, a CapturedStep falls back to eager launches until
    `capture()` is called again.  Idempotent; registered with `atexit` so that no graph survives into interpreter finalisation."""
    owners = list(_graph_owners)
    if not owners:
        return
    cuda_up = torch.cuda.is_available() and torch.cuda.is_initialized()
    if cuda_up:
        torch.cuda.synchronize()
    for o in owners:
        o.release()
    del owners
    gc.collect()
    if cuda_up:
        torch.cuda.synchronize()


atexit.register(release_captured_graphs)


class FlatParams:
    """Flatten a module's trainable parameters and gradients into two contiguous buffers (views stay live).

    While attached, the HIP ops write parameter gradients straight into `self.grad` (ops.GradSink — its docstring states the
    contract: zero through `zero_grad()`; wrap `torch.autograd.grad(...)` / `backward(inputs=...)` over attached parameters in
    `with ops.suspend_grad_sinks():`)."""

    def __init__(self, module):
        seen, params = set(), []
        for p in module.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append(p)
        if not params:
            raise ValueError("module has no trainable parameters")
        dev, dt = params[0].device, params[0].dtype
        total = sum(p.numel() for p in params)
        self.params = params
        self.flat = torch.empty(total, device=dev, dtype=dt)
        self.grad = torch.zeros(total, device=dev, dtype=dt)
        off = 0
        for p in params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
            # the ops' backward writes this parameter's gradient straight into the view (ops.GradSink): no AccumulateGrad add
            p._mri3d_grad_sink = ops.GradSink(p.grad)
            off += n

    def zero_grad(self):
        self.grad.zero_()
        off = 0
        for p in self.params:  # re-attach views if a caller dropped them (optimizer.zero_grad(set_to_none=True))
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + off * self.grad.element_size():
                p.grad = self.grad[off:off + n].view(p.shape)
            sink = getattr(p, "_mri3d_grad_sink", None)
            if sink is None or sink.view.data_ptr() != p.grad.data_ptr():
                p._mri3d_grad_sink = sink = ops.GradSink(p.grad)
            sink.fresh = True      # zeroed: the next gradient of this parameter is written, not added
            off += n

    def all_reduce(self, group=None):
        """Sum gradients over ranks in one collective.  Returns the factor the optimizer must scale by."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            return 1.0 / dist.get_world_size(group)
        return 1.0


class SyncBatchNorm:
    """Synchronised BatchNorm statistics over the data-parallel ranks (SURVEY §8e "parity mode"; the default is local
    statistics): inside `with parallel.SyncBatchNorm(group):` every training-mode BatchNorm3d of the HIP path sums its
    (count, sum x, sum x^2) forward and (sum dy, sum dy*xhat) backward over the ranks, so world_size ranks x B volumes see the
    statistics one device would see with world_size*B volumes.  Two float64 all-reduces of 2C(+1) numbers per layer and step."""

    def __init__(self, group=None):
        self.group = group
        self._prev = None

    def all_reduce(self, t):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def __enter__(self):
        self._prev = ops.set_sync_batchnorm(self)
        return self

    def __exit__(self, *exc):
        ops.set_sync_batchnorm(self._prev)
        return False


class FlatAdam:
    """Adam / AdamW over a FlatParams buffer in ONE HIP kernel (mri3d_adam_step).  Matches torch.optim.AdamW
    (decoupled=True; segmentation/routine.py:358) or torch.optim.Adam with L2 weight_decay (decoupled=False;
    classification/routine.py:271) single-tensor arithmetic."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, decoupled=True):
        self.flat = flat
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, params=flat.params)]
        self.decoupled = decoupled
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self, grad_scale=1.0):
        if not self.flat.flat.is_cuda:
            raise RuntimeError("FlatAdam runs on the HIP kernel only; parameters must live on a ROCm device")
        g = self.param_groups[0]
        self.step_count += 1
        L = _lib.lib()
        p = ctypes.c_void_p
        _lib.check(L.mri3d_adam_step(p(self.flat.flat.data_ptr()), p(self.flat.grad.data_ptr()), p(self.m.data_ptr()),
                                     p(self.v.data_ptr()), self.flat.flat.numel(), float(g["lr"]), float(g["betas"][0]),
                                     float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self.step_count,
                                     float(grad_scale), 1 if self.decoupled else 0,
                                     p(torch.cuda.current_stream().cuda_stream)), "adam_step")


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, local_rank, world_size).  No-op (0, 0, 1) when WORLD_SIZE is unset or 1."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            # RCCL: bind the rank to its device BEFORE the group exists and hand the device to the group, so that the
            # communicator is created eagerly on that device (no lazy first-collective guess, no barrier() on "the current
            # device" of a rank that has not selected one yet)
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def barrier(local_rank=None, group=None):
    """dist.barrier() that names the rank's device for the RCCL backend (a device-less NCCL barrier picks a device by rank
    modulo device count and warns / can hang when that is not the rank's own); a no-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) <= 1:
        return
    if dist.get_backend(group) == "nccl":
        dev = torch.cuda.current_device() if local_rank is None else int(local_rank)
        dist.barrier(group=group, device_ids=[dev])
    else:
        dist.barrier(group=group)


def shard_range(n_items, rank, world):
    """Contiguous [lo, hi) slice of n_items for `rank` (independent volumes: no data-path collective)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class CapturedStep:
    """`zero_grad -> loss = loss_fn() -> loss.backward()` recorded ONCE into a hipGraph and replayed per step.

    A training step of the U-Net is ~300 kernel launches issued from Python autograd; at bf16 speeds the host becomes the
    bound (20 % of the wall time measured as launch gaps).  All shapes are static, every kernel of this library is
    enqueued on torch's current stream and all memory comes from torch's caching allocator, so the whole forward +
    backward can be stream-captured (torch.cuda.graph == hipStreamBeginCapture) and replayed with one launch.  The
    gradient all-reduce and the optimizer step stay outside the graph (the collective is not captured, and the Adam
    bias correction takes the host-side step count).

    Usage:   cap = CapturedStep(flat, lambda: ops.softmax_dice_loss(model(x), t));  loss = cap.run()
    The inputs referenced by `loss_fn` must be static tensors (update them in place between steps).
    """

    def __init__(self, flat, loss_fn, warmup=3):
        self.flat = flat
        self.loss_fn = loss_fn
        self.graph = None
        self.loss = None
        self.warmup = warmup
        _graph_owners.add(self)

    def release(self):
        """Drop the captured graph (and the static loss tensor that lives in its pool); `run()` launches eagerly afterwards."""
        self.graph = None
        self.loss = None

    reset = release

    def _eager(self):
        self.flat.zero_grad()
        loss = self.loss_fn()
        loss.backward()
        return loss

    def capture(self):
        # warm-up on a side stream (allocator pools, lazy workspace growth, hipFuncSetAttribute) as torch recommends
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self.warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.loss = self._eager()
        self.graph = g
        return self

    def run(self):
        if self.graph is None:
            return self._eager()
        self.graph.replay()
        return self.loss


def _ensure_attached(flat):
    """Make every parameter's .grad the view into flat.grad again (a caller's optimizer.zero_grad(set_to_none=True) or an eager
    step in between drops / replaces them) WITHOUT zeroing: a graph replay has just written the gradients there."""
    off = 0
    for p in flat.params:
        n = p.numel()
        if p.grad is None or p.grad.data_ptr() != flat.grad.data_ptr() + off * flat.grad.element_size():
            p.grad = flat.grad[off:off + n].view(p.shape)
        sink = getattr(p, "_mri3d_grad_sink", None)
        if sink is None or sink.view.data_ptr() != p.grad.data_ptr():
            p._mri3d_grad_sink = ops.GradSink(p.grad)
        off += n


class StepCache:
    """What makes the drop-in loops fast by default (segmentation/routine.py::run_epoch, classification/routine.py::
    run_one_epoch): per model, a cache of hipGraph-captured `zero_grad -> forward -> loss [-> backward]` sequences keyed by the
    input signature.  The first batch of a new shape is captured (its warm-up runs are undone: module buffers and the RNG state
    are restored, so BatchNorm running statistics / num_batches_tracked advance exactly once per batch, as in the reference
    loop); later batches copy into the static inputs and replay.  Anything that cannot be captured (CPU modules or tensors — the
    oracle-driven CPU tests —, a capture error) runs eagerly: same kernels, same results, only more launch overhead.

        cache = StepCache.of(model)
        out, loss = cache.run(inputs, targets, loss_fn, backward=True)     # loss_fn(outputs, targets) -> scalar tensor
        optimizer.step()                                                    # gradients are in p.grad (views of one flat buffer)

    `out` and `loss` are static tensors, valid until the next run() with the same signature."""

    MAX_ENTRIES = 8   # captured graphs kept per model (least recently used one is dropped: each holds its activations)

    def __init__(self, model):
        self.model = model
        self.flat = None
        self.entries = {}
        self.disabled = False
        self.replays = self.captures = self.eager_runs = 0
        _graph_owners.add(self)

    def clear(self):
        """Drop every captured entry (graphs, static inputs / outputs, their allocator pools); the next run() re-captures."""
        self.entries.clear()

    release = clear

    @staticmethod
    def of(model):
        c = getattr(model, "_mri3d_step_cache", None)
        if c is None:
            c = StepCache(model)
            object.__setattr__(model, "_mri3d_step_cache", c)   # plain attribute: not a sub-module, not in the state_dict
        return c

    def _capturable(self, inputs, targets):
        if self.disabled or not torch.cuda.is_available() or not inputs.is_cuda or (targets is not None and not targets.is_cuda):
            return False
        ps = list(self.model.parameters())
        return bool(ps) and all(p.is_cuda for p in ps)

    def _signature(self, inputs, targets, loss_fn, backward):
        m = self.model
        modes = tuple(mod.training for mod in m.modules())
        grads = tuple(p.requires_grad for p in m.parameters())
        if isinstance(loss_fn, torch.nn.Module) and not any(True for _ in loss_fn.parameters()) and not any(True for _ in loss_fn.buffers()):
            loss_key = (type(loss_fn), repr(loss_fn))     # e.g. a fresh nn.CrossEntropyLoss() per train() call is the same loss
        else:
            loss_key = loss_fn
        return (tuple(inputs.shape), inputs.dtype, None if targets is None else (tuple(targets.shape), targets.dtype),
                loss_key, bool(backward), torch.is_grad_enabled(), ops.autocast_dtype(), hash(modes), hash(grads),
                inputs.device.index)

    def _eager(self, inputs, targets, loss_fn, backward):
        self.eager_runs += 1
        out = self.model(inputs)
        loss = loss_fn(out, targets) if loss_fn is not None else None
        if backward:
            loss.backward()
        return out, loss

    def run(self, inputs, targets=None, loss_fn=None, backward=False):
        if not self._capturable(inputs, targets):
            return self._eager(inputs, targets, loss_fn, backward)
        key = self._signature(inputs, targets, loss_fn, backward)
        e = self.entries.get(key)
        if e is None:
            try:
                e = self._capture(inputs, targets, loss_fn, backward)
            except Exception as exc:  # noqa: BLE001 — capture support is a property of the runtime / the model, not of the data
                import warnings
                warnings.warn("mri3d: hipGraph capture of the training step failed (%s: %s); running eagerly from now on"
                              % (type(exc).__name__, exc))
                self.disabled = True
                torch.cuda.synchronize()
                if backward:
                    for p in self.model.parameters():
                        p.grad = None
                return self._eager(inputs, targets, loss_fn, backward)
            if len(self.entries) >= self.MAX_ENTRIES:
                del self.entries[next(iter(self.entries))]
            self.entries[key] = e
        else:
            self.entries[key] = self.entries.pop(key)     # most recently used last
        e["x"].copy_(inputs)
        if targets is not None:
            e["t"].copy_(targets)
        e["graph"].replay()
        self.replays += 1
        if backward:
            _ensure_attached(self.flat)
        return e["out"], e["loss"]

    def _capture(self, inputs, targets, loss_fn, backward):
        m = self.model
        if backward and self.flat is None:
            self.flat = FlatParams(m)
        x = inputs.clone()
        t = targets.clone() if targets is not None else None
        grad_mode = torch.is_grad_enabled()

        def body():
            if backward:
                self.flat.zero_grad()
            with torch.set_grad_enabled(grad_mode):
                out = m(x)
                loss = loss_fn(out, t) if loss_fn is not None else None
                if backward:
                    loss.backward()
            return out, loss

        # the warm-up runs (allocator pools, lazy workspaces) must leave no trace: buffers (BatchNorm running statistics,
        # num_batches_tracked) and the generator state are put back before the graph's first replay handles this batch
        bufs = [b.detach().clone() for b in m.buffers()]
        rng = torch.cuda.get_rng_state(inputs.device)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                body()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out, loss = body()
        with torch.no_grad():
            for b, saved in zip(m.buffers(), bufs):
                b.copy_(saved)
        torch.cuda.set_rng_state(rng, inputs.device)
        self.captures += 1
        return {"graph": g, "x": x, "t": t, "out": out, "loss": loss}
