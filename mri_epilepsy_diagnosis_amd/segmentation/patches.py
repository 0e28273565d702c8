"""Patch pipeline on the device — SURVEY §8 row f3: the windows TorchIO cuts either side of the training path.

Reference call sites (the arithmetic itself is TorchIO's, absent here: "parity unpinned", see oracle/patches.py):
  * training on random 64^3 patches: `torchio.Queue(subjects_dataset, max_length, samples_per_volume, patch_size,
    sampler_class=torchio.sampler.ImageSampler, shuffle_subjects, shuffle_patches)` wrapped in a DataLoader
    (segmentation/routine.py:150-178; segmentation/pretraining_3d_unet.ipynb cell 24) -> `Queue` / `ImageSampler` below;
  * whole-volume inference on a grid of overlapping 64^3 windows: `GridSampler(sample, patch_size, patch_overlap)`,
    `GridAggregator(sample, patch_overlap).add_batch(labels, locations)`, `.get_output_tensor()`
    (pretraining_3d_unet.ipynb cell 26) -> `GridSampler` / `GridAggregator` below.

MI355X-first: the subject volumes stay resident in HBM (a 160x192x160 fp32 volume is 19.7 MB; 288 GB hold thousands), a
whole batch of windows is cut by ONE `mri3d_extract_patches` launch per image (origins travel as kernel arguments), and the
predicted label windows are written back by `mri3d_aggregate_patches_{u8,argmax}`, the second taking the arg-max of the
logits in flight so no label tensor is materialised.  Window origins are host integers; there is no CPU data path.
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..ops import _ptr, _stream, _dt

MRI = "MRI"
LABEL = "LABEL"
DATA = "data"  # torchio.DATA
LOCATION = "location"


# ------------------------------------------------------------------ window arithmetic (host integers)
def _axis_starts(size, window, step):
    """Window origins along one axis: every `step` voxels while the window fits, then one flush with the far end; when that
    makes exactly two, a third at their rounded mean (the NiftyNet/TorchIO grid)."""
    size, window, step = max(int(size), 0), max(int(window), 1), max(int(step), 1)
    starts, s = [], 0
    while s + window <= size:
        starts.append(s)
        s += step
    last = max(size - window, 0)
    if last not in starts:
        starts.append(last)
    starts.sort()
    if len(starts) == 2:
        mid = int(np.round((starts[0] + starts[1]) / 2.0))  # round-half-even, as np.round
        if mid not in starts:
            starts.append(mid)  # appended after the sorted pair
    return starts


def grid_locations(shape, patch_size, patch_overlap):
    """int32 [n, 6] = (i0, j0, k0, i1, j1, k1) of the inference grid, in TorchIO's order (np.meshgrid 'xy' indexing of the
    three axes' origins: the second axis varies slowest, then the first, then the third)."""
    shape, patch, border = _triple(shape), _triple(patch_size), _triple(patch_overlap)
    for s, p in zip(shape, patch):
        if p > s:
            raise ValueError("patch %s larger than the volume %s" % (patch, shape))
    a, b, c = (_axis_starts(shape[i], patch[i], patch[i] - 2 * border[i]) for i in range(3))
    loc = np.empty((len(a) * len(b) * len(c), 6), dtype=np.int32)
    n = 0
    for j in b:
        for i in a:
            for k in c:
                loc[n] = (i, j, k, i + patch[0], j + patch[1], k + patch[2])
                n += 1
    return loc


def _triple(v):
    if isinstance(v, (int, np.integer)):
        return int(v), int(v), int(v)
    v = tuple(int(x) for x in v)
    if len(v) != 3:
        raise ValueError("expected 3 values, got %r" % (v,))
    return v


def _table(volume_index, locations):
    loc = np.asarray(locations)
    t = np.empty((loc.shape[0], 4), dtype=np.int32)
    t[:, 0] = volume_index
    t[:, 1:] = loc[:, :3]
    return np.ascontiguousarray(t)


# ------------------------------------------------------------------ device ops
def _require_volume(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s: needs a ROCm device tensor (got %s); there is no CPU fallback"
                           % (what, getattr(t, "device", type(t))))
    if t.element_size() not in (1, 2, 4, 8):
        raise RuntimeError("%s: unsupported element size %d" % (what, t.element_size()))


def extract_patches(volumes, table, patch_size):
    """volumes: contiguous device tensor (S, D, H, W) of any 1/2/4/8-byte dtype; table: int32 [P, 4] host array of
    (volume index, d0, h0, w0).  Returns (P, 1, pd, ph, pw) of the same dtype — one kernel launch per 64 windows."""
    _require_volume(volumes, "extract_patches")
    if volumes.dim() != 4 or not volumes.is_contiguous():
        raise RuntimeError("extract_patches: volumes must be a contiguous (S, D, H, W) tensor, got %s" % (tuple(volumes.shape),))
    table = np.ascontiguousarray(table, dtype=np.int32)
    if table.ndim != 2 or table.shape[1] != 4 or table.shape[0] == 0:
        raise ValueError("extract_patches: table must be [P>0, 4]")
    pd, ph, pw = _triple(patch_size)
    s, d, h, w = (int(v) for v in volumes.shape)
    out = torch.empty((table.shape[0], 1, pd, ph, pw), dtype=volumes.dtype, device=volumes.device)
    L = _lib.lib()
    _lib.check(L.mri3d_extract_patches(_ptr(volumes), volumes.element_size(), s, d, h, w,
                                       table.ctypes.data_as(ctypes.c_void_p), table.shape[0], pd, ph, pw, _ptr(out),
                                       _stream()), "extract_patches")
    return out


class GridSampler:
    """torchio.inference.GridSampler(sample, patch_size, patch_overlap): the windows of one subject, here cut on the device.
    `sample` is the subject dict {MRI: {DATA: (1, D, H, W) device tensor}, ...}; every entry that has a DATA tensor is cut."""

    def __init__(self, sample, patch_size, patch_overlap):
        self.sample = sample
        self.patch_size = _triple(patch_size)
        self.patch_overlap = _triple(patch_overlap)
        self._images = {k: v[DATA] for k, v in sample.items() if isinstance(v, dict) and DATA in v}
        if not self._images:
            raise ValueError("GridSampler: the sample holds no image")
        first = next(iter(self._images.values()))
        if first.dim() != 4 or first.shape[0] != 1:
            raise ValueError("GridSampler: images must be (1, D, H, W), got %s" % (tuple(first.shape),))
        self.shape = tuple(int(v) for v in first.shape[1:])
        self.locations = grid_locations(self.shape, self.patch_size, self.patch_overlap)
        self._resident = {k: v.contiguous() for k, v in self._images.items()}

    def __len__(self):
        return self.locations.shape[0]

    def _cut(self, idx):
        loc = self.locations[idx]
        table = _table(0, loc)
        out = {k: {DATA: extract_patches(v, table, self.patch_size)} for k, v in self._resident.items()}
        out[LOCATION] = torch.from_numpy(loc.astype(np.int64))
        return out

    def __getitem__(self, index):
        """One window, as the TorchIO dataset yields it to a DataLoader: {name: {DATA: (1, pd, ph, pw)}, 'location': (6,)}."""
        if not 0 <= index < len(self):
            raise IndexError(index)
        b = self._cut(slice(index, index + 1))
        return {k: ({DATA: v[DATA][0]} if k != LOCATION else v[0]) for k, v in b.items()}

    def batches(self, batch_size):
        """What DataLoader(grid_sampler, batch_size) yields, each batch cut by one launch per image."""
        for first in range(0, len(self), int(batch_size)):
            yield self._cut(slice(first, min(first + int(batch_size), len(self))))


class GridAggregator:
    """torchio.inference.GridAggregator(sample, patch_overlap): collects label windows into the whole volume (uint8 in HBM).
    Every window is cropped by the overlap on all six faces; later windows overwrite earlier ones; the outermost
    `overlap` voxels of the volume are never written and stay 0."""

    def __init__(self, sample, patch_overlap):
        first = next(v[DATA] for v in sample.values() if isinstance(v, dict) and DATA in v)
        if not first.is_cuda:
            raise RuntimeError("GridAggregator: the sample must live on the ROCm device; there is no CPU fallback")
        self.shape = tuple(int(v) for v in first.shape[1:])
        self.patch_overlap = _triple(patch_overlap)
        self._output = torch.zeros(self.shape, dtype=torch.uint8, device=first.device)

    def _loc(self, locations):
        loc = locations.cpu().numpy() if isinstance(locations, torch.Tensor) else np.asarray(locations)
        if loc.ndim != 2 or loc.shape[1] != 6:
            raise ValueError("locations must be [B, 6]")
        return loc

    def add_batch(self, windows, locations):
        """windows: (B, 1, pd, ph, pw) or (B, pd, ph, pw) integer label tensor on the device (e.g. ops.argmax_mask)."""
        _require_volume(windows, "GridAggregator.add_batch")
        loc = self._loc(locations)
        if windows.dim() == 5:
            if windows.shape[1] != 1:
                raise ValueError("add_batch takes single-channel label windows; pass logits to add_batch_logits")
            windows = windows[:, 0]
        if windows.dtype != torch.uint8:
            windows = windows.to(torch.uint8)
        windows = windows.contiguous()
        b, pd, ph, pw = (int(v) for v in windows.shape)
        self._check(loc, b, (pd, ph, pw))
        table = _table(0, loc)
        bd, bh, bw = self.patch_overlap
        d, h, w = self.shape
        L = _lib.lib()
        _lib.check(L.mri3d_aggregate_patches_u8(_ptr(windows), table.ctypes.data_as(ctypes.c_void_p), b, pd, ph, pw, bd, bh, bw,
                                                _ptr(self._output), 1, d, h, w, _stream()), "aggregate_patches_u8")

    def add_batch_logits(self, logits, locations):
        """`labels = logits.argmax(dim=1, keepdim=True); add_batch(labels, locations)` in one kernel: logits is the model's
        (B, C, pd, ph, pw) output in its native channels-last storage (fp32 or bf16)."""
        if not logits.is_cuda or logits.dim() != 5:
            raise RuntimeError("add_batch_logits: needs a (B, C, D, H, W) ROCm device tensor; there is no CPU fallback")
        if logits.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError("add_batch_logits: logits must be float32 or bfloat16, got %s" % logits.dtype)
        loc = self._loc(locations)
        x = logits.detach().permute(0, 2, 3, 4, 1)
        if not x.is_contiguous():
            x = x.contiguous()
        b, pd, ph, pw, c = (int(v) for v in x.shape)
        self._check(loc, b, (pd, ph, pw))
        table = _table(0, loc)
        bd, bh, bw = self.patch_overlap
        d, h, w = self.shape
        L = _lib.lib()
        _lib.check(L.mri3d_aggregate_patches_argmax(_ptr(x), c, c, _dt(x), table.ctypes.data_as(ctypes.c_void_p), b, pd, ph, pw,
                                                    bd, bh, bw, _ptr(self._output), 1, d, h, w, _stream()),
                   "aggregate_patches_argmax")

    def _check(self, loc, b, patch):
        if loc.shape[0] != b:
            raise ValueError("%d windows but %d locations" % (b, loc.shape[0]))
        if np.any(loc[:, 3:] - loc[:, :3] != np.asarray(patch)):
            raise ValueError("locations do not match the window size %s" % (patch,))

    def get_output_tensor(self, dtype=torch.float32):
        """(1, D, H, W) like TorchIO (float32 there); pass dtype=torch.uint8 to get the resident mask without a cast."""
        out = self._output if dtype == torch.uint8 else self._output.to(dtype)
        return out[None]


# ------------------------------------------------------------------ random training patches
class ImageSampler:
    """torchio.sampler.ImageSampler(sample, patch_size): an endless stream of uniformly random windows of one subject —
    origin per axis uniform in [0, size - patch].  Here it yields window ORIGINS; `Queue` cuts them in batches."""

    def __init__(self, shape, patch_size, rng):
        self.shape, self.patch_size, self.rng = _triple(shape), _triple(patch_size), rng
        for s, p in zip(self.shape, self.patch_size):
            if p > s:
                raise ValueError("patch %s larger than the volume %s" % (self.patch_size, self.shape))

    def __iter__(self):
        return self

    def draw(self, n):
        """int64 [n, 3] window origins in one call of the generator."""
        hi = np.asarray(self.shape, dtype=np.int64) - np.asarray(self.patch_size, dtype=np.int64) + 1
        return self.rng.integers(0, hi, size=(int(n), 3))

    def __next__(self):
        return tuple(int(v) for v in self.draw(1)[0])


class Queue:
    """torchio.Queue restated for HBM-resident subjects.

    `subjects_dataset`: a sequence of subject dicts {MRI: {DATA: (1,D,H,W)}, LABEL: {DATA: (1,D,H,W)}} of one common shape,
    on the device.  The queue holds at most `max_length` windows: it is refilled from the next
    `max_length // samples_per_volume` subjects (`samples_per_volume` random windows each; subjects reshuffled per pass when
    `shuffle_subjects`), shuffled when `shuffle_patches`, and popped from the end, as TorchIO's does.  The random stream is
    numpy's `default_rng(seed)`, not TorchIO's — which windows are drawn is a property of the seed, not of the reference.
    `num_workers` is accepted for signature compatibility; nothing is loaded, so there is nothing to parallelise.
    """

    def __init__(self, subjects_dataset, max_length, samples_per_volume, patch_size, sampler_class=ImageSampler,
                 num_workers=0, shuffle_subjects=True, shuffle_patches=True, seed=0):
        self.subjects = list(subjects_dataset)
        if not self.subjects:
            raise ValueError("Queue: no subjects")
        self.max_length, self.samples_per_volume = int(max_length), int(samples_per_volume)
        if self.samples_per_volume < 1 or self.max_length < self.samples_per_volume:
            raise ValueError("Queue: max_length must hold at least one subject's samples")
        self.patch_size = _triple(patch_size)
        self.sampler_class = sampler_class
        self.shuffle_subjects, self.shuffle_patches = bool(shuffle_subjects), bool(shuffle_patches)
        self.rng = np.random.default_rng(seed)
        self.names = [k for k, v in self.subjects[0].items() if isinstance(v, dict) and DATA in v]
        shape0 = tuple(self.subjects[0][self.names[0]][DATA].shape)
        self._resident = {}
        for name in self.names:
            vols = [s[name][DATA] for s in self.subjects]
            for v in vols:
                _require_volume(v, "Queue")
                if tuple(v.shape) != shape0 or v.dim() != 4 or v.shape[0] != 1:
                    raise ValueError("Queue: every image must be (1, D, H, W) of one common shape (CropOrPad first); got %s vs %s"
                                     % (tuple(v.shape), shape0))
            self._resident[name] = torch.cat(vols, dim=0).contiguous()  # (S, D, H, W) in HBM
        self.shape = shape0[1:]
        self.patches_list = np.empty((0, 4), dtype=np.int32)  # (subject index, d0, h0, w0); popped from the end
        self._order, self._next = [], 0

    def __len__(self):
        return len(self.subjects) * self.samples_per_volume

    def _next_subject(self):
        if self._next >= len(self._order):
            self._order = list(self.rng.permutation(len(self.subjects))) if self.shuffle_subjects \
                else list(range(len(self.subjects)))
            self._next = 0
        s = int(self._order[self._next])
        self._next += 1
        return s

    def fill(self):
        n_subjects = min(self.max_length // self.samples_per_volume, len(self.subjects))
        spv = self.samples_per_volume
        new = np.empty((n_subjects * spv, 4), dtype=np.int32)
        for i in range(n_subjects):
            sampler = self.sampler_class(self.shape, self.patch_size, self.rng)
            new[i * spv:(i + 1) * spv, 0] = self._next_subject()
            if hasattr(sampler, "draw"):
                new[i * spv:(i + 1) * spv, 1:] = sampler.draw(spv)
            else:
                it = iter(sampler)
                for j in range(spv):
                    new[i * spv + j, 1:] = next(it)
        self.patches_list = np.concatenate([self.patches_list, new])
        if self.shuffle_patches:
            self.patches_list = self.patches_list[self.rng.permutation(len(self.patches_list))]

    def _pop(self, n):
        parts, need = [], n
        while need > 0:
            if len(self.patches_list) == 0:
                self.fill()
            k = min(need, len(self.patches_list))
            parts.append(self.patches_list[len(self.patches_list) - k:][::-1])  # pop() order: last first
            self.patches_list = self.patches_list[:len(self.patches_list) - k]
            need -= k
        return np.ascontiguousarray(np.concatenate(parts))

    def _cut(self, table):
        out = {name: {DATA: extract_patches(v, table, self.patch_size)} for name, v in self._resident.items()}
        ini = table[:, 1:].astype(np.int64)
        out[LOCATION] = torch.from_numpy(np.concatenate([ini, ini + np.asarray(self.patch_size)], axis=1))
        out["subject"] = torch.from_numpy(table[:, 0].astype(np.int64))
        return out

    def __getitem__(self, _):
        """One window (the DataLoader protocol of the reference: the index is ignored, the queue decides)."""
        b = self._cut(self._pop(1))
        return {k: ({DATA: v[DATA][0]} if isinstance(v, dict) else v[0]) for k, v in b.items()}

    def batches(self, batch_size, drop_last=False):
        """One epoch of what DataLoader(queue, batch_size) yields: len(queue) windows, each batch cut by one launch per image."""
        left = len(self)
        while left > 0:
            n = min(int(batch_size), left)
            if n < int(batch_size) and drop_last:
                return
            yield self._cut(self._pop(n))
            left -= n
