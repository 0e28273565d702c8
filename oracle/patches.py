"""CPU restatement of the TorchIO patch pipeline the reference drives (TEST ORACLE — only tests/, smoke() and bench.py's
cpu_baseline may import this package).

PARITY UNPINNED.  The arithmetic lives in `torchio` (PyPI, F. Pérez-García), a third-party dependency that is neither in
/root/reference nor installed here, and the reference pins no version (no requirements file; the API it calls —
`torchio.Queue(sampler_class=torchio.sampler.ImageSampler)`, `torchio.inference.GridSampler(sample, patch_size,
patch_overlap)`, `GridAggregator(sample, patch_overlap)` — is that of torchio ≈ 0.1x, June 2020).  The reference holds no
test or golden vector for it.  What is restated here is that era's published algorithm (itself taken from NiftyNet's
grid sampler), anchored on the reference's call sites:
  * segmentation/routine.py:150-178, segmentation/pretraining_3d_unet.ipynb cell 24: random 64^3 training patches,
    `samples_per_volume` per subject, through a shuffling queue;
  * pretraining_3d_unet.ipynb cell 26: grid inference with patch 64^3, overlap 4, `labels = logits.argmax(dim=1,
    keepdim=True)`, `aggregator.add_batch(labels, locations)`, `aggregator.get_output_tensor()`.

Grid: along each axis the windows start every `patch - 2*overlap` voxels while they fit, plus one last window flush with the
far end; if that leaves exactly two starts, a third is added at their rounded mean.  Locations are (i0,j0,k0,i1,j1,k1).
Aggregation: every window is cropped by `overlap` voxels on all six faces and assigned into the output volume with plain
slicing, in batch order — a later window overwrites an earlier one where they overlap, and the outermost `overlap` voxels of
the volume are never written (they stay 0).
"""
import numpy as np


def enumerate_step_points(starting, ending, win_size, step_size):
    starting = max(int(starting), 0)
    ending = max(int(ending), 0)
    win_size = max(int(win_size), 1)
    step_size = max(int(step_size), 1)
    if starting > ending:
        starting, ending = ending, starting
    points = []
    while starting + win_size <= ending:
        points.append(starting)
        starting += step_size
    points.append(max(ending - win_size, 0))
    points = np.unique(points).flatten()
    if len(points) == 2:
        points = np.append(points, np.round(np.mean(points)))
    _, first = np.unique(points, return_index=True)
    return points[np.sort(first)]


def grid_locations(shape, patch_size, patch_overlap):
    """int32 [n, 6] windows covering a volume of `shape` (meshgrid order of the three axes' starts, as numpy yields it)."""
    shape = tuple(int(v) for v in shape)
    patch = tuple(int(v) for v in patch_size)
    border = tuple(int(v) for v in patch_overlap)
    steps = [enumerate_step_points(0, shape[i], patch[i], max(patch[i] - 2 * border[i], 0)) for i in range(3)]
    starts = np.asanyarray(np.meshgrid(*steps)).reshape((3, -1)).T
    loc = np.zeros((starts.shape[0], 6), dtype=np.int32)
    loc[:, :3] = starts
    for i in range(3):
        loc[:, 3 + i] = starts[:, i] + patch[i]
    assert np.all(loc[:, 3:].max(axis=0) <= np.asarray(shape)), "window larger than the volume"
    return loc


def extract(volume, locations):
    """[n, pd, ph, pw] windows of a (D,H,W) array."""
    return np.stack([volume[i0:i1, j0:j1, k0:k1] for i0, j0, k0, i1, j1, k1 in np.asarray(locations)])


def aggregate(shape, windows, locations, patch_overlap, out=None):
    """Sequential add_batch: crop each window by the border and slice-assign it; returns the (D,H,W) uint8 volume."""
    out = np.zeros(shape, dtype=np.uint8) if out is None else out
    b = tuple(int(v) for v in patch_overlap)
    for win, (i0, j0, k0, i1, j1, k1) in zip(np.asarray(windows), np.asarray(locations)):
        pd, ph, pw = win.shape
        out[i0 + b[0]:i1 - b[0], j0 + b[1]:j1 - b[1], k0 + b[2]:k1 - b[2]] = win[b[0]:pd - b[0], b[1]:ph - b[1],
                                                                                 b[2]:pw - b[2]]
    return out


def random_locations(shape, patch_size, n, rng):
    """ImageSampler: a uniformly random window origin per axis in [0, size - patch] (numpy Generator `rng`)."""
    shape = np.asarray(shape, dtype=np.int64)
    patch = np.asarray(patch_size, dtype=np.int64)
    assert np.all(patch <= shape), "patch larger than the volume"
    ini = np.stack([rng.integers(0, shape[i] - patch[i] + 1, size=n) for i in range(3)], axis=1)
    return np.concatenate([ini, ini + patch], axis=1).astype(np.int32)
