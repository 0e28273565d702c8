"""SURVEY §8f row 3 — window arithmetic of the patch pipeline (host integers): the product's grid must equal the CPU
restatement of TorchIO's grid sampler (oracle/patches.py, "parity unpinned": TorchIO is absent), and the restatement itself
must satisfy the properties the algorithm promises."""
import numpy as np
import pytest
import torch

from mri_epilepsy_diagnosis_amd.segmentation import patches as P
from oracle import patches as O

SHAPES = [((160, 192, 160), 64, 4), ((64, 64, 64), 64, 4), ((65, 64, 70), 64, 4), ((96, 80, 72), 32, 4),
          ((100, 90, 130), (48, 32, 64), (4, 2, 8)), ((70, 70, 70), 64, 0), ((128, 128, 128), 64, 0),
          ((33, 40, 47), 16, 3), ((20, 21, 22), (20, 8, 5), (0, 1, 2)), ((57, 64, 71), 8, 0)]


@pytest.mark.parametrize("shape,patch,overlap", SHAPES)
def test_grid_locations_match_oracle(shape, patch, overlap):
    pt, ov = P._triple(patch), P._triple(overlap)
    got, want = P.grid_locations(shape, pt, ov), O.grid_locations(shape, pt, ov)
    assert got.dtype == np.int32 and np.array_equal(got, want)
    # every voxel further than `overlap` from the volume's faces is covered by a cropped window
    cover = np.zeros(shape, bool)
    for i0, j0, k0, i1, j1, k1 in got:
        cover[i0 + ov[0]:i1 - ov[0], j0 + ov[1]:j1 - ov[1], k0 + ov[2]:k1 - ov[2]] = True
    inner = tuple(slice(o, s - o) for s, o in zip(shape, ov))
    assert cover[inner].all()


def test_grid_known_answers():
    # 160 / 64 / overlap 4: step 56 -> 0, 56 and the flush window at 96; 192 -> 0, 56, 112, 128
    assert P._axis_starts(160, 64, 56) == [0, 56, 96]
    assert P._axis_starts(192, 64, 56) == [0, 56, 112, 128]
    assert P._axis_starts(64, 64, 56) == [0]
    assert P._axis_starts(70, 64, 64) == [0, 6, 3]          # two starts -> a third at their mean, appended last
    assert P._axis_starts(65, 64, 56) == [0, 1]             # mean 0.5 rounds (half-even) onto an existing start
    assert len(P.grid_locations((160, 192, 160), 64, 4)) == 36
    with pytest.raises(ValueError):
        P.grid_locations((32, 64, 64), 64, 4)


def test_oracle_round_trip_without_overlap():
    rng = np.random.default_rng(0)
    vol = rng.integers(0, 3, (48, 40, 56)).astype(np.uint8)
    loc = O.grid_locations(vol.shape, (16, 8, 8), (0, 0, 0))
    assert np.array_equal(O.aggregate(vol.shape, O.extract(vol, loc), loc, (0, 0, 0)), vol)
    # with a border, the interior is reproduced and the outer strip stays zero
    loc = O.grid_locations(vol.shape, (16, 16, 16), (2, 3, 1))
    out = O.aggregate(vol.shape, O.extract(vol, loc), loc, (2, 3, 1))
    assert np.array_equal(out[2:-2, 3:-3, 1:-1], vol[2:-2, 3:-3, 1:-1])
    out[2:-2, 3:-3, 1:-1] = 0
    assert not out.any()


def test_oracle_later_window_wins():
    shape = (8, 8, 8)
    wins = np.stack([np.full((6, 6, 6), 1, np.uint8), np.full((6, 6, 6), 2, np.uint8)])
    loc = np.array([[0, 0, 0, 6, 6, 6], [2, 2, 2, 8, 8, 8]], np.int32)
    out = O.aggregate(shape, wins, loc, (1, 1, 1))
    assert out[1, 1, 1] == 1 and out[3, 3, 3] == 2 and out[6, 6, 6] == 2 and out[0, 0, 0] == 0 and out[7, 7, 7] == 0


def test_no_cpu_fallback():
    sample = {P.MRI: {P.DATA: torch.zeros(1, 16, 16, 16)}}
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.GridSampler(sample, 8, 0)[0]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.GridAggregator(sample, 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.Queue([sample], 8, 2, 8)


def test_image_sampler_range():
    rng = np.random.default_rng(1)
    s = P.ImageSampler((20, 16, 9), (8, 16, 3), rng)
    pts = np.array([next(s) for _ in range(500)])
    assert pts.min(axis=0).tolist() == [0, 0, 0] and pts.max(axis=0).tolist() == [12, 0, 6]
    with pytest.raises(ValueError):
        P.ImageSampler((8, 8, 8), 9, rng)
