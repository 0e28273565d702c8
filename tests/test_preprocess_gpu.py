"""SURVEY §8f row 2 — histogram standardisation on the device: exact order statistics, numpy-identical percentiles and the
float64 landmark map must reproduce the reference's `normalize` (classification/train_ENC_CLF.ipynb cell 9) BIT FOR BIT;
the oracle is pinned to the reference by tests/golden/hist_std.npz."""
import hashlib
import os

import numpy as np
import pytest
import torch

from mri_epilepsy_diagnosis_amd.classification import preprocessing as P
from oracle import preprocessing as O_PRE
from util import GOLDEN, load_golden

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.Generator(np.random.PCG64(3))
    yield "normal", rng.normal(0, 1, 100003).astype(np.float32)
    yield "ties", rng.integers(-3, 4, 50000).astype(np.float32)
    yield "zeros_mixed", np.concatenate([np.zeros(4000, np.float32), -np.zeros(3000, np.float32),
                                         rng.normal(0, 1e-3, 5000).astype(np.float32)])
    yield "tiny", np.array([3.0, -1.0, 2.0], np.float32)
    yield "one", np.array([7.5], np.float32)
    yield "wide", (rng.normal(0, 1, 70001) * np.exp(rng.normal(0, 8, 70001))).astype(np.float32)
    yield "denormal", (rng.normal(0, 1, 9000) * 1e-41).astype(np.float32)


@pytest.mark.parametrize("name,x", list(_cases()), ids=[c[0] for c in _cases()])
def test_order_statistics_exact(name, x):
    srt = np.sort(x)
    n = len(x)
    ranks = sorted(set([0, n - 1, n // 2, n // 3, min(n - 1, 7), max(0, n - 2)] + list(np.linspace(0, n - 1, 20).astype(int))))
    got = P.order_statistics(torch.from_numpy(x).cuda(), ranks).cpu().numpy()
    assert np.array_equal(got, srt[ranks])          # values (-0.0 == 0.0 compare equal; both are legitimate at a tie)
    # permutation invariance + repeated ranks
    got2 = P.order_statistics(torch.from_numpy(x[::-1].copy()).cuda(), ranks[::-1] + ranks[:2]).cpu().numpy()
    assert np.array_equal(got2, srt[ranks[::-1] + ranks[:2]])


@pytest.mark.parametrize("name,x", list(_cases()), ids=[c[0] for c in _cases()])
def test_percentile_identical_to_numpy(name, x):
    q = np.array([1.0, 10, 20, 25, 30, 40, 50, 60, 70, 75, 80, 90, 99, 0, 100, 33.3])
    got = P.percentile(torch.from_numpy(x).cuda(), q)
    ref = np.percentile(x, q)
    assert got.dtype == ref.dtype == np.float64
    assert np.array_equal(got, ref), (got - ref)


def test_normalize_bit_exact_vs_oracle_and_reference_golden():
    g = load_golden("hist_std.npz")
    shipped = np.load(os.path.join(GOLDEN, "fcd_train_data_landmarks.npy"))
    for name, lm in (("shipped", shipped), ("mono", g["mono_landmarks"])):
        for i, row in enumerate(g["cases"]):
            vol = O_PRE.synthetic_t1(int(row[0]), tuple(int(v) for v in row[1:]))
            out = P.normalize(torch.from_numpy(vol).cuda(), lm)
            assert out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == vol.shape
            o = out.cpu().numpy()
            assert hashlib.sha256(o.tobytes()).hexdigest() == str(g[name + "_sha256"][i]), (name, i)   # == the reference
            assert np.array_equal(o, O_PRE.normalize(vol, lm))


def test_normalize_full_size_volume_and_mask_and_cutoff():
    lm = load_golden("hist_std.npz")["mono_landmarks"]
    vol = O_PRE.synthetic_t1(99, (160, 192, 160))
    out = P.normalize(torch.from_numpy(vol).cuda(), lm).cpu().numpy()
    assert np.array_equal(out, O_PRE.normalize(vol, lm))
    small = O_PRE.synthetic_t1(5, (20, 24, 18))
    mask = small > small.mean()
    out = P.normalize(torch.from_numpy(small).cuda(), lm, mask=torch.from_numpy(mask), cutoff=(0.05, 0.95)).cpu().numpy()
    assert np.array_equal(out, O_PRE.normalize(small, lm, mask=mask, cutoff=(0.05, 0.95)))


def test_default_collate_matches_reference_layout():
    lm = load_golden("hist_std.npz")["mono_landmarks"]
    vols = [torch.from_numpy(O_PRE.synthetic_t1(s, (1, 12, 14, 10))) for s in (1, 2, 3)]
    X, y, dom = P.default_collate([(vols[0], 1, 4), (vols[1], 0, 2), (vols[2], 1, 17)], lm)
    assert X.is_cuda and tuple(X.shape) == (3, 1, 12, 14, 10) and y.tolist() == [1, 0, 1] and dom.tolist() == [4, 2, 17]
    for k in range(3):
        assert np.array_equal(X[k].cpu().numpy(), O_PRE.normalize(vols[k].numpy(), lm))


def test_no_cpu_fallback():
    with pytest.raises(RuntimeError):
        P.normalize(torch.zeros(2, 2, 2), np.arange(13.0))
    with pytest.raises(RuntimeError):
        P.order_statistics(torch.zeros(4), [0])


def test_z_normalize_matches_torch_restatement():
    """ZNormalization(masking_method=mean) — third-party (TorchIO) arithmetic, "parity unpinned": checked against the
    torch-CPU restatement to fp32 rounding of the two statistics (1e-6 relative) and through its defining properties."""
    for seed, shape in ((1, (1, 24, 20, 28)), (2, (160, 192, 160))):
        vol = O_PRE.synthetic_t1(seed, shape)
        y, stats = P.z_normalize(torch.from_numpy(vol).cuda())
        ref = O_PRE.z_normalize(vol)
        yc = y.cpu().numpy()
        assert np.abs(yc - ref).max() <= 2e-6 * np.abs(ref).max()
        m = vol > np.float32(vol.astype(np.float64).mean())
        st = stats.cpu().numpy()
        assert st[1] == m.sum()
        assert abs(yc[m].astype(np.float64).mean()) < 1e-5 and abs(yc[m].astype(np.float64).std(ddof=1) - 1.0) < 1e-5


@pytest.mark.parametrize("src,tgt", [((20, 24, 18), (16, 30, 18)), ((7, 9, 11), (12, 4, 11)), ((160, 192, 160), (176, 208, 160)),
                                     ((13, 13, 13), (6, 21, 14))])
def test_crop_or_pad_exact(src, tgt):
    rng = np.random.Generator(np.random.PCG64(4))
    x = rng.normal(size=(2, 1) + src).astype(np.float32)
    y = P.crop_or_pad(torch.from_numpy(x).cuda(), tgt)
    assert tuple(y.shape) == (2, 1) + tgt
    assert np.array_equal(y.cpu().numpy(), O_PRE.crop_or_pad(x, tgt))
