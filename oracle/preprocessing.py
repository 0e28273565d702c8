"""CPU restatement of the reference's histogram standardisation (TEST ORACLE — only tests/, smoke() and bench.py's
cpu_baseline may import this package).

`normalize`: classification/train_ENC_CLF.ipynb cell 9 (np.percentile at 13 percentiles of all voxels -> 10 linear segments
between the landmarks of segmentation/weights/fcd_train_data_landmarks.npy, evaluated in float64, stored as float32).
Pinned by executing the notebook cell itself in the authoring container (oracle/gen_golden.py hist_std ->
tests/golden/hist_std.npz).
"""
import numpy as np

DEFAULT_CUTOFF = 0.01, 0.99


def _standardize_cutoff(cutoff):
    cutoff = np.asarray(cutoff, dtype=np.float64).copy()
    cutoff[0] = max(0., cutoff[0])
    cutoff[1] = min(1., cutoff[1])
    cutoff[0] = np.min([cutoff[0], 0.09])
    cutoff[1] = np.max([cutoff[1], 0.91])
    return cutoff


def _get_percentiles(percentiles_cutoff):
    quartiles = np.arange(25, 100, 25).tolist()
    deciles = np.arange(10, 100, 10).tolist()
    return np.array(sorted(set(list(percentiles_cutoff) + quartiles + deciles)))


def normalize(array, landmarks, mask=None, cutoff=None, epsilon=1e-5):
    cutoff_ = DEFAULT_CUTOFF if cutoff is None else cutoff
    shape = array.shape
    data = np.asarray(array).reshape(-1).astype(np.float32)
    if mask is None:
        mask = np.ones_like(data, bool)
    mask = np.asarray(mask).reshape(-1)
    range_to_use = [0, 1, 2, 4, 5, 6, 7, 8, 10, 11, 12]
    percentiles = _get_percentiles(100 * np.array(_standardize_cutoff(cutoff_)))
    percentile_values = np.percentile(data[mask], percentiles)
    range_mapping = np.asarray(landmarks)[range_to_use]
    range_perc = percentile_values[range_to_use]
    diff_mapping = np.diff(range_mapping)
    diff_perc = np.diff(range_perc)
    diff_perc[diff_perc < epsilon] = np.inf
    affine_map = np.zeros([2, len(range_to_use) - 1])
    affine_map[0] = diff_mapping / diff_perc
    affine_map[1] = range_mapping[:-1] - affine_map[0] * range_perc[:-1]
    bin_id = np.digitize(data, range_perc[1:-1], right=False)
    new_img = affine_map[0, bin_id] * data + affine_map[1, bin_id]
    return new_img.reshape(shape).astype(np.float32)


def synthetic_t1(seed, shape):
    """Skull-stripped-T1-like intensities: ~45 % exact-zero background, tissue = |N(600, 250)| with a bright tail."""
    rng = np.random.Generator(np.random.PCG64(seed))
    tissue = np.abs(rng.normal(600.0, 250.0, size=shape)) + 50.0 * rng.random(shape) ** 4
    return np.where(rng.random(shape) < 0.45, 0.0, tissue).astype(np.float32)


def z_normalize(array):
    """TorchIO ZNormalization(masking_method=mean) restated with torch CPU ops (third-party, "parity unpinned")."""
    import torch
    t = torch.as_tensor(np.asarray(array, dtype=np.float32))
    mask = t > t.mean()
    values = t[mask]
    return ((t - values.mean()) / values.std()).numpy()


def crop_or_pad(array, target_shape, fill=0.0):
    """TorchIO CropOrPad restated: per axis, ini = floor(|diff| / 2) cropped or padded in front, the rest at the end."""
    a = np.asarray(array)
    for ax, tgt in zip((-3, -2, -1), target_shape):
        n = a.shape[ax]
        if n > tgt:
            ini = (n - tgt) // 2
            a = np.take(a, np.arange(ini, ini + tgt), axis=ax)
        elif n < tgt:
            ini = (tgt - n) // 2
            pad = [(0, 0)] * a.ndim
            pad[ax] = (ini, tgt - n - ini)
            a = np.pad(a, pad, constant_values=fill)
    return np.ascontiguousarray(a)
