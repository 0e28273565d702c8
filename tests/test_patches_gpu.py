"""SURVEY §8f row 3 — the patch pipeline on the device against the CPU restatement (oracle/patches.py; TorchIO itself is
absent, "parity unpinned").  Byte work: every comparison is bit-exact."""
import numpy as np
import pytest
import torch

from mri_epilepsy_diagnosis_amd import ops
from mri_epilepsy_diagnosis_amd.segmentation import patches as P
from mri_epilepsy_diagnosis_amd.unet import UNet
from oracle import patches as O

pytestmark = pytest.mark.gpu


def _table(rng, nvol, shape, patch, n):
    loc = np.stack([rng.integers(0, s - p + 1, n) for s, p in zip(shape, patch)], axis=1)
    return np.concatenate([rng.integers(0, nvol, (n, 1)), loc], axis=1).astype(np.int32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.uint8, torch.bfloat16, torch.int64, torch.int16])
@pytest.mark.parametrize("nvol,shape,patch,n", [(3, (40, 36, 50), (16, 12, 20), 150), (1, (20, 21, 22), (20, 21, 22), 1),
                                                (2, (33, 17, 9), (1, 5, 9), 70), (4, (64, 64, 64), (32, 32, 32), 16)])
def test_extract_patches_bit_exact(dtype, nvol, shape, patch, n):
    rng = np.random.default_rng(hash((nvol, n)) % 1000)
    g = torch.Generator().manual_seed(n)
    vols = (torch.randn((nvol,) + shape, generator=g) * 50).to(dtype)
    table = _table(rng, nvol, shape, patch, n)
    got = P.extract_patches(vols.cuda(), table, patch).cpu()
    assert got.shape == (n, 1) + patch and got.dtype == dtype
    for p, (v, d0, h0, w0) in enumerate(table):
        want = vols[v, d0:d0 + patch[0], h0:h0 + patch[1], w0:w0 + patch[2]]
        assert torch.equal(got[p, 0].view(torch.uint8 if dtype == torch.uint8 else dtype), want), p


def test_extract_rejects_windows_outside_the_volume():
    vols = torch.zeros(2, 8, 8, 8, device="cuda")
    for bad in ([[0, 1, 0, 0]], [[2, 0, 0, 0]], [[0, 0, 0, -1]], [[0, 0, 5, 0]]):
        with pytest.raises(RuntimeError, match="leaves the"):
            P.extract_patches(vols, np.array(bad, np.int32), (8, 4, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.extract_patches(vols.cpu(), np.array([[0, 0, 0, 0]], np.int32), (8, 4, 8))


GRIDS = [((96, 80, 72), (32, 32, 32), (4, 4, 4), 16), ((65, 64, 70), (64, 64, 64), (4, 4, 4), 5),
         ((100, 90, 130), (48, 32, 64), (4, 2, 8), 7), ((70, 70, 70), (64, 64, 64), (0, 0, 0), 16),
         ((57, 64, 71), (8, 8, 8), (0, 0, 0), 200), ((33, 40, 47), (16, 16, 16), (3, 3, 3), 100),
         ((160, 192, 160), (64, 64, 64), (4, 4, 4), 16)]


@pytest.mark.parametrize("shape,patch,overlap,batch", GRIDS)
def test_grid_sample_and_aggregate_labels(shape, patch, overlap, batch):
    rng = np.random.default_rng(sum(shape))
    vol = rng.integers(0, 4, shape).astype(np.uint8)
    sample = {P.LABEL: {P.DATA: torch.from_numpy(vol)[None].cuda()}}
    sampler = P.GridSampler(sample, patch, overlap)
    loc = O.grid_locations(shape, patch, overlap)
    assert np.array_equal(sampler.locations, loc)
    agg = P.GridAggregator(sample, overlap)
    # perturb every window by its index so that overlapping windows disagree and the write order matters
    wins_all = (O.extract(vol, loc) + (np.arange(len(loc), dtype=np.uint8) * 7)[:, None, None, None]).astype(np.uint8)
    n = 0
    for b in sampler.batches(batch):
        k = b[P.LOCATION].shape[0]
        assert np.array_equal(b[P.LOCATION].numpy(), loc[n:n + k])
        assert np.array_equal(b[P.LABEL][P.DATA].cpu().numpy()[:, 0], O.extract(vol, loc[n:n + k]))
        agg.add_batch(torch.from_numpy(wins_all[n:n + k])[:, None].cuda(), b[P.LOCATION])
        n += k
    assert n == len(loc) == len(sampler)
    want = O.aggregate(shape, wins_all, loc, overlap)
    got = agg.get_output_tensor(torch.uint8)
    assert got.shape == (1,) + shape and np.array_equal(got[0].cpu().numpy(), want)
    assert agg.get_output_tensor().dtype == torch.float32
    one = sampler[len(sampler) - 1]
    assert one[P.LABEL][P.DATA].shape == (1,) + patch and np.array_equal(one[P.LOCATION].numpy(), loc[-1])


def test_aggregate_round_trip_without_overlap():
    vol = torch.randint(0, 2, (1, 128, 64, 192), dtype=torch.uint8, device="cuda")
    sample = {P.LABEL: {P.DATA: vol}}
    sampler, agg = P.GridSampler(sample, 64, 0), P.GridAggregator(sample, 0)
    for b in sampler.batches(5):
        agg.add_batch(b[P.LABEL][P.DATA], b[P.LOCATION])
    assert torch.equal(agg.get_output_tensor(torch.uint8), vol)


@pytest.mark.parametrize("n", [3, 64, 65, 150])
def test_aggregate_random_overlapping_windows_later_wins(n):
    rng = np.random.default_rng(n)
    shape, patch, overlap = (40, 36, 50), (16, 12, 20), (2, 1, 3)
    ini = np.stack([rng.integers(0, s - p + 1, n) for s, p in zip(shape, patch)], axis=1)
    loc = np.concatenate([ini, ini + np.asarray(patch)], axis=1).astype(np.int32)
    wins = rng.integers(1, 255, (n,) + patch).astype(np.uint8)
    sample = {P.MRI: {P.DATA: torch.zeros((1,) + shape, device="cuda")}}
    agg = P.GridAggregator(sample, overlap)
    agg.add_batch(torch.from_numpy(wins).cuda(), loc)                      # one call: ordering resolved inside the launches
    assert np.array_equal(agg.get_output_tensor(torch.uint8)[0].cpu().numpy(), O.aggregate(shape, wins, loc, overlap))
    agg2 = P.GridAggregator(sample, overlap)
    for i in range(0, n, 7):                                               # many calls: ordering by the stream
        agg2.add_batch(torch.from_numpy(wins[i:i + 7]).cuda(), loc[i:i + 7])
    assert torch.equal(agg2.get_output_tensor(torch.uint8), agg.get_output_tensor(torch.uint8))


@pytest.mark.parametrize("dtype,c", [(torch.float32, 2), (torch.bfloat16, 2), (torch.float32, 3)])
def test_aggregate_argmax_of_logits(dtype, c):
    shape, patch, overlap = (40, 40, 48), (16, 16, 16), (2, 2, 2)
    loc = O.grid_locations(shape, patch, overlap)
    g = torch.Generator().manual_seed(c)
    logits = torch.randn((len(loc), c) + patch, generator=g).to(dtype)
    logits[:, :, ::3, 1::2] = 0.25                                           # ties: the first maximum must win
    logits[0, 1, 5, 5, 5] = float("nan")
    sample = {P.MRI: {P.DATA: torch.zeros((1,) + shape, device="cuda")}}
    agg = P.GridAggregator(sample, overlap)
    dev = logits.cuda().contiguous(memory_format=torch.channels_last_3d)
    for i in range(0, len(loc), 16):
        agg.add_batch_logits(dev[i:i + 16], loc[i:i + 16])
    labels = logits.float().argmax(dim=1).numpy().astype(np.uint8)
    assert np.array_equal(agg.get_output_tensor(torch.uint8)[0].cpu().numpy(), O.aggregate(shape, labels, loc, overlap))
    # the two-step form the notebook writes gives the same volume
    agg2 = P.GridAggregator(sample, overlap)
    agg2.add_batch(ops.argmax_mask(dev)[:, None], loc)
    assert torch.equal(agg2.get_output_tensor(torch.uint8), agg.get_output_tensor(torch.uint8))


def test_aggregate_rejects_bad_windows():
    sample = {P.MRI: {P.DATA: torch.zeros((1, 16, 16, 16), device="cuda")}}
    wins = torch.zeros((1, 1, 8, 8, 8), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError, match="leaves the"):
        P.GridAggregator(sample, 1).add_batch(wins, np.array([[9, 0, 0, 17, 8, 8]]))
    with pytest.raises(RuntimeError, match="leaves nothing"):
        P.GridAggregator(sample, 4).add_batch(wins, np.array([[0, 0, 0, 8, 8, 8]]))
    with pytest.raises(ValueError):
        P.GridAggregator(sample, 1).add_batch(wins, np.array([[0, 0, 0, 8, 8, 9]]))


def _subjects(n, shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [{P.MRI: {P.DATA: torch.randn((1,) + shape, generator=g).cuda()},
             P.LABEL: {P.DATA: (torch.rand((1,) + shape, generator=g) < 0.2).float().cuda()}} for _ in range(n)]


@pytest.mark.parametrize("shuffle", [False, True])
def test_queue_yields_windows_of_the_resident_subjects(shuffle):
    shape, patch = (40, 48, 36), (16, 16, 16)
    subjects = _subjects(5, shape)
    q = P.Queue(subjects, max_length=12, samples_per_volume=4, patch_size=patch, shuffle_subjects=shuffle,
                shuffle_patches=shuffle, seed=3)
    assert len(q) == 20
    seen, total = [], 0
    for b in q.batches(6):
        x, y, loc, sub = b[P.MRI][P.DATA], b[P.LABEL][P.DATA], b[P.LOCATION].numpy(), b["subject"].numpy()
        assert x.shape[1:] == (1,) + patch and x.shape == y.shape and x.is_cuda
        for i in range(x.shape[0]):
            i0, j0, k0, i1, j1, k1 = loc[i]
            assert (i1 - i0, j1 - j0, k1 - k0) == patch
            assert torch.equal(x[i, 0], subjects[sub[i]][P.MRI][P.DATA][0, i0:i1, j0:j1, k0:k1])
            assert torch.equal(y[i, 0], subjects[sub[i]][P.LABEL][P.DATA][0, i0:i1, j0:j1, k0:k1])
        seen += sub.tolist()
        total += x.shape[0]
    assert total == 20
    if not shuffle:
        # a fill takes 12 // 4 = 3 subjects in order, windows are popped from the end of the list
        assert seen[:12] == [2] * 4 + [1] * 4 + [0] * 4 and seen[12:] == [0] * 4 + [4] * 4
    # the same seed replays the same windows; a single __getitem__ is one of them
    q2 = P.Queue(subjects, 12, 4, patch, shuffle_subjects=shuffle, shuffle_patches=shuffle, seed=3)
    first = next(iter(q2.batches(6)))
    q3 = P.Queue(subjects, 12, 4, patch, shuffle_subjects=shuffle, shuffle_patches=shuffle, seed=3)
    item = q3[0]
    assert item[P.MRI][P.DATA].shape == (1,) + patch
    assert torch.equal(item[P.MRI][P.DATA], first[P.MRI][P.DATA][0]) and torch.equal(item[P.LOCATION], first[P.LOCATION][0])


def test_grid_inference_of_the_unet_matches_the_sequential_aggregator():
    """pretraining_3d_unet.ipynb cell 26 end to end: HIP model on grid windows -> arg-max -> aggregator."""
    torch.manual_seed(0)
    shape, patch, overlap = (48, 72, 40), (32, 32, 32), (4, 4, 4)
    model = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=8,
                 normalization="batch", upsampling_type="linear", padding=True, activation="PReLU").cuda().eval()
    sample = {P.MRI: {P.DATA: torch.randn((1,) + shape).cuda()}}
    sampler, agg = P.GridSampler(sample, patch, overlap), P.GridAggregator(sample, overlap)
    labels, locs = [], []
    with torch.no_grad():
        for b in sampler.batches(4):
            logits = model(b[P.MRI][P.DATA])
            agg.add_batch_logits(logits, b[P.LOCATION])
            labels.append(logits.float().cpu().argmax(dim=1).numpy().astype(np.uint8))
            locs.append(b[P.LOCATION].numpy())
    want = O.aggregate(shape, np.concatenate(labels), np.concatenate(locs), overlap)
    got = agg.get_output_tensor()
    assert got.dtype == torch.float32 and np.array_equal(got[0].cpu().numpy(), want.astype(np.float32))
    assert 0 < want.sum() < want.size
