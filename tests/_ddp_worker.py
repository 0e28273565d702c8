"""Worker for test_data_parallel_gloo_world2_matches_single_process (CPU, gloo)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mri_epilepsy_diagnosis_amd import parallel  # noqa: E402
from oracle import losses, unet_recon  # noqa: E402
from util import seeded_rand, seeded_randn  # noqa: E402


def grads_for(volumes, reduce):
    torch.manual_seed(0)
    m = unet_recon.UNetRecon(out_channels_first_layer=8)
    m.eval()  # frozen BN statistics: volumes are independent, so DP == big batch up to the 1/world scale
    fp = parallel.FlatParams(m)
    fp.zero_grad()
    for i in volumes:
        x = seeded_randn(100 + i, (1, 1, 16, 16, 16))
        t = (seeded_rand(200 + i, (1, 1, 16, 16, 16)) < 0.1).float()
        losses.softmax_dice_loss(m(x), t).backward()
    scale = fp.all_reduce() if reduce else 1.0 / len(volumes)
    return fp.grad * scale


if __name__ == "__main__":
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    torch.set_num_threads(2)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    r, _, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard_range(world, rank, world)
    g = grads_for(range(lo, hi), reduce=True)
    torch.save(g, os.path.join(out, "grad_rank%d.pt" % rank))
    # the collective behind parallel.SyncBatchNorm: a float64 sum over the ranks
    packed = parallel.SyncBatchNorm().all_reduce(torch.full((5,), float(rank + 1), dtype=torch.float64))
    assert packed.dtype == torch.float64 and torch.equal(packed, torch.full((5,), world * (world + 1) / 2.0, dtype=torch.float64))
    if rank == 0:
        torch.save(grads_for(range(world), reduce=False), os.path.join(out, "grad_single.pt"))
    dist.barrier()
    dist.destroy_process_group()
