"""Worker for test_two_ranks_on_one_gpu_*: two processes share cuda:0, collectives go through gloo (RCCL needs one device per
rank).  Each rank trains ONE step of the HIP U-Net on its own volume with synchronised BatchNorm; rank 0 also runs the same
step on both volumes in a single process for comparison."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_epilepsy_diagnosis_amd import ops, parallel  # noqa: E402
from mri_epilepsy_diagnosis_amd.unet import UNet  # noqa: E402


def make():
    torch.manual_seed(0)
    return UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=2, out_channels_first_layer=8,
                normalization="batch", upsampling_type="linear", padding=True, activation="PReLU").cuda().train()


def data(i):
    g = torch.Generator().manual_seed(100 + i)
    x = torch.randn(1, 1, 16, 16, 16, generator=g)
    t = (torch.rand(1, 1, 16, 16, 16, generator=g) < 0.3).float()
    return x.cuda(), t.cuda()


def step(model, xs, ts, world_scale, sync):
    flat = parallel.FlatParams(model)
    opt = parallel.FlatAdam(flat, lr=1e-3, weight_decay=0.01, decoupled=True)
    flat.zero_grad()
    x, t = torch.cat(xs), torch.cat(ts)
    if sync:
        with parallel.SyncBatchNorm():
            loss = ops.softmax_dice_loss(model(x), t)
            loss.backward()
    else:
        loss = ops.softmax_dice_loss(model(x), t)
        loss.backward()
    scale = flat.all_reduce() if world_scale is None else world_scale
    opt.step(scale)
    bufs = torch.cat([b.detach().float().reshape(-1) for b in model.buffers()])
    return flat.flat.detach().cpu().clone(), flat.grad.detach().cpu().clone() * scale, bufs.cpu()


if __name__ == "__main__":
    rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), "0"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, t = data(rank)
    params, grads, bufs = step(make(), [x], [t], None, sync=True)
    torch.save({"params": params, "grads": grads, "bufs": bufs}, os.path.join(out, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        xs, ts = zip(*[data(i) for i in range(world)])
        # single process, both volumes in one batch.  The Dice loss is a mean over (volume, class), so the sum of the two
        # per-rank losses' gradients scaled by 1/world equals the gradient of the batch loss.
        params, grads, bufs = step(make(), list(xs), list(ts), 1.0, sync=False)
        torch.save({"params": params, "grads": grads, "bufs": bufs}, os.path.join(out, "single.pt"))
