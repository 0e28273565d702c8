"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

REL_TOL = 1e-3  # north_star: "within 1e-3 rel fp32"

AE_KWARGS_93_6_4 = dict(  # classification/train_ENC_CLF.ipynb cell 17
    c_in=1, is_skip=False, deapth=3, c_base=8, inc_size=2, reduce_size=False,
    down_block_kwargs=dict(conv_k=6, conv_pad=2, conv_s=2, maxpool_k=2, maxpool_s=2, batch_norm=True, act="l_relu"),
    up_block_kwargs=dict(up="upsample", scale=4, scale_mode="nearest", conv_k=3, conv_pad=1, conv_s=1, batch_norm=False,
                         act="l_relu"))
DISC_KWARGS = dict(c_in=32, c_out=64, conv_k=3, conv_s=1, conv_pad=0, l_in=64, l_out=32, batch_norm=True, act="relu",
                   n_domains=18, p_drop=0.5)
CLF_KWARGS = dict(c_in=32, c_out=64, conv_k=3, conv_s=1, conv_pad=0, l_in=64, l_out=32, batch_norm=True, act="relu",
                  p_drop=0.5, n_class=2)


def seeded_randn(seed, shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


def seeded_rand(seed, shape):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g)


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).detach().double().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.numel() == 0:
        return 0.0
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def assert_close(a, b, rel=REL_TOL, what=""):
    e = rel_err(a, b)
    assert e <= rel, "%s: max-norm relative error %.3e > %.1e" % (what, e, rel)


def sample(t, n=4096):
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // n)
    return f[::stride].cpu().numpy().copy(), stride


def grad_norms(model):
    return np.array([p.grad.detach().double().norm().item() if p.grad is not None else -1.0
                     for _, p in model.named_parameters()])


def param_checksum(model):
    return np.array([p.detach().double().sum().item() for p in model.state_dict().values() if p.dtype.is_floating_point])


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def load_ckpt(name):
    return torch.load(os.path.join(GOLDEN, "ckpt", name), weights_only=True, map_location="cpu")


def to_ncdhw(t):
    """Logical-NCDHW contiguous CPU copy of a (possibly channels-last) device tensor."""
    return t.detach().cpu().contiguous(memory_format=torch.contiguous_format)


def kink_and_pool_margins(model64, x64):
    """(min |x| over every PReLU input, min (largest - second largest) over every MaxPool3d(2) window) of one forward pass —
    how far the model is from its non-differentiable points on this input."""
    import torch.nn as nn
    acts, pools, hooks = [], [], []

    def pool_hook(mod, inp, out):
        x = inp[0].detach()
        n, c, d, h, w = x.shape
        win = x.view(n, c, d // 2, 2, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 6, 3, 5, 7).reshape(n, c, d // 2, h // 2, w // 2, 8)
        top = win.topk(2, dim=-1).values
        pools.append((top[..., 0] - top[..., 1]).min().item())
    for mod in model64.modules():
        if isinstance(mod, nn.PReLU):
            hooks.append(mod.register_forward_hook(lambda mo, inp, out: acts.append(inp[0].detach().abs().min().item())))
        elif isinstance(mod, nn.MaxPool3d):
            hooks.append(mod.register_forward_hook(pool_hook))
    with torch.no_grad():
        model64(x64)
    for h in hooks:
        h.remove()
    return min(acts), min(pools)
