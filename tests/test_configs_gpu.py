"""GPU parity at BASELINE.json's OWN configuration sizes — one test per config, each against the CPU oracle
run live on the same seeded inputs AND against fixtures recorded from the imported reference modules
(`oracle/gen_golden.py configs`):

  cfg2  unet.UNet c0=8, 1 x 160x192x160 fp32: shipped checkpoint, eval mask bit-exact, train step (logits/loss/grads/BN stats)
  cfg3  AE(**93_6_4) batch 4 x 160x192x160: MSE step; encoder + classifier head (conv_pad=1, l_in=768) CE step
  cfg4  the cfg2 step inside the bf16 storage region at 2 x 160x192x160, against the fp32 HIP run
  cfg5  CNN(32^3) + Linear(128,2): batch 64 (per-GPU share of 512/8) vs oracle + reference fixture; batch 512 properties
  a10   Modified3DUNet in TRAIN mode with the Dropout3d masks injected
"""
import hashlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mri_epilepsy_diagnosis_amd import ops, parallel
from mri_epilepsy_diagnosis_amd.classification.models import AE_model as P_AE, cnn_model as P_CNN
from mri_epilepsy_diagnosis_amd.segmentation.models.modified_3dunet import Modified3DUNet
from mri_epilepsy_diagnosis_amd.unet import UNet
from oracle import ae_model as O_AE, cnn_model as O_CNN, losses, modified_3dunet as O_M, unet_recon
from test_models_gpu import _compare
from util import (AE_KWARGS_93_6_4, CLF_KWARGS, assert_close, grad_norms, load_ckpt, load_golden, sample, seeded_rand,
                  seeded_randn, to_ncdhw)

pytestmark = pytest.mark.gpu
DEV = "cuda"
FULL = (160, 192, 160)


def _unet(c0=8):
    return UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=c0,
                normalization="batch", upsampling_type="linear", padding=True, activation="PReLU")


# ------------------------------------------------------------------------------------------------ cfg2 volume size
def test_cfg2_unet_full_volume_eval_mask_bit_exact_and_train_step_vs_oracle():
    """unet.UNet c0=8 with the reference's shipped checkpoint on ONE 1x160x192x160 volume: (1) eval logits vs the oracle run live
    on the host and vs the fixture, arg-max mask sha256 equal (bit-exact mask, north_star); (2) a train step: logits and loss
    <= 1e-3, every gradient tensor judged against the fp64 oracle, BatchNorm running statistics, recorded gradient norms."""
    gold = load_golden("unet_c8_ckpt_160x192x160.npz")
    sd = load_ckpt("whole_im_train_seg_parc_epoch_7.pth")
    x = seeded_randn(161, (1, 1) + FULL)
    tgt = (seeded_rand(162, (1, 1) + FULL) < 0.1).float()
    orc = unet_recon.UNetRecon(out_channels_first_layer=8)
    orc.load_state_dict(sd, strict=True)
    prod = _unet(8)
    prod.load_state_dict(sd, strict=True)
    prod.to(DEV).eval()
    orc.eval()
    with torch.no_grad():
        lo_p = prod(x.to(DEV))
        lo_o = orc(x)
    assert_close(to_ncdhw(lo_p), lo_o, rel=1e-3, what="eval logits vs oracle (full volume)")
    assert_close(sample(to_ncdhw(lo_p))[0], gold["eval_sample"], rel=1e-3, what="eval logits vs fixture")
    mask = ops.argmax_mask(lo_p).cpu().numpy()
    ref_mask = lo_o.argmax(dim=1).to(torch.uint8).numpy()
    assert hashlib.sha256(ref_mask.tobytes()).hexdigest() == str(gold["mask_sha256"])          # the oracle reproduces its fixture
    n_diff = int((mask != ref_mask).sum())
    assert n_diff == 0, "%d of %d mask voxels differ from the oracle's" % (n_diff, mask.size)
    assert hashlib.sha256(mask.tobytes()).hexdigest() == str(gold["mask_sha256"]) and int(mask.sum()) == int(gold["mask_sum"])
    del lo_p, lo_o
    torch.cuda.empty_cache()
    # train step (fresh module objects: _compare loads orc's state into prod)
    prod = _unet(8)
    _compare(prod, orc, x, lambda o: ops.softmax_dice_loss(o, tgt.to(DEV)), lambda o: losses.softmax_dice_loss(o, tgt.to(o.dtype)), True)
    np.testing.assert_allclose(float(gold["loss"]), losses.softmax_dice_loss(orc.train()(x), tgt).item(), rtol=1e-5)
    # recorded gradient norms: 1e-2 (the fp64-anchored per-tensor check above is the strict one; the fp32 CPU oracle itself is
    # 3e-3..8e-3 from its fp64 twin on the smallest of these tensors at 4.9 M voxels)
    ok = gold["grad_norms"] > 1e-3 * gold["grad_norms"].max()
    np.testing.assert_allclose(grad_norms(prod)[ok], gold["grad_norms"][ok], rtol=1e-2)


# ------------------------------------------------------------------------------------------------ cfg3
def test_cfg3_autoencoder_mse_step_batch4_full_volume():
    """classification full AE (AE_model.py:123-210, 93_6_4 kwargs), batch 4 x 160x192x160, MSE reconstruction step
    (train_AE.ipynb cell 9) vs the oracle and the fixture recorded from the imported reference module."""
    torch.manual_seed(0)
    orc = O_AE.AE(**AE_KWARGS_93_6_4)
    torch.manual_seed(0)
    prod = P_AE.AE(**AE_KWARGS_93_6_4)
    x = seeded_randn(131, (4, 1) + FULL)
    _compare(prod, orc, x, lambda o: F.mse_loss(o, x.to(DEV)), lambda o: F.mse_loss(o, x.to(o.dtype)), True,
             gold=load_golden("cfg3_ae_mse_b4_160.npz"))


def test_cfg3_encoder_and_classifier_head_ce_step_batch4_full_volume():
    """Encoder (AE(**93_6_4).enc) + Classificator(conv_pad=1, l_in=768) cross-entropy step on 4 x 160x192x160."""
    gold = load_golden("cfg3_enc_clf_ce_b4_160.npz")
    ckw = dict(CLF_KWARGS, conv_pad=1, l_in=768, p_drop=0.0)
    x = seeded_randn(131, (4, 1) + FULL)
    y = torch.tensor([0, 1, 1, 0])
    out = {}
    for tag, A, dev in (("o", O_AE, "cpu"), ("p", P_AE, DEV)):
        torch.manual_seed(0)
        enc, clf = A.AE(**AE_KWARGS_93_6_4).enc, A.Classificator(**ckw)
        enc.to(dev).train(); clf.to(dev).train()
        lat, sizes = enc(x.to(dev))
        logits = clf(lat)
        loss = F.cross_entropy(logits, y.to(dev))
        loss.backward()
        out[tag] = (lat.detach(), logits.detach(), loss.detach(), enc, clf, sizes)
    lat_p = to_ncdhw(out["p"][0])
    assert tuple(lat_p.shape) == (4, 32, 2, 3, 2)
    assert [list(s) for s in out["p"][5]] == gold["sizes"].tolist()
    assert_close(lat_p, out["o"][0], what="latent vs oracle")
    assert_close(lat_p, gold["latent"], what="latent vs reference fixture")
    assert_close(out["p"][1].cpu(), gold["logits"], what="head logits vs reference fixture")
    np.testing.assert_allclose(out["p"][2].item(), float(gold["loss"]), rtol=1e-3)
    assert torch.equal(out["p"][1].argmax(1).cpu(), torch.from_numpy(gold["logits"]).argmax(1))
    for key, idx in (("grad_norms_enc", 3), ("grad_norms_clf", 4)):
        gn = grad_norms(out["p"][idx])
        ok = gold[key] > 1e-3 * gold[key].max()
        np.testing.assert_allclose(gn[ok], gold[key][ok], rtol=3e-2)
    # per-tensor gradients against the oracle's
    for mo, mp in ((out["o"][3], out["p"][3]), (out["o"][4], out["p"][4])):
        gmax = max(p.grad.abs().max().item() for p in mo.parameters())
        for (k, po), pp in zip(mo.named_parameters(), mp.parameters()):
            e = (pp.grad.cpu() - po.grad).abs().max().item()
            assert e <= 3e-2 * po.grad.abs().max().item() + 1e-5 * gmax, (k, e)


# ------------------------------------------------------------------------------------------------ cfg5
def _cnn(mod):
    return torch.nn.Sequential(mod.CNN(input_shape=(32, 32, 32), n_filters=16, n_blocks=3), torch.nn.Linear(128, 2))


def test_cfg5_cnn_32cube_batch64_train_step():
    """cnn_model.CNN (cnn_model.py:104-175) + Linear(128, 2) on 64 x 1x32^3 patches (the per-GPU share of batch 512 over 8 GPUs),
    cross-entropy step vs the oracle and the reference fixture."""
    torch.manual_seed(0)
    orc = _cnn(O_CNN)
    torch.manual_seed(0)
    prod = _cnn(P_CNN)
    x = seeded_randn(151, (64, 1, 32, 32, 32))
    y = torch.arange(64) % 2
    _compare(prod, orc, x, lambda o: F.cross_entropy(o, y.to(DEV)), lambda o: F.cross_entropy(o, y), True,
             gold=load_golden("cfg5_cnn_b64_32.npz"))


def test_cfg5_cnn_batch512_properties_and_batch_split_consistency():
    """The whole cfg5 batch (512 patches) on one GPU: finite, deterministic run to run, and — in eval mode, where BatchNorm does
    not couple patches — no patch's output depends on its neighbours in the batch: a permuted batch gives the permuted rows bit
    for bit (same kernels, same per-voxel summation order), and a batch-64 forward of the first 64 patches agrees to fp32 rounding
    (the dispatcher may pick another kernel for the smaller launch — the fp32 marching kernel takes 16-channel layers from 4 M
    voxels — so the summation order, not the data a row sees, differs)."""
    torch.manual_seed(0)
    m = _cnn(P_CNN).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(512, 1, 32, 32, 32, device=DEV, generator=g)
    y = (torch.arange(512, device=DEV) % 2)
    m.train()
    res = []
    for _ in range(2):
        for mod in m.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.reset_running_stats()
        m.zero_grad(set_to_none=True)
        out = m(x)
        loss = F.cross_entropy(out, y)
        loss.backward()
        res.append((out.detach().clone(), loss.detach().clone(), [p.grad.clone() for p in m.parameters()]))
    assert tuple(res[0][0].shape) == (512, 2) and torch.isfinite(res[0][0]).all()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)
    m.eval()
    with torch.no_grad():
        full = m[0](x)
        perm = torch.randperm(512, device=DEV, generator=g)
        shuffled = m[0](x[perm].contiguous())
        part = m[0](x[:64].contiguous())
    assert torch.equal(shuffled, full[perm])
    scale = float(full.abs().max())
    assert float((full[:64] - part).abs().max()) <= 1e-4 * scale   # fp32 rounding of another summation order, nothing more


# ------------------------------------------------------------------------------------------------ cfg4 (bf16 region, per-GPU share)
def test_cfg4_bf16_step_full_size_against_fp32_hip_run():
    """BASELINE configs[3] per-GPU share: the U-Net step on 2 x 160x192x160 inside the bf16 storage region (bf16 activations, fp32
    accumulate / parameters / statistics / loss) against the SAME step in fp32 on the HIP path: loss within 1e-2 relative,
    logits within 5e-2 of the fp32 range, per-tensor gradient cosine > 0.95, and masks that disagree only where the fp32 logit
    margin is inside bf16 resolution."""
    torch.manual_seed(0)
    m = _unet(8).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(1234)
    x = torch.randn(2, 1, *FULL, device=DEV, generator=g)
    t = (torch.rand(2, 1, *FULL, device=DEV, generator=g) < 0.1).float()
    state = {k: v.clone() for k, v in m.state_dict().items()}

    def run(bf16):
        m.load_state_dict(state)
        m.train()
        m.zero_grad(set_to_none=True)
        with ops.autocast(enabled=bf16):
            lo = m(x)
            loss = ops.softmax_dice_loss(lo, t)
        loss.backward()
        grads = [p.grad.detach().float().clone() for p in m.parameters()]
        m.eval()
        with torch.no_grad(), ops.autocast(enabled=bf16):
            le = m(x[:1])
            mask = ops.argmax_mask(le)
        return lo.detach().float(), loss.item(), grads, le.detach().float(), mask

    lo32, l32, g32, le32, mk32 = run(False)
    lo16, l16, g16, le16, mk16 = run(True)
    assert abs(l16 - l32) <= 1e-2 * abs(l32), (l16, l32)
    rng = (lo32.max() - lo32.min()).item()
    assert (lo16 - lo32).abs().max().item() <= 5e-2 * rng
    gmax = max(a.abs().max().item() for a in g32)
    for k, (a, b) in enumerate(zip(g32, g16)):
        if a.abs().max().item() < 1e-4 * gmax:       # analytically-zero gradients (conv biases feeding a train-mode BatchNorm)
            continue
        cos = torch.nn.functional.cosine_similarity(a.flatten(), b.flatten(), dim=0).item()
        assert cos > 0.95, (k, cos)
    # masks: any disagreement must sit where the fp32 margin is within bf16 resolution of the logit magnitude
    diff = mk32 != mk16
    margin = (le32[:, 1] - le32[:, 0]).abs()
    scale = le32.abs().amax(dim=1)
    assert diff.float().mean().item() < 2e-2
    if diff.any():
        assert bool((margin[diff] <= 4 * 2.0 ** -8 * scale[diff] + 2e-2 * (le32.max() - le32.min())).all())


# ------------------------------------------------------------------------------------------------ a10 train mode
def _masks(seed, n, widths, p=0.6):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(n, c, generator=g) >= p).float() for c in widths]


class _InjectedDropout3d(torch.nn.Module):
    def __init__(self, p, masks):
        super().__init__()
        self.p, self.masks, self.calls = p, masks, 0

    def forward(self, x):
        keep = self.masks[self.calls].to(x)
        self.calls += 1
        return x * (keep / (1.0 - self.p)).view(*keep.shape, 1, 1, 1)


def test_a10_modified3dunet_train_mode_with_injected_dropout_masks(monkeypatch):
    """Modified3DUNet.forward in TRAIN mode (Dropout3d(0.6) active, modified_3dunet.py:114-116): the five Bernoulli channel
    masks are injected into both sides — the oracle's Dropout3d module and the product's `ops.dropout3d` (whose HIP volume pass
    still runs) — so outputs, loss and gradients are comparable; also against the fixture recorded from the reference module."""
    gold = load_golden("modified3dunet_train_b8_48.npz")
    masks = _masks(171, 2, [8, 16, 32, 64, 128])
    torch.manual_seed(0)
    orc = O_M.Modified3DUNet(1, 2, 8)
    orc.dropout3d = _InjectedDropout3d(0.6, masks)
    prod = Modified3DUNet(1, 2, 8)
    state = {"calls": 0}

    def injected(x, p, training):
        assert training and p == 0.6
        keep = masks[state["calls"] % 5].to(x.device)
        state["calls"] += 1
        return ops._ScaleInstanceFn.apply(x, keep / (1.0 - p))

    monkeypatch.setattr(ops, "dropout3d", injected)
    x = seeded_randn(172, (2, 1, 48, 32, 32))
    tgt = (seeded_rand(173, (2, 1, 48, 32, 32)) < 0.2).float()

    def loss_o(o):
        return losses.softmax_dice_loss(o, tgt.to(o.dtype))

    # _compare runs the oracle twice (fp64 copy, then fp32): restart its mask sequence per forward
    orig_forward = O_M.Modified3DUNet.forward

    def fwd(self, inp):
        self.dropout3d.calls = 0
        return orig_forward(self, inp)

    monkeypatch.setattr(O_M.Modified3DUNet, "forward", fwd)
    _compare(prod, orc, x, lambda o: ops.softmax_dice_loss(o, tgt.to(DEV)), loss_o, True, gold=gold, grad_rel=3e-2)
    assert state["calls"] == 5
