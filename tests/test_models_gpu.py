"""GPU parity suite, model level: the HIP-backed drop-in models against (a) the CPU oracle run live on the same
seeded inputs and (b) the committed golden vectors recorded from the reference modules — forward outputs, loss,
per-parameter gradient norms, BatchNorm running statistics, bit-exact argmax masks — plus size-independent
properties at BASELINE.json's full 160x192x160 size."""
import copy
import hashlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from mri_epilepsy_diagnosis_amd import ops, parallel
from mri_epilepsy_diagnosis_amd.classification import routine as clf_routine
from mri_epilepsy_diagnosis_amd.classification.models import AE_model as P_AE, cnn_model as P_CNN
from mri_epilepsy_diagnosis_amd.segmentation import routine
from mri_epilepsy_diagnosis_amd.segmentation.models.modified_3dunet import Modified3DUNet
from mri_epilepsy_diagnosis_amd.unet import UNet
from oracle import ae_model as O_AE, cnn_model as O_CNN, losses, modified_3dunet as O_M, unet_recon
from util import (AE_KWARGS_93_6_4, CLF_KWARGS, DISC_KWARGS, assert_close, grad_norms, load_ckpt, load_golden, rel_err,
                  sample, seeded_rand, seeded_randn, to_ncdhw)

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _unet(c0=8):
    return UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=c0,
                normalization="batch", upsampling_type="linear", padding=True, activation="PReLU")


def _step(model, x, loss_fn, train):
    model.train(train)
    model.zero_grad(set_to_none=True)
    out = model(x)
    out = out[0] if isinstance(out, tuple) else out
    loss = loss_fn(out)
    loss.backward()
    return out.detach(), loss.detach()


def _compare(prod, orc, x, loss_prod, loss_orc, train, gold=None, rel=1e-3, grad_rel=3e-2):
    """prod on GPU vs orc on CPU on the same input; optionally also vs the golden record."""
    prod.load_state_dict(orc.state_dict())
    prod.to(DEV)
    o64 = copy.deepcopy(orc).double()
    _step(o64, x.double(), loss_orc, train)   # loss_orc must be dtype-agnostic
    out_o, l_o = _step(orc, x, loss_orc, train)
    out_p, l_p = _step(prod, x.to(DEV), loss_prod, train)
    out_p = to_ncdhw(out_p) if out_p.dim() == 5 else out_p.cpu()
    assert_close(out_p, out_o, rel=rel, what="output")
    assert_close(l_p.cpu(), l_o, rel=rel, what="loss")
    # Outputs/loss: 1e-3 (north_star).  Gradients are judged against an fp64 run of the oracle.  Per operator the HIP
    # kernels are as accurate as torch's fp32 CPU kernels (tools/op_error_audit.py: 1e-7..2e-6 either way), but a
    # gradient that has travelled back through ten normalisation layers carries those 1e-6 perturbations amplified by
    # the cancellation in sum(du * xhat): both fp32 paths land 1e-4..1e-2 from the fp64 truth, tensor by tensor.
    # On top of that the models are NOT smooth: max-pool arg-max near-ties and PReLU kinks flip under a 1e-6 input
    # perturbation and move individual gradient tensors by 1e-2 even in exact arithmetic
    # (tests/test_oracle_golden.py::test_reference_gradients_are_discontinuous_at_fp32_noise_level).
    # So: the HIP gradient must be within grad_rel (3e-2, max-norm per tensor) of the truth
    # (max-norm, per tensor) OR no further from it than 4x what PyTorch's own fp32 CPU path (the reference's
    # arithmetic) is.  The second clause covers tensors whose true gradient is zero by construction (a conv bias feeding
    # a train-mode BatchNorm), where every fp32 implementation returns rounding noise.
    gmax = max(p.grad.abs().max().item() for p in o64.parameters() if p.grad is not None)
    for (k, p64), po, pp in zip(o64.named_parameters(), orc.parameters(), prod.parameters()):
        if p64.grad is None:
            assert pp.grad is None, k
            continue
        e_cpu = (po.grad.double() - p64.grad).abs().max().item()
        e_hip = (pp.grad.cpu().double() - p64.grad).abs().max().item()
        allowed = max(grad_rel * p64.grad.abs().max().item(), 4.0 * e_cpu, 1e-5 * gmax)
        assert e_hip <= allowed, "grad of %s: |hip-fp64| %.3e > allowed %.3e (|cpu32-fp64| %.3e)" % (k, e_hip, allowed, e_cpu)
    for (k, bo), (_, bp) in zip(orc.named_buffers(), prod.named_buffers()):
        if bo.dtype.is_floating_point:
            assert_close(bp.cpu(), bo, rel=1e-4, what="buffer " + k)
        else:
            assert torch.equal(bp.cpu(), bo), k
    if gold is not None:
        smp, stride = sample(out_p)
        assert stride == int(gold["out_stride"])
        assert_close(smp, gold["out_sample"], rel=rel, what="output vs golden")
        np.testing.assert_allclose(l_p.item(), float(gold["loss"]), rtol=rel)
        gn = grad_norms(prod)
        ok = gold["grad_norms"] > 1e-3 * gold["grad_norms"].max()
        np.testing.assert_allclose(gn[ok], gold["grad_norms"][ok], rtol=grad_rel)


# ------------------------------------------------------------------------------------------------ U-Net
def test_unet_gradients_meet_1e_3_against_fp64_on_an_input_with_a_margin():
    """north_star's 1e-3, for GRADIENTS, where the model is differentiable with a margin.  The general
    comparison above accepts 3e-2 per tensor because max-pool near-ties and PReLU kinks flip under fp32 rounding noise.  Here
    the oracle's BatchNorm scale / shift (gamma 4, beta 12: activations 3 sigma away from the kink) and a searched seed give an
    input on which every PReLU input is at least 1e-3 from zero and every MaxPool3d window's maximum leads by at least 1e-3
    (asserted on the float64 oracle): every parameter gradient of the HIP model must then be within 1e-3 (max-norm, per
    tensor) of the float64 gradient."""
    from util import kink_and_pool_margins
    torch.manual_seed(24)
    orc = unet_recon.UNetRecon(out_channels_first_layer=8)
    with torch.no_grad():
        for mod in orc.modules():
            if isinstance(mod, torch.nn.BatchNorm3d):
                mod.bias.fill_(12.0)
                mod.weight.fill_(4.0)
        orc.encoder.encoding_blocks[0].conv1.conv_layer.bias.fill_(7.2)
    x = torch.randn(1, 1, 8, 8, 16, generator=torch.Generator().manual_seed(1024))
    tgt = (seeded_rand(7, (1, 1, 8, 8, 16)) < 0.3).float()
    o64 = copy.deepcopy(orc).double().train()
    act_margin, pool_margin = kink_and_pool_margins(o64, x.double())
    assert act_margin >= 1e-3 and pool_margin >= 1e-3, (act_margin, pool_margin)
    o64 = copy.deepcopy(orc).double()   # (the margin pass advanced the running statistics of the copy above)
    _step(o64, x.double(), lambda o: losses.softmax_dice_loss(o, tgt.double()), True)
    prod = _unet(8)
    prod.load_state_dict(orc.state_dict())
    prod.to(DEV)
    _step(prod, x.to(DEV), lambda o: ops.softmax_dice_loss(o, tgt.to(DEV)), True)
    gmax = max(p.grad.abs().max().item() for p in o64.parameters() if p.grad is not None)
    worst = 0.0
    for (k, p64), pp in zip(o64.named_parameters(), prod.parameters()):
        g64 = p64.grad
        if g64.abs().max().item() < 1e-6 * gmax:
            continue   # zero by construction (a conv bias in front of a train-mode BatchNorm): rounding noise in any fp32 implementation
        e = (pp.grad.cpu().double() - g64).abs().max().item() / g64.abs().max().item()
        worst = max(worst, e)
        assert e <= 1e-3, "grad of %s: max-norm relative error %.3e against float64" % (k, e)
    print("worst per-tensor gradient error against float64: %.2e" % worst)


def test_unet_checkpoint_eval_logits_and_bit_exact_mask():
    gold = load_golden("unet_c8_ckpt_32.npz")
    m = _unet(8)
    m.load_state_dict(load_ckpt("whole_im_train_seg_parc_epoch_7.pth"), strict=True)
    m.to(DEV).eval()
    x = seeded_randn(61, (1, 1, 32, 32, 32))
    with torch.no_grad():
        lo = m(x.to(DEV))
    assert_close(sample(to_ncdhw(lo))[0], gold["eval_sample"], what="eval logits vs golden")
    mask = ops.argmax_mask(lo).cpu().numpy()
    assert hashlib.sha256(mask.tobytes()).hexdigest() == str(gold["mask_sha256"])   # bit-exact argmax mask
    assert int(mask.sum()) == int(gold["mask_sum"])


def test_unet_checkpoint_train_step_vs_oracle_and_golden():
    gold = load_golden("unet_c8_ckpt_32.npz")
    orc = unet_recon.UNetRecon(out_channels_first_layer=8)
    orc.load_state_dict(load_ckpt("whole_im_train_seg_parc_epoch_7.pth"), strict=True)
    x = seeded_randn(61, (1, 1, 32, 32, 32))
    tgt = (seeded_rand(62, (1, 1, 32, 32, 32)) < 0.1).float()
    prod = _unet(8)
    _compare(prod, orc, x, lambda o: ops.softmax_dice_loss(o, tgt.to(DEV)), lambda o: losses.softmax_dice_loss(o, tgt.to(o.dtype)), True)
    ok = gold["grad_norms"] > 1e-3 * gold["grad_norms"].max()
    np.testing.assert_allclose(grad_norms(prod)[ok], gold["grad_norms"][ok], rtol=5e-3)
    bn = prod.encoder.encoding_blocks[0].conv2.norm_layer
    assert_close(bn.running_mean.cpu(), gold["running_mean_b0c2"], rel=1e-4)
    assert_close(bn.running_var.cpu(), gold["running_var_b0c2"], rel=1e-4)
    assert int(bn.num_batches_tracked) == 928 * 7 + 1


@pytest.mark.parametrize("c0,shape", [(8, (2, 1, 32, 48, 32)), (16, (1, 1, 32, 32, 32)), (8, (1, 1, 20, 28, 36))])
def test_unet_fresh_train_step_vs_oracle(c0, shape):
    torch.manual_seed(0)
    orc = unet_recon.UNetRecon(out_channels_first_layer=c0)
    x = seeded_randn(5, shape)
    tgt = (seeded_rand(6, shape) < 0.1).float()
    _compare(_unet(c0), orc, x, lambda o: ops.softmax_dice_loss(o, tgt.to(DEV)),
             lambda o: losses.softmax_dice_loss(o, tgt.to(o.dtype)), True)


def test_unet_loss_trajectory_matches_golden():
    """routine.run_epoch on the device (AdamW, 3 seeded iterations) reproduces the recorded CPU trajectory."""
    gold = load_golden("unet_c8_traj_32.npz")
    batches = [{routine.MRI: {routine.DATA: seeded_randn(70 + i, (1, 1, 32, 32, 32))},
                routine.LABEL: {routine.DATA: (seeded_rand(80 + i, (1, 1, 32, 32, 32)) < 0.1).float()}} for i in range(3)]
    model, opt, _ = routine.get_model_and_optimizer(DEV, out_channels_first_layer=8)
    got = routine.run_epoch(1, routine.Action.TRAIN, batches, model, opt)
    np.testing.assert_allclose(got, gold["losses"], rtol=1e-3)


def test_unet_flat_adam_data_parallel_step_matches_torch_adamw():
    torch.manual_seed(0)
    a, b = _unet(8).to(DEV), _unet(8)
    b.load_state_dict(a.state_dict()); b.to(DEV)
    fp = parallel.FlatParams(b)
    oa, ob = torch.optim.AdamW(a.parameters()), parallel.FlatAdam(fp)
    for it in range(2):
        x = seeded_randn(30 + it, (1, 1, 16, 16, 16)).to(DEV)
        t = (seeded_rand(40 + it, (1, 1, 16, 16, 16)) < 0.1).float().to(DEV)
        oa.zero_grad(); ops.softmax_dice_loss(a(x), t).backward(); oa.step()
        ob.zero_grad(); ops.softmax_dice_loss(b(x), t).backward(); ob.step(fp.all_reduce())
    # Same kernels, same gradients at step 1 (bit for bit); the two optimizers round differently at the 1e-7 level, and
    # Adam's first steps are sign-like (update ~ lr * g/|g|): a gradient entry that is analytically zero (conv biases in
    # front of BatchNorm) can flip sign on that noise and move by 2*lr.  So: every entry within the 2-step Adam bound, and
    # all but a sliver of them equal to 1e-5.
    lr, steps = 1e-3, 2
    tot = off = 0
    for pa, pb in zip(a.parameters(), b.parameters()):
        d = (pa - pb).abs()
        assert d.max().item() <= 2 * lr * steps * 1.05 + 1e-6
        tot += d.numel()
        off += int((d > 1e-5).sum().item())
    assert off <= 0.01 * tot, (off, tot)


# ------------------------------------------------------------------------------------------------ classification
@pytest.mark.parametrize("name,seed,shape", [("ae_93_6_4_64", 31, (2, 1, 64, 64, 64)), ("ae_93_6_4_odd", 32, (2, 1, 72, 80, 68))])
def test_ae_reconstruction_step(name, seed, shape):
    torch.manual_seed(0)
    orc = O_AE.AE(**AE_KWARGS_93_6_4)
    torch.manual_seed(0)
    prod = P_AE.AE(**AE_KWARGS_93_6_4)
    x = seeded_randn(seed, shape)
    _compare(prod, orc, x, lambda o: F.mse_loss(o, x.to(DEV)), lambda o: F.mse_loss(o, x.to(o.dtype)), True, gold=load_golden(name + ".npz"))


def test_encoder_clf_disc_checkpoints_192():
    gold = load_golden("enc_clf_disc_ckpt_192.npz")
    enc = P_AE.AE(**AE_KWARGS_93_6_4).enc
    clf, disc = P_AE.Classificator(**CLF_KWARGS), P_AE.Discriminator(**DISC_KWARGS)
    enc.load_state_dict(load_ckpt("encoder_93_6_4.pth"), strict=True)
    clf.load_state_dict(load_ckpt("clf_93_6_4.pth"), strict=True)
    disc.load_state_dict(load_ckpt("disc_93_6_4.pth"), strict=True)
    enc.to(DEV).eval(); clf.to(DEV).eval(); disc.to(DEV).eval()
    x = seeded_randn(41, (1, 1, 192, 192, 192)).to(DEV)
    with torch.no_grad():
        lat, sizes = enc(x)
        c, d = clf(lat), disc(lat)
    assert [list(s) for s in sizes] == gold["sizes"].tolist()
    assert_close(to_ncdhw(lat), gold["latent"], what="latent")
    assert_close(c.cpu(), gold["clf"], what="clf logits")
    assert_close(d.cpu(), gold["disc"], what="disc logits")
    assert c.argmax(1).item() == int(np.argmax(gold["clf"]))


def test_classifier_head_fails_at_160x192x160_like_reference():
    enc = P_AE.AE(**AE_KWARGS_93_6_4).enc.to(DEV).eval()
    clf = P_AE.Classificator(**CLF_KWARGS).to(DEV).eval()
    with torch.no_grad():
        lat, _ = enc(torch.zeros(1, 1, 160, 192, 160, device=DEV))
        assert tuple(lat.shape) == (1, 32, 2, 3, 2)
        with pytest.raises(RuntimeError, match="greater than actual input size"):
            clf(lat)


def test_adversarial_step_vs_oracle():
    """One fader-style batch (train_ENC_CLF.ipynb cell 16) on device vs the same loop with oracle modules on CPU."""
    kw = dict(DISC_KWARGS, conv_pad=1, l_in=64)
    ckw = dict(CLF_KWARGS, conv_pad=1, l_in=64)
    mods = {}
    for tag, A in (("o", O_AE), ("p", P_AE)):
        torch.manual_seed(0)
        mods[tag] = (A.AE(**AE_KWARGS_93_6_4).enc, A.Discriminator(**kw), A.Classificator(**ckw))
    for mo, mp in zip(mods["o"], mods["p"]):
        mp.load_state_dict(mo.state_dict()); mp.to(DEV)
    x = seeded_randn(9, (4, 1, 64, 64, 64))
    y, dom = torch.tensor([0, 1, 1, 0]), torch.tensor([3, 0, 17, 5])
    res = {}
    for tag, dev in (("o", "cpu"), ("p", DEV)):
        enc, disc, clf = mods[tag]
        for m in (disc, clf):  # dropout off for a deterministic comparison
            for s in m.modules():
                if isinstance(s, torch.nn.Dropout):
                    s.p = 0.0
        o1 = torch.optim.Adam(list(enc.parameters()) + list(clf.parameters()), lr=7e-4, weight_decay=1e-4)
        o2 = torch.optim.Adam(disc.parameters(), lr=5e-4, weight_decay=1e-4)
        ce = torch.nn.CrossEntropyLoss()
        res[tag] = clf_routine.adversarial_step(enc, disc, clf, x.to(dev), y.to(dev), dom.to(dev), ce, ce, o1, o2, 0.05, 18, n_d=2)
    for a, b in zip(res["p"], res["o"]):
        assert_close(a.cpu(), b, rel=1e-3)
    # Adam's first step moves every element by ~lr*sign(g): elements whose true gradient is rounding noise may move in
    # opposite directions, so parameters are compared with an absolute tolerance of 2.5*lr.
    for po, pp in zip(mods["o"][0].parameters(), mods["p"][0].parameters()):
        assert (pp.detach().cpu() - po.detach()).abs().max().item() <= 2.5 * 7e-4


@pytest.mark.parametrize("name,cls,kw,shape", [
    ("cnn_32", "CNN", dict(input_shape=(32, 32, 32), n_filters=16, n_blocks=3), (4, 1, 32, 32, 32)),
    ("voxresnet_32", "VoxResNet", dict(input_shape=(32, 32, 32), n_filters=8, n_blocks=3), (3, 1, 32, 32, 32)),
    ("dilatedcnn_180", "DilatedCNN", dict(input_shape=(180, 180, 180), n_channels=2), (2, 1, 180, 180, 180)),
])
def test_cnn_family_train_step(name, cls, kw, shape):
    torch.manual_seed(0)
    orc = getattr(O_CNN, cls)(**kw)
    torch.manual_seed(0)
    prod = getattr(P_CNN, cls)(**kw)
    x = seeded_randn(21, shape)
    y = torch.arange(shape[0]) % 2
    _compare(prod, orc, x, lambda o: F.cross_entropy(o[:, :2], y.to(DEV)), lambda o: F.cross_entropy(o[:, :2], y), True,
             gold=load_golden(name + ".npz"), grad_rel=3e-2)


# ------------------------------------------------------------------------------------------------ Modified3DUNet
def test_modified3dunet_eval_step():
    torch.manual_seed(0)
    orc = O_M.Modified3DUNet(1, 2, 8)
    prod = Modified3DUNet(1, 2, 8)
    x = seeded_randn(11, (1, 1, 32, 32, 32))
    tgt = (seeded_rand(12, (1, 1, 32, 32, 32)) < 0.2).float()
    _compare(prod, orc, x, lambda o: ops.softmax_dice_loss(o, tgt.to(DEV)), lambda o: losses.softmax_dice_loss(o, tgt.to(o.dtype)), False,
             gold=load_golden("modified3dunet_b8_32.npz"), grad_rel=3e-2)


def test_modified3dunet_train_mode_runs_with_dropout():
    m = Modified3DUNet(1, 2, 8).to(DEV).train()
    out = m(seeded_randn(1, (2, 1, 32, 32, 32)).to(DEV))
    out.float().mean().backward()
    assert tuple(out.shape) == (2, 2, 32, 32, 32) and torch.isfinite(out).all()


# ------------------------------------------------------------------------------------------------ unet3d.py blocks
@pytest.mark.parametrize("norm", ["gn", "bn", "in"])
def test_unet3d_blocks_vs_oracle_and_golden(norm):
    from mri_epilepsy_diagnosis_amd.segmentation.models import unet3d as P_U3
    from oracle import unet3d_blocks as O_U3
    gold = load_golden("unet3d_blocks.npz")
    torch.manual_seed(0); od = O_U3.ConvD(4, 8, norm=norm)
    torch.manual_seed(0); pd_ = P_U3.ConvD(4, 8, norm=norm)
    x = seeded_randn(1, (2, 4, 16, 16, 16))
    _compare(pd_, od, x, lambda o: o.square().mean(), lambda o: o.square().mean(), True)
    with torch.no_grad():
        pd_.load_state_dict(od.state_dict())   # _compare updated BN running stats in both; re-sync and compare to golden
    torch.manual_seed(0); fresh = P_U3.ConvD(4, 8, norm=norm).to(DEV)
    assert_close(to_ncdhw(fresh(x.to(DEV))).flatten()[::37], gold["convd_" + norm], what="ConvD vs golden")
    torch.manual_seed(0); ou = O_U3.ConvU(8, norm=norm)
    torch.manual_seed(0); pu = P_U3.ConvU(8, norm=norm).to(DEV)
    prev, xin = seeded_randn(2, (2, 4, 16, 16, 16)), seeded_randn(3, (2, 16, 8, 8, 8))
    yo = ou(xin, prev)
    yp = pu(xin.to(DEV), prev.to(DEV))
    assert_close(to_ncdhw(yp), yo.detach(), what="ConvU vs oracle")
    assert_close(to_ncdhw(yp).flatten()[::37], gold["convu_" + norm], what="ConvU vs golden")
    yo.square().mean().backward(); yp.square().mean().backward()
    for (k, po), pp in zip(ou.named_parameters(), pu.parameters()):
        assert_close(pp.grad.cpu(), po.grad, rel=5e-3, what="ConvU grad " + k)


# ------------------------------------------------------------------------------------------------ full-size properties
def test_full_size_unet_properties_160x192x160():
    """BASELINE configs[1] size (batch 2, fp32): shapes, finiteness, run-to-run bit determinism, loss in range,
    eval-mode argmax consistency between the fused mask kernel and the logits, gradient/bias identities."""
    torch.manual_seed(0)
    m = _unet(8).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(1234)
    x = torch.randn(2, 1, 160, 192, 160, device=DEV, generator=g)
    t = (torch.rand(2, 1, 160, 192, 160, device=DEV, generator=g) < 0.1).float()

    def step():
        m.zero_grad(set_to_none=True)
        lo = m(x)
        loss = ops.softmax_dice_loss(lo, t)
        loss.backward()
        return lo.detach(), loss.detach(), [p.grad.clone() for p in m.parameters()]

    m.train()
    lo1, l1, g1 = step()
    for mod in m.modules():  # undo the running-stat update so that the second pass sees identical state
        if isinstance(mod, torch.nn.BatchNorm3d):
            mod.reset_running_stats()
    lo2, l2, g2 = step()
    assert tuple(lo1.shape) == (2, 2, 160, 192, 160) and torch.isfinite(lo1).all()
    assert 0.0 < l1.item() < 1.0
    assert torch.equal(lo1, lo2) and torch.equal(l1, l2)          # deterministic kernels: bit-identical reruns
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    # conv bias feeding a train-mode BatchNorm gets (numerically) zero gradient; classifier bias grads sum to ~0
    cb = m.decoder.decoding_blocks[1].conv2.conv_layer.bias.grad
    cw = m.decoder.decoding_blocks[1].conv2.conv_layer.weight.grad
    assert cb.abs().max().item() <= 1e-3 * cw.abs().max().item() + 1e-6
    assert abs(m.classifier.conv_layer.bias.grad.sum().item()) <= 1e-4 * m.classifier.conv_layer.bias.grad.abs().max().item() + 1e-7
    m.eval()
    with torch.no_grad():
        lo = m(x[:1])
        mask = ops.argmax_mask(lo)
    ref = (lo[:, 1] > lo[:, 0]).to(torch.uint8)
    assert torch.equal(mask, ref)


def test_full_size_layer_parity_48_16_slab():
    """The heaviest layer (decoder 48->16 3x3x3) on a full-resolution slab (in-plane 192x160), against torch CPU."""
    x = seeded_randn(1, (1, 48, 6, 192, 160))
    conv = torch.nn.Conv3d(48, 16, 3, padding=1)
    xr = x.clone().requires_grad_(True)
    yr = conv(xr)
    gy = seeded_randn(2, tuple(yr.shape))
    yr.backward(gy)
    xd = x.to(DEV).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    w = conv.weight.detach().to(DEV).requires_grad_(True)
    b = conv.bias.detach().to(DEV).requires_grad_(True)
    yd = ops.conv3d(xd, w, b, 1, 1, 1)
    yd.backward(gy.to(DEV))
    assert_close(to_ncdhw(yd), yr, what="y")
    assert_close(to_ncdhw(xd.grad), xr.grad, what="dx")
    assert_close(w.grad.cpu(), conv.weight.grad, what="dw")
    assert_close(b.grad.cpu(), conv.bias.grad, what="db")


@pytest.mark.parametrize("bf16", [False, True])
def test_captured_step_replays_eager_step_bit_exactly(bf16):
    """parallel.CapturedStep: forward+backward recorded once into a hipGraph must reproduce the eagerly launched step
    (same kernels, deterministic reductions => identical loss and flat gradient), also after the inputs change in place."""
    torch.manual_seed(3)
    net = _unet(8).to(DEV)
    flat = parallel.FlatParams(net)
    x = torch.randn(1, 1, 32, 32, 32, device=DEV)
    t = (torch.rand(1, 1, 32, 32, 32, device=DEV) < 0.2).float()

    def loss_fn():
        with ops.autocast(enabled=bf16):
            return ops.softmax_dice_loss(net(x), t)

    cap = parallel.CapturedStep(flat, loss_fn).capture()
    state = {k: v.clone() for k, v in net.state_dict().items()}   # BN running statistics advance with every step
    for trial in range(2):
        if trial == 1:
            x.copy_(torch.randn_like(x))
            t.copy_((torch.rand_like(t) < 0.3).float())
        net.load_state_dict(state)
        l_g = cap.run().clone()
        g_g = flat.grad.clone()
        rm_g = net.encoder.encoding_blocks[0].conv2.norm_layer.running_mean.clone()
        net.load_state_dict(state)
        l_e = cap._eager().clone()
        assert torch.equal(l_g, l_e), (l_g.item(), l_e.item())
        assert torch.equal(g_g, flat.grad)
        assert torch.equal(rm_g, net.encoder.encoding_blocks[0].conv2.norm_layer.running_mean)
    assert g_g.abs().max().item() > 0


def test_validate_dsc_asd_device_counts_equal_host_metrics():
    """validate_dsc_asd (segmentation/routine.py:217-237): Dice / IoU from the on-device overlap counts must equal the
    reference's host arithmetic on the downloaded masks, sample by sample, with the shipped checkpoint's predictions."""
    from oracle import metrics as O_MET
    net = _unet(8)
    net.load_state_dict(load_ckpt("whole_im_train_seg_parc_epoch_7.pth"))
    net.to(DEV)
    loader = routine.synthetic_loader(3, 1, (32, 48, 32), seed=5, foreground=0.3)
    dsc, asd_mean, asd_std, iou = routine.validate_dsc_asd(net, loader)
    assert len(dsc) == 3
    net.eval()
    for k, batch in enumerate(routine.synthetic_loader(3, 1, (32, 48, 32), seed=5, foreground=0.3)):
        with torch.no_grad():
            pred = ops.argmax_mask(net(batch[routine.MRI][routine.DATA].to(DEV)))[0].cpu().numpy()
        gt = batch[routine.LABEL][routine.DATA].numpy().astype(np.uint8)[0][0]
        assert dsc[k] == O_MET.dice_coefficient(gt, pred)
        assert iou[k] == O_MET.iou_score(pred, gt)
        ref_a, ref_b = O_MET.average_surface_distance(gt, pred, load_golden("surface_asd.npz")["area_table"])
        assert np.isclose(asd_mean[k], ref_a, rtol=1e-12, atol=0, equal_nan=True)
        assert np.isclose(asd_std[k], ref_b, rtol=1e-12, atol=0, equal_nan=True)
    # the host path (surface metrics requested) gives the same Dice / IoU
    d2, _, _, i2 = routine.validate_dsc_asd(net, routine.synthetic_loader(3, 1, (32, 48, 32), seed=5, foreground=0.3),
                                            surface_metrics=lambda s, p: (0.0, 0.0))
    assert d2 == dsc and i2 == iou


def test_flat_params_gradient_sinks_match_plain_autograd():
    """parallel.FlatParams makes the ops write parameter gradients straight into the flat buffer (ops.GradSink) instead of
    through AccumulateGrad: same numbers as plain autograd, shared parameters and repeated backward calls accumulate, and a
    dropped .grad falls back to autograd's own path."""
    from mri_epilepsy_diagnosis_amd import parallel
    from mri_epilepsy_diagnosis_amd.segmentation.models.modified_3dunet import Modified3DUNet
    torch.manual_seed(0)
    x = torch.randn(1, 1, 16, 16, 16, device="cuda")
    t = (torch.rand(1, 1, 16, 16, 16, device="cuda") < 0.3).float()
    for make in (lambda: UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=2, out_channels_first_layer=8,
                              normalization="batch", upsampling_type="linear", padding=True, activation="PReLU"),
                 lambda: Modified3DUNet(1, 2, 8)):          # the second one applies some convs twice (shared weights)
        torch.manual_seed(1)
        a = make().cuda()
        torch.manual_seed(1)
        b = make().cuda()
        for m in (a, b):
            m.eval()                                          # no dropout randomness between the two copies
        ops.softmax_dice_loss(a(x), t).backward()
        flat = parallel.FlatParams(b)
        flat.zero_grad()
        ops.softmax_dice_loss(b(x), t).backward()
        for (name, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert pb.grad.data_ptr() >= flat.grad.data_ptr() and torch.equal(pa.grad, pb.grad), name
        # a second backward without zero_grad accumulates; with zero_grad it starts over
        ops.softmax_dice_loss(b(x), t).backward()
        for (name, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert torch.allclose(pb.grad, 2 * pa.grad, rtol=1e-5, atol=1e-9), name    # (g + g1) + g2 vs 2 g: rounding only
        flat.zero_grad()
        ops.softmax_dice_loss(b(x), t).backward()
        for (name, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert torch.equal(pa.grad, pb.grad), name
        # a caller that drops .grad gets autograd's own gradient tensors
        for p in b.parameters():
            p.grad = None
        ops.softmax_dice_loss(b(x), t).backward()
        for (name, pa), pb in zip(a.named_parameters(), b.parameters()):
            assert torch.equal(pa.grad, pb.grad), name


def test_autograd_grad_over_attached_parameters_with_suspended_sinks():
    """GradSink contract: with FlatParams attached, `torch.autograd.grad` inside `ops.suspend_grad_sinks()` returns the gradient
    tensors and leaves `.grad` alone; outside it, backward() fills `.grad` with the same numbers."""
    from mri_epilepsy_diagnosis_amd import nn as mnn
    torch.manual_seed(2)
    m = torch.nn.Sequential(mnn.Conv3d(4, 8, 3, padding=1), mnn.BatchNorm3d(8), mnn.ReLU(), mnn.Conv3d(8, 2, 1)).cuda()
    flat = parallel.FlatParams(m)
    x = torch.randn(2, 4, 6, 7, 9, device="cuda")
    flat.zero_grad()
    m(x).square().mean().backward()
    want = [p.grad.detach().clone() for p in m.parameters()]
    flat.zero_grad()
    with ops.suspend_grad_sinks():
        got = torch.autograd.grad(m(x).square().mean(), list(m.parameters()))
    assert all(g is not None for g in got)
    assert all(torch.allclose(g, w, rtol=1e-5, atol=1e-7) for g, w in zip(got, want))
    assert float(flat.grad.abs().max()) == 0.0          # untouched
    flat.zero_grad()
    m(x).square().mean().backward()                      # and the sinks are live again afterwards
    assert all(torch.allclose(p.grad, w, rtol=1e-5, atol=1e-7) for p, w in zip(m.parameters(), want))


class _OtherRank:
    """Stands in for the second data-parallel rank of a synchronised BatchNorm: adds that rank's contribution (computed
    analytically by the test) to each collective, in call order."""

    def __init__(self, contributions):
        self.contributions, self.calls = list(contributions), 0

    def all_reduce(self, t):
        out = t + self.contributions[self.calls].to(t)
        self.calls += 1
        return out


@pytest.mark.parametrize("warm", [False, True], ids=["cold_running_mean", "warm_running_mean"])
def test_sync_batchnorm_keeps_its_digits_when_the_mean_dwarfs_the_spread(warm):
    """Channels with |mean| = 100 x std (un-normalised intensities), the two ranks' local means apart by a fraction of a std: the
    merged variance must agree with the float64 variance of the pooled batch to 1e-5 relative — with the shift at 0 (first step)
    and at the running mean (every later step).  The wire format is (count, sum(x - s), sum((x - s)^2)), s = running_mean."""
    torch.manual_seed(11)
    c, shape = 8, (6, 10, 12)
    x = torch.randn(4, c, *shape) * 0.05 + 5.0
    x[2:] += 0.02                                   # the other rank sees a slightly different population
    s0 = torch.full((c,), 5.0) if warm else torch.zeros(c)
    xb = x[2:].double()
    red = (0, 2, 3, 4)
    sh = s0.double().view(1, c, 1, 1, 1)
    fwd_b = torch.cat([torch.tensor([float(xb.numel() // c)], dtype=torch.float64), (xb - sh).sum(red), ((xb - sh) ** 2).sum(red)])
    reducer = _OtherRank([fwd_b])
    xd = x[:2].cuda().contiguous(memory_format=torch.channels_last_3d)
    rm, rv = s0.clone().cuda(), torch.ones(c, device="cuda")
    gamma, beta = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    prev = ops.set_sync_batchnorm(reducer)
    try:
        with torch.no_grad():
            y = ops.norm_act(xd, gamma, beta, None, rm, rv, "sync", 1.0, 0.0, None)   # momentum 1: running = this batch's
    finally:
        ops.set_sync_batchnorm(prev)
    mean64, var64 = x.double().mean(red), x.double().var(red, unbiased=True)
    assert torch.allclose(rm.cpu().double(), mean64, rtol=2e-7, atol=0)
    assert torch.allclose(rv.cpu().double(), var64, rtol=1e-5, atol=0), ((rv.cpu().double() - var64) / var64).abs().max()
    want = ((x[:2].double() - mean64.view(1, c, 1, 1, 1)) / x.double().var(red, unbiased=False).sqrt().view(1, c, 1, 1, 1))
    assert torch.allclose(y.cpu().double(), want, rtol=0, atol=2e-4)   # x itself is fp32: (x - mean)/std carries 100 x eps_fp32


@pytest.mark.parametrize("act", [None, "relu"])
def test_sync_batchnorm_equals_the_global_batch(act):
    """SURVEY §8e parity mode: one rank holding half of the batch, with the other half's sums arriving through the reducer,
    must produce what a single device computes on the whole batch — outputs, input gradients, running statistics — and its
    parameter gradients must be its share of the global ones."""
    torch.manual_seed(3)
    c, shape = 8, (6, 5, 7)
    x = torch.randn(4, c, *shape) * 2.0 + 0.5
    dy = torch.randn(4, c, *shape)
    gamma, beta = torch.rand(c) + 0.5, torch.randn(c)

    def run(xs, dys, reducer):
        xd = xs.cuda().contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        g, b = gamma.cuda().requires_grad_(True), beta.cuda().requires_grad_(True)
        rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        prev = ops.set_sync_batchnorm(reducer)
        try:
            y = ops.norm_act(xd, g, b, None, rm, rv, "sync" if reducer is not None else "batch", 0.1, 1e-5, act)
            y.backward(dys.cuda().contiguous(memory_format=torch.channels_last_3d))
        finally:
            ops.set_sync_batchnorm(prev)
        return y.detach().cpu(), xd.grad.cpu(), g.grad.cpu(), b.grad.cpu(), rm.cpu(), rv.cpu()

    y_all, dx_all, dg_all, db_all, rm_all, rv_all = run(x, dy, None)
    # the other rank's contributions, in float64 torch on the CPU
    xb, dyb = x[2:].double(), dy[2:].double()
    red = (0, 2, 3, 4)
    cnt_b = torch.tensor([float(xb.numel() // c)], dtype=torch.float64)
    fwd_b = torch.cat([cnt_b, xb.sum(red), (xb * xb).sum(red)])
    n_all = x.numel() // c
    mean = x.double().mean(red)
    var = x.double().var(red, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    xhat_b = (xb - mean.view(1, c, 1, 1, 1)) * invstd.view(1, c, 1, 1, 1)
    z_b = xhat_b * gamma.double().view(1, c, 1, 1, 1) + beta.double().view(1, c, 1, 1, 1)
    dyp_b = dyb * (z_b > 0) if act == "relu" else dyb
    bwd_b = torch.cat([dyp_b.sum(red), (dyp_b * xhat_b).sum(red)])
    reducer = _OtherRank([fwd_b, bwd_b])
    y_a, dx_a, dg_a, db_a, rm_a, rv_a = run(x[:2], dy[:2], reducer)
    assert reducer.calls == 2
    assert torch.allclose(y_a, y_all[:2], rtol=1e-5, atol=1e-5)
    assert torch.allclose(dx_a, dx_all[:2], rtol=1e-4, atol=2e-6)
    assert torch.allclose(rm_a, rm_all, rtol=1e-5, atol=1e-6) and torch.allclose(rv_a, rv_all, rtol=1e-5, atol=1e-6)
    # parameter gradients are per-rank sums: this rank's share + the other rank's share = the global gradient
    assert torch.allclose(db_a.double() + bwd_b[:c], db_all.double(), rtol=1e-4, atol=1e-4)
    assert torch.allclose(dg_a.double() + bwd_b[c:], dg_all.double(), rtol=1e-4, atol=1e-4)


def test_sync_batchnorm_with_one_rank_is_local_batchnorm_and_reaches_the_modules():
    """world_size 1: the reducer is the identity, so `with parallel.SyncBatchNorm():` must not change a U-Net step."""
    torch.manual_seed(0)
    x = torch.randn(2, 1, 16, 16, 16, device="cuda")
    t = (torch.rand(2, 1, 16, 16, 16, device="cuda") < 0.3).float()
    outs = []
    for sync in (False, True):
        torch.manual_seed(1)
        m = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=2, out_channels_first_layer=8,
                 normalization="batch", upsampling_type="linear", padding=True, activation="PReLU").cuda().train()
        if sync:
            with parallel.SyncBatchNorm() as red:
                assert ops.sync_batchnorm_reducer() is red
                loss = ops.softmax_dice_loss(m(x), t)
                loss.backward()
            assert ops.sync_batchnorm_reducer() is None
        else:
            loss = ops.softmax_dice_loss(m(x), t)
            loss.backward()
        outs.append((loss.item(), [p.grad.clone() for p in m.parameters()], [b.clone() for b in m.buffers()]))
    assert abs(outs[0][0] - outs[1][0]) < 1e-6
    for ga, gb in zip(outs[0][1], outs[1][1]):
        assert torch.allclose(ga, gb, rtol=2e-4, atol=1e-6)
    for ba, bb in zip(outs[0][2], outs[1][2]):
        assert torch.allclose(ba.float(), bb.float(), rtol=1e-5, atol=1e-6)


def test_two_ranks_on_one_gpu_with_sync_batchnorm_match_one_process_with_both_volumes(tmp_path):
    """The data-parallel path on the real kernels: two processes (sharing this GPU, gloo for the collectives) each take one
    volume, synchronise their BatchNorm statistics, all-reduce the flat gradient and step the fused AdamW — and end up with
    the parameters, gradients and running statistics of ONE process that trained on both volumes as a batch of two."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "_ddp_gpu_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", PYTHONPATH=root)
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(tmp_path)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    r0, r1, one = (torch.load(tmp_path / n) for n in ("rank0.pt", "rank1.pt", "single.pt"))
    assert torch.equal(r0["params"], r1["params"]) and torch.equal(r0["grads"], r1["grads"])     # replicas stay in lock-step
    assert torch.allclose(r0["bufs"], one["bufs"], rtol=1e-5, atol=1e-6)                         # BatchNorm running statistics
    gscale = one["grads"].abs().max().item()
    assert (r0["grads"] - one["grads"]).abs().max().item() <= 2e-4 * gscale
    # one AdamW step moves every parameter by ~lr in the direction of the gradient's sign: compare where the gradient is not tiny
    big = one["grads"].abs() > 1e-3 * gscale
    assert torch.allclose(r0["params"][big], one["params"][big], rtol=0, atol=2e-4)


def test_bench_multi_rank_path_rehearsed_on_one_gpu():
    """`bench.py --gpus 2` as the driver launches it (torch.distributed.run, one process per rank), rehearsed on this one-GPU
    box: both ranks share cuda:0 and gloo carries the collectives (MRI3D_BENCH_ONE_GPU_REHEARSAL).  Checks the contract of the
    JSON line for N > 1 — exactly one line, from rank 0, whole-job aggregate, weak scaling — not its value."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MRI3D_BENCH_ONE_GPU_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and abs(out["value"] - 4 * 2 / (out["ms_per_step"] * 2 / 1e3)) < 1e-2 * out["value"]
    # two processes time-slice the one card here, so WHICH operator collects the most event time (and its rate) is arbitrary:
    # only the shape of the roofline object is checked
    assert "cpu_baseline" not in out and out["roofline"]["frac"] >= 0 and out["roofline"]["kernel"]
    assert out["ranks_seen"] == 2 and out["backend"] == "gloo"


def test_bench_gpus_2_self_launches_two_ranks_on_one_gpu():
    """The form the driver's N=1 run used — plain `python bench.py --gpus 2`, no torchrun, no WORLD_SIZE — must start its own
    two ranks (a fresh child process per rank; the parent never touches the GPU) and report n_gpus == ranks_seen == 2."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MRI3D_BENCH_ONE_GPU_REHEARSAL"] = "1"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["config"]["global_batch"] == 4 and out["value"] > 0


# ------------------------------------------------------------------------------------------------ routines capture by default
def _clone_model(m):
    c = copy.deepcopy(m)
    if hasattr(c, "_mri3d_step_cache"):
        object.__delattr__(c, "_mri3d_step_cache")
    return c


def test_run_epoch_captured_by_default_equals_the_eager_loop_bit_for_bit():
    """segmentation.routine.run_epoch replays a captured hipGraph per batch shape (parallel.StepCache); a twin model whose cache
    is disabled runs the reference order eagerly.  Losses, parameters, BatchNorm running statistics and num_batches_tracked
    must be IDENTICAL after train + validate epochs with two alternating batch shapes (same kernels, deterministic reductions;
    the capture's warm-up runs leave no trace)."""
    torch.manual_seed(0)
    a = _unet(8).to(DEV)
    b = _clone_model(a)
    shapes = [(1, 1, 32, 32, 32), (2, 1, 16, 32, 16), (1, 1, 32, 32, 32), (2, 1, 16, 32, 16), (1, 1, 32, 32, 32)]
    batches = [{routine.MRI: {routine.DATA: seeded_randn(90 + i, s)},
                routine.LABEL: {routine.DATA: (seeded_rand(95 + i, s) < 0.2).float()}} for i, s in enumerate(shapes)]
    oa, ob = torch.optim.AdamW(a.parameters()), torch.optim.AdamW(b.parameters())
    parallel.StepCache.of(b).disabled = True
    la = routine.run_epoch(1, routine.Action.TRAIN, batches, a, oa)
    lb = routine.run_epoch(1, routine.Action.TRAIN, batches, b, ob)
    va = routine.run_epoch(1, routine.Action.VALIDATE, batches[:2], a, oa)
    vb = routine.run_epoch(1, routine.Action.VALIDATE, batches[:2], b, ob)
    ca, cb = parallel.StepCache.of(a), parallel.StepCache.of(b)
    assert ca.captures == 4 and ca.replays == 7 and ca.eager_runs == 0      # 2 shapes x (train, validate)
    assert cb.captures == 0 and cb.eager_runs == 7
    assert np.array_equal(la, lb) and np.array_equal(va, vb)
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k
    assert int(a.encoder.encoding_blocks[0].conv2.norm_layer.num_batches_tracked) == 5


def test_classification_run_one_epoch_captured_equals_eager_on_the_autoencoder_encoder():
    """classification.routine.run_one_epoch (a12) with the separable-conv encoder + head: captured replay == eager loop."""
    kw = dict(CLF_KWARGS, conv_pad=1, l_in=64, p_drop=0.0)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.enc, self.clf = P_AE.AE(**AE_KWARGS_93_6_4).enc, P_AE.Classificator(**kw)

        def forward(self, x):
            return self.clf(self.enc(x)[0])

    torch.manual_seed(0)
    a = Net().to(DEV)
    b = _clone_model(a)
    parallel.StepCache.of(b).disabled = True
    g = torch.Generator().manual_seed(4)
    loader = [(torch.randn(4, 1, 64, 64, 64, generator=g), torch.tensor([0, 1, 1, 0]), torch.arange(4)) for _ in range(3)]
    res = []
    for m in (a, b):
        opt = torch.optim.Adam(m.parameters(), 1e-3, weight_decay=0.01)
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=2, threshold=0.001)
        res.append(clf_routine.run_one_epoch(m, loader, torch.nn.CrossEntropyLoss(), True, DEV, opt, sch, False))
        with torch.no_grad():
            res[-1] += clf_routine.run_one_epoch(m, loader[:1], torch.nn.CrossEntropyLoss(), False, DEV, opt, sch, False)
    assert parallel.StepCache.of(a).captures == 2 and parallel.StepCache.of(a).eager_runs == 0
    for x, y in zip(res[0], res[1]):
        assert np.array_equal(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))
    for (k, x), (_, y) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(x, y), k


def test_step_cache_replay_matches_eager_for_the_full_autoencoder_and_modified3dunet():
    """Graph-vs-eager bit equality beyond the U-Net: the full AE (MSE) and Modified3DUNet in eval mode."""
    for make, shape, loss in ((lambda: P_AE.AE(**AE_KWARGS_93_6_4), (2, 1, 64, 64, 64), lambda o, t: F.mse_loss(o, t)),
                              (lambda: Modified3DUNet(1, 2, 8), (1, 1, 32, 32, 32), lambda o, t: o.float().square().mean())):
        torch.manual_seed(0)
        a = make().to(DEV)
        b = _clone_model(a)
        if isinstance(a, Modified3DUNet):
            a.eval(), b.eval()                                  # dropout off: the two copies see the same arithmetic
        x = seeded_randn(77, shape).to(DEV)
        ca, cb = parallel.StepCache.of(a), parallel.StepCache.of(b)
        cb.disabled = True
        for it in range(3):
            xi = x * (1.0 + 0.1 * it)
            oa, la = ca.run(xi, xi, loss, backward=True)
            oa, la = (oa[0] if isinstance(oa, tuple) else oa).clone(), la.clone()
            for p in b.parameters():
                p.grad = None
            ob, lb = cb.run(xi, xi, loss, backward=True)
            ob = ob[0] if isinstance(ob, tuple) else ob
            assert torch.equal(oa, ob) and torch.equal(la, lb)
            for (k, pa), pb in zip(a.named_parameters(), b.parameters()):
                assert torch.equal(pa.grad, pb.grad), k
        assert ca.captures == 1 and ca.replays == 3


def test_unet_split_decoder_conv_equals_the_shared_concat_buffer_scheme():
    """The decoder's first convolution reads (skip, upsampled) as two dense tensors (ops.conv3d_cat, the default since round 2) or,
    with `shared_concat_buffers`, one 3C-channel buffer that both producers wrote into (round 1): same kernels, same order of the
    sums — logits, loss and every parameter gradient must be bit-identical."""
    torch.manual_seed(4)
    from mri_epilepsy_diagnosis_amd.unet import UNet
    m = UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=8, normalization="batch",
             upsampling_type="linear", padding=True, activation="PReLU").cuda().train()
    x = torch.randn(2, 1, 48, 64, 80, device="cuda")
    t = (torch.rand(2, 1, 48, 64, 80, device="cuda") < 0.2).float()
    res = []
    for shared in (False, True):
        m.shared_concat_buffers = shared
        m.zero_grad(set_to_none=True)
        bufs = [b.detach().clone() for b in m.buffers()]
        y = m(x)
        loss = ops.softmax_dice_loss(y, t)
        loss.backward()
        res.append((y.detach().clone(), loss.detach().clone(), [p.grad.detach().clone() for p in m.parameters()]))
        with torch.no_grad():
            for b, s in zip(m.buffers(), bufs):
                b.copy_(s)                                    # same running statistics for the second pass
    (y0, l0, g0), (y1, l1, g1) = res
    assert torch.equal(y0, y1) and torch.equal(l0, l1)
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
