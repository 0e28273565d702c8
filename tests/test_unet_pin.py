"""Pins oracle/unet_recon.py's forward semantics with REFERENCE-HELD data: the BatchNorm running statistics inside
the reference's shipped `unet.UNet` checkpoints are averages of the batch statistics the REAL upstream forward produced during
training.  A restatement with the right layer / concat order reproduces them on a T1-like volume; wrong variants do not.

Input: pseudo-T1 rebuilt from the reference's grey-matter template (tests/golden/mni152_gm_u8.npz, 8-bit quantised data file of
detection/MNI152_T1_1mm_brain_gray.nii.gz) exactly as oracle/pin_unet.py builds it.  tests/golden/unet_pin.json holds the sweep
over ALL 18 shipped checkpoints x 6 variants (run in the authoring container); this test re-runs the committed checkpoint
(whole_im_train_seg_parc_epoch_7.pth) on the host and checks both the live numbers and the recorded sweep.

What the statistics can and cannot tell (recorded in DESIGN.md §2):  concat order (skip first) and conv -> BN -> PReLU order are
pinned decisively; max- vs average pooling weakly (encoder variances); the upsampling mode NOT at all (nearest / trilinear /
align_corners differ by < 0.01) — that one stays pinned by the PyTorch warning text recorded in the reference's notebook outputs
(SURVEY §8c)."""
import json
import os

import numpy as np
import torch

from oracle import pin_unet, unet_recon
from util import GOLDEN, load_ckpt

RECON = "recon (skip first, trilinear, conv-BN-PReLU)"


def test_recorded_sweep_over_all_shipped_checkpoints_separates_the_variants():
    d = json.load(open(os.path.join(GOLDEN, "unet_pin.json")))
    assert len(d["checkpoints"]) == 18
    for name, e in d["checkpoints"].items():
        rec, swap, order = e[RECON], e["upsampled first in the concat"], e["BatchNorm after PReLU"]
        # the restatement: decoder batch means within 0.2 running standard deviations, variances within a factor e^0.45 = 1.57
        assert rec["decoder"][0] < 0.2 and rec["decoder"][1] < 0.45, (name, rec["decoder"])
        # swapped concat order: decoder variances off by a factor > e^0.75 = 2.1 in EVERY checkpoint, and > 2.4x the restatement's score
        assert swap["decoder"][1] > 0.75 and swap["decoder"][1] > 2.4 * rec["decoder"][1], (name, swap["decoder"], rec["decoder"])
        # BatchNorm after the activation: means off by > 0.7 running standard deviations everywhere
        assert order["all"][0] > 0.65 and order["all"][0] > 5 * rec["all"][0], (name, order["all"], rec["all"])
        # the upsampling mode is NOT identifiable from these statistics (stated, so that nobody reads more into this pin)
        for v in ("nearest upsampling", "trilinear align_corners=True"):
            assert abs(e[v]["decoder"][1] - rec["decoder"][1]) < 0.03, (name, v)


def test_committed_checkpoint_batch_statistics_match_its_running_statistics_live():
    gm = np.load(os.path.join(GOLDEN, "mni152_gm_u8.npz"))["gm"].astype(np.float32) / 255.0
    x = torch.from_numpy(pin_unet.pseudo_t1(gm))[None, None]
    assert tuple(x.shape[2:]) == (184, 220, 184)
    m = unet_recon.UNetRecon(out_channels_first_layer=8)
    m.load_state_dict(load_ckpt("whole_im_train_seg_parc_epoch_7.pth"), strict=True)
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    sc = {}
    for name in (RECON, "upsampled first in the concat", "BatchNorm after PReLU"):
        with torch.no_grad():
            rec, _ = pin_unet.forward_stats(m, x, **pin_unet.VARIANTS[name])
        s = pin_unet.scores(rec)
        sc[name] = (pin_unet.summarise(s), pin_unet.summarise(s, "dec"))
    (all_r, dec_r), (all_s, dec_s), (all_o, dec_o) = sc[RECON], sc["upsampled first in the concat"], sc["BatchNorm after PReLU"]
    assert dec_r[0] < 0.2 and dec_r[1] < 0.45, dec_r
    assert dec_s[1] > 0.75 and dec_s[1] > 2.4 * dec_r[1], (dec_s, dec_r)
    assert all_o[0] > 0.65 and all_o[0] > 5 * all_r[0], (all_o, all_r)
    # and they are the numbers of the recorded sweep for this checkpoint
    e = json.load(open(os.path.join(GOLDEN, "unet_pin.json")))["checkpoints"]["whole_im_train_seg_parc_epoch_7.pth"]
    np.testing.assert_allclose(dec_r, e[RECON]["decoder"], rtol=2e-2)
    np.testing.assert_allclose(dec_s, e["upsampled first in the concat"]["decoder"], rtol=2e-2)
