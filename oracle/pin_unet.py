"""Pin `unet.UNet`'s forward semantics with REFERENCE-HELD data — test infrastructure, authoring container only.

The upstream `unet` package is absent, so oracle/unet_recon.py restates its published forward pass.  What the reference itself
holds about that forward pass, beyond the state_dict schema, are the BatchNorm RUNNING STATISTICS inside its 18 shipped
checkpoints (segmentation/weights/*.pth): exponential averages of the per-layer batch statistics that the REAL upstream forward
produced on the authors' T1 volumes (batch 1, 928 volumes per epoch).  A restatement with the right layer order / concat order /
upsampling reproduces those statistics on a T1-like volume, layer by layer; a wrong one drifts from them at the first layer
downstream of the mistake.

    python -m oracle.pin_unet            # all checkpoints x all variants, full 184x220x184 volume  -> tests/golden/unet_pin.json
                                         # + tests/golden/mni152_gm_u8.npz (input of the committed CPU test)

Input: a pseudo-T1 built from the reference's grey-matter template (detection/MNI152_T1_1mm_brain_gray.nii.gz, a data file):
GM 70, enclosed white matter 110, sigma 0.7 blur, z-normalised over the voxels above the mean (TorchIO ZNormalization(mean), as the
training pipeline of segmentation/pretraining_3d_unet.ipynb cell 9 does), zero-padded to a multiple of 4.
Score of a variant on a checkpoint: over the BatchNorm layers, mean over channels of
    |batch_mean - running_mean| / sqrt(running_var)      ("dm", in running standard deviations)
    |log(batch_var / running_var)|                       ("dv")
"""
import gzip
import json
import os
import struct
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

VARIANTS = {
    "recon (skip first, trilinear, conv-BN-PReLU)": dict(),
    "upsampled first in the concat": dict(skip_first=False),
    "nearest upsampling": dict(up_mode="nearest"),
    "trilinear align_corners=True": dict(align_corners=True),
    "BatchNorm after PReLU": dict(norm_after_act=True),
    "average pooling": dict(pool="avg"),
}


def read_nifti(path):
    raw = gzip.open(path, "rb").read()
    dims = struct.unpack("<8h", raw[40:56])
    dtype, = struct.unpack("<h", raw[70:72])
    vox_offset, = struct.unpack("<f", raw[108:112])
    assert dtype == 16, dtype  # float32
    n = dims[1] * dims[2] * dims[3]
    a = np.frombuffer(raw, dtype="<f4", count=n, offset=int(vox_offset))
    return a.reshape(dims[3], dims[2], dims[1]).copy()   # NIfTI is x-fastest


def pseudo_t1(gm):
    from scipy import ndimage
    g = gm / max(float(gm.max()), 1e-6)
    brain = ndimage.binary_fill_holes(g > 0.2)
    wm = brain & ~(g > 0.2)
    vol = 70.0 * g + 110.0 * wm.astype(np.float32) * (1.0 - g)
    vol = ndimage.gaussian_filter(vol.astype(np.float32), 0.7)
    m = vol > vol.mean()
    vol = (vol - vol[m].mean()) / vol[m].std()
    pad = [(0, (-s) % 4) for s in vol.shape]
    return np.pad(vol, pad).astype(np.float32)


def forward_stats(model, x, skip_first=True, up_mode="trilinear", align_corners=False, norm_after_act=False, pool="max"):
    """Run UNetRecon's modules by hand with the given variant; BatchNorm in training arithmetic (batch statistics, as during the
    reference's training).  Returns [(layer name, batch mean, batch var (biased), running_mean, running_var)]."""
    rec = []

    def block(name, b, t):
        t = b.conv_layer(t)
        if norm_after_act and b.activation_layer is not None:
            t = b.activation_layer(t)
        if b.norm_layer is not None:
            bn = b.norm_layer
            mean = t.mean(dim=(0, 2, 3, 4))
            var = t.var(dim=(0, 2, 3, 4), unbiased=False)
            rec.append((name, mean, var, bn.running_mean, bn.running_var))
            t = (t - mean.view(1, -1, 1, 1, 1)) * torch.rsqrt(var.view(1, -1, 1, 1, 1) + bn.eps)
            t = t * bn.weight.view(1, -1, 1, 1, 1) + bn.bias.view(1, -1, 1, 1, 1)
        if not norm_after_act and b.activation_layer is not None:
            t = b.activation_layer(t)
        return t

    skips = []
    for i, eb in enumerate(model.encoder.encoding_blocks):
        x = block("enc%d.conv1" % i, eb.conv1, x)
        x = block("enc%d.conv2" % i, eb.conv2, x)
        skips.append(x)
        x = F.max_pool3d(x, 2) if pool == "max" else F.avg_pool3d(x, 2)
    x = block("bottom.conv1", model.bottom_block.conv1, x)
    x = block("bottom.conv2", model.bottom_block.conv2, x)
    for i, (s, db) in enumerate(zip(reversed(skips), model.decoder.decoding_blocks)):
        if up_mode == "nearest":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        else:
            x = F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=align_corners)
        x = torch.cat((s, x) if skip_first else (x, s), dim=1)
        x = block("dec%d.conv1" % i, db.conv1, x)
        x = block("dec%d.conv2" % i, db.conv2, x)
    return rec, model.classifier(x)


def scores(rec):
    out = {}
    for name, mean, var, rm, rv in rec:
        dm = ((mean - rm).abs() / rv.sqrt()).mean().item()
        dv = (var / rv).log().abs().mean().item()
        out[name] = (dm, dv)
    return out


def summarise(sc, layers=None):
    ks = [k for k in sc if layers is None or k.startswith(layers)]
    return float(np.mean([sc[k][0] for k in ks])), float(np.mean([sc[k][1] for k in ks]))


def main():
    sys.path.insert(0, ROOT)
    from oracle import unet_recon
    torch.set_num_threads(int(os.environ.get("PIN_THREADS", "8")))
    gm = read_nifti(os.path.join(REF, "detection", "MNI152_T1_1mm_brain_gray.nii.gz"))
    # the committed fixture is the template quantised to 8 bits (0.8 MB); everything below is computed from THAT, so that the
    # CPU test (tests/test_unet_pin.py) rebuilds exactly this input
    q = np.round(gm / gm.max() * 255).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "mni152_gm_u8.npz"), gm=q)
    vol = pseudo_t1(q.astype(np.float32) / 255.0)
    print("pseudo-T1", vol.shape, float(vol.min()), float(vol.max()))
    x_full = torch.from_numpy(vol)[None, None]
    wdir = os.path.join(REF, "segmentation", "weights")
    results = {"volume": list(vol.shape), "checkpoints": {}}
    for f in sorted(os.listdir(wdir)):
        if not f.endswith(".pth"):
            continue
        sd = torch.load(os.path.join(wdir, f), weights_only=True, map_location="cpu")
        m = unet_recon.UNetRecon(out_channels_first_layer=sd["encoder.encoding_blocks.0.conv1.conv_layer.weight"].shape[0])
        m.load_state_dict(sd, strict=True)
        entry = {}
        for vname, kw in VARIANTS.items():
            with torch.no_grad():
                rec, _ = forward_stats(m, x_full, **kw)
            sc = scores(rec)
            entry[vname] = {"all": summarise(sc), "decoder": summarise(sc, "dec"), "per_layer": {k: [round(a, 4), round(b, 4)] for k, (a, b) in sc.items()}}
        results["checkpoints"][f] = entry
        base = entry["recon (skip first, trilinear, conv-BN-PReLU)"]
        print("%-46s recon: all dm %.3f dv %.3f | decoder dm %.3f dv %.3f" % (f, *base["all"], *base["decoder"]))
        for vname in list(VARIANTS)[1:]:
            e = entry[vname]
            print("    %-40s all dm %.3f dv %.3f | decoder dm %.3f dv %.3f" % (vname, *e["all"], *e["decoder"]))
    json.dump(results, open(os.path.join(OUT, "unet_pin.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
