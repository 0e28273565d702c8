"""oracle/ — CPU restatement (plain torch.nn / torch.nn.functional on CPU tensors, fp32) of the reference's
volumetric hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and only as the
checker / reported CPU baseline — never as something measured as the product or shipped.  The product package
(mri_epilepsy_diagnosis_amd) never imports it and has no CPU path of its own.

Parity pinning:
  * ae_model.py, cnn_model.py, modified_3dunet.py restate in-repo reference modules that ARE importable in the
    authoring container; oracle/gen_golden.py imports the reference, checks the restatements against it on seeded
    inputs (identical parameters, forward outputs and parameter gradients) and writes tests/golden/*.npz.
  * unet_recon.py restates the third-party PyPI package `unet` (F. Pérez-García; no version pinned by the
    reference, ~0.7.x by date) whose source is NOT in /root/reference.  It is pinned by the reference's own
    shipped checkpoints (strict state_dict load of segmentation/weights/*.pth) and by the recorded upsampling warning;
    functional parity against upstream `unet` itself is "parity unpinned" (no reference test or golden tensor exists).
"""
