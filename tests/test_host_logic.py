"""CPU suite: host-side logic of the product package — state_dict compatibility with the reference's shipped
checkpoints, the routine.py-style loops (driven here with the CPU oracle model + oracle loss injected, since the
product ops have no CPU path), loud failure on CPU tensors, flat-gradient data parallelism over gloo (world_size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mri_epilepsy_diagnosis_amd import ops, parallel
from mri_epilepsy_diagnosis_amd.classification import routine as clf_routine
from mri_epilepsy_diagnosis_amd.classification.models import AE_model as P_AE, cnn_model as P_CNN
from mri_epilepsy_diagnosis_amd.segmentation import routine
from mri_epilepsy_diagnosis_amd.segmentation.models.modified_3dunet import Modified3DUNet
from mri_epilepsy_diagnosis_amd.unet import UNet
from oracle import losses, unet_recon
from util import AE_KWARGS_93_6_4, CLF_KWARGS, DISC_KWARGS, ROOT, load_ckpt, load_golden


def _unet(c0=8):
    return UNet(in_channels=1, out_classes=2, dimensions=3, num_encoding_blocks=3, out_channels_first_layer=c0,
                normalization="batch", upsampling_type="linear", padding=True, activation="PReLU")


def test_unet_state_dict_schema_and_strict_load():
    m = _unet(8)
    sd = load_ckpt("whole_im_train_seg_parc_epoch_7.pth")
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in m.parameters()) == 246412
    assert sum(p.numel() for p in _unet(16).parameters()) == 983564
    # alias keys share storage: loading conv_layer.* also sets block.0.*
    blk = m.encoder.encoding_blocks[0].conv2
    assert blk.conv_layer is blk.block[0] and blk.norm_layer is blk.block[1] and blk.activation_layer is blk.block[2]


def test_unet_seeded_init_equals_oracle_init():
    torch.manual_seed(0); a = _unet(8)
    torch.manual_seed(0); b = unet_recon.UNetRecon(out_channels_first_layer=8)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa) == list(sb)
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k


def test_classification_checkpoints_strict_load():
    enc = P_AE.AE(**AE_KWARGS_93_6_4).enc
    enc.load_state_dict(load_ckpt("encoder_93_6_4.pth"), strict=True)
    P_AE.Classificator(**CLF_KWARGS).load_state_dict(load_ckpt("clf_93_6_4.pth"), strict=True)
    P_AE.Discriminator(**DISC_KWARGS).load_state_dict(load_ckpt("disc_93_6_4.pth"), strict=True)
    assert sum(p.numel() for p in enc.parameters()) == 20296


def test_product_ops_refuse_cpu_tensors():
    x = torch.randn(1, 1, 8, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.conv3d(x, torch.randn(2, 1, 3, 3, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _unet(8)(torch.randn(1, 1, 8, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.softmax_dice_loss(torch.randn(1, 2, 4, 4, 4), torch.zeros(1, 1, 4, 4, 4))
    with pytest.raises(RuntimeError):
        Modified3DUNet(1, 2, 8)(torch.randn(1, 1, 16, 16, 16))
    with pytest.raises(RuntimeError):
        P_CNN.CNN(input_shape=(16, 16, 16))(torch.randn(2, 1, 16, 16, 16))


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch it."""
    for top in ("mri_epilepsy_diagnosis_amd", "tools", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                    src = open(os.path.join(d, f)).read()
                    assert "import oracle" not in src and "from oracle" not in src, os.path.join(d, f)
    # the two root scripts import it in exactly one function each
    for name, func in (("bench.py", "def cpu_baseline"), ("__graft_entry__.py", "def smoke")):
        src = open(os.path.join(ROOT, name)).read()
        head, _, tail = src.partition(func)
        assert "from oracle" not in head and "import oracle" not in head, name
        assert "from oracle" in tail, name


def test_prepare_batch_binarises_like_reference():
    lab = torch.tensor([0, 8, 9, 255, 1000, 2035, 1, 3], dtype=torch.float32).view(1, 1, 2, 2, 2).repeat(2, 1, 1, 1, 1)
    batch = {routine.MRI: {routine.DATA: torch.zeros(2, 1, 2, 2, 2)}, routine.LABEL: {routine.DATA: lab.clone()}}
    _, t = routine.prepare_batch(batch, "cpu")
    # sample 0: LIST_FCD ids (8, 255) -> 1; ids >= 1000 -> 1; label 1 stays; everything else 0
    assert t[0].flatten().tolist() == [0, 1, 0, 1, 1, 1, 1, 0]
    # sample 1: the reference applies LIST_FCD to targets[0][0] only (routine.py:192)
    assert t[1].flatten().tolist() == [0, 0, 0, 0, 1, 1, 1, 0]


def test_run_epoch_and_train_call_order_and_trajectory(tmp_path):
    """routine.train with the CPU oracle model + oracle loss reproduces the recorded seeded loss trajectory."""
    gold = load_golden("unet_c8_traj_32.npz")
    from util import seeded_rand, seeded_randn
    batches = [{routine.MRI: {routine.DATA: seeded_randn(70 + i, (1, 1, 32, 32, 32))},
                routine.LABEL: {routine.DATA: (seeded_rand(80 + i, (1, 1, 32, 32, 32)) < 0.1).float()}} for i in range(3)]
    torch.manual_seed(0)
    m = unet_recon.UNetRecon(out_channels_first_layer=8)
    opt = torch.optim.AdamW(m.parameters())
    losses_ = routine.run_epoch(1, routine.Action.TRAIN, batches, m, opt, loss_fn=losses.softmax_dice_loss)
    np.testing.assert_allclose(losses_, gold["losses"], rtol=1e-5)
    assert m.training
    v = routine.run_epoch(1, routine.Action.VALIDATE, batches[:1], m, opt, loss_fn=losses.softmax_dice_loss)
    assert not m.training and v.shape == (1,)

    class Sched:
        calls = []

        def step(self, x):
            self.calls.append(float(x))

    s = Sched()
    tr, va = routine.train(2, batches[:1], batches[1:2], m, opt, s, "stem", save_epoch=2, verbose=False,
                           loss_fn=losses.softmax_dice_loss, weights_dir=str(tmp_path))
    assert len(tr) == 2 and len(va) == 2 and len(s.calls) == 2 and abs(s.calls[-1] - va[-1]) < 1e-12
    assert os.listdir(tmp_path) == ["stem_epoch_2.pth"]
    assert list(torch.load(tmp_path / "stem_epoch_2.pth", weights_only=True)) == list(m.state_dict())


def test_get_model_and_optimizer_defaults():
    model, opt, sched = routine.get_model_and_optimizer("cpu", out_channels_first_layer=8)
    assert isinstance(opt, torch.optim.AdamW) and opt.defaults["lr"] == 1e-3 and opt.defaults["weight_decay"] == 1e-2
    assert isinstance(sched, torch.optim.lr_scheduler.ReduceLROnPlateau) and sched.factor == 0.1 and sched.patience == 3
    torch.manual_seed(0)
    ref = unet_recon.UNetRecon(out_channels_first_layer=8)
    assert torch.equal(model.classifier.conv_layer.weight, ref.classifier.conv_layer.weight)


def test_dice_helpers_match_oracle():
    g = load_golden("dice_known.npz")
    p = torch.softmax(torch.from_numpy(g["logits"]), dim=1)
    np.testing.assert_allclose(routine.get_dice_loss(p, torch.from_numpy(g["target"])).numpy(), g["per_channel"], rtol=1e-6)
    assert routine.get_iou_score(np.array([1, 1, 0, 0]), np.array([1, 0, 1, 0])) == pytest.approx(1 / 3)
    assert np.isnan(routine.compute_dice_coefficient(np.zeros(4, bool), np.zeros(4, bool)))


def test_stratified_batch_indices_and_adv_loss():
    idx = np.arange(10)
    lab = np.array([0, 0, 0, 0, 0, 0, 0, 1, 1, 1])
    out = clf_routine.stratified_batch_indices(idx, lab)
    assert sorted(out.tolist()) == idx.tolist() and lab[out][0] == 1  # minority class leads each stride
    g = load_golden("adv_loss.npz")
    v = clf_routine.adv_loss(torch.from_numpy(g["domain"]), torch.from_numpy(g["logits"]), 18).item()
    np.testing.assert_allclose(v, float(g["adv"]), rtol=1e-6)


def test_flat_params_views_and_shard_range():
    torch.manual_seed(0)
    m = unet_recon.UNetRecon(out_channels_first_layer=8)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    fp = parallel.FlatParams(m)
    assert fp.flat.numel() == 246412
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    x = torch.randn(1, 1, 16, 16, 16)
    losses.softmax_dice_loss(m(x), (torch.rand(1, 1, 16, 16, 16) < 0.1).float()).backward()
    p0 = fp.params[0]
    assert p0.grad.data_ptr() == fp.grad.data_ptr() and fp.grad.abs().sum() > 0
    fp.zero_grad()
    assert fp.grad.abs().sum() == 0 and p0.grad.data_ptr() == fp.grad.data_ptr()
    spans = [parallel.shard_range(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_data_parallel_gloo_world2_matches_single_process(tmp_path):
    """Two gloo ranks, one volume each, flat-gradient all-reduce == one process seeing both volumes (InstanceNorm-free
    check uses BN in eval mode so that batch statistics do not couple the volumes)."""
    script = os.path.join(ROOT, "tests", "_ddp_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(tmp_path)], env=env) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    g0 = torch.load(tmp_path / "grad_rank0.pt")
    g1 = torch.load(tmp_path / "grad_rank1.pt")
    ref = torch.load(tmp_path / "grad_single.pt")
    assert torch.equal(g0, g1)
    np.testing.assert_allclose(g0.numpy(), ref.numpy(), rtol=1e-4, atol=1e-7)


def _run_bench(argv, env_extra=None, drop=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT") + tuple(drop)}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT, capture_output=True,
                          text=True, timeout=600)


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO torchrun in the command and no WORLD_SIZE in the environment must itself start two
    ranks (it used to run one GPU silently).  --launch-check runs the launcher, the rendezvous, a barrier and
    two all-reduces over gloo and prints rank 0's line; no model or kernel is involved (there is no GPU here)."""
    import json
    res = _run_bench(["--gpus", "2", "--launch-check"])
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["ranks_counted_by_all_reduce"] == 2
    assert out["max_over_ranks"] == 2.0 and out["backend"] == "gloo" and out["launch_check"] is True
    # every child maps LOCAL_RANK -> its own device and draws its volumes from seed 1234 + rank (bench.rank_plan, the function
    # main() reads the same two things from)
    assert out["rank_plans"] == [{"rank": r, "local_rank": r, "device_index": r, "data_seed": 1234 + r} for r in range(2)]


def test_bench_rank_plan_and_barrier_helper():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.rank_plan(3, 3) == {"rank": 3, "local_rank": 3, "device_index": 3, "data_seed": 1237}
    assert b.rank_plan(3, 3, rehearsal=True)["device_index"] == 0
    from mri_epilepsy_diagnosis_amd import parallel
    parallel.barrier(0)          # no process group: a no-op, not an error
    parallel.release_captured_graphs()   # nothing captured: a no-op


def test_bench_refuses_a_world_size_that_is_not_gpus():
    res = _run_bench(["--gpus", "2", "--launch-check"], env_extra={"WORLD_SIZE": "1"})
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr and not res.stdout.strip()
    res = _run_bench(["--gpus", "1", "--launch-check"], env_extra={"WORLD_SIZE": "2", "RANK": "0"})
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the loud failure of the N-rank run on a box without GPUs")
def test_bench_launcher_propagates_rank_failure():
    """Without a GPU every rank of the real bench refuses to run; the launcher must exit non-zero and print no JSON line."""
    res = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert res.returncode != 0 and not [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert "ROCm device" in res.stderr


def test_fused_operator_predicates_decline_host_tensors_without_touching_the_library():
    """ops.upsample_conv3d_supported / ops.conv3d_pair_supported are pure host decisions: a CPU tensor, a non-integer scale, a
    strided convolution after the upsampling, an input that needs its gradient or an active autocast region all mean "run the
    two operators" — decided before any C-ABI call."""
    import types

    import torch

    from mri_epilepsy_diagnosis_amd import ops

    x = torch.zeros(1, 8, 2, 3, 4)
    w = torch.zeros(1, 8, 3, 1, 1)
    assert not ops.upsample_conv3d_supported(x, w, 4, 1, (1, 0, 0), 1)          # host tensor
    assert not ops.upsample_conv3d_supported(x, w, 2.5, 1, (1, 0, 0), 1)        # non-integer scale
    assert not ops.upsample_conv3d_supported(x, w, 4, 2, (1, 0, 0), 1)          # strided convolution
    assert not ops.upsample_conv3d_supported(x, torch.zeros(1, 4, 3, 1, 1), 4, 1, 0, 1)   # channel mismatch
    c1 = types.SimpleNamespace(weight=torch.zeros(8, 1, 6, 1, 1), bias=None, stride=(2, 1, 1), padding=(2, 0, 0), dilation=(1, 1, 1))
    c2 = types.SimpleNamespace(weight=torch.zeros(8, 8, 1, 6, 1), bias=None, stride=(1, 2, 1), padding=(0, 2, 0), dilation=(1, 1, 1))
    assert not ops.conv3d_pair_supported(torch.zeros(1, 1, 8, 8, 8), c1, c2)    # host tensor
    assert not ops.conv3d_pair_supported(torch.zeros(1, 1, 8, 8), c1, c2)       # not 5-D

