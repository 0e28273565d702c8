"""SURVEY §8 row a12 — `run_one_epoch` / `train` / `create_model_opt` of classification/routine.py (:15-52, :55-159, :253-279).

The loops are model-agnostic host logic, so on the CPU they are driven with a tiny torch model and compared with
tests/golden/clf_routine.npz, which `oracle/gen_golden.py clf_routine` recorded by executing the REFERENCE's own functions
(extracted with `ast`; the module itself needs comet_ml / IPython) on the same seeded batches: every logged loss, the returned
tuple, the learning rate after the per-batch `scheduler.step(loss)` calls, the trained and the saved parameters — bit for bit.
Stated divergence: the reference initialises `patience` but increments `patience_`, so it raises
UnboundLocalError whenever epoch 0 does not improve the validation metric, and ALWAYS without a validation loader; the product
keeps the intended behaviour (one counter) — the fixture records that the reference raises there.
The GPU part runs the product loops on the HIP-backed VoxResNet."""
import os

import numpy as np
import pytest
import torch

from mri_epilepsy_diagnosis_amd.classification import routine as R
from mri_epilepsy_diagnosis_amd.classification.models import cnn_model as P_CNN
from oracle.gen_golden import RecordingExperiment, accuracy, clf_loaders, clf_tiny_model
from util import assert_close, load_golden, seeded_randn


def _opt_sched(m, lr=5e-2, wd=0.01):
    opt = torch.optim.Adam(m.parameters(), lr, weight_decay=wd)
    return opt, torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=2, threshold=0.001)


@pytest.mark.parametrize("tag,kw", [("A", dict(max_epoch=8, max_patience=2, eps=3e-3)),      # stops on patience
                                    ("E", dict(max_epoch=8, max_patience=50, eps=0.62))])   # stops on train loss < eps
def test_train_reproduces_the_reference_run_bit_for_bit(tag, kw, tmp_path):
    gold = load_golden("clf_routine.npz")
    m = clf_tiny_model()
    opt, sch = _opt_sched(m)
    ex = RecordingExperiment()
    path = str(tmp_path / "m.pth")
    ret = R.train(m, opt, sch, clf_loaders(1, 4, 8), clf_loaders(2, 2, 8), "cpu", accuracy, verbose=0, model_save_path=path,
                  experiment=ex, **kw)
    assert np.array_equal(np.array([float(v) for v in ret]), gold[tag + "_ret"])
    assert [c[0] for c in ex.calls] == gold[tag + "_log_names"].tolist()        # same log calls in the same order ...
    assert np.array_equal(np.array([c[1] for c in ex.calls]), gold[tag + "_log_values"])  # ... with the same values
    assert opt.param_groups[0]["lr"] == float(gold[tag + "_lr"])             # per-batch scheduler.step(loss) (routine.py:35)
    assert np.array_equal(torch.cat([p.detach().flatten() for p in m.parameters()]).numpy(), gold[tag + "_params"])
    saved = torch.load(path, weights_only=True)
    assert np.array_equal(torch.cat([v.flatten() for v in saved.values()]).numpy(), gold[tag + "_saved"])
    if tag == "A":   # early stop on patience leaves the file of the last completed epoch, not the final parameters
        assert opt.param_groups[0]["lr"] < 5e-2


def test_train_where_the_reference_raises_on_patience_(tmp_path):
    """Epoch 0 does not improve (B) / no validation loader (C): the reference dies on its unbound `patience_`; the product runs
    the intended logic: counts patience, stops when it is out, returns the last metrics."""
    gold = load_golden("clf_routine.npz")
    assert str(gold["B_raised"]).startswith("UnboundLocalError") and str(gold["C_raised"]).startswith("UnboundLocalError")
    for val in (clf_loaders(3, 1, 8, flip=True), None):
        m = clf_tiny_model()
        with torch.no_grad():
            m[1].weight.copy_(torch.stack([-torch.ones(64), torch.ones(64)]))
            m[1].bias.zero_()
        opt = torch.optim.Adam(m.parameters(), 1e-5)
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=2, threshold=0.001)
        ret = R.train(m, opt, sch, clf_loaders(1, 2, 8), val, "cpu", accuracy, verbose=0, max_epoch=5, max_patience=2,
                      model_save_path=str(tmp_path / "x.pth"))
        assert ret[0] is not None and ret[1] == 1.0                     # the confidently-right model: train accuracy 1
        assert ret[2] is None and ret[3] is None      # validation numbers are only recorded on improvement (routine.py:126-129)
    # B ran max_patience epochs then stopped: the flipped validation set never beats best_metric = 0
    calls = []

    def metric(t, p):
        calls.append(len(t))
        return accuracy(t, p)

    m = clf_tiny_model()
    with torch.no_grad():
        m[1].weight.copy_(torch.stack([-torch.ones(64), torch.ones(64)]))
        m[1].bias.zero_()
    opt = torch.optim.Adam(m.parameters(), 1e-5)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=2, threshold=0.001)
    R.train(m, opt, sch, clf_loaders(1, 2, 8), clf_loaders(3, 1, 8, flip=True), "cpu", metric, verbose=0, max_epoch=9, max_patience=2)
    assert len(calls) == 2 * 2                                           # (train + val metric) x 2 epochs, then patience is out


def test_run_one_epoch_matches_reference_and_steps_the_scheduler_per_batch():
    gold = load_golden("clf_routine.npz")
    m = clf_tiny_model()
    opt = torch.optim.SGD(m.parameters(), 1e-3)
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=2, threshold=0.001)
    losses, probs, targets = R.run_one_epoch(m, clf_loaders(5, 1, 8) * 12, torch.nn.CrossEntropyLoss(), True, "cpu", opt, sch, False)
    assert np.array_equal(np.array([float(v) for v in losses]), gold["R_losses"])
    assert np.array_equal(np.array(probs), gold["R_probs"]) and np.array_equal(np.array(targets), gold["R_targets"])
    assert sch.last_epoch == int(gold["R_sched_last_epoch"]) == 12 and sch.num_bad_epochs == int(gold["R_sched_num_bad"])
    assert m.training
    # evaluation: no optimizer / scheduler step, model left in eval mode, same return structure
    before = [p.detach().clone() for p in m.parameters()]
    with torch.no_grad():
        l2, p2, t2 = R.run_one_epoch(m, clf_loaders(5, 1, 8), torch.nn.CrossEntropyLoss(), False, "cpu", opt, sch, False)
    assert not m.training and sch.last_epoch == 12 and len(l2) == 1 and len(p2) == 8 and len(t2) == 8
    assert all(torch.equal(a, b) for a, b in zip(before, m.parameters()))


def test_create_model_opt_transfer_structure_matches_reference():
    """transfer=True (routine.py:262-273): everything frozen, a fresh seeded Linear(128, 2) replaces the last child, Adam over the
    new layer only (lr, weight_decay 0.01), ReduceLROnPlateau(min, 0.5, patience, 1e-3)."""
    gold = load_golden("clf_routine.npz")
    torch.manual_seed(0)
    base = P_CNN.VoxResNet(input_shape=(32, 32, 32), n_filters=8, n_blocks=3)
    assert sum(p.numel() for p in base.parameters()) == int(gold["T_n_base"])
    model, opt, sch = R.create_model_opt(base, transfer=True, lr=1e-5, patience=2)
    inner = list(model.children())[0]
    assert len(list(inner.children())) == len(gold["T_names"])     # Sequential(*modules) renames the children "0".."N"
    assert [k for k, _ in inner.named_children()] == gold["T_names"].tolist()
    last = list(inner.children())[-1]
    assert np.array_equal(last.weight.detach().numpy(), gold["T_last_weight"])      # torch.manual_seed(0) inside, then Linear(128, 2)
    assert np.array_equal(last.bias.detach().numpy(), gold["T_last_bias"])
    assert sum(p.numel() for p in model.parameters() if p.requires_grad) == int(gold["T_n_trainable"]) == 258
    assert sum(p.numel() for p in model.parameters()) == int(gold["T_n_params"])
    assert sum(p.numel() for g in opt.param_groups for p in g["params"]) == int(gold["T_opt_numel"]) == 258
    assert [opt.defaults["lr"], opt.defaults["weight_decay"], sch.factor, sch.patience, sch.threshold] == gold["T_opt"].tolist()
    # not transfer: Adam over every parameter, same scheduler
    torch.manual_seed(0)
    base = P_CNN.VoxResNet(input_shape=(32, 32, 32), n_filters=8, n_blocks=3)
    m2, o2, s2 = R.create_model_opt(base, transfer=False, lr=3e-5, patience=4)
    assert m2 is base and o2.defaults["lr"] == 3e-5 and o2.defaults["weight_decay"] == 0.01 and s2.patience == 4
    assert sum(p.numel() for g in o2.param_groups for p in g["params"]) == int(gold["T_n_base"])


@pytest.mark.gpu
def test_create_model_opt_transfer_forward_and_training_on_device(tmp_path):
    """The transfer model on the HIP path: eval forward equals the reference's recorded output; one `train` epoch moves only the
    new head; `model_load_path` round-trips a saved state_dict."""
    gold = load_golden("clf_routine.npz")
    torch.manual_seed(0)
    base = P_CNN.VoxResNet(input_shape=(32, 32, 32), n_filters=8, n_blocks=3)
    path = str(tmp_path / "base.pth")
    torch.save(base.state_dict(), path)
    torch.manual_seed(5)
    other = P_CNN.VoxResNet(input_shape=(32, 32, 32), n_filters=8, n_blocks=3)     # different init: the load must overwrite it
    model, opt, sch = R.create_model_opt(other, model_load_path=path, transfer=True, lr=1e-2, patience=2)
    model.to("cuda").eval()
    with torch.no_grad():
        out = model(seeded_randn(181, (2, 1, 32, 32, 32)).to("cuda"))
    assert_close(out.cpu(), gold["T_out"], rel=1e-3, what="transfer model eval output vs the reference's")
    frozen = [p.detach().clone() for p in model.parameters() if not p.requires_grad]
    head = [p.detach().clone() for p in model.parameters() if p.requires_grad]
    g = torch.Generator().manual_seed(9)
    loader = [(torch.randn(4, 1, 32, 32, 32, generator=g), torch.tensor([0, 1, 1, 0]), torch.arange(4)) for _ in range(3)]
    ret = R.train(model, opt, sch, loader, loader[:1], "cuda", accuracy, verbose=0, max_epoch=1)
    assert ret[0] is not None and np.isfinite(ret[0]) and sch.last_epoch == 3
    assert all(torch.equal(a, b) for a, b in zip(frozen, [p for p in model.parameters() if not p.requires_grad]))
    assert any(not torch.equal(a, b) for a, b in zip(head, [p for p in model.parameters() if p.requires_grad]))


@pytest.mark.gpu
def test_run_one_epoch_on_device_vs_cpu_oracle_model():
    """run_one_epoch with the HIP-backed CNN vs the same loop driving the CPU oracle CNN: same losses / probabilities to 1e-3."""
    from oracle import cnn_model as O_CNN
    res = {}
    g = torch.Generator().manual_seed(3)
    loader = [(torch.randn(4, 1, 32, 32, 32, generator=g), torch.tensor([0, 1, 0, 1]), torch.arange(4)) for _ in range(2)]
    for tag, mod, dev in (("o", O_CNN, "cpu"), ("p", P_CNN, "cuda")):
        torch.manual_seed(0)
        m = torch.nn.Sequential(mod.CNN(input_shape=(32, 32, 32), n_filters=16, n_blocks=3), torch.nn.Linear(128, 2))
        opt = torch.optim.Adam(m.parameters(), 1e-5, weight_decay=0.01)
        sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=2, threshold=0.001)
        res[tag] = R.run_one_epoch(m, loader, torch.nn.CrossEntropyLoss(), True, dev, opt, sch, False)
    np.testing.assert_allclose(np.array(res["p"][0], dtype=np.float64), np.array(res["o"][0], dtype=np.float64), rtol=1e-3)
    np.testing.assert_allclose(np.array(res["p"][1]), np.array(res["o"][1]), rtol=1e-3, atol=1e-4)
    assert res["p"][2] == res["o"][2]
