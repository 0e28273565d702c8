"""Host-side mirror of the reference's classification training loops, driving the HIP-backed models.

  classification/routine.py  run_one_epoch :15, train :55, stratified_batch_indices :162, create_model_opt :253
  classification/train_ENC_CLF.ipynb  adv_loss / main_loss (cell 14), adversarial encoder+classifier step (cell 16)

Loss heads operate on (N, <=64) tensors with torch ops (CrossEntropy / log_softmax: not on the volumetric path,
SURVEY.md §2.1); everything 5-D inside the models is HIP.  Reference quirks handled as the
*intended* behaviour and documented here: the reference initialises `patience` but increments `patience_`
(UnboundLocalError when epoch 0 does not improve) — one counter is used; `stats.mode(labels)[0][0]` breaks on
SciPy >= 1.11 — np.bincount is used; `scheduler.step(loss)` is called per batch, as in the reference.
cross_val_score (:182-251, sklearn CV bookkeeping) is out of scope.
"""
import time

import numpy as np
import torch
import torch.nn as tnn
import torch.nn.functional as F

from .. import nn as mnn
from .. import parallel


def run_one_epoch(model, loader, criterion, train, device, optimizer=None, scheduler=None, experiment=False):
    """classification/routine.py:15-52.  On the device, forward + criterion (+ backward) of a batch shape are captured once
    into a hipGraph and replayed (parallel.StepCache); the optimizer and the per-batch `scheduler.step(loss)` stay eager."""
    model.to(device)
    model.train(train)
    cache = parallel.StepCache.of(model)
    losses, probs, targets = [], [], []
    for data, target, _ in loader:
        data = data.to(device, dtype=torch.float)
        target = target.long().to(device)
        step = train and optimizer is not None
        if step:
            optimizer.zero_grad()
        outputs, loss = cache.run(data, target, criterion, backward=step)
        if step:
            optimizer.step()
            if scheduler is not None:
                scheduler.step(loss)
        losses.append(loss.data.cpu().numpy())
        probs.extend(F.softmax(outputs, dim=-1).cpu().data.numpy()[:, 1])
        targets.extend(list(target.cpu().data.numpy()))
        if experiment:
            experiment.log_metric("train_loss" if train else "val_loss", losses[-1])
        del data, target, outputs, loss
    return losses, probs, targets


def train(model, optimizer, scheduler, train_dataloader, val_dataloader, device, metric, verbose=0,
          model_save_path=None, max_epoch=20, eps=3e-3, max_patience=10, experiment=False):
    criterion = tnn.CrossEntropyLoss()
    patience, best_metric = 0, 0
    epoch_train_loss, epoch_train_metric, epoch_val_loss, epoch_val_metric = [], [], [], []
    last_train_loss = last_train_metric = last_val_loss = last_val_metric = None
    for epoch in range(max_epoch):
        start_time = time.time()
        tl, tp, tt = run_one_epoch(model, train_dataloader, criterion, True, device, optimizer, scheduler, experiment)
        if val_dataloader is not None:
            with torch.no_grad():
                vl, vp, vt = run_one_epoch(model, val_dataloader, criterion, False, device, optimizer, scheduler,
                                           experiment)
        epoch_train_loss.append(np.mean(tl))
        epoch_train_metric.append(metric(tt, tp))
        if experiment:
            experiment.log_metrics({"mean_train_loss": np.mean(tl), "train_metric": metric(tt, tp)}, epoch=epoch)
        if val_dataloader is not None:
            epoch_val_loss.append(np.mean(vl))
            epoch_val_metric.append(metric(vt, vp))
            if experiment:
                experiment.log_metrics({"mean_val_loss": np.mean(vl), "val_metric": metric(vt, vp)}, epoch=epoch)
        if verbose:
            print("Epoch {} of {} took {:.3f}s".format(epoch + 1, max_epoch, time.time() - start_time))
            print("  training loss (in-iteration): \t{:.6f}".format(epoch_train_loss[-1]))
            if val_dataloader is not None:
                print("  validation loss: \t\t\t{:.6f}".format(epoch_val_loss[-1]))
        improved = (epoch_val_metric[-1] > best_metric) if val_dataloader is not None \
            else (epoch_train_metric[-1] >= best_metric)
        if improved:
            patience = 0
            best_metric = epoch_val_metric[-1] if val_dataloader is not None else epoch_train_metric[-1]
            last_train_metric, last_train_loss = epoch_train_metric[-1], epoch_train_loss[-1]
            if val_dataloader is not None:
                last_val_metric, last_val_loss = epoch_val_metric[-1], epoch_val_loss[-1]
            if model_save_path is not None:
                torch.save(model.state_dict(), model_save_path)
        else:
            patience += 1
        if patience >= max_patience:
            print("Early stopping! Patience is out.")
            break
        if epoch_train_loss[-1] < eps:
            print("Early stopping! Train loss < eps.")
            break
        last_train_metric, last_train_loss = epoch_train_metric[-1], epoch_train_loss[-1]
        if model_save_path is not None:
            torch.save(model.state_dict(), model_save_path)
    return last_train_loss, last_train_metric, last_val_loss, last_val_metric


def stratified_batch_indices(indices, labels):
    indices, labels = np.asarray(indices), np.asarray(labels)
    dominating = np.bincount(labels.astype(np.int64)).argmax()
    idx0, idx1 = indices[labels == dominating], indices[labels != dominating]
    step = np.ceil(len(idx0) / max(len(idx1), 1)) + 1
    result, j0, j1 = [], 0, 0
    for i in range(len(indices)):
        if (i % step == 0 or j0 == len(idx0)) and j1 < len(idx1):
            result.append(idx1[j1]); j1 += 1
        else:
            result.append(idx0[j0]); j0 += 1
    return np.array(result)


def create_model_opt(model, model_load_path=None, input_shape=(192, 192, 192), n_fc_units=192, transfer=False, lr=1e-5,
                     patience=2):
    torch.manual_seed(0)
    np.random.seed(0)
    if model_load_path is not None:
        model.load_state_dict(torch.load(model_load_path))
    if transfer:
        for p in model.parameters():
            p.requires_grad = False
        last = tnn.Linear(128, 2)
        modules = list(list(model.children())[0].children())[:-1] + [last]
        model = tnn.Sequential(mnn.FusedSequential(*modules))
        opt = torch.optim.Adam(last.parameters(), lr, weight_decay=0.01)
    else:
        opt = torch.optim.Adam(model.parameters(), lr, weight_decay=0.01)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.5, patience=patience, threshold=0.001)
    return model, opt, scheduler


# ------------------------------------------------------------------ adversarial encoder + classifier step (train_ENC_CLF.ipynb)


def adv_loss(y, pred_logits, n_domains):
    """-mean((1 - onehot(domain)) * log_softmax(disc(latent)))  (cell 14)."""
    onehot = torch.zeros((y.shape[0], n_domains), dtype=torch.int32, device=pred_logits.device)
    onehot.scatter_(1, y.view(-1, 1).to(pred_logits.device), 1)
    return -torch.mean((1 - onehot) * F.log_softmax(pred_logits, dim=1))


def main_loss(pred_logits_clf, y, pred_logits_domain, domain, criterion_enc_clf, lambda_t, n_domains):
    loss_clf = criterion_enc_clf(pred_logits_clf, y)
    loss_adversarial = adv_loss(domain, pred_logits_domain, n_domains)
    return loss_clf + lambda_t * loss_adversarial, loss_adversarial


def adversarial_step(encoder, disc, clf, X, y, domain, criterion_enc_clf, criterion_disc, optimizer_enc_clf,
                     optimizer_disc, lambda_t, n_domains, n_d=1):
    """One batch of the fader-style loop (cell 16): (1) encoder frozen/eval, `n_d` discriminator updates on the
    latent; (2) encoder+classifier update against CE + lambda * adversarial loss with the discriminator frozen.
    Costs two encoder forwards and one encoder backward on the full volume — the hot part."""
    encoder.eval()
    for p in encoder.parameters():
        p.requires_grad = False
    disc.train()
    latent = encoder(X)[0]
    for _ in range(n_d):
        optimizer_disc.zero_grad()
        loss_disc = criterion_disc(disc(latent), domain)
        loss_disc.backward()
        optimizer_disc.step()
    for p in encoder.parameters():
        p.requires_grad = True
    encoder.train(); clf.train(); disc.eval()
    for p in disc.parameters():
        p.requires_grad = False
    optimizer_enc_clf.zero_grad()
    latent, _ = encoder(X)
    loss, loss_adv = main_loss(clf(latent), y, disc(latent), domain, criterion_enc_clf, lambda_t, n_domains)
    loss.backward()
    optimizer_enc_clf.step()
    for p in disc.parameters():
        p.requires_grad = True
    return loss.detach(), loss_disc.detach(), loss_adv.detach()
