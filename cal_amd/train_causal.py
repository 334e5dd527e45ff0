"""Causal training / evaluation loops with the reference's function surface
(train_causal.py:11-61, 162-223): same loss (KL to uniform + 2x NLL, weights
args.c / args.o / args.co), Adam + cosine schedule, same returned tuples.

Differences that do not change results: the five ``.item()`` host syncs per
iteration (train_causal.py:186-191) are replaced by on-device accumulators read
once per epoch, and ``CosineAnnealingLR`` is built without the ``verbose``
keyword current torch rejects.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.optim import Adam
from torch.optim.lr_scheduler import CosineAnnealingLR

from .data import DataLoader


def _device(args=None):
    """train_causal.py:10: ``cuda`` when available, else ``cpu`` -- where the reference's CPU plumbing run
    (BASELINE.json configs[0]) lands; ``args.device`` (not a reference flag) pins it.  On the CPU the models run the
    operator-level path on libcalhost.so (the host implementation of the same C-ABI symbols); nothing GPU-resident is
    ever computed there."""
    forced = getattr(args, "device", None)
    return torch.device(forced if forced else ("cuda" if torch.cuda.is_available() else "cpu"))


def _check_engine(model):
    """Raise if any step since the last check flagged its batch as invalid (the read-back above already synchronised)."""
    eng = getattr(model, "_engine", None)
    if eng is not None:
        eng.check_status()


def num_graphs(data):
    """utils.py:12-16."""
    if data.batch is not None:
        return data.num_graphs
    return data.x.size(0)


def causal_loss(c_logs, o_logs, co_logs, y, num_classes, args):
    """train_causal.py:176-183."""
    one_hot_target = y.view(-1)
    uniform_target = torch.ones_like(c_logs, dtype=torch.float) / num_classes
    c_loss = F.kl_div(c_logs, uniform_target, reduction="batchmean")
    o_loss = F.nll_loss(o_logs, one_hot_target)
    co_loss = F.nll_loss(co_logs, one_hot_target)
    loss = args.c * c_loss + args.o * o_loss + args.co * co_loss
    return loss, c_loss, o_loss, co_loss


def train_causal_epoch(model, optimizer, loader, device, args, grad_sync=None):
    """train_causal.py:162-200.  ``grad_sync`` (optional callable) runs between
    backward and the optimizer step -- the data-parallel gradient all-reduce."""
    model.train()
    acc = torch.zeros(5, dtype=torch.float64, device=device)   # loss, c, o, co, correct
    for it, data in enumerate(loader):
        optimizer.zero_grad()
        data = data.to(device)
        c_logs, o_logs, co_logs = model(data, eval_random=args.with_random)
        loss, c_loss, o_loss, co_loss = causal_loss(c_logs, o_logs, co_logs, data.y, model.num_classes, args)
        pred_o = o_logs.max(1)[1]
        loss.backward()
        if grad_sync is not None:
            grad_sync()
        n = num_graphs(data)
        with torch.no_grad():
            acc += torch.stack([loss.detach() * n, c_loss.detach() * n, o_loss.detach() * n,
                                co_loss.detach() * n,
                                pred_o.eq(data.y.view(-1)).sum().to(loss.dtype)]).to(torch.float64)
        optimizer.step()
    num = len(loader.dataset)
    total_loss, total_loss_c, total_loss_o, total_loss_co, correct_o = (acc / num).tolist()
    _check_engine(model)
    return total_loss, total_loss_c, total_loss_o, total_loss_co, correct_o


def eval_acc_causal(model, loader, device, args):
    """train_causal.py:202-223."""
    model.eval()
    eval_random = args.eval_random
    acc = torch.zeros(3, dtype=torch.float64, device=device)
    for data in loader:
        data = data.to(device)
        with torch.no_grad():
            c_logs, o_logs, co_logs = model(data, eval_random=eval_random)
            y = data.y.view(-1)
            acc += torch.stack([co_logs.max(1)[1].eq(y).sum(), c_logs.max(1)[1].eq(y).sum(),
                                o_logs.max(1)[1].eq(y).sum()]).to(torch.float64)
    n = len(loader.dataset)
    acc_co, acc_c, acc_o = (acc / n).tolist()
    _check_engine(model)
    return acc_co, acc_c, acc_o


def train_causal_syn(train_set, val_set, test_set, model_func=None, args=None, log=print):
    """train_causal.py:11-61."""
    device = _device(args)
    train_loader = DataLoader(train_set, args.batch_size, shuffle=True)
    val_loader = DataLoader(val_set, args.batch_size, shuffle=False)
    test_loader = DataLoader(test_set, args.batch_size, shuffle=False)
    if args.feature_dim == -1:
        args.feature_dim = args.max_degree
    model = model_func(args.feature_dim, args.num_classes).to(device)
    optimizer = Adam(model.parameters(), lr=args.lr)
    lr_scheduler = CosineAnnealingLR(optimizer, T_max=args.epochs, eta_min=args.min_lr, last_epoch=-1)
    best_val_acc, update_test_acc_co, update_test_acc_c, update_test_acc_o, update_epoch = 0, 0, 0, 0, 0
    history = []
    for epoch in range(1, args.epochs + 1):
        train_loss, loss_c, loss_o, loss_co, train_acc_o = train_causal_epoch(model, optimizer, train_loader, device, args)
        val_acc_co, val_acc_c, val_acc_o = eval_acc_causal(model, val_loader, device, args)
        test_acc_co, test_acc_c, test_acc_o = eval_acc_causal(model, test_loader, device, args)
        lr_scheduler.step()
        if val_acc_o > best_val_acc:
            best_val_acc = val_acc_o
            update_test_acc_co = test_acc_co
            update_test_acc_c = test_acc_c
            update_test_acc_o = test_acc_o
            update_epoch = epoch
        history.append(dict(epoch=epoch, loss=train_loss, loss_c=loss_c, loss_o=loss_o, loss_co=loss_co,
                            train_acc_o=train_acc_o, val_acc_o=val_acc_o, test_acc_o=test_acc_o))
        log("BIAS:[{:.2f}] | Model:[{}] Epoch:[{}/{}] Loss:[{:.4f}={:.4f}+{:.4f}+{:.4f}] Train:[{:.2f}] val:[{:.2f}] "
            "Test:[{:.2f}] | Update Test:[co:{:.2f},c:{:.2f},o:{:.2f}] at Epoch:[{}] | lr:{:.6f}".format(
                args.bias, args.model, epoch, args.epochs, train_loss, loss_c, loss_o, loss_co,
                train_acc_o * 100, val_acc_o * 100, test_acc_o * 100, update_test_acc_co * 100,
                update_test_acc_c * 100, update_test_acc_o * 100, update_epoch,
                optimizer.param_groups[0]["lr"]))
    log("syd: BIAS:[{:.2f}] | Val acc:[{:.2f}] Test acc:[co:{:.2f},c:{:.2f},o:{:.2f}] at epoch:[{}]".format(
        args.bias, val_acc_o * 100, update_test_acc_co * 100, update_test_acc_c * 100,
        update_test_acc_o * 100, update_epoch))
    return model, history


def train_causal_real(dataset=None, model_func=None, args=None, log=print):
    """train_causal.py:63-160: stratified k-fold cross-validation on a TU dataset (cal_amd/tu.py): per fold a fresh model
    and Adam(lr, weight_decay), every epoch one training pass and one evaluation of the test fold; the reported test
    accuracy is taken at the epoch whose fold-mean test accuracy is highest (``test_acc`` / ``test_acc_c``) and at the
    one whose fold-mean ``test_acc_o`` is highest (``test_acc_o``), mean and std over folds.
    Returns a dict with those figures and the per-fold / per-epoch tensors."""
    from .tu import k_fold
    device = _device(args)
    train_accs, test_accs, test_accs_c, test_accs_o = [], [], [], []
    random_guess = 1.0 / dataset.num_classes
    for fold, (train_idx, test_idx, val_idx) in enumerate(zip(*k_fold(dataset, args.folds, args.epoch_select))):
        best_test_acc, best_epoch, best_test_acc_c, best_test_acc_o = 0, 0, 0, 0
        train_loader = DataLoader(dataset[train_idx], args.batch_size, shuffle=True)
        test_loader = DataLoader(dataset[test_idx], args.batch_size, shuffle=False)
        model = model_func(dataset.num_features, dataset.num_classes).to(device)
        optimizer = Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
        for epoch in range(1, args.epochs + 1):
            train_loss, loss_c, loss_o, loss_co, train_acc = train_causal_epoch(model, optimizer, train_loader, device, args)
            test_acc, test_acc_c, test_acc_o = eval_acc_causal(model, test_loader, device, args)
            train_accs.append(train_acc)
            test_accs.append(test_acc)
            test_accs_c.append(test_acc_c)
            test_accs_o.append(test_acc_o)
            if test_acc > best_test_acc:
                best_test_acc, best_epoch, best_test_acc_c, best_test_acc_o = test_acc, epoch, test_acc_c, test_acc_o
            log("Causal | dataset:[{}] fold:[{}] | Epoch:[{}/{}] Loss:[{:.4f}={:.4f}+{:.4f}+{:.4f}] Train:[{:.4f}] "
                "Test:[{:.2f}] Test_o:[{:.2f}] Test_c:[{:.2f}] (RG:{:.2f}) | Best Test:[{:.2f}] at Epoch:[{}] | "
                "Test_o:[{:.2f}] Test_c:[{:.2f}]".format(
                    getattr(args, "dataset", dataset.name), fold, epoch, args.epochs, train_loss, loss_c, loss_o, loss_co,
                    train_acc * 100, test_acc * 100, test_acc_o * 100, test_acc_c * 100, random_guess * 100,
                    best_test_acc * 100, best_epoch, best_test_acc_o * 100, best_test_acc_c * 100))
    shape = (args.folds, args.epochs)
    train_acc = torch.tensor(train_accs, dtype=torch.float64).view(shape)
    test_acc = torch.tensor(test_accs, dtype=torch.float64).view(shape)
    test_acc_c = torch.tensor(test_accs_c, dtype=torch.float64).view(shape)
    test_acc_o = torch.tensor(test_accs_o, dtype=torch.float64).view(shape)
    sel = test_acc.mean(dim=0).argmax().repeat(args.folds)
    sel_o = test_acc_o.mean(dim=0).argmax().repeat(args.folds)
    ar = torch.arange(args.folds)
    pick, pick_c, pick_o = test_acc[ar, sel], test_acc_c[ar, sel], test_acc_o[ar, sel_o]

    def std(t):
        return t.std().item() if t.numel() > 1 else 0.0

    res = dict(train_acc_mean=train_acc[:, -1].mean().item(), test_acc_mean=pick.mean().item(), test_acc_std=std(pick),
               test_acc_c_mean=pick_c.mean().item(), test_acc_c_std=std(pick_c), test_acc_o_mean=pick_o.mean().item(),
               test_acc_o_std=std(pick_o), random_guess=random_guess, train_acc=train_acc, test_acc=test_acc,
               test_acc_c=test_acc_c, test_acc_o=test_acc_o)
    log("sydall Final: Causal | Dataset:[{}] Model:[{}] | Test Acc: {:.2f}±{:.2f} | OTest: {:.2f}±{:.2f}, CTest: {:.2f}±{:.2f} "
        "(RG:{:.2f})".format(getattr(args, "dataset", dataset.name), getattr(args, "model", "?"), res["test_acc_mean"] * 100,
                             res["test_acc_std"] * 100, res["test_acc_o_mean"] * 100, res["test_acc_o_std"] * 100,
                             res["test_acc_c_mean"] * 100, res["test_acc_c_std"] * 100, random_guess * 100))
    return res
