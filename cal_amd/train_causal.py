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


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("cal_amd needs an MI355X (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    return torch.device("cuda")


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
    return acc_co, acc_c, acc_o


def train_causal_syn(train_set, val_set, test_set, model_func=None, args=None, log=print):
    """train_causal.py:11-61."""
    device = _device()
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
