"""Causal training / evaluation loops with the reference's function surface
(train_causal.py:11-61, 162-223): same loss (KL to uniform + 2x NLL, weights
args.c / args.o / args.co), Adam + cosine schedule, same returned tuples.

Differences that do not change results: the five ``.item()`` host syncs per
iteration (train_causal.py:186-191) are replaced by on-device accumulators read
once per epoch, and ``CosineAnnealingLR`` is built without the ``verbose``
keyword current torch rejects.

On a GPU the loop body of ``train_causal_epoch`` -- forward, the 3-term loss,
backward and the optimizer step (train_causal.py:173-192) -- is ONE
``cal_engine_step`` call per mini-batch when the model is engine-backed and the
optimizer is a plain Adam over its parameters (``cal_amd.optim.bind``): same
signature, same returned tuple, the optimizer object and any LR scheduler keep
working (its moments are views of the engine's, its ``step`` counters are kept in
sync).  Anything else -- another optimizer, a ``grad_sync`` hook, a CPU model, a
model variant the engine does not cover -- runs the statement-by-statement loop.
``train_causal_syn`` / ``train_causal_real`` feed it from a device-resident
dataset with on-GPU collate (``cal_amd.device_data``) instead of the per-graph
Python collate.
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
        from .engine import check_loss_labels
        check_loss_labels()


def num_graphs(data):
    """utils.py:12-16."""
    if data.batch is not None:
        return data.num_graphs
    return data.x.size(0)


def causal_loss(c_logs, o_logs, co_logs, y, num_classes, args):
    """train_causal.py:176-183.  The three heads of an engine-backed model arrive as consecutive blocks of one buffer: the four
    values and the gradient w.r.t. the heads then come from ONE launch and one autograd node (``engine.fused_causal_loss``)
    instead of ~20 small torch launches; anything else takes the torch formulation below."""
    if c_logs.is_cuda:
        from .engine import fused_causal_loss
        res = fused_causal_loss(c_logs, o_logs, co_logs, y, num_classes, args.c, args.o, args.co)
        if res is not None:
            return res
    one_hot_target = y.view(-1)
    uniform_target = torch.ones_like(c_logs, dtype=torch.float) / num_classes
    c_loss = F.kl_div(c_logs, uniform_target, reduction="batchmean")
    o_loss = F.nll_loss(o_logs, one_hot_target)
    co_loss = F.nll_loss(co_logs, one_hot_target)
    loss = args.c * c_loss + args.o * o_loss + args.co * co_loss
    return loss, c_loss, o_loss, co_loss


def _loader(dataset, batch_size, shuffle, device):
    """``DataLoader(dataset, batch_size, shuffle)`` (train_causal.py:13-15,72-73): on a GPU the dataset is made
    device-resident once and every mini-batch is assembled by one collate kernel; the host ``DataLoader`` elsewhere.  The
    batches hold the same graphs as the host loader's; for datasets of small graphs (mean <= 40 nodes) their ORDER inside a
    mini-batch is the tile packing's (``Batch.order`` = dataset indices in batch order) -- the loops here only sum losses and
    count hits over a batch, and the intervention permutation is random, so no result depends on it; with Python's RNG
    stream unchanged the permutation pairs other graphs than the reference would for the same seed."""
    if device.type == "cuda" and len(dataset) > 0:
        from .device_data import DeviceDataset, DeviceLoader
        graphs = dataset if isinstance(dataset, (list, tuple)) else [dataset[i] for i in range(len(dataset))]
        return DeviceLoader(DeviceDataset(graphs, device=device), batch_size, shuffle=shuffle, pack="small")
    return DataLoader(dataset, batch_size, shuffle=shuffle)


def _fused_epoch(model, binding, loader, device, args):
    """The loop of train_causal.py:171-192 with its body as one engine call per mini-batch (module docstring)."""
    eng = binding.engine
    eng.wc, eng.wo, eng.wco = float(args.c), float(args.o), float(args.co)
    binding.resync()
    acc = torch.zeros(5, dtype=torch.float64, device=device)      # loss, c, o, co (each x graphs), correct
    weights = {}
    device_perm = bool(getattr(args, "device_perm", False))       # not a reference flag: draw the permutation on the GPU
    if device_perm and getattr(eng, "_perm_counter", None) is None:
        import random
        eng.set_perm_rng(random.getrandbits(63), torch.zeros(1, dtype=torch.int64, device=device))
    stage = eng.perm_stage()
    try:
        for data in loader:
            if eng.peek_status():        # host-mapped mirror, no sync: an earlier step flagged its batch (and updated nothing
                eng.check_status()       # since): raise now, not at the end of the epoch
            data = data.to(device)
            n = num_graphs(data)
            shuffles = bool(args.with_random and (model.with_random if model._gate_on_with_random else True))
            if device_perm and shuffles and n <= 1024:
                stats = eng.train_step(data, None, adam=True, draw_perm=True)
            else:
                # model.py:147-152: Python's RNG on the host, exactly the reference's stream
                perm = stage.put(model.intervention_list(n, args.with_random))
                stats = eng.train_step(data, perm, adam=True)
            binding.stepped()
            w = weights.get(n)
            if w is None:
                w = weights[n] = torch.tensor([n, n, n, n, 1], dtype=torch.float64, device=device)
            acc.addcmul_(stats.to(torch.float64), w)
    finally:
        binding.flush()
    return acc


def train_causal_epoch(model, optimizer, loader, device, args, grad_sync=None):
    """train_causal.py:162-200.  ``grad_sync`` (optional callable) runs between
    backward and the optimizer step -- the data-parallel gradient all-reduce."""
    model.train()
    device = torch.device(device)
    binding = None
    if grad_sync is None and device.type == "cuda" and getattr(model, "use_engine", False) \
            and not getattr(args, "no_fused_step", False):
        from .optim import bind
        binding = bind(optimizer, model)
    if binding is not None:
        acc = _fused_epoch(model, binding, loader, device, args)
    else:
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


def _engine_for_eval(model, device, args):
    """The model's step engine when an evaluation pass can count its hits on the device (engine-backed model on the GPU)."""
    if torch.device(device).type != "cuda" or not getattr(model, "use_engine", False) or getattr(args, "no_fused_step", False):
        return None
    get = getattr(model, "engine", None)
    return get() if callable(get) else None


def eval_acc_causal(model, loader, device, args):
    """train_causal.py:202-223.  Engine-backed models: one eval-mode forward call per mini-batch whose readout kernel also
    counts the three heads' hits (stats[4:7] = o, c, co) -- no log-prob copies, no argmax / compare / sum launches."""
    model.eval()
    eval_random = args.eval_random
    eng = _engine_for_eval(model, device, args)
    if eng is not None:
        hits = torch.zeros(3, dtype=torch.float64, device=device)              # o, c, co
        shuffles = bool(eval_random and (model.with_random if model._gate_on_with_random else True))
        stage = None
        for data in loader:
            data = data.to(device)
            perm = None                                                        # identity (model.py:147-152 with the gate off)
            if shuffles:
                if stage is None:
                    stage = eng.perm_stage()
                perm = stage.put(model.intervention_list(num_graphs(data), eval_random))
            eng.forward(data, perm, training=False)
            hits.add_(eng.buffer("stats", 8)[4:7])
        n = len(loader.dataset)
        acc_o, acc_c, acc_co = (hits / n).tolist()
        _check_engine(model)
        return acc_co, acc_c, acc_o
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


_SYN_EPOCH_LINE = ("BIAS:[{:.2f}] | Model:[{}] Epoch:[{}/{}] Loss:[{:.4f}={:.4f}+{:.4f}+{:.4f}] Train:[{:.2f}] val:[{:.2f}] "
                   "Test:[{:.2f}] | Update Test:[co:{:.2f},c:{:.2f},o:{:.2f}] at Epoch:[{}] | lr:{:.6f}")
_SYN_FINAL_LINE = "syd: BIAS:[{:.2f}] | Val acc:[{:.2f}] Test acc:[co:{:.2f},c:{:.2f},o:{:.2f}] at epoch:[{}]"


def train_causal_syn(train_set, val_set, test_set, model_func=None, args=None, log=print):
    """train_causal.py:11-61: Adam + cosine schedule over ``args.epochs`` epochs of the SPMotif split, one training pass and
    an evaluation of the validation and test sets per epoch; the reported test accuracies are those of the epoch with the
    best ``val_acc_o`` so far (model selection on the validation set, train_causal.py:29-35).  The two log lines are the
    reference's, character for character (they are what a user of main_syn.py greps).  Returns (model, per-epoch history)."""
    from .optim import EngineAdam
    device = _device(args)
    loaders = {name: _loader(ds, args.batch_size, name == "train", device)
               for name, ds in (("train", train_set), ("val", val_set), ("test", test_set))}
    if args.feature_dim == -1:
        args.feature_dim = args.max_degree                                      # train_causal.py:17-18
    model = model_func(args.feature_dim, args.num_classes).to(device)
    optimizer = EngineAdam(model.parameters(), lr=args.lr)
    schedule = CosineAnnealingLR(optimizer, T_max=args.epochs, eta_min=args.min_lr, last_epoch=-1)
    picked = dict(val=0.0, co=0.0, c=0.0, o=0.0, epoch=0)                        # test accuracies at the best validation epoch
    history = []
    for epoch in range(1, args.epochs + 1):
        loss, loss_c, loss_o, loss_co, train_acc_o = train_causal_epoch(model, optimizer, loaders["train"], device, args)
        _, _, val_acc_o = eval_acc_causal(model, loaders["val"], device, args)
        test_acc_co, test_acc_c, test_acc_o = eval_acc_causal(model, loaders["test"], device, args)
        schedule.step()
        if val_acc_o > picked["val"]:
            picked = dict(val=val_acc_o, co=test_acc_co, c=test_acc_c, o=test_acc_o, epoch=epoch)
        history.append(dict(epoch=epoch, loss=loss, loss_c=loss_c, loss_o=loss_o, loss_co=loss_co,
                            train_acc_o=train_acc_o, val_acc_o=val_acc_o, test_acc_o=test_acc_o))
        log(_SYN_EPOCH_LINE.format(args.bias, args.model, epoch, args.epochs, loss, loss_c, loss_o, loss_co,
                                   train_acc_o * 100, val_acc_o * 100, test_acc_o * 100, picked["co"] * 100,
                                   picked["c"] * 100, picked["o"] * 100, picked["epoch"],
                                   optimizer.param_groups[0]["lr"]))
    # "Val acc" is the LAST epoch's val_acc_o, as in the reference (train_causal.py:55-57), not the best one
    log(_SYN_FINAL_LINE.format(args.bias, val_acc_o * 100, picked["co"] * 100, picked["c"] * 100, picked["o"] * 100,
                               picked["epoch"]))
    return model, history


def train_causal_real(dataset=None, model_func=None, args=None, log=print):
    """train_causal.py:63-160: stratified k-fold cross-validation on a TU dataset (cal_amd/tu.py): per fold a fresh model
    and Adam(lr, weight_decay), every epoch one training pass and one evaluation of the test fold; the reported test
    accuracy is taken at the epoch whose fold-mean test accuracy is highest (``test_acc`` / ``test_acc_c``) and at the
    one whose fold-mean ``test_acc_o`` is highest (``test_acc_o``), mean and std over folds.
    Returns a dict with those figures and the per-fold / per-epoch tensors."""
    from .optim import EngineAdam
    from .tu import k_fold
    device = _device(args)
    train_accs, test_accs, test_accs_c, test_accs_o = [], [], [], []
    random_guess = 1.0 / dataset.num_classes
    for fold, (train_idx, test_idx, val_idx) in enumerate(zip(*k_fold(dataset, args.folds, args.epoch_select))):
        best_test_acc, best_epoch, best_test_acc_c, best_test_acc_o = 0, 0, 0, 0
        train_loader = _loader(dataset[train_idx], args.batch_size, True, device)
        test_loader = _loader(dataset[test_idx], args.batch_size, False, device)
        model = model_func(dataset.num_features, dataset.num_classes).to(device)
        optimizer = EngineAdam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
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
