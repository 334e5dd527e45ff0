"""The reference-shaped training loop on the GPU (train_causal.py:11-61,162-200): ``train_causal_epoch(model, optimizer,
loader, device, args)`` with an ``Adam`` object, as the reference's entry scripts drive it.

* the one-call fused body (``cal_engine_step`` per mini-batch behind the unchanged signature) returns the reference's
  tuple, equal to the CPU oracle's epoch on the same batches / the same Python-RNG permutations, and leaves the optimizer
  object in the state the statement-by-statement loop leaves it in (moments, step counters, parameters);
* LR schedulers act through ``param_groups[0]["lr"]`` as with any torch optimizer;
* ``EngineAdam.step()`` after ``loss.backward()`` on the nn.Module surface equals ``torch.optim.Adam.step()``;
* the log lines of ``train_causal_syn`` are the reference's (train_causal.py:37-61)."""
import argparse
import copy
import random
import re

import numpy as np
import pytest
import torch

from oracle import cal_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _args(**kw):
    d = dict(layers=3, hidden=128, with_random=True, without_node_attention=False, without_edge_attention=False,
             fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5, eval_random=False)
    d.update(kw)
    return argparse.Namespace(**d)


def _model(name, sd, args, nfeat=10, ncls=4):
    from cal_amd import model as M
    m = getattr(M, name)(nfeat, ncls, args)
    m.load_state_dict(sd)
    m = m.to(DEV)
    if name == "CausalGAT":
        for c in m.convs:
            c.dropout = 0.0
    return m


def _graphs(n, seed=11):
    from cal_amd import spmotif
    return spmotif.train_mix(n, seed=seed)


@pytest.mark.parametrize("name", ["CausalGCN", "CausalGAT"])
def test_fused_epoch_equals_the_oracle_epoch(name):
    """One epoch (3 batches of 32 + one of 7) through train_causal_epoch's fused body vs oracle.CpuTrainer stepping the same
    batches with the same ``random.shuffle`` permutations: the returned tuple (train_causal.py:194-200) and the post-epoch
    parameters."""
    from cal_amd.data import Batch, DataLoader
    from cal_amd.device_data import DeviceDataset, DeviceLoader
    from cal_amd.optim import EngineAdam
    from cal_amd.train_causal import train_causal_epoch
    gs = _graphs(103)
    args = _args(layers=2, hidden=64)
    torch.manual_seed(3)
    sd = O.init_state(name, 10, 4, hidden=64, layers=2, heads=4)
    m = _model(name, {k: v.clone() for k, v in sd.items()}, args)
    # eps = 1e-3: with the default 1e-8 the first Adam steps move every element by ~lr * sign(gradient), so elements whose
    # gradient is rounding noise take opposite +-lr steps in two correct fp32 implementations and the trajectories drift
    # apart by more than the 1e-4 the losses are compared at (measured: 1.2e-4 after four steps at lr 2e-3)
    opt = EngineAdam(m.parameters(), lr=2e-3, eps=1e-3)
    loader = DeviceLoader(DeviceDataset(gs), 32, shuffle=False)
    random.seed(99)
    out = train_causal_epoch(m, opt, loader, torch.device(DEV), args)
    assert getattr(opt, "_cal_binding", None) is not None            # the fused body ran
    # oracle: same batches, same Python-RNG stream
    tr = O.CpuTrainer(name, {k: v.clone() for k, v in sd.items()}, 4, lr=2e-3, layers=2, heads=4,
                      **({"gat_dropout": 0.0} if name == "CausalGAT" else {}))
    tr.opt.param_groups[0]["eps"] = 1e-3
    random.seed(99)
    tot = np.zeros(5)
    for s in range(0, len(gs), 32):
        b = Batch.from_data_list(gs[s:s + 32])
        perm = torch.tensor(O.intervention_perm(b.num_graphs, True, True, name))
        loss, lc, lo, lco, logits = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
        n = b.num_graphs
        tot += [loss.item() * n, lc.item() * n, lo.item() * n, lco.item() * n,
                logits[1].max(1)[1].eq(b.y.view(-1)).sum().item()]
    ref = tot / len(gs)
    assert np.allclose(out[:4], ref[:4], atol=1e-4), (out, ref)
    assert abs(out[4] - ref[4]) <= 1.0 / len(gs) + 1e-9              # at most one borderline argmax
    # four Adam steps: an element whose gradient two correct fp32 implementations round differently (a ReLU at its
    # boundary) may have moved differently -- by at most ~lr per step; all but a handful must agree closely
    for k, p in m.named_parameters():
        d = (p.detach().cpu() - tr.sd[k].detach()).abs()
        tol = 2e-4 + 1e-3 * tr.sd[k].detach().abs()
        assert d.max().item() < 4 * 2e-3, (k, d.max().item())
        assert (d > tol).float().mean().item() < 2e-3, (k, (d > tol).float().mean().item())
    # the optimizer object saw four steps
    assert all(float(opt.state[p]["step"]) == 4.0 for p in m.parameters())


def test_fused_epoch_and_statement_loop_leave_the_same_optimizer_state():
    """Same data, identity permutation: fused body vs ``no_fused_step`` (model(data) -> torch loss -> backward ->
    optimizer.step()); afterwards BOTH continue with one statement-by-statement step of a plain torch.optim.Adam built from
    the first optimizer's state_dict -- the moments / counters the fused path left must be what Adam expects."""
    from cal_amd.device_data import DeviceDataset, DeviceLoader
    from cal_amd.train_causal import causal_loss, train_causal_epoch
    gs = _graphs(96, seed=5)
    torch.manual_seed(1)
    sd = O.init_state("CausalGCN", 10, 4, hidden=64, layers=2)
    res = {}
    for kind in ("fused", "statement"):
        args = _args(layers=2, hidden=64, with_random=False, no_fused_step=(kind == "statement"))
        m = _model("CausalGCN", {k: v.clone() for k, v in sd.items()}, args)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, weight_decay=1e-4, eps=1e-3)      # (eps: see the test above)
        loader = DeviceLoader(DeviceDataset(gs), 32, shuffle=False)
        out = train_causal_epoch(m, opt, loader, torch.device(DEV), args)
        assert (getattr(opt, "_cal_binding", None) is not None) == (kind == "fused")
        st = opt.state_dict()
        steps = {float(s["step"]) for s in st["state"].values()}
        assert steps == {3.0}, steps
        # one more step the plain way, driven by the SAME optimizer object
        b = next(iter(loader))
        opt.zero_grad()
        c, o, co = m(b, eval_random=False)
        loss, *_ = causal_loss(c, o, co, b.y, 4, args)
        loss.backward()
        opt.step()
        res[kind] = (out, {k: p.detach().cpu().clone() for k, p in m.named_parameters()},
                     {k: opt.state[p]["exp_avg_sq"].detach().cpu().clone() for k, p in m.named_parameters()},
                     {float(opt.state[p]["step"]) for p in m.parameters()})
    assert np.allclose(res["fused"][0], res["statement"][0], atol=2e-5)
    assert res["fused"][3] == res["statement"][3] == {4.0}
    for k in res["fused"][1]:
        assert torch.allclose(res["fused"][1][k], res["statement"][1][k], atol=5e-5, rtol=1e-3), k
        a, b_ = res["fused"][2][k], res["statement"][2][k]
        assert torch.allclose(a, b_, atol=1e-7 + 1e-3 * float(b_.abs().max()), rtol=1e-2), k


def test_optimizer_restore_after_the_bind_reaches_the_fused_step():
    """``optimizer.load_state_dict()`` between two fused epochs (round-3 advisor): the restored moments are fresh tensors, the
    engine must adopt them (``Binding.rehome``) instead of stepping on with its own -- the second epoch after a restore of the
    post-epoch-1 state must reproduce the uninterrupted second epoch bit for bit, and after a restore of a ZEROED state it
    must differ from it."""
    from cal_amd.device_data import DeviceDataset, DeviceLoader
    from cal_amd.train_causal import train_causal_epoch
    gs = _graphs(64, seed=7)
    torch.manual_seed(2)
    sd = O.init_state("CausalGCN", 10, 4, hidden=64, layers=2)
    args = _args(layers=2, hidden=64, with_random=False)
    outs = {}
    for kind in ("straight", "restored", "zeroed"):
        m = _model("CausalGCN", {k: v.clone() for k, v in sd.items()}, args)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        loader = DeviceLoader(DeviceDataset(gs), 32, shuffle=False)
        train_causal_epoch(m, opt, loader, torch.device(DEV), args)
        assert getattr(opt, "_cal_binding", None) is not None
        if kind != "straight":
            st = copy.deepcopy(opt.state_dict())
            if kind == "zeroed":
                for s_ in st["state"].values():
                    s_["exp_avg"].zero_(); s_["exp_avg_sq"].zero_()
            opt.load_state_dict(st)
            p0 = next(iter(m.parameters()))
            assert opt.state[p0]["exp_avg"].data_ptr() != m._engine.exp_avg.data_ptr()        # fresh tensors, not the engine's
        train_causal_epoch(m, opt, loader, torch.device(DEV), args)
        p0 = next(iter(m.parameters()))
        assert opt.state[p0]["exp_avg"].data_ptr() == m._engine.exp_avg.data_ptr()            # re-homed
        assert {float(opt.state[p]["step"]) for p in m.parameters()} == {4.0}
        outs[kind] = torch.cat([p.detach().flatten().cpu() for p in m.parameters()])
    assert torch.equal(outs["straight"], outs["restored"])
    assert not torch.equal(outs["straight"], outs["zeroed"])


def test_fused_causal_loss_equals_the_torch_formulation():
    """``causal_loss`` on the three heads of an engine-backed model is ONE launch and one autograd node (cal_causal_loss): its
    four values and the parameter gradients behind ``loss.backward()`` equal the torch formulation of train_causal.py:176-183
    (F.kl_div batchmean + two F.nll_loss) applied to the same heads -- also when the loss is scaled before backward and
    when a single term is differentiated on its own."""
    import torch.nn.functional as F
    from cal_amd.data import Batch
    from cal_amd.train_causal import causal_loss
    gs = _graphs(24, seed=3)
    bd = Batch.from_data_list(gs).to(DEV)
    args = _args(layers=2, hidden=64)
    torch.manual_seed(4)
    sd = O.init_state("CausalGCN", 10, 4, hidden=64, layers=2)
    perm = torch.randperm(24)

    def torch_loss(c, o, co, y):
        u = torch.ones_like(c) / 4
        lc, lo, lco = F.kl_div(c, u, reduction="batchmean"), F.nll_loss(o, y), F.nll_loss(co, y)
        return args.c * lc + args.o * lo + args.co * lco, lc, lo, lco

    res = {}
    for kind in ("fused", "torch", "fused_scaled", "torch_scaled", "fused_term", "torch_term"):
        m = _model("CausalGCN", {k: v.clone() for k, v in sd.items()}, args)
        m.train()
        c, o, co = m(bd, eval_random=True, perm=perm)
        if kind.startswith("fused"):
            out = causal_loss(c, o, co, bd.y, 4, args)
            assert type(out[0].grad_fn).__name__.startswith("_FusedCausalLoss"), type(out[0].grad_fn).__name__
        else:
            out = torch_loss(c, o, co, bd.y.view(-1))
        if kind.endswith("scaled"):
            (out[0] * 0.25).backward()
        elif kind.endswith("term"):
            (out[0] + 0.3 * out[2]).backward()
        else:
            out[0].backward()
        res[kind] = ([float(t.detach()) for t in out], {k: p.grad.detach().cpu().clone() for k, p in m.named_parameters() if p.grad is not None})
    for a, b in (("fused", "torch"), ("fused_scaled", "torch_scaled"), ("fused_term", "torch_term")):
        assert np.allclose(res[a][0], res[b][0], atol=1e-6, rtol=1e-6), (a, res[a][0], res[b][0])
        assert res[a][1].keys() == res[b][1].keys()
        for k in res[a][1]:
            ga, gb = res[a][1][k], res[b][1][k]
            assert torch.allclose(ga, gb, atol=1e-6 + 1e-5 * float(gb.abs().max()), rtol=1e-4), (a, k)


def test_fused_causal_loss_flags_a_label_outside_the_class_range():
    """F.nll_loss in the reference (train_causal.py:178-180) stops with a device assert on a label outside [0, C); the fused
    loss gives such a graph zero loss, so it raises the sticky flag and ``check_loss_labels`` (called where the loops read their
    statistics back) turns it into an error (advisor, round 4)."""
    from cal_amd import _lib
    from cal_amd.data import Batch
    from cal_amd.engine import check_loss_labels
    from cal_amd.train_causal import causal_loss
    gs = _graphs(8, seed=5)
    bd = Batch.from_data_list(gs).to(DEV)
    args = _args(layers=2, hidden=64)
    torch.manual_seed(1)
    m = _model("CausalGCN", O.init_state("CausalGCN", 10, 4, hidden=64, layers=2), args)
    m.train()
    c, o, co = m(bd, eval_random=False)
    causal_loss(c, o, co, bd.y, 4, args)
    check_loss_labels()                                   # all labels in range: nothing
    y_bad = bd.y.clone()
    y_bad[3] = 7
    c, o, co = m(bd, eval_random=False)
    out = causal_loss(c, o, co, y_bad, 4, args)
    assert type(out[0].grad_fn).__name__.startswith("_FusedCausalLoss")
    with pytest.raises(_lib.CalError, match="label outside"):
        check_loss_labels()
    check_loss_labels()                                   # cleared by the raise


def test_lr_scheduler_reaches_the_fused_step():
    """CosineAnnealingLR rewrites param_groups[0]['lr'] (train_causal.py:22,29); the next fused epoch must use it: two
    models, one stepped with lr = 0 after the schedule, stay / move accordingly."""
    from cal_amd.device_data import DeviceDataset, DeviceLoader
    from cal_amd.optim import EngineAdam
    from cal_amd.train_causal import train_causal_epoch
    gs = _graphs(64, seed=8)
    args = _args(layers=2, hidden=64, with_random=False)
    torch.manual_seed(2)
    sd = O.init_state("CausalGCN", 10, 4, hidden=64, layers=2)
    m = _model("CausalGCN", {k: v.clone() for k, v in sd.items()}, args)
    opt = EngineAdam(m.parameters(), lr=1e-3)
    loader = DeviceLoader(DeviceDataset(gs), 32, shuffle=False)
    train_causal_epoch(m, opt, loader, torch.device(DEV), args)
    before = m.convs[0].weight.detach().clone()
    opt.param_groups[0]["lr"] = 0.0                                   # what a scheduler does
    train_causal_epoch(m, opt, loader, torch.device(DEV), args)
    assert torch.equal(before, m.convs[0].weight.detach())
    opt.param_groups[0]["lr"] = 1e-2
    train_causal_epoch(m, opt, loader, torch.device(DEV), args)
    assert (before - m.convs[0].weight.detach()).abs().max().item() > 1e-4


def test_engine_adam_step_equals_torch_adam_on_the_module_surface():
    from cal_amd.data import Batch
    from cal_amd.optim import EngineAdam
    from cal_amd.train_causal import causal_loss
    gs = _graphs(32, seed=4)
    args = _args(layers=2, hidden=64, with_random=False)
    torch.manual_seed(7)
    sd = O.init_state("CausalGCN", 10, 4, hidden=64, layers=2)
    out = {}
    for kind, cls in (("engine", EngineAdam), ("torch", torch.optim.Adam)):
        m = _model("CausalGCN", {k: v.clone() for k, v in sd.items()}, args).train()
        opt = cls(m.parameters(), lr=3e-3, betas=(0.8, 0.99), eps=1e-3, weight_decay=1e-3)
        b = Batch.from_data_list(gs).to(DEV)
        for _ in range(3):
            opt.zero_grad()
            c, o, co = m(b, eval_random=False)
            loss, *_ = causal_loss(c, o, co, b.y, 4, args)
            loss.backward()
            opt.step()
        if kind == "engine":
            assert getattr(opt, "_cal_binding", None) is not None and opt._cal_binding.engine_step == 3.0
            assert {float(s["step"]) for s in opt.state_dict()["state"].values()} == {3.0}
        out[kind] = {k: p.detach().cpu().clone() for k, p in m.named_parameters()}
    for k in out["engine"]:
        assert torch.allclose(out["engine"][k], out["torch"][k], atol=2e-5, rtol=1e-3), k


def test_train_causal_syn_log_lines_and_device_loader():
    """train_causal_syn on the GPU: device-resident loaders, fused epochs, and the two log lines of train_causal.py:37-61."""
    from functools import partial
    from cal_amd import model as M
    from cal_amd.train_causal import train_causal_syn
    train, val, test = _graphs(96, seed=1), _graphs(32, seed=2), _graphs(40, seed=3)
    args = _args(batch_size=32, feature_dim=-1, max_degree=10, num_classes=4, lr=1e-3, epochs=2, min_lr=1e-6,
                 bias=0.9, model="CausalGCN")
    torch.manual_seed(5)
    random.seed(5)
    lines = []
    model, history = train_causal_syn(train, val, test, model_func=partial(M.CausalGCN, args=args), args=args, log=lines.append)
    assert len(lines) == 3
    pat = (r"^BIAS:\[0\.90\] \| Model:\[CausalGCN\] Epoch:\[\d/2\] Loss:\[\d+\.\d{4}=\d+\.\d{4}\+\d+\.\d{4}\+\d+\.\d{4}\] "
           r"Train:\[\d+\.\d{2}\] val:\[\d+\.\d{2}\] Test:\[\d+\.\d{2}\] \| Update Test:\[co:\d+\.\d{2},c:\d+\.\d{2},o:\d+\.\d{2}\] "
           r"at Epoch:\[\d\] \| lr:0\.\d{6}$")
    assert re.match(pat, lines[0]), lines[0]
    assert re.match(pat, lines[1]), lines[1]
    assert re.match(r"^syd: BIAS:\[0\.90\] \| Val acc:\[\d+\.\d{2}\] Test acc:\[co:\d+\.\d{2},c:\d+\.\d{2},o:\d+\.\d{2}\] at epoch:\[\d\]$", lines[2]), lines[2]
    assert lines[0][-11:] in ("lr:0.000500", "lr:0.000501") and lines[1].endswith("lr:0.000001")   # cosine: midpoint, then eta_min
    # "Val acc" of the final line is the LAST epoch's val_acc_o (train_causal.py:55-57), not the best epoch's
    assert "Val acc:[%.2f]" % (history[-1]["val_acc_o"] * 100) in lines[2]
    assert getattr(model, "_engine", None) is not None
    for h in history:
        assert abs(h["loss"] - (0.5 * h["loss_c"] + h["loss_o"] + 0.5 * h["loss_co"])) < 1e-5


@pytest.mark.parametrize("name,batch", [("CausalGCN", 32), ("CausalGAT", 24), ("CausalGCN", 300)])
def test_eval_acc_causal_counts_on_the_device_like_the_statement_loop(name, batch):
    """eval_acc_causal (train_causal.py:202-223): the engine path counts the three heads' hits inside the readout kernel
    (stats[4:7]); the statement-by-statement loop (``no_fused_step``) takes argmax / eq / sum of the returned log-probs.
    Same tuple (acc_co, acc_c, acc_o), also with ``eval_random`` shuffling (same Python-RNG stream)."""
    from cal_amd.device_data import DeviceDataset, DeviceLoader
    from cal_amd.train_causal import eval_acc_causal
    gs = _graphs(2 * batch + 5, seed=21)
    torch.manual_seed(4)
    sd = O.init_state(name, 10, 4, hidden=64, layers=2, heads=4)
    for eval_random in (False, True):
        out = {}
        for kind in ("engine", "statement"):
            args = _args(layers=2, hidden=64, eval_random=eval_random, no_fused_step=(kind == "statement"))
            m = _model(name, {k: v.clone() for k, v in sd.items()}, args)
            loader = DeviceLoader(DeviceDataset(gs), batch, shuffle=False, pack=False)
            random.seed(31)
            out[kind] = eval_acc_causal(m, loader, torch.device(DEV), args)
        assert len(out["engine"]) == 3 and all(0.0 <= v <= 1.0 for v in out["engine"])
        # identical up to one graph per head whose two best classes tie within rounding
        for a, b in zip(out["engine"], out["statement"]):
            assert abs(a - b) <= 1.0 / len(gs) + 1e-12, (eval_random, out)
