"""libcalhost.so -- the plain-C++ HOST implementation of the operator-level C-ABI (VERDICT r1 row b') --
against the oracle, on CPU tensors, in this GPU-less container: every operator (forward and autograd backward), the three
causal models end to end (logits 1e-4, gradients), the dropout mask, and BASELINE.json configs[0] -- `main_syn.py --model
CausalGCN --bias 0.9`, batch 32, SPMotif node_num 15 on the CPU -- through `train_causal_syn`.  The same wrappers
(`cal_amd.ops`) route CUDA tensors to libcalhip.so; these tests are the host twin of tests/test_gpu_ops.py /
test_gpu_model.py."""
import argparse

import numpy as np
import pytest
import torch

from oracle import cal_oracle as O
from tests.helpers import random_graph_batch, ref_batch, ref_graphs

LOGIT_TOL = 1e-4


def _args(**kw):
    d = dict(layers=2, hidden=32, with_random=True, without_node_attention=False, without_edge_attention=False,
             fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
    d.update(kw)
    return argparse.Namespace(**d)


def _plan(b):
    from cal_amd.plan import GraphPlan
    x = b.x if b.x is not None else b.feat
    return GraphPlan(b.edge_index, x.size(0), b.batch, b.num_graphs, validate=True)


@pytest.mark.parametrize("weighted,improved,relu", [(False, False, True), (True, False, False), (True, True, True)])
def test_gcn_aggregate_host_matches_oracle(weighted, improved, relu):
    from cal_amd import ops
    b = random_graph_batch(num_graphs=6, n_lo=1, n_hi=14, p=0.3, feat=8, seed=3, self_loops=True, directed=True)
    plan = _plan(b)
    torch.manual_seed(0)
    h = torch.randn(b.x.size(0), 8, requires_grad=True)
    w = torch.rand(b.edge_index.size(1), requires_grad=True) if weighted else None
    bias = torch.randn(8, requires_grad=True)
    out = ops.gcn_aggregate(h, plan, w, bias, improved, relu)
    h2 = h.detach().clone().requires_grad_(True)
    w2 = w.detach().clone().requires_grad_(True) if weighted else None
    b2 = bias.detach().clone().requires_grad_(True)
    ei, norm = O.gcn_norm(b.edge_index, b.x.size(0), w2, improved)
    ref = O.propagate_add(ei, h2, norm) + b2
    ref = torch.relu(ref) if relu else ref
    assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5)
    g = torch.randn_like(out)
    out.backward(g); ref.backward(g)
    assert torch.allclose(h.grad, h2.grad, atol=1e-5, rtol=1e-4) and torch.allclose(bias.grad, b2.grad, atol=1e-5, rtol=1e-4)
    if weighted:
        assert torch.allclose(w.grad, w2.grad, atol=1e-5, rtol=1e-4)


def test_attention_pool_and_linear_host_match_torch():
    from cal_amd import ops
    b = random_graph_batch(num_graphs=5, n_lo=2, n_hi=12, p=0.35, feat=12, seed=5)
    plan = _plan(b)
    torch.manual_seed(1)
    N, H = b.x.size(0), 12
    x = torch.randn(N, H, requires_grad=True)
    We, be = torch.randn(2, 2 * H, requires_grad=True), torch.randn(2, requires_grad=True)
    Wn, bn = torch.randn(2, H, requires_grad=True), torch.randn(2, requires_grad=True)
    Wl, bl = torch.randn(7, H, requires_grad=True), torch.randn(7, requires_grad=True)
    att = ops.edge_attention(x, We, be, plan)
    xc, xo, natt = ops.node_attention_split(x, Wn, bn)
    pooled = ops.add_pool(xc + 2 * xo, plan)
    y = ops.linear(pooled, Wl, bl, relu=True)
    loss = (att[0] * torch.arange(att.size(1))).sum() + y.pow(2).sum()
    loss.backward()
    got = [t.grad.clone() for t in (x, We, be, Wn, bn, Wl, bl)]
    for t in (x, We, be, Wn, bn, Wl, bl):
        t.grad = None
    row, col = b.edge_index
    att_r = torch.softmax(torch.cat([x[row], x[col]], -1) @ We.t() + be, -1).t()
    na = torch.softmax(x @ Wn.t() + bn, -1)
    pooled_r = O.global_add_pool(na[:, :1] * x + 2 * na[:, 1:] * x, b.batch, b.num_graphs)
    y_r = torch.relu(pooled_r @ Wl.t() + bl)
    assert torch.allclose(att, att_r, atol=1e-6) and torch.allclose(natt, na, atol=1e-6) and torch.allclose(y, y_r, atol=1e-4, rtol=1e-5)
    ((att_r[0] * torch.arange(att.size(1))).sum() + y_r.pow(2).sum()).backward()
    for a, t in zip(got, (x, We, be, Wn, bn, Wl, bl)):
        assert torch.allclose(a, t.grad, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("p", [0.0, 0.3])
def test_gat_aggregate_host_matches_oracle(p):
    from cal_amd import ops
    b = random_graph_batch(num_graphs=5, n_lo=1, n_hi=12, p=0.3, feat=4, seed=9, self_loops=True, directed=True)
    plan = _plan(b)
    K, D = 2, 8
    torch.manual_seed(2)
    x = torch.randn(b.x.size(0), 6)
    W = torch.randn(6, K * D, requires_grad=True)
    att = (0.3 * torch.randn(1, K, 2 * D)).requires_grad_(True)
    bias = torch.randn(K * D, requires_grad=True)
    z = ops.matmul(x, W)
    out = ops.gat_aggregate(z, att, bias, plan, K, 0.2, p, 77, True)
    mask = None
    if p > 0:
        full = ops.gat_dropout_mask(77, plan, K, p)
        row, col = b.edge_index
        keep_e = (row != col).nonzero().view(-1)
        mask = torch.cat([full[keep_e], full[plan.E:]], 0)
    W2, a2, b2 = (t.detach().clone().requires_grad_(True) for t in (W, att, bias))
    ref = torch.relu(O.gat_conv(x, b.edge_index, W2, a2, b2, K, 0.2, p, True, mask))
    assert torch.allclose(out, ref, atol=1e-5, rtol=1e-5)
    g = torch.randn_like(out)
    out.backward(g); ref.backward(g)
    for a, r in ((W, W2), (att, a2), (bias, b2)):
        assert torch.allclose(a.grad, r.grad, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("name", ["CausalGCN", "CausalGAT", "CausalGIN"])
def test_causal_models_on_the_host_library_match_oracle(name):
    """The nn.Module surface on CPU tensors: operator-level path on libcalhost.so, torch BatchNorm / loss / autograd."""
    from cal_amd import model as M
    from cal_amd.train_causal import causal_loss
    ids = list(range(12))
    b = ref_batch(ids)
    torch.manual_seed(4)
    sd = O.init_state(name, 10, 4, hidden=32, layers=2, heads=4)
    args = _args()
    m = getattr(M, name)(10, 4, args)
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=name != "CausalGIN")      # (GINConv keeps an `eps` buffer)
    m.train()
    if name == "CausalGAT":
        for c in m.convs:
            c.dropout = 0.0
    perm = torch.randperm(len(ids))
    c, o, co = m(ref_batch(ids), eval_random=True, perm=perm)
    assert not c.is_cuda and getattr(m, "_engine", None) is None
    loss, *_ = causal_loss(c, o, co, b.y, 4, args)
    loss.backward()
    tr = O.CpuTrainer(name, {k: v.clone() for k, v in sd.items()}, 4, lr=1e-3, layers=2, heads=4, gat_dropout=0.0)
    ref_loss, _, _, _, logits = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
    for r, t in zip(logits, (c, o, co)):
        assert (r.detach() - t.detach()).abs().max().item() < LOGIT_TOL
    assert abs(ref_loss.item() - loss.item()) < 1e-5
    for k, p in m.named_parameters():
        gref = tr.sd[k].grad
        if gref is not None:
            assert torch.allclose(p.grad, gref, atol=2e-5, rtol=1e-3), k
    m.eval()
    sde = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.endswith(".eps")}
    with torch.no_grad():
        ev = m(ref_batch(ids), eval_random=False, perm=perm)
    ref = O.causal_forward(name, sde, b.feat, b.edge_index, b.batch, perm=perm, training=False, layers=2, heads=4)
    for r, t in zip(ref, ev):
        assert (r - t).abs().max().item() < LOGIT_TOL


def test_config0_cpu_plumbing_train_causal_syn():
    """BASELINE.json configs[0]: `main_syn.py --model CausalGCN --bias 0.9` on CPU, batch 32, the reference's default
    SPMotif shape (node_num 15) -- one epoch of train_causal_syn through the host library; tuple / log shape of
    train_causal.py:24-61,194-200."""
    from functools import partial
    from cal_amd import model as M, spmotif
    from cal_amd.train_causal import train_causal_syn
    train = spmotif.train_mix(64, node_num=15, seed=1)
    val = spmotif.train_mix(32, node_num=15, seed=2)
    test = spmotif.train_mix(32, node_num=15, seed=3)
    args = _args(layers=3, hidden=128, batch_size=32, feature_dim=-1, max_degree=10, num_classes=4, lr=1e-3, epochs=1,
                 min_lr=1e-6, bias=0.9, model="CausalGCN", eval_random=False, device="cpu")
    torch.manual_seed(5)
    lines = []
    model, history = train_causal_syn(train, val, test, model_func=partial(M.CausalGCN, args=args), args=args, log=lines.append)
    assert not next(model.parameters()).is_cuda and args.feature_dim == 10
    assert len(history) == 1 and len(lines) == 2 and lines[-1].startswith("syd: BIAS:[0.90]")
    h = history[0]
    assert all(np.isfinite(h[k]) for k in ("loss", "loss_c", "loss_o", "loss_co", "train_acc_o", "val_acc_o", "test_acc_o"))
    assert abs(h["loss"] - (0.5 * h["loss_c"] + h["loss_o"] + 0.5 * h["loss_co"])) < 1e-5


from tests.helpers import SWEEP_CASES as _SWEEP  # noqa: E402


@pytest.mark.parametrize("seed,name,kw,case", _SWEEP, ids=[str(s[0]) for s in _SWEEP])
def test_random_shape_sweep_cases_on_the_host_library(seed, name, kw, case):
    """Ten draws of tests/tools/fuzz_host.py kept as fixed cases: (hidden, layers, features, classes, ragged graph sizes
    with single nodes / edgeless graphs / stars / self loops) x model variant, one forward + external loss + backward
    through libcalhost.so, judged against the oracle's fp64 step (8x the fp32 oracle's own distance, floor 1e-4 of scale).
    The ablation flags on a GAT / GIN model change nothing, as in the reference (model.py:236-264,380-409)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import fuzz_host
    assert fuzz_host.run(case, seed, name, kw) == []


def test_causalgin_forward_keeps_the_reference_signature_with_train_type():
    """model.py:234,281-292: ``CausalGIN.forward(data, eval_random=True, train_type="base")``; ``"irm"`` makes the objects head
    return (raw logits, log-probs).  Dead in the reference's loops, but a caller passing it must not get a TypeError."""
    from cal_amd import model as M
    ids = list(range(6))
    torch.manual_seed(8)
    m = M.CausalGIN(10, 4, _args()).eval()
    perm = torch.arange(len(ids))
    with torch.no_grad():
        base = m(ref_batch(ids), True, "base", perm=perm)
        c, (raw, o), co = m(ref_batch(ids), eval_random=True, train_type="irm", perm=perm)
        old = m(ref_batch(ids), True, perm)                      # the base-class argument order still works
    assert torch.allclose(o, torch.log_softmax(raw, dim=-1), atol=1e-6)
    for r, t in zip(base, (c, o, co)):
        assert torch.allclose(r, t, atol=1e-6)
    for r, t in zip(base, old):
        assert torch.allclose(r, t, atol=1e-6)
    m.train()
    c, (raw, o), co = m(ref_batch(ids), True, "irm", perm=perm)
    raw.square().mean().backward()                               # the raw logits are differentiable (an IRM penalty needs that)
    assert m.fc2_o.weight.grad is not None and m.fc2_o.weight.grad.abs().sum().item() > 0
