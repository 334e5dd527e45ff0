"""GPU parity of the nn.Module surface (CausalGCN / CausalGAT) against the CPU
oracle: committed golden fixtures (eval logits, one training step) and seeded
batches at the BASELINE.json config-2 shape.  Logit tolerance 1e-4 (north_star)."""
import argparse
import os
import random

import numpy as np
import pytest
import torch

from oracle import cal_oracle as O
from tests.helpers import GOLDEN, ref_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOGIT_TOL = 1e-4


def _args(**kw):
    d = dict(layers=3, hidden=128, with_random=True, without_node_attention=False,
             without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
    d.update(kw)
    return argparse.Namespace(**d)


def _build(name, nfeat, ncls, args, sd=None, use_engine=None):
    from cal_amd import model as M
    m = getattr(M, name)(nfeat, ncls, args)
    if use_engine is not None:
        m.use_engine = use_engine
    if sd is not None:
        m.load_state_dict(sd)
    return m.to(DEV)


@pytest.mark.parametrize("name,fname,use_engine", [("CausalGCN", "causal_gcn_batch8.npz", False),
                                                   ("CausalGCN", "causal_gcn_batch8.npz", True),
                                                   ("CausalGAT", "causal_gat_batch8.npz", False),
                                                   ("CausalGAT", "causal_gat_batch8.npz", True)])
def test_golden_fixture_eval_and_train_step(name, fname, use_engine):
    """nn.Module surface + torch loss + torch Adam, on the operator-level path and on the native engine
    behind the same autograd surface."""
    fx = np.load(os.path.join(GOLDEN, fname))
    b = ref_batch(list(fx["ids"]))
    sd = {k[3:]: torch.from_numpy(fx[k]).clone() for k in fx.files if k.startswith("sd.")}
    perm = torch.from_numpy(fx["perm"])
    m = _build(name, 10, 4, _args(layers=2, hidden=32), sd, use_engine)
    bd = ref_batch(list(fx["ids"])).to(DEV)
    m.eval()
    with torch.no_grad():
        ev = m(bd, eval_random=False, perm=perm)
    for n, t in zip(("c", "o", "co"), ev):
        assert np.abs(t.cpu().numpy() - fx[f"eval_logits_{n}"]).max() < LOGIT_TOL, n
    # one training step (GAT attention dropout off, as in the fixture)
    m.train()
    if name == "CausalGAT":
        for c in m.convs:
            c.dropout = 0.0
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt.zero_grad()
    logits = m(bd, eval_random=True, perm=perm)
    for n, t in zip(("c", "o", "co"), logits):
        assert np.abs(t.detach().cpu().numpy() - fx[f"train_logits_{n}"]).max() < LOGIT_TOL, n
    loss, lc, lo, lco = O.causal_loss(*logits, bd.y, 4)
    assert np.allclose([loss.item(), lc.item(), lo.item(), lco.item()], fx["loss"], atol=1e-4)
    loss.backward()
    assert m.conv_feat.bias.grad is None or m.conv_feat.bias.grad.abs().max().item() == 0
    for k, p in m.named_parameters():
        g = fx[f"grad.{k}"]
        if g.size == 0:
            continue
        assert np.allclose(p.grad.cpu().numpy(), g, atol=2e-4, rtol=2e-3), k
    opt.step()
    post = m.state_dict()
    for k in post:
        if f"post.{k}" in fx.files:
            a, b = post[k].cpu().numpy(), fx[f"post.{k}"]
            if f"grad.{k}" in fx.files and fx[f"grad.{k}"].size:      # Adam's first step is +-lr where |g| ~ 0
                keep = np.abs(fx[f"grad.{k}"]) > 1e-6
                a, b = a[keep], b[keep]
            assert np.allclose(a, b, atol=2e-4, rtol=1e-3), k


@pytest.mark.parametrize("name", ["CausalGCN", "CausalGAT"])
@pytest.mark.parametrize("cat_or_add", ["add", "cat"])
def test_config2_shape_logits_within_1e4(name, cat_or_add):
    """hidden=128, 3 layers, 32 SPMotif graphs (reference-generated fixtures), train + eval mode."""
    ids = list(range(24)) + [0, 3, 5, 7, 9, 11, 13, 15]
    b = ref_batch(ids)
    torch.manual_seed(7)
    sd = O.init_state(name, 10, 4, hidden=128, layers=3, heads=4, cat_or_add=cat_or_add)
    args = _args(cat_or_add=cat_or_add)
    m = _build(name, 10, 4, args, {k: v.clone() for k, v in sd.items()})
    perm = torch.randperm(len(ids))
    for training in (True, False):
        sdc = {k: v.clone() for k, v in sd.items()}
        ref = O.causal_forward(name, sdc, b.feat, b.edge_index, b.batch, perm=perm, training=training,
                               cat_or_add=cat_or_add, gat_dropout=0.0)
        m.load_state_dict(sd)
        m.train(training)
        if name == "CausalGAT":
            for c in m.convs:
                c.dropout = 0.0
        with torch.no_grad():
            out = m(ref_batch(ids).to(DEV), eval_random=True, perm=perm)
        for r, t in zip(ref, out):
            assert (r - t.cpu()).abs().max().item() < LOGIT_TOL
        if training:    # BN running statistics updated like torch's
            post = m.state_dict()
            for k in ("bn_feat.running_mean", "bnc.running_var", "fc2_bn_co.running_mean"):
                assert torch.allclose(post[k].cpu(), sdc[k], atol=1e-5, rtol=1e-4), k


def test_gat_training_dropout_matches_oracle_with_same_mask():
    from cal_amd import ops
    from cal_amd.plan import plan_of
    ids = [0, 4, 7, 10]
    b = ref_batch(ids)
    torch.manual_seed(3)
    sd = O.init_state("CausalGAT", 10, 4, hidden=32, layers=2, heads=4)
    m = _build("CausalGAT", 10, 4, _args(layers=2, hidden=32), {k: v.clone() for k, v in sd.items()})
    m.train()
    bd = ref_batch(ids).to(DEV)
    plan = plan_of(bd)
    masks = []
    row, col = b.edge_index
    keep_e = (row != col).nonzero().view(-1)
    for i, c in enumerate(m.convs):
        c.seed = 100 + i
        full = ops.gat_dropout_mask(c.seed, plan, 4, 0.2).cpu()
        masks.append(torch.cat([full[keep_e], full[plan.E:]], 0))
    out = m(bd, eval_random=False)
    ref = O.causal_forward("CausalGAT", sd, b.feat, b.edge_index, b.batch, training=True, layers=2,
                           heads=4, gat_dropout=0.2, gat_masks=masks)
    for r, t in zip(ref, out):
        assert (r - t.detach().cpu()).abs().max().item() < LOGIT_TOL


def test_ablation_flags_and_random_perm_path():
    b = ref_batch([0, 4, 7, 10, 13])
    torch.manual_seed(1)
    sd = O.init_state("CausalGCN", 10, 4, hidden=32, layers=2)
    args = _args(layers=2, hidden=32, without_node_attention=True, without_edge_attention=True)
    m = _build("CausalGCN", 10, 4, args, {k: v.clone() for k, v in sd.items()})
    m.eval()
    random.seed(5)
    with torch.no_grad():
        out = m(ref_batch([0, 4, 7, 10, 13]).to(DEV), eval_random=True)
    random.seed(5)
    perm = torch.tensor(O.intervention_perm(5, True, True, "CausalGCN"))
    ref = O.causal_forward("CausalGCN", sd, b.feat, b.edge_index, b.batch, perm=perm, training=False,
                           layers=2, without_node_attention=True, without_edge_attention=True)
    for r, t in zip(ref, out):
        assert (r - t.cpu()).abs().max().item() < LOGIT_TOL


def test_causal_gin_matches_oracle():
    ids = list(range(12))
    b = ref_batch(ids)
    torch.manual_seed(11)
    sd = O.init_state("CausalGIN", 10, 4, hidden=64, layers=2)
    args = _args(layers=2, hidden=64)
    from cal_amd import model as M
    m = M.CausalGIN(10, 4, args)
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.endswith(".eps") for k in missing.missing_keys)
    m = m.to(DEV)
    perm = torch.randperm(len(ids))
    for training in (True, False):
        sdc = {k: v.clone() for k, v in sd.items()}
        ref = O.causal_forward("CausalGIN", sdc, b.feat, b.edge_index, b.batch, perm=perm, training=training, layers=2)
        m.load_state_dict(sd, strict=False)
        m.train(training)
        out = m(ref_batch(ids).to(DEV), eval_random=True, perm=perm)
        for r, t in zip(ref, out):
            assert (r - t.detach().cpu()).abs().max().item() < LOGIT_TOL
    # gradients through the GIN aggregation + MFMA linears
    m.load_state_dict(sd, strict=False)
    m.train()
    tr = O.CpuTrainer("CausalGIN", {k: v.clone() for k, v in sd.items()}, 4, layers=2)
    loss_ref, *_ = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
    bd = ref_batch(ids).to(DEV)
    logits = m(bd, eval_random=True, perm=perm)
    loss, *_ = O.causal_loss(*logits, bd.y, 4)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-4
    for k, p in m.named_parameters():
        g = tr.sd[k].grad
        if g is not None:
            assert torch.allclose(p.grad.cpu(), g, atol=1e-4, rtol=2e-3), k


def test_linear_op_matches_torch():
    from cal_amd import ops
    g = torch.Generator().manual_seed(0)
    # (16391, 64, 132) and (33000, 96, 64) take the 128x128 throughput kernel (gemm_big.hip: >= 16k rows, K % 32 == 0), ragged
    # edges included; the shapes with K, N in {128, 256} and >= 16k rows the weight-resident kernels (gemm_wres.hip: k_wres NN / NT
    # with whole and partial (M % 32 != 0) last row blocks, k_tn for the 256 x 256 weight gradient)
    for (M_, K, N) in [(130, 10, 128), (7000, 128, 128), (64, 128, 4), (257, 96, 33),
                       (20000, 256, 256), (16391, 64, 132), (33000, 96, 64),
                       (20011, 256, 256), (17000, 128, 128), (16500, 128, 256), (16401, 256, 128)]:
        x = torch.randn(M_, K, generator=g).to(DEV).requires_grad_(True)
        w = (torch.randn(N, K, generator=g) * 0.1).to(DEV).requires_grad_(True)
        b = torch.randn(N, generator=g).to(DEV).requires_grad_(True)
        y = ops.linear(x, w, b, relu=True)
        ref = torch.relu(torch.nn.functional.linear(x.detach().cpu(), w.detach().cpu(), b.detach().cpu()))
        assert torch.allclose(y.detach().cpu(), ref, atol=1e-4, rtol=1e-4)
        gy = torch.randn(M_, N, generator=g)
        y.backward(gy.to(DEV))
        xr, wr, br = (t.detach().cpu().requires_grad_(True) for t in (x, w, b))
        torch.relu(torch.nn.functional.linear(xr, wr, br)).backward(gy)
        assert torch.allclose(x.grad.cpu(), xr.grad, atol=2e-4, rtol=1e-3)
        assert torch.allclose(w.grad.cpu(), wr.grad, atol=2e-3, rtol=2e-3)
        assert torch.allclose(b.grad.cpu(), br.grad, atol=2e-3, rtol=2e-3)
        w2 = w.detach().t().contiguous().requires_grad_(True)
        y2 = ops.matmul(x.detach(), w2)
        assert torch.allclose(y2.detach().cpu(), x.detach().cpu() @ w2.detach().cpu(), atol=1e-4, rtol=1e-4)


def test_weight_gradient_gemm_over_node_counts_around_the_split_window():
    """dW[256,256] = X^T dZ (cal_gemm TN -> k_tn, gemm_wres.hip) over node counts whose split-K chunk lands where the 64 x 64
    kernel re-cuts a chunk to its preload depth: left alone that made more slices than slabs (375 for 256 at 47 950 nodes) and
    wrote past the workspace -- found by tests/tools/fuzz_gemm_wres.py in round 5.  Against torch in fp64."""
    from cal_amd import _lib
    from cal_amd.plan import _p, _stream
    g = torch.Generator().manual_seed(3)
    for nodes in (33000, 41000, 47950, 49152):
        x = torch.randn(nodes, 256, generator=g).to(DEV)
        d = torch.randn(nodes, 256, generator=g).to(DEV)
        n_ws = _lib.query("cal_gemm_ws", 256, 256, nodes)
        ws = torch.empty(max(n_ws, 4) + 1024, device=DEV)
        ws[n_ws:].fill_(12345.0)                                   # canary behind the workspace the entry point asked for
        dw = torch.full((256, 256), float("nan"), device=DEV)
        _lib.call("cal_gemm", 1, 0, _p(x), _p(d), _p(dw), None, 0, _p(ws), 256, 256, nodes, _stream())
        ref = x.double().t() @ d.double()
        assert ((dw.double() - ref).abs().max() / ref.abs().max()).item() < 2e-5, nodes
        assert bool((ws[n_ws:] == 12345.0).all().item()), nodes


def test_train_causal_real_k_fold_loop_runs_on_the_engine():
    """train_causal.py:63-160 on a TU-style dataset (synthetic MUTAG-like stand-in): 2 folds x 3 epochs, the reference's
    loop shape (model(data) -> loss -> backward -> torch Adam) on the native engine behind the nn.Module."""
    from cal_amd import model as M, synth, tu
    from cal_amd.train_causal import train_causal_real
    torch.manual_seed(0)
    random.seed(0)
    graphs = synth.tu_like(96, kind="mutag", seed=2)
    # make the label learnable: class = whether the graph has more than the median number of nodes
    med = sorted(g.num_nodes for g in graphs)[48]
    for g in graphs:
        g.y = torch.tensor([int(g.num_nodes > med)])
    ds = tu.TUDataset(graphs, "MUTAG-like")
    args = _args(layers=2, hidden=32)
    args.folds, args.epoch_select, args.batch_size, args.lr, args.weight_decay, args.epochs = 2, "test_max", 32, 5e-3, 0.0, 3
    args.dataset, args.model, args.eval_random = "MUTAG-like", "CausalGCN", False
    logs = []
    res = train_causal_real(ds, lambda nf, nc: M.CausalGCN(nf, nc, args), args, log=logs.append)
    assert res["test_acc"].shape == (2, 3) and len(logs) == 2 * 3 + 1
    assert 0.0 <= res["test_acc_mean"] <= 1.0 and res["random_guess"] == 0.5
    assert res["train_acc"][:, -1].mean().item() > 0.55          # it learns the size label
