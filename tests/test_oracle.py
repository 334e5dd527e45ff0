"""Pin the CPU oracle (no reference tests exist -- SURVEY.md section 4/8c):
sparse vs independent dense restatement, hand-derivable known answers,
invariances, fp64 gradcheck, and the committed regression fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import cal_oracle as O
from oracle import cal_oracle_dense as D
from tests.helpers import GOLDEN, random_graph_batch, ref_batch


# ---------------------------------------------------------------- known answers
def test_gcn_path2_known_answer():
    # 0 - 1, W = I, b = 0, X = I: out = D^-1/2 (A + I) D^-1/2 = [[.5,.5],[.5,.5]]
    ei = torch.tensor([[0, 1], [1, 0]])
    out = O.gcn_conv(torch.eye(2), ei, torch.eye(2), torch.zeros(2))
    assert torch.allclose(out, torch.full((2, 2), 0.5), atol=1e-7)


def test_gcn_star4_known_answer():
    # centre 0 with leaves 1..3: deg = [4,2,2,2]
    ei = torch.tensor([[0, 0, 0, 1, 2, 3], [1, 2, 3, 0, 0, 0]])
    out = O.gcn_conv(torch.eye(4), ei, torch.eye(4), None)
    s = 1.0 / (2.0 * 2 ** 0.5)
    exp = torch.tensor([[.25, s, s, s], [s, .5, 0, 0], [s, 0, .5, 0], [s, 0, 0, .5]])
    assert torch.allclose(out, exp, atol=1e-7)


def test_gcn_triangle_weighted_asymmetric():
    # directed weights: row-degree normalisation on both ends (gcn_conv.py:65-70)
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]])
    w = torch.tensor([3.0, 8.0, 15.0])
    out = O.gcn_conv(torch.eye(3), ei, torch.eye(3), None, edge_weight=w)
    dis = torch.tensor([4.0, 9.0, 16.0]).rsqrt()
    exp = torch.diag(dis * dis)
    exp[1, 0] = dis[0] * 3 * dis[1]     # out[col] += n * x[row]
    exp[2, 1] = dis[1] * 8 * dis[2]
    exp[0, 2] = dis[2] * 15 * dis[0]
    assert torch.allclose(out, exp, atol=1e-7)


def test_gcn_drops_explicit_self_loops_and_their_weights():
    ei = torch.tensor([[0, 0, 1], [0, 1, 0]])
    w = torch.tensor([100.0, 1.0, 1.0])
    a = O.gcn_conv(torch.eye(2), ei, torch.eye(2), None, edge_weight=w)
    b = O.gcn_conv(torch.eye(2), ei[:, 1:], torch.eye(2), None, edge_weight=w[1:])
    assert torch.equal(a, b)


def test_gat_zero_att_is_uniform_mean():
    b = random_graph_batch(3, seed=1)
    n, f = b.x.shape
    h, k = 8, 4
    w = torch.randn(f, h)
    att = torch.zeros(1, k, 2 * (h // k))
    out = O.gat_conv(b.x, b.edge_index, w, att, None, heads=k)
    a = D.dense_weighted_adj(b.edge_index, n, None, torch.float32) + torch.eye(n)
    exp = (a.t() / a.t().sum(1, keepdim=True)) @ (b.x @ w)
    assert torch.allclose(out, exp, atol=1e-5)


def test_add_pool_column_sums():
    x = torch.arange(12.0).view(6, 2)
    batch = torch.tensor([0, 0, 1, 1, 1, 2])
    out = O.global_add_pool(x, batch)
    assert torch.equal(out, torch.stack([x[:2].sum(0), x[2:5].sum(0), x[5:].sum(0)]))


# ---------------------------------------------------------- sparse vs dense
@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("loops", [False, True])
def test_gcn_sparse_vs_dense(seed, weighted, loops):
    b = random_graph_batch(4, seed=seed, self_loops=loops, dtype=torch.float64, directed=weighted)
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(b.x.size(1), 7, generator=g, dtype=torch.float64)
    bias = torch.randn(7, generator=g, dtype=torch.float64)
    ew = torch.rand(b.edge_index.size(1), generator=g, dtype=torch.float64) if weighted else None
    s = O.gcn_conv(b.x, b.edge_index, w, bias, ew)
    d = D.gcn_conv_dense(b.x, b.edge_index, w, bias, ew)
    assert torch.allclose(s, d, atol=1e-10)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_gat_sparse_vs_dense(seed):
    b = random_graph_batch(4, seed=seed, self_loops=(seed == 1), dtype=torch.float64)
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(b.x.size(1), 8, generator=g, dtype=torch.float64)
    att = torch.randn(1, 4, 4, generator=g, dtype=torch.float64)
    bias = torch.randn(8, generator=g, dtype=torch.float64)
    s = O.gat_conv(b.x, b.edge_index, w, att, bias, heads=4)
    d = D.gat_conv_dense(b.x, b.edge_index, w, att, bias, heads=4)
    assert torch.allclose(s, d, atol=1e-10)


def test_pool_and_edge_attention_dense():
    b = random_graph_batch(4, seed=3, dtype=torch.float64)
    assert torch.allclose(O.global_add_pool(b.x, b.batch),
                          D.global_add_pool_dense(b.x, b.batch, 4), atol=1e-12)
    h = b.x.size(1)
    w = torch.randn(2, 2 * h, dtype=torch.float64)
    bb = torch.randn(2, dtype=torch.float64)
    row, col = b.edge_index
    ref = torch.softmax(torch.cat([b.x[row], b.x[col]], -1) @ w.t() + bb, -1)
    assert torch.allclose(ref, D.edge_attention_dense(b.x, b.edge_index, w, bb), atol=1e-12)


# ------------------------------------------------------------------ properties
def _model_inputs(model, dtype=torch.float64, seed=0, hidden=16):
    b = random_graph_batch(5, seed=seed, dtype=dtype)
    sd = O.init_state(model, b.x.size(1), 3, hidden=hidden, layers=2, heads=4, dtype=dtype, seed=seed)
    return b, sd


@pytest.mark.parametrize("model", ["CausalGCN", "CausalGAT", "CausalGIN"])
def test_masks_partition_and_shapes(model):
    b, sd = _model_inputs(model)
    logits, inter = O.causal_forward(model, sd, b.x, b.edge_index, b.batch, layers=2,
                                     return_intermediates=True)
    assert all(t.shape == (5, 3) for t in logits)
    assert torch.allclose(inter["edge_att"].sum(1), torch.ones(b.edge_index.size(1), dtype=torch.float64))
    assert torch.allclose(inter["node_att"].sum(1), torch.ones(b.x.size(0), dtype=torch.float64))
    for t in logits:
        assert torch.allclose(t.exp().sum(1), torch.ones(5, dtype=torch.float64))


@pytest.mark.parametrize("model", ["CausalGCN", "CausalGAT"])
def test_node_relabel_and_edge_order_invariance(model):
    b, sd = _model_inputs(model, seed=4)
    base = O.causal_forward(model, sd, b.x, b.edge_index, b.batch, layers=2)
    g = torch.Generator().manual_seed(7)
    # shuffle edge order
    pe = torch.randperm(b.edge_index.size(1), generator=g)
    out = O.causal_forward(model, sd, b.x, b.edge_index[:, pe], b.batch, layers=2)
    for u, v in zip(base, out):
        assert torch.allclose(u, v, atol=1e-10)
    # relabel nodes inside each graph (keeps `batch` sorted)
    n = b.x.size(0)
    newpos = torch.empty(n, dtype=torch.long)
    for gi in range(b.num_graphs):
        idx = (b.batch == gi).nonzero().view(-1)
        newpos[idx] = idx[torch.randperm(idx.numel(), generator=g)]
    x2 = torch.empty_like(b.x)
    x2[newpos] = b.x
    out = O.causal_forward(model, sd, x2, newpos[b.edge_index], b.batch, layers=2)
    for u, v in zip(base, out):
        assert torch.allclose(u, v, atol=1e-10)


def test_batched_equals_per_graph_in_eval_mode():
    b, sd = _model_inputs("CausalGCN", seed=5)
    whole = O.causal_forward("CausalGCN", sd, b.x, b.edge_index, b.batch, layers=2)
    for gi in range(b.num_graphs):
        nm = b.batch == gi
        lo = int(nm.nonzero()[0])
        em = nm[b.edge_index[0]]
        one = O.causal_forward("CausalGCN", sd, b.x[nm], b.edge_index[:, em] - lo,
                               torch.zeros(int(nm.sum()), dtype=torch.long), layers=2)
        # co head with identity perm is per-graph too
        for u, v in zip(whole, one):
            assert torch.allclose(u[gi], v[0], atol=1e-10)


def test_intervention_perm_gating():
    import random
    random.seed(0)
    assert O.intervention_perm(6, True, False) == list(range(6))
    assert O.intervention_perm(6, False, True, "CausalGCN") == list(range(6))
    random.seed(0)
    p = O.intervention_perm(6, False, True, "CausalGAT")      # model.py:435: no with_random gate
    random.seed(0)
    l = list(range(6)); random.shuffle(l)
    assert p == l


def test_loss_formula():
    torch.manual_seed(0)
    c = torch.log_softmax(torch.randn(5, 4), -1)
    o = torch.log_softmax(torch.randn(5, 4), -1)
    co = torch.log_softmax(torch.randn(5, 4), -1)
    y = torch.tensor([0, 1, 2, 3, 0])
    loss, lc, lo, lco = O.causal_loss(c, o, co, y, 4)
    kl = (0.25 * (np.log(0.25) - c)).sum() / 5
    assert torch.allclose(lc, kl, atol=1e-6)
    assert torch.allclose(lo, -o[torch.arange(5), y].mean(), atol=1e-6)
    assert torch.allclose(loss, 0.5 * lc + lo + 0.5 * lco, atol=1e-6)


# ------------------------------------------------------------------ gradcheck
def test_gradcheck_weighted_gcn():
    b = random_graph_batch(2, n_hi=6, seed=8, dtype=torch.float64, directed=True)
    e = b.edge_index.size(1)
    x = b.x.clone().requires_grad_(True)
    w = torch.randn(b.x.size(1), 3, dtype=torch.float64, requires_grad=True)
    bias = torch.randn(3, dtype=torch.float64, requires_grad=True)
    ew = (torch.rand(e, dtype=torch.float64) + 0.1).requires_grad_(True)
    assert torch.autograd.gradcheck(lambda *a: O.gcn_conv(a[0], b.edge_index, a[1], a[2], a[3]),
                                    (x, w, bias, ew), atol=1e-6)


def test_gradcheck_gat():
    b = random_graph_batch(2, n_hi=6, seed=9, dtype=torch.float64)
    x = b.x.clone().requires_grad_(True)
    w = torch.randn(b.x.size(1), 4, dtype=torch.float64, requires_grad=True)
    att = torch.randn(1, 2, 4, dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(lambda *a: O.gat_conv(a[0], b.edge_index, a[1], a[2], None, heads=2),
                                    (x, w, att), atol=1e-6)


# ------------------------------------------------------- regression fixtures
@pytest.mark.parametrize("model,fname", [("CausalGCN", "causal_gcn_batch8.npz"),
                                         ("CausalGAT", "causal_gat_batch8.npz")])
def test_oracle_matches_committed_fixture(model, fname):
    fx = np.load(os.path.join(GOLDEN, fname))
    b = ref_batch(list(fx["ids"]))
    sd = {k[3:]: torch.from_numpy(fx[k]).clone() for k in fx.files if k.startswith("sd.")}
    perm = torch.from_numpy(fx["perm"])
    ev = O.causal_forward(model, {k: v.clone() for k, v in sd.items()}, b.feat, b.edge_index,
                          b.batch, perm=perm, training=False, layers=2, heads=4)
    for n, t in zip(("c", "o", "co"), ev):
        assert np.allclose(t.numpy(), fx[f"eval_logits_{n}"], atol=1e-5)
    tr = O.CpuTrainer(model, sd, 4, lr=1e-3, layers=2, heads=4, gat_dropout=0.0)
    loss, lc, lo, lco, _ = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
    assert np.allclose([loss.item(), lc.item(), lo.item(), lco.item()], fx["loss"], atol=1e-5)
    assert tr.sd["conv_feat.bias"].grad is None          # gfn bias never gets a grad
    for k in tr.names:
        assert np.allclose(tr.sd[k].detach().numpy(), fx[f"post.{k}"], atol=1e-5), k


@pytest.mark.parametrize("model", ["CausalGCN", "CausalGIN", "CausalGAT"])
def test_ablation_flags_only_act_in_causalgcn(model):
    """model.py:99-107 -- only CausalGCN.forward replaces the attentions by constant 0.5 masks when
    --without_node_attention / --without_edge_attention are given; CausalGIN.forward (model.py:236-264) and
    CausalGAT.forward (model.py:380-409) have no such branch, so the flags change nothing there."""
    b = random_graph_batch(num_graphs=6, feat=5, seed=3)
    sd = O.init_state(model, 5, 3, hidden=16, layers=2, heads=4, seed=1)
    kw = dict(layers=2, heads=4, gat_dropout=0.0, training=False)
    base = O.causal_forward(model, sd, b.x, b.edge_index, b.batch, **kw)
    flagged = O.causal_forward(model, sd, b.x, b.edge_index, b.batch, without_node_attention=True,
                               without_edge_attention=True, **kw)
    same = all(torch.equal(a, c) for a, c in zip(base, flagged))
    assert same == (model != "CausalGCN")
