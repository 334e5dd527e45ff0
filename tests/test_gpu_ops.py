"""GPU parity of the operator-level C-ABI kernels against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import cal_oracle as O
from tests.helpers import random_graph_batch, ref_batch

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = dict(atol=2e-5, rtol=1e-4)


def _plan(b):
    from cal_amd.plan import GraphPlan
    x = b.x if b.x is not None else b.feat
    return GraphPlan(b.edge_index.to(DEV), x.size(0), b.batch.to(DEV), b.num_graphs, validate=True)


def _cases():
    yield "rand", random_graph_batch(6, seed=0, feat=32)
    yield "rand_loops_directed", random_graph_batch(6, seed=1, feat=20, self_loops=True, directed=True)
    yield "odd_width", random_graph_batch(4, seed=2, feat=7)
    b = ref_batch([0, 4, 7, 10, 13, 16, 19, 22, 25, 30])
    g = torch.Generator().manual_seed(5)
    b.x = torch.randn(b.feat.size(0), 128, generator=g)
    yield "spmotif128", b
    b2 = ref_batch([24, 29])
    b2.x = torch.randn(b2.feat.size(0), 256, generator=g)
    yield "spmotif256", b2


CASES = list(_cases())


def test_plan_build_matches_numpy():
    for name, b in CASES:
        p = _plan(b)
        ei = b.edge_index.numpy()
        N = p.N
        keep = ei[0] != ei[1]
        eids = np.nonzero(keep)[0]
        for which, key, other in (("dst", 1, 0), ("src", 0, 1)):
            order = eids[np.argsort(ei[key][eids], kind="stable")]
            rowptr = np.zeros(N + 1, np.int64)
            np.add.at(rowptr, ei[key][eids] + 1, 1)
            rowptr = np.cumsum(rowptr)
            assert np.array_equal(getattr(p, f"rowptr_{which}").cpu().numpy(), rowptr), name
            nnz = int(rowptr[-1])
            assert np.array_equal(getattr(p, f"eid_{which}").cpu().numpy()[:nnz], order), name
            assert np.array_equal(getattr(p, f"nbr_{which}").cpu().numpy()[:nnz], ei[other][order]), name
        gptr = p.gptr.cpu().numpy()
        assert np.array_equal(gptr, np.searchsorted(b.batch.numpy(), np.arange(b.num_graphs + 1)))


def test_plan_flags_bad_indices():
    from cal_amd.plan import GraphPlan
    ei = torch.tensor([[0, 5], [1, 0]], device=DEV)
    with pytest.raises(IndexError):
        GraphPlan(ei, 3, validate=True)
    with pytest.raises(ValueError):
        GraphPlan(torch.tensor([[0], [1]], device=DEV), 3, torch.tensor([1, 0, 1], device=DEV), 2, validate=True)


@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("improved", [False, True])
def test_gcn_aggregate_fwd_bwd(weighted, relu, improved):
    from cal_amd import ops
    for name, b in CASES:
        p = _plan(b)
        g = torch.Generator().manual_seed(11)
        h = b.x.clone().requires_grad_(True)
        H = h.size(1)
        bias = torch.randn(H, generator=g).requires_grad_(True)
        w = (torch.rand(b.edge_index.size(1), generator=g) + 0.05).requires_grad_(True) if weighted else None
        ref = O.gcn_conv(h, b.edge_index, torch.eye(H), bias, w, improved=improved)
        ref = torch.relu(ref) if relu else ref
        gout = torch.randn(ref.shape, generator=g)
        ref.backward(gout)

        hd = b.x.to(DEV).requires_grad_(True)
        bd = bias.detach().to(DEV).requires_grad_(True)
        wd = w.detach().to(DEV).requires_grad_(True) if weighted else None
        out = ops.gcn_aggregate(hd, p, wd, bd, improved, relu)
        out.backward(gout.to(DEV))
        assert torch.allclose(out.detach().cpu(), ref.detach(), **TOL), name
        assert torch.allclose(hd.grad.cpu(), h.grad, **TOL), name
        assert torch.allclose(bd.grad.cpu(), bias.grad, atol=1e-4, rtol=1e-4), name
        if weighted:
            assert torch.allclose(wd.grad.cpu(), w.grad, atol=5e-5, rtol=1e-3), name


def test_gcn_conv_module_matches_oracle_and_caches():
    from cal_amd.gcn_conv import GCNConv
    _, b = CASES[0]
    conv = GCNConv(32, 16).to(DEV)
    with torch.no_grad():
        conv.bias.uniform_(-1, 1)
    out = conv(b.x.to(DEV), b.edge_index.to(DEV))
    ref = O.gcn_conv(b.x, b.edge_index, conv.weight.detach().cpu(), conv.bias.detach().cpu())
    assert torch.allclose(out.detach().cpu(), ref, **TOL)
    gfn = GCNConv(32, 16, gfn=True).to(DEV)
    assert torch.allclose(gfn(b.x.to(DEV), b.edge_index.to(DEV)).cpu(), b.x @ gfn.weight.detach().cpu(), **TOL)


def test_empty_and_isolated_inputs():
    from cal_amd import ops
    from cal_amd.plan import GraphPlan
    # graph with no edges at all: out = h + bias (only the added self loop, deg = 1)
    x = torch.randn(5, 8)
    p = GraphPlan(torch.zeros(2, 0, dtype=torch.long, device=DEV), 5, torch.zeros(5, dtype=torch.long, device=DEV), 1)
    out = ops.gcn_aggregate(x.to(DEV), p, None, None)
    assert torch.allclose(out.cpu(), x, atol=1e-6)
    assert torch.allclose(ops.add_pool(x.to(DEV), p).cpu(), x.sum(0, keepdim=True), atol=1e-5)
    # a batch whose middle graph is empty (no nodes)
    batch = torch.tensor([0, 0, 2, 2, 2], device=DEV)
    p = GraphPlan(torch.tensor([[0, 1], [1, 0]], device=DEV), 5, batch, 3)
    pooled = ops.add_pool(x.to(DEV), p).cpu()
    assert torch.allclose(pooled[1], torch.zeros(8)) and torch.allclose(pooled[2], x[2:].sum(0), atol=1e-5)


def test_edge_attention_fwd_bwd():
    from cal_amd import ops
    for name, b in CASES:
        p = _plan(b)
        g = torch.Generator().manual_seed(3)
        H = b.x.size(1)
        x = b.x.clone().requires_grad_(True)
        W = (torch.randn(2, 2 * H, generator=g) * 0.3).requires_grad_(True)
        bb = torch.randn(2, generator=g).requires_grad_(True)
        row, col = b.edge_index
        ref = torch.softmax(torch.nn.functional.linear(torch.cat([x[row], x[col]], -1), W, bb), -1).t()
        gout = torch.randn(ref.shape, generator=g)
        # weights of explicit self-loop edges never reach a conv: zero their gradient
        gout[:, row == col] = 0
        ref.backward(gout)
        xd, Wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, W, bb))
        att = ops.edge_attention(xd, Wd, bd, p)
        att.backward(gout.to(DEV))
        assert torch.allclose(att.detach().cpu(), ref.detach(), **TOL), name
        assert torch.allclose(xd.grad.cpu(), x.grad, atol=5e-5, rtol=1e-3), name
        assert torch.allclose(Wd.grad.cpu(), W.grad, atol=2e-4, rtol=1e-3), name
        assert torch.allclose(bd.grad.cpu(), bb.grad, atol=2e-4, rtol=1e-3), name


def test_node_attention_split_fwd_bwd():
    from cal_amd import ops
    for name, b in CASES:
        g = torch.Generator().manual_seed(4)
        H = b.x.size(1)
        x = b.x.clone().requires_grad_(True)
        W = (torch.randn(2, H, generator=g) * 0.3).requires_grad_(True)
        bb = torch.randn(2, generator=g).requires_grad_(True)
        a = torch.softmax(torch.nn.functional.linear(x, W, bb), -1)
        rc, ro = a[:, :1] * x, a[:, 1:] * x
        g1, g2 = torch.randn(rc.shape, generator=g), torch.randn(rc.shape, generator=g)
        (rc * g1 + ro * g2).sum().backward()
        xd, Wd, bd = (t.detach().to(DEV).requires_grad_(True) for t in (x, W, bb))
        xc, xo, att = ops.node_attention_split(xd, Wd, bd)
        (xc * g1.to(DEV) + xo * g2.to(DEV)).sum().backward()
        assert torch.allclose(xc.detach().cpu(), rc.detach(), **TOL), name
        assert torch.allclose(xo.detach().cpu(), ro.detach(), **TOL), name
        assert torch.allclose(att.cpu(), a.detach(), **TOL)
        assert torch.allclose(xd.grad.cpu(), x.grad, atol=5e-5, rtol=1e-3), name
        assert torch.allclose(Wd.grad.cpu(), W.grad, atol=3e-4, rtol=1e-3), name
        assert torch.allclose(bd.grad.cpu(), bb.grad, atol=3e-4, rtol=1e-3), name


def test_add_pool_fwd_bwd():
    from cal_amd import ops
    for name, b in CASES:
        p = _plan(b)
        x = b.x.clone().requires_grad_(True)
        ref = O.global_add_pool(x, b.batch, b.num_graphs)
        gout = torch.randn(ref.shape)
        ref.backward(gout)
        xd = b.x.to(DEV).requires_grad_(True)
        out = ops.add_pool(xd, p)
        out.backward(gout.to(DEV))
        assert torch.allclose(out.detach().cpu(), ref.detach(), atol=1e-4, rtol=1e-5), name
        assert torch.equal(xd.grad.cpu(), x.grad), name


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("p_drop", [0.0, 0.3])
def test_gat_aggregate_fwd_bwd(relu, p_drop):
    from cal_amd import ops
    for name, b in CASES:
        H = b.x.size(1)
        if H % 4:
            continue
        K = 4
        D = H // K
        if D & (D - 1):
            continue
        p = _plan(b)
        g = torch.Generator().manual_seed(6)
        z = (b.x * 0.5).clone().requires_grad_(True)
        att = (torch.randn(1, K, 2 * D, generator=g) * 0.3).requires_grad_(True)
        bias = torch.randn(H, generator=g).requires_grad_(True)
        seed = 1234
        mask = None
        if p_drop > 0:
            full = ops.gat_dropout_mask(seed, p, K, p_drop).cpu()
            row, col = b.edge_index
            keep_e = (row != col).nonzero().view(-1)
            mask = torch.cat([full[keep_e], full[p.E:]], 0)
            frac = full.mean().item()
            assert abs(frac - (1 - p_drop)) < 0.08
        ref = O.gat_conv(z, b.edge_index, torch.eye(H), att, bias, K, 0.2, p_drop, p_drop > 0, mask)
        ref = torch.relu(ref) if relu else ref
        gout = torch.randn(ref.shape, generator=g)
        ref.backward(gout)
        zd, ad, bd = (t.detach().to(DEV).requires_grad_(True) for t in (z, att, bias))
        out = ops.gat_aggregate(zd, ad, bd, p, K, 0.2, p_drop, seed, relu)
        out.backward(gout.to(DEV))
        assert torch.allclose(out.detach().cpu(), ref.detach(), atol=5e-5, rtol=1e-4), name
        assert torch.allclose(zd.grad.cpu(), z.grad, atol=1e-4, rtol=1e-3), name
        assert torch.allclose(ad.grad.cpu(), att.grad, atol=5e-4, rtol=1e-3), name
        assert torch.allclose(bd.grad.cpu(), bias.grad, atol=2e-4, rtol=1e-3), name
