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


@pytest.mark.parametrize("p_drop", [0.0, 0.25])
def test_gat_h256_four_heads_hub_rows_and_isolated_nodes(p_drop):
    """The H = 256 / 4-head kernels keep a row's slots in the lanes of ONE wave: rows of more than 64 slots (hubs: two sweeps in the
    by-destination backward, several 64-slot chunks everywhere), rows with no edge at all (only the added loop, GATConv's
    add_self_loops), an input self loop (dropped) and a directed edge, all against the oracle's GATConv (model.py:340,390)."""
    from cal_amd import ops
    from cal_amd.data import Batch, Data
    g = torch.Generator().manual_seed(21)
    ds = []
    # (40, 14..16) / (40, 31..32): stars WITHOUT extra edges at the hub -- in-degree exactly 14 / 15 / 16 / 31 / 32, i.e. 15, 16 (the
    # (head, slot) lane layout of the by-destination backward and one 16-slot hash group), 17, 32 and 33 slots with the node's own loop
    for n, hub_deg in ((200, 150), (90, 70), (5, 0), (40, 14), (40, 15), (40, 16), (40, 31), (40, 32)):
        src, dst = [], []
        for v in range(1, hub_deg + 1):                      # star around node 0, both directions
            src += [0, v]; dst += [v, 0]
        extra = torch.randint(1 if hub_deg else 0, n - 1, (2, 3 * n // 2), generator=g)   # (node n - 1 stays isolated)
        keep = extra[0] != extra[1]
        src += extra[0][keep].tolist(); dst += extra[1][keep].tolist()
        src += [2] if n > 3 else []; dst += [2] if n > 3 else []                            # an input self loop
        ei = torch.tensor([src, dst], dtype=torch.long)
        ei = torch.unique(ei, dim=1)
        ds.append(Data(x=torch.randn(n, 256, generator=g), edge_index=ei, y=torch.zeros(1, dtype=torch.long)))
    b = Batch.from_data_list(ds)
    K, D, H = 4, 64, 256
    pl = _plan(b)
    deg_dst = (pl.rowptr_dst[1:] - pl.rowptr_dst[:-1]).cpu()
    assert int(deg_dst.max()) > 128 and int((deg_dst > 63).sum()) >= 2 and int((deg_dst == 0).sum()) >= 2
    assert all(int((deg_dst == d).sum()) >= 1 for d in (14, 15, 16, 31, 32))
    z = (b.x * 0.5).clone().requires_grad_(True)
    att = (torch.randn(1, K, 2 * D, generator=g) * 0.3).requires_grad_(True)
    bias = torch.randn(H, generator=g).requires_grad_(True)
    seed, mask = 77, None
    if p_drop > 0:
        full = ops.gat_dropout_mask(seed, pl, K, p_drop).cpu()
        row, col = b.edge_index
        keep_e = (row != col).nonzero().view(-1)
        mask = torch.cat([full[keep_e], full[pl.E:]], 0)
    ref = torch.relu(O.gat_conv(z, b.edge_index, torch.eye(H), att, bias, K, 0.2, p_drop, p_drop > 0, mask))
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    zd, ad, bd = (t.detach().to(DEV).requires_grad_(True) for t in (z, att, bias))
    out = ops.gat_aggregate(zd, ad, bd, pl, K, 0.2, p_drop, seed, True)
    out.backward(gout.to(DEV))
    assert torch.allclose(out.detach().cpu(), ref.detach(), atol=5e-5, rtol=1e-4)
    assert torch.allclose(zd.grad.cpu(), z.grad, atol=1e-4, rtol=1e-3)
    assert torch.allclose(ad.grad.cpu(), att.grad, atol=1e-3, rtol=1e-3)
    assert torch.allclose(bd.grad.cpu(), bias.grad, atol=2e-4, rtol=1e-3)


def _big_plan(N, deg, seed=0):
    from cal_amd.plan import GraphPlan
    g = torch.Generator().manual_seed(seed)
    # block-diagonal random graph: 32 graphs, neighbours inside the own graph (like a mini-batch)
    per = N // 32
    dst = torch.arange(N).repeat_interleave(deg)
    src = (dst // per) * per + torch.randint(0, per, (N * deg,), generator=g)
    batch = (torch.arange(N) // per).clamp(max=31)
    return GraphPlan(torch.stack([src, dst]).to(DEV), N, batch.to(DEV), 32)


def test_full_size_properties_config5_shape():
    """BASELINE.json config-5 per-GPU shape (160k nodes, ~800k edges, H = 256): size-independent
    checks where the CPU oracle would be too slow -- linearity, adjointness of the forward and
    transposed aggregation (<A x, y> == <x, A^T y>), row sums, pooling conservation, GAT attention
    rows summing to one (out = z when all z rows are equal)."""
    from cal_amd import _lib, ops
    from cal_amd.plan import _p, _stream
    N, H, deg = 160000, 256, 5
    p = _big_plan(N, deg)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, generator=g).to(DEV)
    y = torch.randn(N, H, generator=g).to(DEV)
    w = (torch.rand(p.E, generator=g) + 0.1).to(DEV)
    ax = ops.gcn_aggregate(x, p, w)
    ay = ops.gcn_aggregate(y, p, w)
    # linearity
    axy = ops.gcn_aggregate(2.0 * x - 0.5 * y, p, w)
    assert torch.allclose(axy, 2.0 * ax - 0.5 * ay, atol=2e-4, rtol=1e-4)
    # adjointness through autograd: d/dx <A x, y> = A^T y
    xg = x.clone().requires_grad_(True)
    (ops.gcn_aggregate(xg, p, w) * y).sum().backward()
    lhs = (ax.double() * y.double()).sum()
    rhs = (x.double() * xg.grad.double()).sum()
    assert abs(lhs - rhs).item() < 1e-6 * max(1.0, abs(lhs).item())
    # unweighted: A 1 has the analytic value sum_e dis_r dis_c + dis_i^2
    ones = torch.ones(N, 4, device=DEV)
    a1 = ops.gcn_aggregate(ones, p, None)[:, 0]
    dis, norm = p.unit_norm()
    ref = torch.zeros(N, device=DEV).index_add_(0, p.col32.long(), norm[:p.E]) + dis[:N] ** 2
    assert torch.allclose(a1, ref, atol=1e-5)
    # pooling conserves the column sums
    pooled = ops.add_pool(x, p)
    assert torch.allclose(pooled.sum(0), x.sum(0), atol=5e-2, rtol=1e-4)
    # GAT: attention coefficients of a row sum to one -> constant z is reproduced (+ bias 0)
    K, D = 4, 64
    zc = torch.randn(1, H, generator=g).to(DEV).expand(N, H).contiguous()
    att = (torch.randn(1, K, 2 * D, generator=g) * 0.2).to(DEV)
    out = ops.gat_aggregate(zc, att, None, p, K)
    assert torch.allclose(out, zc, atol=1e-4, rtol=1e-4)
    # and the GAT backward is the adjoint of the (fixed-alpha) forward in z for uniform attention
    att0 = torch.zeros(1, K, 2 * D, device=DEV)
    zg = x.clone().requires_grad_(True)
    (ops.gat_aggregate(zg, att0, None, p, K) * y).sum().backward()
    out0 = ops.gat_aggregate(x, att0, None, p, K)
    lhs = (out0.double() * y.double()).sum()
    rhs = (x.double() * zg.grad.double()).sum()
    assert abs(lhs - rhs).item() < 1e-5 * max(1.0, abs(lhs).item())
