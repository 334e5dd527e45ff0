"""GPU parity of the native CausalGCN step engine (cal_amd/csrc/engine.hip) against the CPU oracle:
golden fixture, BASELINE.json config-2 scale (128 SPMotif graphs, hidden 128, 3 layers), eval-mode
forward, multi-step training, and size-independent properties at full size."""
import argparse
import os

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


def _engine(sd, args, nfeat=10, ncls=4, lr=1e-3, **kw):
    from cal_amd import model as M
    from cal_amd.engine import StepEngine
    m = M.CausalGCN(nfeat, ncls, args)
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    return m, StepEngine(m, lr=lr, **kw)


def _close(a, b, atol, rtol):
    return np.allclose(a, b, atol=atol, rtol=rtol)


def test_golden_fixture_train_step():
    fx = np.load(os.path.join(GOLDEN, "causal_gcn_batch8.npz"))
    sd = {k[3:]: torch.from_numpy(fx[k]).clone() for k in fx.files if k.startswith("sd.")}
    m, eng = _engine(sd, _args(layers=2, hidden=32))
    bd = ref_batch(list(fx["ids"])).to(DEV)
    perm = torch.from_numpy(fx["perm"]).to(DEV)
    # eval-mode forward first (running statistics, no updates)
    ev = eng.forward(bd, perm, training=False)
    for n, t in zip(("c", "o", "co"), ev):
        assert np.abs(t.cpu().numpy() - fx[f"eval_logits_{n}"]).max() < LOGIT_TOL, n
    stats = eng.train_step(bd, perm, adam=True).cpu().numpy()
    lp = eng.buffer("logp", 3 * 8 * 4).view(3, 8, 4).cpu().numpy()
    for i, n in enumerate(("c", "o", "co")):
        assert np.abs(lp[i] - fx[f"train_logits_{n}"]).max() < LOGIT_TOL, n
    assert np.allclose(stats[:4], fx["loss"], atol=1e-4)
    assert m.conv_feat.bias.grad.abs().max().item() == 0          # never receives a gradient
    for k, p in m.named_parameters():
        g = fx[f"grad.{k}"]
        if g.size:
            assert _close(p.grad.cpu().numpy(), g, 2e-5, 1e-3), k
    post = m.state_dict()
    for k in post:
        if f"post.{k}" in fx.files and "running" in k:
            assert _close(post[k].cpu().numpy(), fx[f"post.{k}"], 1e-4, 1e-4), k
    # Adam: compare where the gradient is not numerically zero (sign of ~1e-8 grads is noise)
    for k, p in m.named_parameters():
        g = fx[f"grad.{k}"]
        if g.size:
            mask = np.abs(g) > 1e-6
            assert _close(post[k].cpu().numpy()[mask], fx[f"post.{k}"][mask], 2e-5, 1e-4), k


def _config2_batch(n_graphs=128, seed=11):
    from cal_amd import spmotif
    from cal_amd.data import Batch
    gs = spmotif.train_mix(n_graphs, seed=seed)
    return Batch.from_data_list(gs), Batch.from_data_list(gs).to(DEV)


def test_config2_scale_step_matches_oracle():
    """Full headline shape: N ~ 7.3k nodes, E ~ 25k edges, B = 128 (partial-row reductions, split-K
    weight gradients, edge tiles all exercised)."""
    b, bd = _config2_batch()
    torch.manual_seed(5)
    sd = O.init_state("CausalGCN", 10, 4, hidden=128, layers=3)
    m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args())
    perm = torch.randperm(128)
    tr = O.CpuTrainer("CausalGCN", {k: v.clone() for k, v in sd.items()}, 4, lr=1e-3, layers=3)
    loss, lc, lo, lco, logits = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
    stats = eng.train_step(bd, perm.to(DEV), adam=True).cpu().numpy()
    lp = eng.buffer("logp", 3 * 128 * 4).view(3, 128, 4).cpu()
    for r, t in zip(logits, lp):
        assert (r.detach() - t).abs().max().item() < LOGIT_TOL
    assert np.allclose(stats[:4], [loss.item(), lc.item(), lo.item(), lco.item()], atol=1e-4)
    correct = (logits[1].argmax(1) == b.y).sum().item()
    assert int(stats[4]) == correct
    for k, p in m.named_parameters():
        gref = tr.sd[k].grad
        if gref is not None:
            assert torch.allclose(p.grad.cpu(), gref, atol=5e-5, rtol=2e-3), k
    post = m.state_dict()
    for k in post:
        if "running" in k:
            assert torch.allclose(post[k].cpu(), tr.sd[k], atol=1e-3, rtol=1e-4), k


def test_eval_forward_and_module_path_agree():
    b, bd = _config2_batch(64, seed=3)
    torch.manual_seed(9)
    sd = O.init_state("CausalGCN", 10, 4, hidden=64, layers=2)
    for k in sd:                       # non-trivial running statistics
        if k.endswith("running_mean"):
            sd[k] = torch.randn_like(sd[k]) * 0.1
        if k.endswith("running_var"):
            sd[k] = torch.rand_like(sd[k]) + 0.5
    m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(layers=2, hidden=64))
    perm = torch.randperm(64)
    ref = O.causal_forward("CausalGCN", {k: v.clone() for k, v in sd.items()}, b.feat, b.edge_index, b.batch,
                           perm=perm, training=False, layers=2)
    out = eng.forward(bd, perm.to(DEV), training=False)
    for r, t in zip(ref, out):
        assert (r - t.cpu()).abs().max().item() < LOGIT_TOL
    m.eval()
    with torch.no_grad():
        mod = m(bd, eval_random=False, perm=perm)          # operator-level path on the same weights
    for r, t in zip(mod, out):
        assert (r - t).abs().max().item() < LOGIT_TOL
    # eval forward must not touch running statistics
    for k, v in m.state_dict().items():
        if "running" in k or "num_batches" in k:
            assert torch.equal(v.cpu(), sd[k])


def test_three_steps_track_the_oracle():
    b, bd = _config2_batch(32, seed=21)
    torch.manual_seed(2)
    sd = O.init_state("CausalGCN", 10, 4, hidden=32, layers=3)
    m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(hidden=32), lr=1e-2)
    tr = O.CpuTrainer("CausalGCN", {k: v.clone() for k, v in sd.items()}, 4, lr=1e-2, layers=3)
    g = torch.Generator().manual_seed(0)
    for step in range(3):
        perm = torch.randperm(32, generator=g)
        loss, *_ = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
        stats = eng.train_step(bd, perm.to(DEV), adam=True).cpu()
        assert abs(stats[0].item() - loss.item()) < 2e-3 * (step + 1), step
    assert int(eng.step_count.item()) == 3
    assert int(m.bn_feat.num_batches_tracked.item()) == 3


def test_trainer_graph_replay_matches_eager_engine():
    from cal_amd import model as M
    from cal_amd.trainer import CausalTrainer
    _, b1 = _config2_batch(32, seed=1)
    _, b2 = _config2_batch(32, seed=2)
    torch.manual_seed(4)
    sd = O.init_state("CausalGCN", 10, 4, hidden=32, layers=2)
    outs = []
    for use_graph in (False, True):
        m = M.CausalGCN(10, 4, _args(layers=2, hidden=32))
        m.load_state_dict(sd)
        tr = CausalTrainer(m.to(DEV), _args(layers=2, hidden=32), lr=1e-2, use_graph=use_graph)
        tr.reserve_for([b1, b2])
        losses = []
        for i in range(6):
            perm = torch.arange(32).roll(i)
            losses.append(tr.step(b1 if i % 2 == 0 else b2, perm)[0].item())
        outs.append((losses, tr.flat_p.clone()))
    assert np.allclose(outs[0][0], outs[1][0], atol=1e-6)
    assert torch.allclose(outs[0][1], outs[1][1], atol=1e-6)
    assert outs[0][0][-1] < outs[0][0][0]            # it trains


def test_properties_at_full_size():
    """Size-independent invariants on the headline batch: masks partition, log-probs normalise,
    identity permutation makes the co head per-graph, run-to-run determinism."""
    b, bd = _config2_batch()
    torch.manual_seed(1)
    sd = O.init_state("CausalGCN", 10, 4, hidden=128, layers=3)
    m, eng = _engine(sd, _args())
    N, E = bd.feat.size(0), bd.edge_index.size(1)
    perm = torch.randperm(128, device=DEV)
    out1 = [t.clone() for t in eng.forward(bd, perm, training=True)]
    att = eng.buffer("att", 2 * E).view(2, E)
    nsl = bd.edge_index[0] != bd.edge_index[1]
    assert torch.allclose(att[:, nsl].sum(0), torch.ones(int(nsl.sum()), device=DEV), atol=1e-6)
    an = eng.buffer("anode", 2 * N).view(N, 2)
    assert torch.allclose(an.sum(1), torch.ones(N, device=DEV), atol=1e-6)
    for t in out1:
        assert torch.allclose(t.exp().sum(1), torch.ones(128, device=DEV), atol=1e-5)
    # run to run: the default path adds the BatchNorm sums with fp64 atomics into four accumulator rows (order-dependent in the
    # last bits of a double): equal to 1e-6 ...
    state = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    out2 = eng.forward(bd, perm, training=True)
    for u, v in zip(out1, out2):
        assert torch.allclose(u, v, atol=1e-6)
    m.load_state_dict(state)


def test_deterministic_engine_is_bit_reproducible():
    """``StepEngine(deterministic=True)`` / ``CausalTrainer(deterministic=True)`` (atomic-free, deterministic
    reductions): every cross-row sum is a fixed-order sum of partial rows -- two runs of three training steps from the same
    state give the same BITS (log-probs, loss statistics, every parameter after Adam), and the same as the default path to 1e-5."""
    b, bd = _config2_batch()
    torch.manual_seed(1)
    sd = O.init_state("CausalGCN", 10, 4, hidden=128, layers=3)
    perm = torch.randperm(128, device=DEV)
    runs = []
    for det in (True, True, False):
        m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(), deterministic=det)
        assert eng.deterministic == det
        stats = [eng.train_step(bd, perm, adam=True).clone() for _ in range(3)]
        runs.append((eng.buffer("logp", 3 * 128 * 4).clone(), torch.stack(stats), eng.flat_p.detach().clone()))
        eng.check_status()
    for u, v in zip(runs[0], runs[1]):
        assert torch.equal(u, v)                                   # bit for bit
    for u, v in zip(runs[0], runs[2]):
        assert torch.allclose(u, v, atol=1e-5, rtol=1e-5)          # the striped accumulators: same values to rounding


def _ragged_batch(seed, nfeat, sizes):
    """Graphs of very different sizes: single node, no edges, a star, random ones, explicit self loops."""
    from cal_amd.data import Batch, Data
    g = torch.Generator().manual_seed(seed)
    ds = []
    for i, n in enumerate(sizes):
        if n == 1:
            ei = torch.zeros(2, 0, dtype=torch.long)
        elif i % 4 == 1:
            ei = torch.zeros(2, 0, dtype=torch.long)                      # edgeless graph
        elif i % 4 == 2:
            leaves = torch.arange(1, n)
            ei = torch.cat([torch.stack([torch.zeros_like(leaves), leaves]), torch.stack([leaves, torch.zeros_like(leaves)])], 1)
        else:
            a = torch.rand(n, n, generator=g) < 0.2
            a = a | a.t()
            a.fill_diagonal_(False)
            a[0, 0] = True                                               # explicit self loop: dropped by the convs
            ei = a.nonzero().t().contiguous()
        ds.append(Data(x=torch.randn(n, nfeat, generator=g), edge_index=ei, y=torch.randint(0, 3, (1,), generator=g)))
    return Batch.from_data_list(ds)


@pytest.mark.parametrize("hidden,layers,nfeat,ncls,sizes", [
    (64, 1, 7, 3, [1, 5, 9, 33, 2, 17, 64, 3]),
    (256, 2, 10, 2, [40, 1, 70, 12, 90]),
    (32, 4, 3, 5, [6, 6, 6]),
    (128, 0, 10, 4, [20, 30, 25, 8]),
    # fused-readout corner cases: B not a multiple of 16 (several 16-graph row blocks, clamped MFMA tiles),
    # hidden/4 not a divisor of 256 (LDS statistics pass), many classes (large fc2 tile, one lane per score)
    # per-graph fused convolution: 64-node variant (two workgroups per CU), 128-node variant with one and two
    # row tiles per wave (65..96 and 97..128 nodes), and a graph beyond 128 nodes (falls back to GEMM + aggregation)
    (64, 2, 6, 3, [64, 1, 33, 2, 50]),
    (128, 2, 10, 4, [100, 70, 128, 65, 3]),
    (64, 1, 5, 2, [150, 20, 7]),
    (48, 1, 5, 3, [4, 2, 7] * 12 + [5]),
    (80, 2, 4, 40, [3, 5] * 50),
    (16, 1, 6, 2, [2, 3] * 100),
    # NCI1-like: B > 256 (1024-thread loss kernel, GEMM readout) and a wide one-hot-style feature matrix (bn_feat statistics
    # through partial rows: k_colstats + k_stats_final)
    (128, 2, 139, 2, [20 + (i * 7) % 21 for i in range(300)]),
])
def test_ragged_and_odd_shapes(hidden, layers, nfeat, ncls, sizes):
    torch.manual_seed(hidden + layers)
    b = _ragged_batch(hidden, nfeat, sizes)
    bd = _ragged_batch(hidden, nfeat, sizes).to(DEV)
    b.y = b.y % ncls
    bd.y = bd.y % ncls
    sd = O.init_state("CausalGCN", nfeat, ncls, hidden=hidden, layers=layers)
    g = torch.Generator().manual_seed(7)
    for k in list(sd):
        if k.endswith(".bias") or ("bn" in k and k.endswith(".weight")):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
    m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(hidden=hidden, layers=layers), nfeat, ncls)
    B = len(sizes)
    perm = torch.randperm(B)
    tr = O.CpuTrainer("CausalGCN", {k: v.clone() for k, v in sd.items()}, ncls, lr=1e-3, layers=layers)
    loss, lc, lo, lco, logits = tr.step(b.x, b.edge_index, b.batch, b.y, perm=perm)
    stats = eng.train_step(bd, perm.to(DEV), adam=False).cpu().numpy()
    lp = eng.buffer("logp", 3 * B * ncls).view(3, B, ncls).cpu()
    for r, t in zip(logits, lp):
        assert (r.detach() - t).abs().max().item() < LOGIT_TOL
    assert np.allclose(stats[:4], [loss.item(), lc.item(), lo.item(), lco.item()], atol=1e-4)
    for k, p in m.named_parameters():
        gref = tr.sd[k].grad
        if gref is not None:
            assert torch.allclose(p.grad.cpu(), gref, atol=1e-4, rtol=3e-3), k


def test_module_surface_reference_loop_on_engine():
    """The reference's own loop shape (train_causal.py:162-200): model(data) -> torch loss ->
    loss.backward() -> torch Adam -> zero_grad(), with CausalGCN running on the engine behind the
    autograd surface; must track the oracle trainer step by step, and eval must not train."""
    from cal_amd import model as M
    from cal_amd.train_causal import causal_loss
    b, bd = _config2_batch(48, seed=8)
    torch.manual_seed(12)
    sd = O.init_state("CausalGCN", 10, 4, hidden=64, layers=2)
    args = _args(layers=2, hidden=64)
    m = M.CausalGCN(10, 4, args)
    m.load_state_dict(sd)
    m = m.to(DEV)
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    tr = O.CpuTrainer("CausalGCN", {k: v.clone() for k, v in sd.items()}, 4, lr=1e-2, layers=2)
    g = torch.Generator().manual_seed(1)
    for step in range(3):
        perm = torch.randperm(48, generator=g)
        ref_loss, *_ = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
        m.train()
        opt.zero_grad()                                   # set_to_none: the views are re-attached in backward
        c, o, co = m(bd, eval_random=True, perm=perm)
        assert getattr(m, "_engine", None) is not None and c.requires_grad
        loss, *_ = causal_loss(c, o, co, bd.y, 4, args)
        loss.backward()
        opt.step()
        assert abs(loss.item() - ref_loss.item()) < 3e-3 * (step + 1), step
        m.eval()
        with torch.no_grad():
            ev = m(bd, eval_random=False)
        ref_ev = O.causal_forward("CausalGCN", {k: v.clone() for k, v in tr.sd.items()}, b.feat, b.edge_index, b.batch,
                                  training=False, layers=2)
        for r, t in zip(ref_ev, ev):
            assert (r.detach() - t.cpu()).abs().max().item() < 5e-3
    assert int(m.bn_feat.num_batches_tracked.item()) == 3


def test_device_randperm_is_a_fresh_uniform_permutation():
    """cal_randperm (model.py:147-152 on the device): always a permutation, a new one per call, reproducible
    from (seed, counter), and unbiased enough that every position sees every value."""
    from cal_amd import _lib
    from cal_amd.plan import _p, _stream
    for B in (1, 2, 7, 128, 1000, 4096):
        cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
        out = torch.empty(B, dtype=torch.long, device=DEV)
        seen = []
        for k in range(4):
            _lib.call("cal_randperm", _p(out), B, 1234, _p(cnt), _stream())
            p = out.cpu()
            assert torch.equal(p.sort().values, torch.arange(B)), B
            seen.append(p.clone())
        assert int(cnt.item()) == 4
        if B >= 7:
            assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
        cnt.fill_(1)                                        # same (seed, counter) -> same permutation
        _lib.call("cal_randperm", _p(out), B, 1234, _p(cnt), _stream())
        assert torch.equal(out.cpu(), seen[1])
    B, trials = 8, 4000
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    outs = torch.empty(trials, B, dtype=torch.long, device=DEV)
    for k in range(trials):
        _lib.call("cal_randperm", _p(outs[k]), B, 99, _p(cnt), _stream())
    counts = torch.zeros(B, B)
    o = outs.cpu()
    for pos in range(B):
        counts[pos] = torch.bincount(o[:, pos], minlength=B).float()
    assert (counts - trials / B).abs().max() < 6 * (trials / B) ** 0.5      # ~6 sigma of a binomial cell


def test_engine_draws_the_same_permutations_as_cal_randperm():
    """Mode bit 16: the step's first kernel draws the intervention permutation (no launch of its own); for the same
    (seed, counter) it is cal_randperm's permutation, the counter advances once per step, and the step equals one that
    was handed that permutation explicitly."""
    from cal_amd import _lib
    from cal_amd.plan import _p, _stream
    _, bd = _config2_batch(24, seed=2)
    sd = O.init_state("CausalGCN", 10, 4, hidden=32, layers=2)
    m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(layers=2, hidden=32))
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    eng.set_perm_rng(777, cnt)
    ref_cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    ref = torch.empty(24, dtype=torch.long, device=DEV)
    for k in range(3):
        stats = eng.train_step(bd, None, adam=False, draw_perm=True).clone()
        g = eng.flat_g.clone()
        _lib.call("cal_randperm", _p(ref), 24, 777, _p(ref_cnt), _stream())
        assert torch.equal(eng.drawn_perm(24), ref) and int(cnt.item()) == k + 1
        m2, eng2 = _engine({k_: v.clone() for k_, v in sd.items()}, _args(layers=2, hidden=32))
        stats2 = eng2.train_step(bd, ref.clone(), adam=False)
        assert torch.equal(stats, stats2) and torch.equal(g, eng2.flat_g)


def test_trainer_device_perm_graph_trains_and_redraws():
    """Default trainer path: the permutation is drawn inside the captured graph (no host upload); an explicit
    host permutation still works on the same trainer and matches the eager engine bit for bit."""
    from cal_amd import model as M
    from cal_amd.trainer import CausalTrainer
    _, b1 = _config2_batch(32, seed=1)
    torch.manual_seed(4)
    sd = O.init_state("CausalGCN", 10, 4, hidden=32, layers=2)
    m = M.CausalGCN(10, 4, _args(layers=2, hidden=32))
    m.load_state_dict(sd)
    tr = CausalTrainer(m.to(DEV), _args(layers=2, hidden=32), lr=1e-2, use_graph=True)
    tr.reserve_for([b1])
    losses, perms = [], []
    for i in range(8):
        losses.append(tr.step(b1)[0].item())
        perms.append(tr.engine.drawn_perm(32).cpu().clone())     # drawn by the step's first kernel
        assert torch.equal(perms[-1].sort().values, torch.arange(32))
    assert len({tuple(p.tolist()) for p in perms}) == 8          # a fresh draw on every replay
    assert losses[-1] < losses[0]
    # explicit permutation on the same trainer -> second captured variant; equals an eager trainer from the same state
    snap = {k: v.clone() for k, v in m.state_dict().items()}
    ea, eq, st = tr.engine.exp_avg.clone(), tr.engine.exp_avg_sq.clone(), tr.engine.step_count.clone()
    perm = torch.arange(32).roll(3)
    l_graph = tr.step(b1, perm)[0].item()
    p_graph = tr.flat_p.clone()
    m2 = M.CausalGCN(10, 4, _args(layers=2, hidden=32))
    m2.load_state_dict(snap)
    tr2 = CausalTrainer(m2.to(DEV), _args(layers=2, hidden=32), lr=1e-2, use_graph=False)
    tr2.engine.exp_avg.copy_(ea); tr2.engine.exp_avg_sq.copy_(eq); tr2.engine.step_count.copy_(st)
    l_eager = tr2.step(b1, perm)[0].item()
    assert abs(l_graph - l_eager) < 1e-6
    assert torch.allclose(p_graph, tr2.flat_p, atol=1e-6)


def test_fused_conv_matches_unfused_and_flags_bad_bounds():
    """The per-graph fused convolution against the GEMM + aggregation path on the same batch (bounds withheld),
    and the status word when the host's per-graph bounds are too small."""
    torch.manual_seed(11)
    sizes = [60, 33, 64, 17, 48, 5]
    b = _ragged_batch(128, 10, sizes)
    bd = _ragged_batch(128, 10, sizes).to(DEV)
    bd.y = bd.y % 4
    assert bd.max_nodes == 64 and bd.max_edges > 0
    sd = O.init_state("CausalGCN", 10, 4, hidden=128, layers=2)
    perm = torch.randperm(len(sizes)).to(DEV)
    res = []
    for fused in (True, False):
        m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(hidden=128, layers=2), 10, 4)
        eng.fused = fused                                   # False: layout withheld -> unfused kernels
        stats = eng.train_step(bd, perm, adam=False).cpu().clone()
        res.append((stats, eng.flat_g.clone(), eng.buffer("logp", 3 * len(sizes) * 4).clone()))
    assert torch.allclose(res[0][0], res[1][0], atol=1e-5)
    assert torch.allclose(res[0][2], res[1][2], atol=2e-5)
    assert torch.allclose(res[0][1], res[1][1], atol=2e-5, rtol=1e-3)
    # a bound that is too small for the kernel variant it selects: the engine flags it (bit 3) instead of
    # reading outside its tiles
    big = _ragged_batch(128, 10, [100, 20]).to(DEV)
    big.y = big.y % 4
    m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(hidden=128, layers=2), 10, 4)
    big.max_nodes, big.max_edges = 50, 10                    # claims the 64-node variant fits
    eng.train_step(big, torch.arange(2, device=DEV), adam=False)
    assert int(eng.buffer("status", 1, dtype=torch.int32).item()) & 8


def test_per_graph_plan_equals_generic_plan_and_flags_violations():
    """k_plan_graph (one kernel, needs the collate's node/edge offsets and no self loops) against plan.hip's generic
    five-launch build on the same batch: identical CSR views, graph pointers and degrees; and the status word when
    the host's promises do not hold."""
    torch.manual_seed(21)
    sizes = [60, 1, 33, 128, 17, 2, 90]

    from cal_amd.data import Batch, Data

    def sparse_batch():            # ~4 out-edges per node, no self loops, duplicate edges allowed
        g = torch.Generator().manual_seed(5)
        ds = []
        for n in sizes:
            if n == 1:
                ei = torch.zeros(2, 0, dtype=torch.long)
            else:
                src = torch.arange(n).repeat_interleave(4)
                dst = (src + 1 + torch.randint(0, n - 1, (src.numel(),), generator=g)) % n
                ei = torch.stack([src, dst])[:, torch.randperm(src.numel(), generator=g)]
            ds.append(Data(x=torch.randn(n, 6, generator=g), edge_index=ei, y=torch.randint(0, 3, (1,), generator=g)))
        return Batch.from_data_list(ds)

    bd = sparse_batch().to(DEV)
    assert bd.no_self_loops and bd.ptr.is_cuda and bd.edge_ptr.is_cuda
    sd = O.init_state("CausalGCN", 6, 3, hidden=64, layers=1)
    N, E, B = bd.x.size(0), bd.edge_index.size(1), len(sizes)
    perm = torch.arange(B, device=DEV)
    views = []
    for fast in (True, False):
        m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(hidden=64, layers=1), 6, 3)
        keep = bd.no_self_loops
        if not fast:
            bd.no_self_loops = False                       # withhold the promise -> generic build
        eng.forward(bd, perm, training=True)
        bd.no_self_loops = keep
        assert int(eng.buffer("status", 1, dtype=torch.int32).item()) == 0
        v = {k: eng.buffer(k, n, dtype=torch.int32).clone() for k, n in
             (("rowptr_dst", N + 1), ("nbr_dst", E), ("eid_dst", E), ("rowptr_src", N + 1), ("nbr_src", E), ("eid_src", E),
              ("gptr", B + 1), ("eptr", B + 1))}
        v["dis_unit"] = eng.buffer("dis_unit", N).clone()
        views.append(v)
    for k in views[0]:
        assert torch.equal(views[0][k], views[1][k]), k
    # promises that do not hold: a self loop, and a batch vector that disagrees with node_ptr
    m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(hidden=64, layers=1), 6, 3)
    bad = sparse_batch().to(DEV)
    bad.edge_index[:, 5] = bad.edge_index[0, 5]            # edge 5 becomes a self loop; the flag still says there is none
    eng.forward(bad, perm, training=True)
    assert int(eng.buffer("status", 1, dtype=torch.int32).item()) & 32
    bad2 = sparse_batch().to(DEV)
    bad2.batch[0] = 1
    eng.forward(bad2, perm, training=True)
    assert int(eng.buffer("status", 1, dtype=torch.int32).item()) & 2


def test_step_sequence_equals_single_steps():
    """CausalTrainer.step_sequence (several train steps per hipGraph launch) against the same steps launched one by one:
    identical parameters (the permutations come from the same device counter stream)."""
    from cal_amd import model as M
    from cal_amd.trainer import CausalTrainer
    import random
    _, b1 = _config2_batch(32, seed=1)
    _, b2 = _config2_batch(32, seed=2)
    _, b3 = _config2_batch(24, seed=3)
    sd = O.init_state("CausalGCN", 10, 4, hidden=32, layers=2)
    outs = []
    for seqmode in (False, True):
        random.seed(7)
        m = M.CausalGCN(10, 4, _args(layers=2, hidden=32))
        m.load_state_dict(sd)
        tr = CausalTrainer(m.to(DEV), _args(layers=2, hidden=32), lr=1e-2, use_graph=True)
        tr.reserve_for([b1, b2, b3])
        for b in (b1, b2, b3):
            tr.prepare(b)
        tr._perm_counter.zero_()                  # captures / warm-ups advanced it: same stream for both runs
        snap = {k: v.clone() for k, v in m.state_dict().items()}
        if seqmode:
            tr.step_sequence([b1, b2, b3])        # capture pass (also a real pass): rewind everything it changed
            m.load_state_dict(snap)
            tr.engine.exp_avg.zero_(); tr.engine.exp_avg_sq.zero_(); tr.engine.step_count.zero_()
            tr._perm_counter.zero_()
        for rep in range(2):
            if seqmode:
                stats = tr.step_sequence([b1, b2, b3])
            else:
                for b in (b1, b2, b3):
                    stats = tr.step(b)
        outs.append((stats.clone(), tr.flat_p.clone()))
    assert torch.allclose(outs[0][0], outs[1][0], atol=1e-6)
    assert torch.allclose(outs[0][1], outs[1][1], atol=1e-6)


@pytest.mark.parametrize("name,kw", [
    ("CausalGCN", dict(cat_or_add="cat")), ("CausalGAT", dict(cat_or_add="cat")),
    ("CausalGCN", dict(without_node_attention=True)), ("CausalGCN", dict(without_edge_attention=True)),
    ("CausalGCN", dict(without_node_attention=True, without_edge_attention=True, cat_or_add="cat")),
    ("CausalGIN", dict()), ("CausalGIN", dict(cat_or_add="cat")),
])
@pytest.mark.parametrize("fused", [True, False])
def test_model_variants_on_the_engine(name, kw, fused):
    """Every causal variant `opts.get_model` can build from the CLI flags runs on the step engine: CausalGIN
    (GINConv(Linear, BN, ReLU, Linear, ReLU) backbone layers, model.py:188-194), `--cat_or_add cat` (fc1_bn_co / fc1_co 2H wide, model.py:65-69,153-154) and the two ablation flags (constant 0.5
    node / edge masks, model.py:99-107, no gradient into the switched-off attention MLP) -- one train step against the
    oracle, through the per-graph fused kernels and through the unfused chain."""
    from cal_amd import model as M
    from cal_amd.engine import StepEngine, supported
    ids = list(range(20))
    b, bd = ref_batch(ids), ref_batch(ids).to(DEV)
    torch.manual_seed(23)
    sd = O.init_state(name, 10, 4, hidden=64, layers=2, heads=4, cat_or_add=kw.get("cat_or_add", "add"))
    args = _args(layers=2, hidden=64, **kw)
    m = getattr(M, name)(10, 4, args)
    m.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=name != "CausalGIN")      # (GINConv keeps an `eps` buffer)
    m = m.to(DEV).train()
    if name == "CausalGAT":
        for c in m.convs:
            c.dropout = 0.0
    assert supported(m)
    eng = StepEngine(m, lr=1e-3)
    eng.fused = fused
    perm = torch.randperm(len(ids))
    okw = {k: v for k, v in kw.items()}
    tr = O.CpuTrainer(name, {k: v.clone() for k, v in sd.items()}, 4, lr=1e-3, layers=2, heads=4, gat_dropout=0.0, **okw)
    loss, lc, lo, lco, logits = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
    stats = eng.train_step(bd, perm.to(DEV), adam=True).cpu().numpy()
    lp = eng.buffer("logp", 3 * len(ids) * 4).view(3, len(ids), 4).cpu()
    for r, t in zip(logits, lp):
        assert (r.detach() - t).abs().max().item() < LOGIT_TOL
    assert np.allclose(stats[:4], [loss.item(), lc.item(), lo.item(), lco.item()], atol=1e-4)
    eng.check_status()
    for k, p in m.named_parameters():
        gref = tr.sd[k].grad
        if gref is None:                       # switched-off attention MLPs / conv_feat.bias: no gradient in the reference
            assert float(p.grad.abs().max()) == 0.0, k
        else:
            assert torch.allclose(p.grad.cpu(), gref, atol=5e-5, rtol=2e-3), k
            # the Adam update (inside k_finish on this path), where the gradient is not numerically zero: every
            # parameter of every variant is finished by some task of that kernel or covered by an update-only range
            mask = gref.abs() > 1e-5
            assert torch.allclose(p.detach().cpu()[mask], tr.sd[k].detach()[mask], atol=3e-5, rtol=1e-3), k
    # eval-mode forward of the same variant
    m.eval()
    sde = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if not k.endswith(".eps")}
    ref = O.causal_forward(name, sde, b.feat, b.edge_index, b.batch, perm=perm, training=False, layers=2, heads=4, **okw)
    out = eng.forward(bd, perm.to(DEV), training=False)
    for r, t in zip(ref, out):
        assert (r - t.cpu()).abs().max().item() < LOGIT_TOL


def test_one_launch_readout_and_in_kernel_adam_equal_the_separate_launches():
    """k_ro_step (readout forward + backward as one launch) and Adam inside k_finish against the launch sequences they
    replace (CAL_AMD_RO_STEP=0: k_ro_fwd_a/fwd_b/bwd_a/bwd_b, CAL_AMD_ADAM_FUSED=0: k_adam): two Adam steps on the same
    batch, same permutation; logits, losses, every gradient, parameter, moment and the step counter agree to rounding
    (the logits are summed over the hidden chunks in a different order, nothing else differs)."""
    b, bd = _config2_batch(96, seed=23)
    sd = O.init_state("CausalGCN", 10, 4, hidden=128, layers=3)
    perm = torch.randperm(96, generator=torch.Generator().manual_seed(3)).to(DEV)
    runs = []
    for env in ({"CAL_AMD_RO_STEP": "0", "CAL_AMD_ADAM_FUSED": "0"}, {}):
        old = {k: os.environ.get(k) for k in ("CAL_AMD_RO_STEP", "CAL_AMD_ADAM_FUSED")}
        os.environ.update(env)
        try:
            m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(), lr=1e-2)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        out = []
        for _ in range(2):
            stats = eng.train_step(bd, perm, adam=True).cpu().numpy().copy()
            out.append((stats, eng.buffer("logp", 3 * 96 * 4).cpu().numpy().copy()))
        runs.append((out, {k: p.detach().cpu().numpy().copy() for k, p in m.named_parameters()},
                     {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters()},
                     (eng.exp_avg.cpu().numpy().copy(), eng.exp_avg_sq.cpu().numpy().copy(), float(eng.step_count.item()))))
        eng.check_status()
    (o0, p0, g0, m0), (o1, p1, g1, m1) = runs
    # step 1 differs by the summation order of the logits only; step 2 starts from parameters that went through Adam at
    # lr = 1e-2, where the sign of a ~1e-8 gradient is rounding noise (the golden-fixture test masks those the same way)
    for it, ((s0, l0), (s1, l1)) in enumerate(zip(o0, o1)):
        assert np.abs(l0 - l1).max() < (2e-5 if it == 0 else 1e-3), it
        assert np.allclose(s0[:4], s1[:4], atol=2e-5 if it == 0 else 1e-3), it
    # (second-step gradients: all but a handful of elements -- a ReLU whose pre-activation sits within rounding of zero takes
    #  a different side in the two sequences once the parameters differ in the last bit)
    for k in p0:
        big = np.abs(g0[k]) > 1e-5
        bad = ~np.isclose(g0[k][big], g1[k][big], atol=1e-5, rtol=2e-2)
        assert bad.sum() <= max(2, int(2e-3 * bad.size)), (k, int(bad.sum()), int(bad.size))
        assert np.allclose(g0[k][big], g1[k][big], atol=2e-4, rtol=0.2), k
    assert m0[2] == m1[2] == 2.0                  # both sequences advanced the step counter once per step
    assert np.allclose(m0[0], m1[0], atol=2e-5, rtol=2e-2)
    assert np.allclose(m0[1], m1[1], atol=1e-8, rtol=5e-2)


@pytest.mark.parametrize("seed,case", [
    (2126, (128, 1, 1, 10, [34, 53])),
    (2110, (128, 1, 139, 4, [54, 6])),
    (2086, (128, 4, 3, 2, [20, 12])),
])
def test_batches_of_two_or_three_graphs_track_the_fp64_step(seed, case):
    """The last batch of an epoch can hold 2-3 graphs: the readout BatchNorms then normalise 2-3 values per column
    (sigma << mean in most columns) and the gradient behind them is the small remainder of a cancellation.  Judged
    against the same step in fp64: the engine may be at most 8x farther from it than the fp32 oracle (floor 1e-4 of the
    tensor's scale).  Cases found by tests/tools/fuzz_engine.py: fp32 sums of squares in k_ro_step's statistics put
    these gradients 10-100 % off before they were shifted by a pivot."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import fuzz_engine
    assert fuzz_engine.run(case, seed) == []


def _sweep_cases():
    from tests.helpers import SWEEP_CASES
    # (99012: a GAT backbone with head width 20 -- not an engine shape, StepEngine says so and the trainer takes the operator path)
    return [c for c in SWEEP_CASES if c[0] != 99012]


@pytest.mark.parametrize("seed,name,kw,case", _sweep_cases(), ids=[str(c[0]) for c in _sweep_cases()])
def test_random_shape_sweep_cases_on_the_engine(seed, name, kw, case):
    """The ten fixed draws of the random-shape sweep (tests/helpers.py: every backbone, add / cat readout, ablation flags,
    ragged sizes up to 129 nodes) through cal_engine_step incl. the in-kernel Adam update and the eval-mode forward of the
    stepped model, judged against the oracle's fp64 step (tests/tools/fuzz_engine.py)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import fuzz_engine
    assert fuzz_engine.run(case, seed, name, kw) == []


@pytest.mark.parametrize("nb", [129, 200, 384, 512])
def test_row_blocked_readout_tracks_the_fp64_step(nb, monkeypatch):
    """129 .. 512 graphs: the one-launch readout runs in row blocks of 128 graphs (k_ro_step<true>: the two BatchNorms'
    statistics and backward sums, the losses and the bias gradients cross the row blocks inside the kernel, the weight
    gradients leave as one slab per row block).  Ragged last blocks (1, 72, 128 rows), every backbone kind's readout is the
    same code, so CausalGCN only; judged like the random-shape sweep against the oracle's fp64 step -- and the same batch
    through the GEMM chain (CAL_AMD_RO_ROWS=0) must pass the same judge."""
    import random
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import fuzz_engine
    rng = random.Random(nb)
    sizes = [rng.randint(2, 24) for _ in range(nb)]
    case = (128, 2, 10, 4, sizes)
    assert fuzz_engine.run(case, 4100 + nb) == []
    monkeypatch.setenv("CAL_AMD_RO_ROWS", "0")
    assert fuzz_engine.run(case, 4100 + nb) == []


def _stage_names():
    from cal_amd import _lib
    h = _lib.lib()
    names, k = [], 1
    while True:
        nm = h.cal_engine_stage_name(k)
        nm = nm.decode() if isinstance(nm, bytes) else nm
        if not nm:
            return names
        names.append(nm)
        k += 1


def test_first_kernel_duties_ride_in_the_plan_kernel(monkeypatch):
    """Round 6: forward + backward steps on the per-graph plan have no k_zero_f64 launch -- k_plan_graph zeroes the fp64 arena
    (all but bn_feat's statistics, which the previous step's k_finish zeroed), advances the counters and draws the in-step
    permutation (engine_plan.hpp: PlanFold).  Three deterministic steps with the fold and with CAL_AMD_FOLD_ZERO=0 give the same
    BITS; the first step of an engine (nothing known about its arena) and a step behind a training forward without backward
    keep the launch; a captured folded step replayed behind such an orphan forward flags status bit 512 and updates nothing."""
    from cal_amd import _lib
    b, bd = _config2_batch()
    torch.manual_seed(1)
    sd = O.init_state("CausalGCN", 10, 4, hidden=128, layers=3)
    perm = torch.randperm(128, device=DEV)
    runs = []
    for fold in ("1", "0"):
        monkeypatch.setenv("CAL_AMD_FOLD_ZERO", fold)
        m, eng = _engine({k: v.clone() for k, v in sd.items()}, _args(), deterministic=True)
        stats, names = [], []
        for _ in range(3):
            stats.append(eng.train_step(bd, perm, adam=True).clone())
            names.append(_stage_names())
        eng.check_status()
        assert "k_zero_f64" in names[0]                                   # first step: the arena is unknown
        assert ("k_zero_f64" in names[1]) == (fold == "0") and ("k_zero_f64" in names[2]) == (fold == "0")
        assert len(names[1]) == len(names[0]) - (1 if fold == "1" else 0)
        runs.append((torch.stack(stats), eng.flat_p.detach().clone(), m.bn_feat.running_mean.clone(), m.bn_feat.running_var.clone()))
        if fold == "1":
            # the in-step permutation draw inside the plan kernel: the same permutation as the stand-alone launch draws
            ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
            eng.set_perm_rng(1234, ctr)
            eng.train_step(bd, None, adam=True, draw_perm=True)
            assert "k_zero_f64" not in _stage_names()
            drawn = eng.drawn_perm(128).clone()
            assert sorted(drawn.tolist()) == list(range(128)) and int(ctr.item()) == 1
            ref = torch.empty(128, dtype=torch.int64, device=DEV)
            ctr2 = torch.zeros(1, dtype=torch.int64, device=DEV)
            from cal_amd.plan import _p, _stream
            _lib.call("cal_randperm", _p(ref), 128, 1234, _p(ctr2), _stream())
            assert torch.equal(drawn, ref)
            # an orphan training forward: the next EAGER step takes the launch again and is still right
            eng.forward(bd, perm, training=True)
            eng.train_step(bd, perm, adam=True)
            assert "k_zero_f64" in _stage_names()
            eng.train_step(bd, perm, adam=True)
            assert "k_zero_f64" not in _stage_names()
            eng.check_status()
            # ... a captured folded step replayed behind one is refused loudly, and the step after it is fine
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                eng.train_step(bd, perm, adam=True)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                eng.train_step(bd, perm, adam=True)
            g.replay()
            torch.cuda.synchronize()
            eng.check_status()
            before = eng.flat_p.detach().clone()
            eng.forward(bd, perm, training=True)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(eng.flat_p.detach(), before)
            with pytest.raises(_lib.CalError, match="0x200"):
                eng.check_status()
            g.replay()
            torch.cuda.synchronize()
            eng.check_status()
            assert not torch.equal(eng.flat_p.detach(), before)
    for u, v in zip(runs[0], runs[1]):
        assert torch.equal(u, v)
