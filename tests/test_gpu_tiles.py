"""Small-graph packing (cal_engine_set_tiles): batches of ~18-30-node graphs (BASELINE.json configs[2..3]: MUTAG / NCI1) run
the per-graph kernels on TILES of consecutive graphs (<= 64 nodes).  A tile is a block-diagonal graph of its own, so only
global_add_pool (model.py:115-116) and its backward see the graphs inside it.  Checked here: the packed step equals the
oracle's and the one-graph-per-workgroup step; graphs without nodes or edges inside a tile; the device collate records the
same tiles as the host collate; an edge that joins two graphs of a tile is flagged, not trained on."""
import argparse

import numpy as np
import pytest
import torch

from oracle import cal_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
LOGIT_TOL = 1e-4


def _args(**kw):
    d = dict(layers=3, hidden=128, with_random=True, without_node_attention=False, without_edge_attention=False,
             fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
    d.update(kw)
    return argparse.Namespace(**d)


def _engine(name, sd, args, nfeat, ncls, tiles=True):
    from cal_amd import model as M
    from cal_amd.engine import StepEngine
    m = getattr(M, name)(nfeat, ncls, args)
    m.load_state_dict(sd, strict=(name != "CausalGIN"))          # (GINConv's eps buffer is not part of the oracle's state)
    m = m.to(DEV).train()
    if name == "CausalGAT":
        for c in m.convs:
            c.dropout = 0.0
    eng = StepEngine(m)
    eng.tiles = "force" if tiles else False          # ("force": also below the batch size from which packing pays)
    return m, eng


@pytest.mark.parametrize("name,kind,nfeat,batch,hidden", [("CausalGCN", "nci1", 139, 96, 128), ("CausalGAT", "mutag", 109, 64, 128),
                                                          ("CausalGIN", "mutag", 109, 50, 64), ("CausalGCN", "mutag", 109, 257, 64)])
def test_packed_step_equals_oracle_and_unpacked_step(name, kind, nfeat, batch, hidden):
    from cal_amd import synth
    from cal_amd.data import Batch
    gs = synth.tu_like(batch, kind=kind, seed=9)
    b, bd = Batch.from_data_list(gs), Batch.from_data_list(gs).to(DEV)
    assert bd.tile_ptr is not None and bd.tile_ptr.numel() - 1 < batch and bd.tile_max_nodes <= 64
    layers = 2 if hidden == 64 else 3
    torch.manual_seed(6)
    sd = O.init_state(name, nfeat, 2, hidden=hidden, layers=layers, heads=4)
    perm = torch.randperm(batch)
    tr = O.CpuTrainer(name, {k: v.clone() for k, v in sd.items()}, 2, lr=1e-3, layers=layers, heads=4,
                      **({"gat_dropout": 0.0} if name == "CausalGAT" else {}))
    loss, lc, lo, lco, logits = tr.step(b.x, b.edge_index, b.batch, b.y, perm=perm)
    # gradients are held against the oracle evaluated in fp64 (a ReLU within rounding of zero makes the fp32 oracle itself
    # differ from it by 6e-4 on conv_feat.weight in the MUTAG-like case; the HIP path, with fp64 cross-row sums, does not)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    tr64 = O.CpuTrainer(name, sd64, 2, lr=1e-3, layers=layers, heads=4, **({"gat_dropout": 0.0} if name == "CausalGAT" else {}))
    tr64.step(b.x.double(), b.edge_index, b.batch, b.y, perm=perm)
    res = {}
    for tiles in (True, False):
        m, eng = _engine(name, {k: v.clone() for k, v in sd.items()}, _args(hidden=hidden, layers=layers), nfeat, 2, tiles=tiles)
        stats = eng.train_step(bd, perm.to(DEV), adam=False).cpu().numpy()
        eng.check_status()
        assert (eng._tiles[1] > 0) == tiles
        lp = eng.buffer("logp", 3 * batch * 2).view(3, batch, 2).cpu().clone()
        pooled = eng.buffer("pooled", 2 * batch * hidden).cpu().clone()
        res[tiles] = (stats, lp, pooled, {k: p.grad.cpu().clone() for k, p in m.named_parameters()})
    for tiles in (True, False):
        stats, lp, _, grads = res[tiles]
        for r, t in zip(logits, lp):
            assert (r.detach() - t).abs().max().item() < LOGIT_TOL, tiles
        assert np.allclose(stats[:4], [loss.item(), lc.item(), lo.item(), lco.item()], atol=1e-4)
        for k, g in grads.items():
            g32, g64 = tr.sd[k].grad, tr64.sd[k].grad
            if g32 is not None:
                e_gpu = (g.double() - g64).abs().max().item()
                e_cpu = (g32.double() - g64).abs().max().item()
                assert e_gpu <= 4 * e_cpu + 1e-5 * max(1.0, g64.abs().max().item()), (k, tiles, e_gpu, e_cpu)
    # packed vs one graph per workgroup: same arithmetic up to the order of a few sums
    assert torch.allclose(res[True][2], res[False][2], atol=1e-4, rtol=1e-5)
    assert (res[True][1] - res[False][1]).abs().max().item() < 2e-5
    for k in res[True][3]:
        a, c = res[True][3][k], res[False][3][k]
        assert torch.allclose(a, c, atol=2e-5 + 1e-4 * float(c.abs().max()), rtol=1e-3), k


def test_causalgin_train_step_above_512_units_matches_oracle():
    """More than 512 units of <= 64-node graphs (one graph per workgroup, B = 600): the per-graph BACKWARD does not apply
    (T <= 512), so the training forward must take the node-level GINConv chain too -- the fused forward leaves only gt1, the
    node-level backward reads gagg / gy (round-3 advisor finding).  Step vs the oracle (model.py:188-194,236-264)."""
    from cal_amd import synth
    from cal_amd.data import Batch
    batch, hidden, layers, nfeat = 600, 64, 2, 109
    gs = synth.tu_like(batch, kind="mutag", seed=11)
    b, bd = Batch.from_data_list(gs), Batch.from_data_list(gs).to(DEV)
    torch.manual_seed(8)
    sd = O.init_state("CausalGIN", nfeat, 2, hidden=hidden, layers=layers, heads=4)
    perm = torch.randperm(batch)
    tr = O.CpuTrainer("CausalGIN", {k: v.clone() for k, v in sd.items()}, 2, lr=1e-3, layers=layers, heads=4)
    loss, lc, lo, lco, logits = tr.step(b.x, b.edge_index, b.batch, b.y, perm=perm)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    tr64 = O.CpuTrainer("CausalGIN", sd64, 2, lr=1e-3, layers=layers, heads=4)
    tr64.step(b.x.double(), b.edge_index, b.batch, b.y, perm=perm)
    for round_ in range(2):          # twice: stale workspace of the first step must not help the second
        m, eng = _engine("CausalGIN", {k: v.clone() for k, v in sd.items()}, _args(hidden=hidden, layers=layers), nfeat, 2, tiles=False)
        stats = eng.train_step(bd, perm.to(DEV), adam=False).cpu().numpy()
        eng.check_status()
        assert eng._tiles[1] == 0
        lp = eng.buffer("logp", 3 * batch * 2).view(3, batch, 2).cpu().clone()
        for r, t in zip(logits, lp):
            assert (r.detach() - t).abs().max().item() < LOGIT_TOL
        assert np.allclose(stats[:4], [loss.item(), lc.item(), lo.item(), lco.item()], atol=1e-4)
        for k, p in m.named_parameters():
            g32, g64 = tr.sd[k].grad, tr64.sd[k].grad
            if g32 is not None:
                e_gpu = (p.grad.cpu().double() - g64).abs().max().item()
                e_cpu = (g32.double() - g64).abs().max().item()
                assert e_gpu <= 4 * e_cpu + 1e-5 * max(1.0, g64.abs().max().item()), (k, e_gpu, e_cpu)


def test_tiles_with_empty_and_edgeless_graphs():
    """A tile may hold a graph without nodes (its pooled row is zero, model.py:115 with no rows) or without edges."""
    from cal_amd import synth
    from cal_amd.data import Batch, Data
    gs = synth.tu_like(12, kind="mutag", seed=3)
    F = gs[0].x.size(1)
    empty = Data(x=torch.zeros(0, F), edge_index=torch.zeros(2, 0, dtype=torch.long), y=torch.tensor([1]))
    lone = Data(x=gs[0].x[:3].clone(), edge_index=torch.zeros(2, 0, dtype=torch.long), y=torch.tensor([0]))
    gl = gs[:2] + [empty] + gs[2:5] + [lone] + gs[5:]          # (a trailing empty graph has no row in global_add_pool's output: batch.max() + 1)
    b, bd = Batch.from_data_list(gl), Batch.from_data_list(gl).to(DEV)
    B = len(gl)
    assert bd.tile_ptr is not None
    torch.manual_seed(1)
    sd = O.init_state("CausalGCN", F, 2, hidden=64, layers=2)
    perm = torch.randperm(B)
    tr = O.CpuTrainer("CausalGCN", {k: v.clone() for k, v in sd.items()}, 2, lr=1e-3, layers=2)
    loss, lc, lo, lco, logits = tr.step(b.x, b.edge_index, b.batch, b.y, perm=perm)
    m, eng = _engine("CausalGCN", {k: v.clone() for k, v in sd.items()}, _args(hidden=64, layers=2), F, 2)
    stats = eng.train_step(bd, perm.to(DEV), adam=False).cpu().numpy()
    eng.check_status()
    assert eng._tiles[1] > 0
    lp = eng.buffer("logp", 3 * B * 2).view(3, B, 2).cpu()
    for r, t in zip(logits, lp):
        assert (r.detach() - t).abs().max().item() < LOGIT_TOL
    assert abs(stats[0] - loss.item()) < 1e-4
    pooled = eng.buffer("pooled", 2 * B * 64).view(2, B, 64)
    assert float(pooled[:, 2].abs().max().item()) == 0.0
    for k, p in m.named_parameters():
        gref = tr.sd[k].grad
        if gref is not None:
            scale = max(1.0, float(gref.abs().max()))
            assert torch.allclose(p.grad.cpu(), gref, atol=1e-4 * scale, rtol=3e-3), k


def test_device_collate_records_the_host_collates_tiles():
    from cal_amd import synth
    from cal_amd.data import Batch
    from cal_amd.device_data import DeviceDataset
    gs = synth.tu_like(80, kind="nci1", seed=4)
    ds = DeviceDataset(gs)
    idx = [5, 9, 2, 70, 33, 34, 35, 1, 0, 79, 40, 41, 42, 43, 44, 45, 46, 47]
    bd = ds.collate(idx)
    bh = Batch.from_data_list([gs[i] for i in idx])
    assert bh.tile_ptr is not None
    assert torch.equal(bd.tile_ptr.cpu(), bh.tile_ptr) and torch.equal(bd.tile_node_ptr.cpu(), bh.tile_node_ptr)
    assert torch.equal(bd.tile_edge_ptr.cpu(), bh.tile_edge_ptr)
    assert (bd.tile_max_nodes, bd.tile_max_edges) == (bh.tile_max_nodes, bh.tile_max_edges)


def test_edge_between_two_graphs_of_a_tile_is_flagged():
    from cal_amd import _lib, synth
    from cal_amd.data import Batch
    gs = synth.tu_like(16, kind="mutag", seed=2)
    bd = Batch.from_data_list(gs).to(DEV)
    assert int(bd.tile_ptr[1]) >= 2                        # the first tile holds at least two graphs
    n0 = int(bd.ptr[1])                                    # first node of graph 1
    bd.edge_index = bd.edge_index.clone()
    bd.edge_index[1, 0] = n0                               # graph 0's first edge now ends in graph 1 (same tile)
    torch.manual_seed(1)
    sd = O.init_state("CausalGCN", gs[0].x.size(1), 2, hidden=64, layers=2)
    m, eng = _engine("CausalGCN", sd, _args(hidden=64, layers=2), gs[0].x.size(1), 2)
    eng.train_step(bd, torch.arange(16, device=DEV), adam=False)
    with pytest.raises(_lib.CalError, match="status"):
        eng.check_status()


@pytest.mark.parametrize("B", [65, 200, 513, 1024])
def test_in_step_permutation_draw_by_several_workgroups_equals_cal_randperm(B):
    """The step's first kernel ranks the permutation with cdiv(B, 64) workgroups (randperm_slice) and its last kernel
    advances the counter: for the same (seed, counter) the result is cal_randperm's permutation, step after step."""
    from cal_amd import _lib, synth
    from cal_amd.data import Batch
    from cal_amd.plan import _p, _stream
    gs = synth.tu_like(B, kind="mutag", seed=B)
    bd = Batch.from_data_list(gs, pack=True).to(DEV)
    F = gs[0].x.size(1)
    torch.manual_seed(2)
    sd = O.init_state("CausalGCN", F, 2, hidden=64, layers=1)
    m, eng = _engine("CausalGCN", sd, _args(hidden=64, layers=1), F, 2)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    eng.set_perm_rng(4242, cnt)
    ref_cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    ref = torch.empty(B, dtype=torch.long, device=DEV)
    for k in range(2):
        eng.train_step(bd, None, adam=False, draw_perm=True)
        _lib.call("cal_randperm", _p(ref), B, 4242, _p(ref_cnt), _stream())
        got = eng.drawn_perm(B)
        assert torch.equal(got, ref) and int(cnt.item()) == k + 1
        assert torch.equal(torch.sort(got).values, torch.arange(B, device=DEV))
    eng.check_status()


@pytest.mark.parametrize("name,F,hidden", [("CausalGCN", 70, 128), ("CausalGCN", 160, 64), ("CausalGIN", 97, 128), ("CausalGIN", 10, 64)])
def test_feature_layer_backward_on_the_matrix_cores(name, F, hidden):
    """k_feat_bwd_mma (65 <= F <= 160, and CausalGIN's feature layer at any F <= 160): one train step vs the oracle evaluated
    in fp32 and fp64 -- conv_feat.weight / bn_feat.* are what the kernel produces."""
    from cal_amd import spmotif
    from cal_amd.data import Batch, Data
    g = torch.Generator().manual_seed(F)
    gs = []
    for d in spmotif.train_mix(40, seed=3):
        n = d.num_nodes
        x = (torch.rand(n, F, generator=g) < 0.2).float() + 0.1 * torch.randn(n, F, generator=g)
        gs.append(Data(x=x, edge_index=d.edge_index, y=d.y))
    b, bd = Batch.from_data_list(gs), Batch.from_data_list(gs).to(DEV)
    layers = 2
    # (seed: with seed 9 the CausalGIN / F = 97 case has one pre-activation of context_convs within 1e-7 of zero; the HIP path
    #  and the oracle land on opposite sides of the ReLU and context_convs.weight differs by 1e-4 in one row -- an error of
    #  neither, DESIGN.md "Tolerances at ill-conditioned shapes")
    torch.manual_seed(9 + F)
    sd = O.init_state(name, F, 4, hidden=hidden, layers=layers, heads=4)
    perm = torch.randperm(40)
    tr = O.CpuTrainer(name, {k: v.clone() for k, v in sd.items()}, 4, lr=1e-3, layers=layers, heads=4)
    loss, lc, lo, lco, logits = tr.step(b.x, b.edge_index, b.batch, b.y, perm=perm)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    tr64 = O.CpuTrainer(name, sd64, 4, lr=1e-3, layers=layers, heads=4)
    tr64.step(b.x.double(), b.edge_index, b.batch, b.y, perm=perm)
    m, eng = _engine(name, {k: v.clone() for k, v in sd.items()}, _args(hidden=hidden, layers=layers), F, 4)
    stats = eng.train_step(bd, perm.to(DEV), adam=False).cpu().numpy()
    eng.check_status()
    lp = eng.buffer("logp", 3 * 40 * 4).view(3, 40, 4).cpu()
    for r, t in zip(logits, lp):
        assert (r.detach() - t).abs().max().item() < LOGIT_TOL
    assert abs(stats[0] - loss.item()) < 1e-4
    for k, p in m.named_parameters():
        g32, g64 = tr.sd[k].grad, tr64.sd[k].grad
        if g32 is not None:
            e_gpu = (p.grad.cpu().double() - g64).abs().max().item()
            e_cpu = (g32.double() - g64).abs().max().item()
            # (floor 2e-5: with these dense random features more activations sit within rounding of a ReLU boundary than with
            #  one-hot degrees; measured 1.03e-5 on bnc.weight for CausalGIN, where the fp32 oracle happens to be 2e-8 away)
            assert e_gpu <= 4 * e_cpu + 2e-5 * max(1.0, g64.abs().max().item()), (k, e_gpu, e_cpu)
