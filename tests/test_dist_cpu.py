"""world_size-2 gloo test of the data-parallel exchange: flat
gradient bucket, one all-reduce, mean over replicas, parameters that never receive a
gradient (conv_feat.bias) tolerated, disjoint loader shards."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 5)
        self.unused = torch.nn.Parameter(torch.zeros(3))     # like conv_feat.bias: grad stays None
        self.b = torch.nn.Linear(5, 2)

    def forward(self, x):
        return self.b(torch.relu(self.a(x)))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cal_amd.trainer import flatten_parameters
    torch.manual_seed(0)
    net = _Net()
    flat_p, flat_g = flatten_parameters(net)
    torch.manual_seed(100 + rank)
    x = torch.randn(7, 6)
    flat_g.zero_()
    net(x).pow(2).sum().backward()
    assert net.unused.grad is not None and torch.all(net.unused.grad == 0)
    assert net.a.weight.grad.untyped_storage().data_ptr() == flat_g.untyped_storage().data_ptr()  # views of the bucket
    dist.all_reduce(flat_g)
    flat_g.mul_(1.0 / world)
    opt = torch.optim.Adam([flat_p], lr=1e-2)
    flat_p.grad = flat_g
    opt.step()
    q.put((rank, flat_g.numpy().copy(), flat_p.detach().numpy().copy(), x.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_bucket_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    (_, g0, p0, x0), (_, g1, p1, x1) = [(r, *map(torch.from_numpy, t)) for r, *t in res]
    assert torch.equal(g0, g1) and torch.equal(p0, p1)                # replicas stay identical
    # single-process reference: mean of the two replica gradients
    torch.manual_seed(0)
    net = _Net()
    from cal_amd.trainer import flat_offsets
    params = list(net.parameters())
    offs, total = flat_offsets(params)                                # every parameter on a 16-byte boundary, zero padding
    assert total == g0.numel() and all(o % 4 == 0 for o in offs)
    ref = []
    for x in (x0, x1):
        net.zero_grad()
        net(x).pow(2).sum().backward()
        flat = torch.zeros(total)
        for p, o in zip(params, offs):
            if p.grad is not None:
                flat[o:o + p.numel()] = p.grad.reshape(-1)
        ref.append(flat)
    assert torch.allclose(g0, (ref[0] + ref[1]) / 2, atol=1e-6)


def _model_worker(rank, world, port, q):
    """CausalGCN itself (operator path on libcalhost.so: CPU tensors) through the flat bucket + one all-reduce."""
    import argparse
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cal_amd import model as M
        from cal_amd.train_causal import causal_loss
        from cal_amd.trainer import flatten_parameters
        from oracle import cal_oracle as O
        from tests.helpers import ref_batch
        args = argparse.Namespace(layers=2, hidden=32, with_random=True, without_node_attention=False, without_edge_attention=False,
                                  fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
        torch.manual_seed(3)
        sd = O.init_state("CausalGCN", 10, 4, hidden=32, layers=2)
        m = M.CausalGCN(10, 4, args)
        m.load_state_dict(sd)
        m.train()
        flat_p, flat_g = flatten_parameters(m)
        b = ref_batch(list(range(6 * rank, 6 * rank + 6)))
        perm = torch.arange(b.num_graphs - 1, -1, -1)
        flat_g.zero_()
        c, o, co = m(b, eval_random=True, perm=perm)
        loss, *_ = causal_loss(c, o, co, b.y, 4, args)
        loss.backward()
        local = flat_g.clone()
        dist.all_reduce(flat_g)
        flat_g.mul_(1.0 / world)
        opt = torch.optim.Adam([flat_p], lr=1e-2)
        flat_p.grad = flat_g
        opt.step()
        # the oracle's gradient on this rank's shard (train_causal.py:173-192 per replica)
        tr = O.CpuTrainer("CausalGCN", {k: v.clone() for k, v in sd.items()}, 4, lr=1e-2, layers=2)
        tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
        from cal_amd.trainer import flat_offsets
        offs, total = flat_offsets(list(m.parameters()))
        ref = torch.zeros(total)
        for (k, p), o in zip(m.named_parameters(), offs):
            if tr.sd[k].grad is not None:
                ref[o:o + p.numel()] = tr.sd[k].grad.reshape(-1)
        q.put((rank, local.numpy().copy(), flat_g.numpy().copy(), flat_p.detach().numpy().copy(), ref.numpy().copy(), None))
    except Exception as exc:
        import traceback
        q.put((rank, None, None, None, None, traceback.format_exc() + repr(exc)))
    dist.barrier()
    dist.destroy_process_group()


def test_causalgcn_replicas_exchange_the_flat_bucket_world2():
    """The N > 1 path on the MODEL (round-3 review: the bucket test ran a toy net): two gloo ranks, each a CausalGCN replica on
    its own shard through the host library, one all-reduce of the flat gradient, mean, Adam -- local gradients equal the
    oracle's on the shard, the exchanged bucket is their mean, replicas end bit-identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_model_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in res:
        assert r[5] is None, r[5]
    (l0, g0, p0, r0), (l1, g1, p1, r1) = [tuple(torch.from_numpy(a) for a in r[1:5]) for r in res]
    assert torch.allclose(l0, r0, atol=2e-5, rtol=1e-3) and torch.allclose(l1, r1, atol=2e-5, rtol=1e-3)      # local == oracle on the shard
    assert torch.equal(g0, g1) and torch.equal(p0, p1)                                                      # replicas identical
    assert torch.allclose(g0, 0.5 * (l0 + l1), atol=1e-7)                                                   # the bucket holds the mean


def test_loader_shards_are_disjoint_and_cover():
    from cal_amd.data import DataLoader
    from tests.helpers import ref_graphs
    gs = ref_graphs()
    shards = []
    for r in range(2):
        dl = DataLoader(gs, 4, shuffle=True, rank=r, world_size=2, generator=torch.Generator().manual_seed(9))
        shards.append(dl._indices())
        assert sum(b.num_graphs for b in dl) == len(shards[-1])
    assert not set(shards[0]) & set(shards[1])
    assert sorted(shards[0] + shards[1]) == list(range(len(gs)))


def test_loader_shards_have_equal_length_for_any_dataset_size():
    """n % world != 0 must not give ranks different step counts (mismatched all-reduce counts hang), and
    without an explicit generator every rank must draw the SAME permutation (seed + epoch), like DistributedSampler."""
    from cal_amd.data import DataLoader, shard_indices
    from tests.helpers import ref_graphs
    gs = ref_graphs(list(range(10)))
    for world in (2, 3, 4):
        for drop_last in (False, True):
            loaders = [DataLoader(gs, 2, shuffle=True, rank=r, world_size=world, drop_last=drop_last, seed=5) for r in range(world)]
            for epoch in range(2):
                shards = [dl._indices() for dl in loaders]
                assert len({len(s) for s in shards}) == 1, (world, drop_last, shards)
                assert len({len(dl) for dl in loaders}) == 1
                flat = sum(shards, [])
                if drop_last:
                    assert len(set(flat)) == len(flat) == (10 // world) * world
                else:
                    assert set(flat) == set(range(10)) and len(flat) == -(-10 // world) * world
                steps = [sum(1 for _ in dl) for dl in loaders]          # iterating advances the epoch on every rank
                assert len(set(steps)) == 1 and steps[0] == len(loaders[0])
            assert loaders[0]._indices() != shard_indices(10, True, 0, world, drop_last, None, 5, 0)   # epoch 2 != epoch 0
