"""cal_amd.optim without a GPU: a CPU-resident model has no step engine, so ``bind`` declines and ``EngineAdam`` is
``torch.optim.Adam`` (the reference's CPU plumbing run, BASELINE.json configs[0], keeps the statement-by-statement loop
on libcalhost.so)."""
import argparse

import torch

from cal_amd import model as M
from cal_amd.data import Batch
from cal_amd.optim import EngineAdam, bind
from cal_amd.spmotif import train_mix
from cal_amd.train_causal import causal_loss, train_causal_epoch, DataLoader


def _args(**kw):
    d = dict(layers=2, hidden=32, with_random=False, without_node_attention=False, without_edge_attention=False,
             fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5, eval_random=False)
    d.update(kw)
    return argparse.Namespace(**d)


def test_bind_declines_on_cpu_and_engine_adam_is_adam():
    args = _args()
    gs = train_mix(16, seed=2)
    outs = []
    for cls in (EngineAdam, torch.optim.Adam):
        torch.manual_seed(0)
        m = M.CausalGCN(10, 4, args).train()
        opt = cls(m.parameters(), lr=1e-2, weight_decay=1e-3)
        assert bind(opt, m) is None
        b = Batch.from_data_list(gs)
        for _ in range(2):
            opt.zero_grad()
            c, o, co = m(b, eval_random=False)
            loss, *_ = causal_loss(c, o, co, b.y, 4, args)
            loss.backward()
            opt.step()
        outs.append({k: p.detach().clone() for k, p in m.named_parameters()})
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k


def test_train_causal_epoch_on_cpu_keeps_the_statement_loop():
    args = _args()
    gs = train_mix(24, seed=3)
    torch.manual_seed(0)
    m = M.CausalGCN(10, 4, args)
    opt = EngineAdam(m.parameters(), lr=1e-3)
    out = train_causal_epoch(m, opt, DataLoader(gs, 8, shuffle=False), torch.device("cpu"), args)
    assert len(out) == 5 and getattr(opt, "_cal_binding", None) is None
    assert abs(out[0] - (0.5 * out[1] + out[2] + 0.5 * out[3])) < 1e-5
    assert {float(s["step"]) for s in opt.state_dict()["state"].values()} == {3.0}
