"""graphs/s of the reference-shaped loop (train_causal.py:162-200: model(data) -> torch loss ->
backward -> torch Adam, incl. its .item() syncs) on the nn.Module surface, engine on / off."""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import model as M, spmotif
from cal_amd.data import Batch
from cal_amd.train_causal import causal_loss
args = argparse.Namespace(layers=3, hidden=128, with_random=True, without_node_attention=False,
                          without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
gs = spmotif.train_mix(8 * 128, seed=1)
batches = [Batch.from_data_list(gs[i * 128:(i + 1) * 128]).to("cuda") for i in range(8)]
for use_engine in (True, False):
    torch.manual_seed(0)
    m = M.CausalGCN(10, 4, args).cuda()
    m.use_engine = use_engine
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    def step(b):
        opt.zero_grad()
        c, o, co = m(b, eval_random=True)
        loss, lc, lo, lco = causal_loss(c, o, co, b.y, 4, args)
        pred = o.max(1)[1]
        correct = pred.eq(b.y.view(-1)).sum().item()
        loss.backward()
        tot = loss.item() + lc.item() + lo.item() + lco.item()
        opt.step()
        return tot
    m.train()
    for i in range(10): step(batches[i % 8])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for i in range(n): step(batches[i % 8])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("reference-shaped loop, engine=%s: %.2f ms/step, %.0f graphs/s" % (use_engine, 1e3 * dt / n, 128 * n / dt))
