"""BASELINE.json configs[4] per-GPU shape: BA(m=2) graphs of 5000 nodes, hidden 256, 32 graphs / GPU
(N = 160 000, E' ~ 800 000): one train step of CausalGAT (operator-level path) and CausalGCN (engine)."""
import argparse, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import model as M
from cal_amd.data import Batch, Data
from cal_amd.spmotif import _ba_edges
from cal_amd.trainer import CausalTrainer

def ba_graph(n, rng, label):
    _, edges = _ba_edges(n, 2, rng)
    e = np.array(edges, dtype=np.int64).T
    ei = np.concatenate([e, e[::-1]], 1)
    deg = np.bincount(ei[0], minlength=n)
    feat = np.zeros((n, 10), np.float32); feat[np.arange(n), np.minimum(deg, 9)] = 1
    return Data(feat=torch.from_numpy(feat), edge_index=torch.from_numpy(ei), y=torch.tensor([label]))

rng = np.random.default_rng(0)
gs = [ba_graph(5000, rng, i % 4) for i in range(32)]
b = Batch.from_data_list(gs).to("cuda")
print("N", b.feat.size(0), "E", b.edge_index.size(1))
for name in ("CausalGCN", "CausalGAT"):
    args = argparse.Namespace(layers=3, hidden=256, with_random=True, without_node_attention=False,
                              without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
    torch.manual_seed(0)
    m = getattr(M, name)(10, 4, args).cuda()
    tr = CausalTrainer(m, args, use_graph=False)
    for _ in range(3): st = tr.step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): st = tr.step(b)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("%s (%s): %.2f ms/step, %.0f graphs/s, loss %.4f, peak mem %.1f GB" % (
        name, "engine" if tr.engine is not None else "op-level", dt * 1e3, 32 / dt, st[0].item(), torch.cuda.max_memory_allocated() / 2**30))
