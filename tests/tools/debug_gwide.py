"""Wide per-graph convolutions (engine_gwide.hpp) against the CPU oracle, intermediate by intermediate (debug aid; lives under
tests/ because only tests may use oracle/).

usage: python tests/tools/debug_gwide.py [B] [H] [L] [seed]     (graphs: the reference-generated node_num = 15 fixtures, ids 24-31)
"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cal_amd import _lib
from cal_amd import model as M
from cal_amd.data import Batch
from cal_amd.engine import StepEngine
from oracle import cal_oracle as O
from tests.helpers import ref_graphs

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
L = int(sys.argv[3]) if len(sys.argv) > 3 else 3
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 5
gs = ref_graphs([24 + (i % 8) for i in range(B)])
for i, g in enumerate(gs):
    g.y = torch.tensor([i % 4])
args = argparse.Namespace(layers=L, hidden=H, with_random=True, without_node_attention=False,
                          without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
torch.manual_seed(seed)
sd = O.init_state("CausalGCN", 10, 4, hidden=H, layers=L)
g = torch.Generator().manual_seed(2)
for k in list(sd):
    if k.endswith(".bias") or ("bn" in k and k.endswith(".weight")):
        sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
m = M.CausalGCN(10, 4, args)
m.load_state_dict(sd)
m = m.cuda().train()
eng = StepEngine(m, lr=1e-3)
b = Batch.from_data_list(gs)
bd = Batch.from_data_list(gs).to("cuda")
print("max_nodes", bd.max_nodes, "N", b.feat.size(0), "E", b.edge_index.size(1))
perm = torch.randperm(B)
tr = O.CpuTrainer("CausalGCN", {k: v.clone() for k, v in sd.items()}, 4, lr=1e-3, layers=L)
sd2 = {k: v.clone() for k, v in sd.items()}
(lc, lo, lco), inter = O.causal_forward("CausalGCN", sd2, b.feat, b.edge_index, b.batch, perm=perm, training=True,
                                        layers=L, return_intermediates=True)
loss, c_loss, o_loss, co_loss, logits = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
stats = eng.train_step(bd, perm.cuda(), adam=True)
torch.cuda.synchronize()
h = _lib.lib()
names = []
k = 1
while True:
    nm = h.cal_engine_stage_name(k)
    nm = nm.decode() if isinstance(nm, bytes) else nm
    if not nm:
        break
    names.append(nm); k += 1
print("stages:", names)
N, E = b.feat.size(0), b.edge_index.size(1)
bad = 0
def err(name, got, ref, tol=None):
    global bad
    got = got.detach().cpu().double(); ref = ref.detach().double()
    e = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    flag = ""
    if tol is not None and not e <= tol * max(1.0, scale):
        flag = "   <-- BAD"; bad += 1
    print("%-28s max|err| %.3e   (ref max %.3e)%s" % (name, e, scale, flag))
err("x (last backbone)", eng.buffer("h", (L + 1) * N * H).view(L + 1, N, H)[L], inter["x"], 1e-4)
err("node_att", eng.buffer("anode", 2 * N).view(N, 2), inter["node_att"], 1e-4)
err("edge_att", eng.buffer("att", 2 * E).view(2, E).t(), inter["edge_att"], 1e-4)
err("hc", eng.buffer("hco", 2 * N * H).view(2, N, H)[0], inter["xc"], 1e-4)
err("ho", eng.buffer("hco", 2 * N * H).view(2, N, H)[1], inter["xo"], 1e-4)
err("pooled c", eng.buffer("pooled", 2 * B * H).view(2, B, H)[0], inter["xc_pool"], 1e-4)
lp = eng.buffer("logp", 3 * B * 4).view(3, B, 4)
for i, (n, r) in enumerate(zip("c o co".split(), logits)):
    err("logp " + n, lp[i], r, 1e-4)
print("stats", stats.tolist(), "oracle", [loss.item(), c_loss.item(), o_loss.item(), co_loss.item()])
for k, p in m.named_parameters():
    gref = tr.sd[k].grad
    if gref is None:
        continue
    err("grad " + k, p.grad, gref, 2e-3)
eng.check_status()
print("BAD" if bad else "OK", bad)
