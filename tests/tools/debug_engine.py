"""Compare every intermediate of the native engine with the CPU oracle (debug aid; lives under tests/ because only tests may use oracle/)."""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cal_amd import model as M
from cal_amd.engine import StepEngine
from oracle import cal_oracle as O
from tests.helpers import ref_batch

H = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ids = [0, 4, 7, 10, 13, 16, 19, 22]
args = argparse.Namespace(layers=L, hidden=H, with_random=True, without_node_attention=False,
                          without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
torch.manual_seed(1)
sd = O.init_state("CausalGCN", 10, 4, hidden=H, layers=L)
g = torch.Generator().manual_seed(2)
for k in list(sd):
    if k.endswith(".bias") or ("bn" in k and k.endswith(".weight")):
        sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
m = M.CausalGCN(10, 4, args)
m.load_state_dict(sd)
m = m.cuda().train()
eng = StepEngine(m, lr=1e-3)
b = ref_batch(ids)
bd = ref_batch(ids).to("cuda")
perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4])
tr = O.CpuTrainer("CausalGCN", {k: v.clone() for k, v in sd.items()}, 4, lr=1e-3, layers=L)
# oracle forward with intermediates (on a copy so BN stats are not double-updated)
sd2 = {k: v.clone() for k, v in sd.items()}
(lc, lo, lco), inter = O.causal_forward("CausalGCN", sd2, b.feat, b.edge_index, b.batch, perm=perm, training=True,
                                        layers=L, return_intermediates=True)
loss, c_loss, o_loss, co_loss, logits = tr.step(b.feat, b.edge_index, b.batch, b.y, perm=perm)
stats = eng.train_step(bd, perm.cuda(), adam=True)
torch.cuda.synchronize()
N, E, B = b.feat.size(0), b.edge_index.size(1), 8
def err(name, got, ref):
    got = got.detach().cpu().double(); ref = ref.detach().double()
    e = (got - ref).abs().max().item()
    print("%-28s max|err| %.3e   (ref max %.3e)" % (name, e, ref.abs().max().item()))
err("x (last backbone)", eng.buffer("h", (L + 1) * N * H).view(L + 1, N, H)[L], inter["x"])
err("node_att", eng.buffer("anode", 2 * N).view(N, 2), inter["node_att"])
err("edge_att", eng.buffer("att", 2 * E).view(2, E).t(), inter["edge_att"])
err("hc", eng.buffer("hco", 2 * N * H).view(2, N, H)[0], inter["xc"])
err("ho", eng.buffer("hco", 2 * N * H).view(2, N, H)[1], inter["xo"])
err("pooled c", eng.buffer("pooled", 2 * B * H).view(2, B, H)[0], inter["xc_pool"])
lp = eng.buffer("logp", 3 * B * 4).view(3, B, 4)
for i, (n, r) in enumerate(zip("c o co".split(), logits)):
    err("logp " + n, lp[i], r)
print("stats", stats.tolist(), "oracle", [loss.item(), c_loss.item(), o_loss.item(), co_loss.item()])
for k, p in m.named_parameters():
    gref = tr.sd[k].grad
    if gref is None:
        continue
    err("grad " + k, p.grad, gref)
for k, v in m.state_dict().items():
    err("post " + k, v.float(), tr.sd[k].float())
