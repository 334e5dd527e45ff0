import argparse, sys, os, random, cProfile, pstats, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cal_amd import model as M, spmotif
from cal_amd.device_data import DeviceDataset, DeviceLoader
from cal_amd.optim import EngineAdam
from cal_amd.train_causal import causal_loss
args = argparse.Namespace(layers=3, hidden=128, with_random=True, without_node_attention=False, without_edge_attention=False,
                          fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
gs = spmotif.train_mix(1024, bias=0.9, node_num=7, seed=1)
dev = torch.device("cuda")
model = M.CausalGCN(10, 4, args).cuda()
opt = EngineAdam(model.parameters(), lr=1e-3)
loader = DeviceLoader(DeviceDataset(gs), 128, shuffle=True)
def epoch():
    for data in loader:
        opt.zero_grad()
        c, o, co = model(data, eval_random=True)
        loss, lc, lo, lco = causal_loss(c, o, co, data.y, 4, args)
        loss.backward()
        v = loss.item()
        opt.step()
epoch(); epoch()
gc.disable()
cProfile.run("epoch(); epoch(); epoch(); epoch()", "/tmp/p.out")
pstats.Stats("/tmp/p.out").sort_stats("tottime").print_stats(28)
