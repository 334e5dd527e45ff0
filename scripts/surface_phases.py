"""Where the statement-by-statement loop on the nn.Module surface spends its time (host wall per phase, no syncs inside)."""
import argparse, sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cal_amd import model as M, spmotif
from cal_amd.data import DataLoader
from cal_amd.device_data import DeviceDataset, DeviceLoader
from cal_amd.optim import EngineAdam
from cal_amd.train_causal import causal_loss
args = argparse.Namespace(layers=3, hidden=128, with_random=True, without_node_attention=False, without_edge_attention=False,
                          fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
gs = spmotif.train_mix(2048, bias=0.9, node_num=7, seed=1)
dev = torch.device("cuda")
import gc
if os.environ.get("NOGC"): gc.freeze(); gc.disable()
for loader_kind in ("device", "host"):
    torch.manual_seed(1); random.seed(1)
    model = M.CausalGCN(10, 4, args).cuda()
    opt = EngineAdam(model.parameters(), lr=1e-3)
    loader = DeviceLoader(DeviceDataset(gs), 128, shuffle=True) if loader_kind == "device" else DataLoader(gs, 128, shuffle=True)
    for sync in (False, True):
        ph = dict(load=0.0, zero=0.0, to=0.0, fwd=0.0, loss=0.0, bwd=0.0, item=0.0, step=0.0)
        mx = dict.fromkeys(ph, 0.0)
        n = 0
        for ep in range(3):
            it = iter(loader)
            while True:
                t0 = time.perf_counter()
                try:
                    data = next(it)
                except StopIteration:
                    break
                t1 = time.perf_counter(); opt.zero_grad()
                t2 = time.perf_counter(); data = data.to(dev)
                t3 = time.perf_counter(); c, o, co = model(data, eval_random=True)
                if sync: torch.cuda.synchronize()
                t4 = time.perf_counter(); loss, lc, lo, lco = causal_loss(c, o, co, data.y, 4, args)
                if sync: torch.cuda.synchronize()
                t5 = time.perf_counter(); loss.backward()
                if sync: torch.cuda.synchronize()
                t6 = time.perf_counter(); v = loss.item()
                t7 = time.perf_counter(); opt.step()
                if sync: torch.cuda.synchronize()
                t8 = time.perf_counter()
                if ep > 0:
                    for k, d in zip(ph, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5, t7 - t6, t8 - t7)):
                        ph[k] += d; mx[k] = max(mx[k], d)
                    n += 1
        print(loader_kind, "sync" if sync else "async", "per step us:", {k: round(1e6 * v / n, 1) for k, v in ph.items()}, "total", round(1e6 * sum(ph.values()) / n, 1), "max", {k: round(1e6 * v) for k, v in mx.items()})
