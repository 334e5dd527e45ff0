"""Phase timestamps inside the fused readout kernels (build with CAL_HIPCC_EXTRA=-DCAL_RO_CLOCKS)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
from cal_amd import _lib, model as M, spmotif
from cal_amd.data import Batch
from cal_amd.engine import StepEngine
args = argparse.Namespace(layers=3, hidden=128, with_random=True, without_node_attention=False,
                          without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
torch.manual_seed(0)
m = M.CausalGCN(10, 4, args).cuda().train()
eng = StepEngine(m)
b = Batch.from_data_list(spmotif.train_mix(128, seed=5)).to("cuda")
perm = torch.randperm(128, device="cuda")
for _ in range(5): eng.train_step(b, perm, adam=False)
if len(sys.argv) > 1:      # graph replay: the timings of a kernel inside a dense launch sequence
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(4): eng.train_step(b, perm, adam=False)
    for _ in range(5): g.replay()
torch.cuda.synchronize()
out = (ctypes.c_longlong * 64)()
f = _lib.lib().cal_debug_ro_clocks
f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int
assert f(out) == 0
v = list(out)
for name, lo, hi in (("gconv", 32, 39), ("fwd_a", 0, 5), ("fwd_b", 6, 10), ("bwd_a", 24, 28), ("bwd_b", 16, 21), ("ro_step", 40, 52)):
    print(name, " ".join("%.2fus" % ((v[k + 1] - v[k]) / 100.0) for k in range(lo, hi)), "total %.2fus" % ((v[hi] - v[lo]) / 100.0))
