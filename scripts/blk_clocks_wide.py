"""Per-workgroup phase timestamps of the wide per-graph kernels (engine_gwide.hpp; build with CAL_HIPCC_EXTRA=-DCAL_BLK_CLOCKS):
runs the engine step truncated after launch site `k` on a batch of node_num = 15 SPMotif graphs.
usage: [BLK_B=32] python scripts/blk_clocks_wide.py <stop> [<stop> ..]"""
import ctypes, os, sys, torch, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import _lib, model as M, spmotif
from cal_amd.data import Batch
from cal_amd.engine import StepEngine
args = argparse.Namespace(layers=3, hidden=128, with_random=True, without_node_attention=False,
                          without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
torch.manual_seed(0)
B = int(os.environ.get("BLK_B", "32"))
m = M.CausalGCN(10, 4, args).cuda().train()
eng = StepEngine(m)
b = Batch.from_data_list(spmotif.train_mix(B, node_num=15, seed=5)).to("cuda")
perm = torch.randperm(B, device="cuda")
h = _lib.lib()
f = h.cal_debug_blk_clocks
f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int
eng.train_step(b, perm, adam=False)
torch.cuda.synchronize()
names, k = [], 1
while True:
    nm = h.cal_engine_stage_name(k)
    nm = nm.decode() if isinstance(nm, bytes) else nm
    if not nm:
        break
    names.append(nm); k += 1
print({i + 1: n for i, n in enumerate(names)})
for stop in [int(a) for a in sys.argv[1:]]:
    h.cal_engine_debug_stop(stop)
    for _ in range(5): eng.train_step(b, perm, adam=False)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 8192)()
    assert f(out) == 0
    t = np.array(list(out), dtype=np.int64).reshape(2048, 4) / 100.0          # entry, mark 2, mark 3, exit
    t = t[t[:, 3] > 0]
    t = t[t[:, 0] > t[:, 0].max() - 200.0]          # the blocks of the latest launch
    t0 = t[:, 0].min()
    d = t[:, 3] - t[:, 0]
    print("stop %d (%s): %d workgroups, starts spread %.2f us, durations min/med/max %.2f/%.2f/%.2f us, last end %.2f us after first start"
          % (stop, names[stop - 1] if stop <= len(names) else "?", len(t), t[:, 0].max() - t0, d.min(), np.median(d), d.max(), t[:, 3].max() - t0))
    if (t[:, 1] > 0).all() and (t[:, 2] > 0).all():
        for name, ia, ib in (("entry->mark2", 0, 1), ("mark2->mark3", 1, 2), ("mark3->exit", 2, 3)):
            x = t[:, ib] - t[:, ia]
            print("   %-14s min/med/max %.2f/%.2f/%.2f us" % (name, x.min(), np.median(x), x.max()))
h.cal_engine_debug_stop(0)
