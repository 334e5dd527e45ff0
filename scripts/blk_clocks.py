"""Per-workgroup start/end timestamps of the instrumented kernel (build with CAL_HIPCC_EXTRA=-DCAL_BLK_CLOCKS):
runs the engine step truncated after launch site `k` so that the last instrumented launch is the one of interest."""
import ctypes, os, sys, torch, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
f = _lib.lib().cal_debug_blk_clocks
f.argtypes = [ctypes.c_void_p]; f.restype = ctypes.c_int
for stop in [int(a) for a in sys.argv[1:]]:
    _lib.lib().cal_engine_debug_stop(stop)
    for _ in range(5): eng.train_step(b, perm, adam=False)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 8192)()
    assert f(out) == 0
    t = np.array(list(out), dtype=np.int64).reshape(4096, 2) / 100.0
    t = t[t[:, 1] > 0]
    # keep the blocks of the latest launch: those whose start is within 100 us of the newest start
    t = t[t[:, 0] > t[:, 0].max() - 100.0]
    t0 = t[:, 0].min()
    d = t[:, 1] - t[:, 0]
    print("stop %d: %d workgroups, starts spread %.2f us, durations min/med/max %.2f/%.2f/%.2f us, last end %.2f us after first start"
          % (stop, len(t), t[:, 0].max() - t0, d.min(), np.median(d), d.max(), t[:, 1].max() - t0))
    order = np.argsort(t[:, 0])
    q = [0, len(t) // 4, len(t) // 2, 3 * len(t) // 4, len(t) - 1]
    print("   start quantiles (us):", " ".join("%.2f" % (t[order[i], 0] - t0) for i in q), " end quantiles:", " ".join("%.2f" % (np.sort(t[:, 1])[i] - t0) for i in q))
_lib.lib().cal_engine_debug_stop(0)
