"""Real (un-profiled) cumulative GPU time of the engine step after each launch site, measured by
truncating the step (cal_engine_debug_stop) and replaying it from a hipGraph."""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import _lib, model as M, spmotif
from cal_amd.data import Batch
from cal_amd.engine import StepEngine
args = argparse.Namespace(layers=3, hidden=128, with_random=True, without_node_attention=False,
                          without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
torch.manual_seed(0)
if os.environ.get("CAL_STAGE_DATA") == "nci1":          # config 4 stand-in: 512 NCI1-like graphs, F = 139, 2 classes
    from cal_amd import synth
    gl = synth.tu_like(512, kind="nci1", seed=5)
    nf, nc = 139, 2
elif os.environ.get("CAL_STAGE_DATA") == "mutag":       # config 3 stand-in: 64 MUTAG-like graphs, F = 109, 2 classes
    from cal_amd import synth
    gl = synth.tu_like(64, kind="mutag", seed=5)
    nf, nc = 109, 2
elif os.environ.get("CAL_STAGE_DATA") == "nn15":        # the reference's default SPMotif shape (opts.py:18): 32 graphs of ~235 nodes
    gl = spmotif.train_mix(32, seed=5, node_num=15)
    nf, nc = 10, 4
else:
    gl = spmotif.train_mix(128, seed=5)
    nf, nc = 10, 4
m = getattr(M, os.environ.get("CAL_STAGE_MODEL", "CausalGCN"))(nf, nc, args).cuda().train()
eng = StepEngine(m)
b = Batch.from_data_list(gl, pack=True).to("cuda")
perm = torch.randperm(len(gl), device="cuda")
eng.train_step(b, perm, adam=False)
torch.cuda.synchronize()
def timed(stop, reps=30, inner=10):
    _lib.lib().cal_engine_debug_stop(stop)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.train_step(b, perm, adam=False)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(inner): eng.train_step(b, perm, adam=False)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / inner * 1e6
nst = int(sys.argv[1]) if len(sys.argv) > 1 else 70
prev = 0.0
full = timed(0)
_lib.lib().cal_engine_debug_stop(0)
eng.train_step(b, perm, adam=False)                      # eager, untruncated: records the launch-site names
torch.cuda.synchronize()
names = [_lib.lib().cal_engine_stage_name(k).decode() for k in range(1, 200)]
names = [n for n in names if n]
nst = min(nst, len(names) + 1)
for k in range(1, nst):
    t = timed(k)
    print("after launch site %2d: %7.1f us  (+%5.1f)  %s" % (k, t, t - prev, names[k - 1] if k - 1 < len(names) else ""))
    if abs(t - full) < 1e-9 or (k > 5 and t >= full * 0.999 and t - prev < 0.05): pass
    prev = t
print("full step (no adam): %.1f us" % full)
