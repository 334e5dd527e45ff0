import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd import model as M
from cal_amd.engine import StepEngine
args = argparse.Namespace(layers=3, hidden=128, with_random=True, without_node_attention=False,
                          without_edge_attention=False, fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)
m = M.CausalGCN(10, 4, args).cuda()
eng = StepEngine(m)
def graph_time(fn, n):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 20 * 1e6
for n in (10, 100, 400):
    t = graph_time(eng.adam, n)
    print("graph of %d x (k_adam[139k] + k_adam_tick): %.1f us -> %.2f us per kernel" % (n, t, t / (2 * n)))
