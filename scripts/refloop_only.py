import sys, os, json, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
wl = bench.WORKLOADS[bench.HEADLINE]
margs = bench.model_args(wl)
gs = list(bench.make_graphs(wl, 16 * wl["batch"], seed=4242)) + list(bench.make_graphs(wl, 5596 - 2048, seed=4243))
if len(sys.argv) > 1:
    cProfile.run("r = bench.reference_loop(wl, margs, gs, 160)", "/tmp/p.out")
    pstats.Stats("/tmp/p.out").sort_stats("tottime").print_stats(16)
else:
    r = bench.reference_loop(wl, margs, gs, int(os.environ.get("STEPS", "160")))
print(json.dumps({k: (round(v["graphs_per_s"]), round(v["ms_per_step"], 3)) for k, v in r.items() if isinstance(v, dict)}))
