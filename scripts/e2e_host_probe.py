#!/usr/bin/env python
"""Where the host-collate leg of bench.py's end_to_end figure spends its time: the same loop (new DataLoader per epoch ->
Batch.to("cuda") -> CausalTrainer.step, eager) with wall-clock per phase (host side only: nothing synchronises inside the
loop), then the same under cProfile.   python scripts/e2e_host_probe.py [workload] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else bench.HEADLINE
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    wl = bench.WORKLOADS[name]
    margs = bench.model_args(wl)
    from cal_amd import model as M
    from cal_amd.data import DataLoader
    from cal_amd.trainer import CausalTrainer
    gs = bench.make_graphs(wl, 16 * wl["batch"], seed=4242)
    torch.manual_seed(1)
    model = getattr(M, wl["model"])(wl["nfeat"], wl["ncls"], margs).cuda()
    tr = CausalTrainer(model, margs, lr=1e-3, use_graph=False)

    def loader(epoch):
        return DataLoader(gs, wl["batch"], shuffle=True, generator=torch.Generator().manual_seed(epoch))

    for b in loader(0):
        tr.step(b.to("cuda"))
    torch.cuda.synchronize()

    def loop(phases):
        n, epoch = 0, 0
        while n < steps:
            epoch += 1
            t = time.perf_counter()
            it = iter(loader(epoch))
            phases["new_loader"] += time.perf_counter() - t
            while True:
                t = time.perf_counter()
                try:
                    b = next(it)
                except StopIteration:
                    break
                t1 = time.perf_counter()
                b = b.to("cuda")
                t2 = time.perf_counter()
                tr.step(b)
                t3 = time.perf_counter()
                phases["collate"] += t1 - t; phases["to"] += t2 - t1; phases["step"] += t3 - t2
                n += 1
                if n >= steps:
                    break
        return n

    for rep in range(2):
        ph = dict(new_loader=0.0, collate=0.0, to=0.0, step=0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = loop(ph)
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("rep %d: %d steps, %.3f ms/step wall (%.3f host-side), per step: %s" % (
            rep, n, 1e3 * dt / n, 1e3 * t_host / n, {k: round(1e3 * v / n, 4) for k, v in ph.items()}), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    loop(dict(new_loader=0.0, collate=0.0, to=0.0, step=0.0))
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)


if __name__ == "__main__":
    main()
