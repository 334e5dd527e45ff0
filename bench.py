#!/usr/bin/env python
"""bench.py -- graphs/sec of the CAL causal train step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (any N: for N > 1 without a launcher it re-execs itself under
                                                            torch.distributed.run, one rank per GPU, 127.0.0.1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1, launcher given)

A "step" = forward (3 heads) + 0.5*KL + 1.0*NLL + 0.5*NLL + backward + Adam
(train_causal.py:173-192) on one pre-collated, HBM-resident mini-batch of
synthetic SPMotif graphs (BASELINE.json configs[1]: CausalGCN, b=0.9, 3 layers,
hidden 128, batch 128, node_num 7 -> ~57 nodes / graph).  Batches shard across
ranks (independent mini-batches per GPU, weak scaling) with one RCCL all-reduce
of the flat gradient bucket per step.

Prints ONE JSON line on rank 0 (see DESIGN.md "measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HEADLINE = "spmotif_b0.9_causalgcn_h128_l3_bs128"
WORKLOADS = {
    # BASELINE.json configs[1] (the headline) first; the others are SURVEY.md 8d's configs 3-5 on synthetic
    # stand-ins (cal_amd/synth.py), selectable with --workload -- the default run measures the headline only
    "spmotif_b0.9_causalgcn_h128_l3_bs128": dict(model="CausalGCN", data="spmotif", node_num=7, hidden=128, layers=3, batch=128, nfeat=10, ncls=4),
    # the reference's DEFAULT SPMotif shape (opts.py:18 node_num = 15, ~235-node graphs) at BASELINE.json configs[0]'s batch of 32
    "spmotif_b0.9_causalgcn_nodenum15_bs32": dict(model="CausalGCN", data="spmotif", node_num=15, hidden=128, layers=3, batch=32, nfeat=10, ncls=4),
    # ... and the reference's defaults TOGETHER (opts.py:18,29: node_num 15 and batch_size 128, ~30 k node rows): what main_syn.py runs
    "spmotif_b0.9_causalgcn_nodenum15_bs128": dict(model="CausalGCN", data="spmotif", node_num=15, hidden=128, layers=3, batch=128, nfeat=10, ncls=4),
    "spmotif_b0.9_causalgat_h128_l3_bs128": dict(model="CausalGAT", data="spmotif", node_num=7, hidden=128, layers=3, batch=128, nfeat=10, ncls=4),
    "spmotif_b0.9_causalgin_h128_l3_bs128": dict(model="CausalGIN", data="spmotif", node_num=7, hidden=128, layers=3, batch=128, nfeat=10, ncls=4),
    "mutaglike_causalgat_h128_l3_bs64": dict(model="CausalGAT", data="mutag", node_num=18, hidden=128, layers=3, batch=64, nfeat=109, ncls=2),
    "nci1like_causalgcn_h128_l3_bs512": dict(model="CausalGCN", data="nci1", node_num=30, hidden=128, layers=3, batch=512, nfeat=139, ncls=2),
    "ba5000_causalgat_h256_l3_bs32": dict(model="CausalGAT", data="ba", node_num=5000, hidden=256, layers=3, batch=32, nfeat=10, ncls=4),
    "ba5000_causalgcn_h256_l3_bs32": dict(model="CausalGCN", data="ba", node_num=5000, hidden=256, layers=3, batch=32, nfeat=10, ncls=4),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="spmotif_b0.9_causalgcn_h128_l3_bs128", choices=list(WORKLOADS))
    ap.add_argument("--mode", default="graph", choices=["graph", "eager"])
    ap.add_argument("--batches", type=int, default=8, help="distinct resident mini-batches per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end (batch assembly included) figure")
    ap.add_argument("--no-sequence", action="store_true", help="one hipGraph launch per step instead of one per pass over the resident batches")
    ap.add_argument("--no-engine", action="store_true", help="operator-level autograd path instead of the native step engine")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; value = median (BASELINE.md section 3)")
    ap.add_argument("--dry", action="store_true", help="launch plumbing only: rendezvous, one collective, the JSON line's shape with value null (no model, no timed region; runs without a GPU over CAL_BENCH_BACKEND=gloo)")
    return ap.parse_args()


def _free_port() -> int:
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a) -> None:
    """`python bench.py --gpus N` (N > 1) without a launcher: re-exec the same command line under torch.distributed.run, one
    rank per GPU, rendezvous on 127.0.0.1 -- the form the driver uses for N = 1 then works for N = 2/4/8 unmodified.  Never
    returns.  Fewer than N visible devices is an error here, before any rank starts (ranks would otherwise share a device
    and RCCL would hang or fail late); CAL_BENCH_BACKEND=gloo lifts the check (test aid: ranks may share devices)."""
    backend = os.environ.get("CAL_BENCH_BACKEND", "nccl")
    if backend == "nccl" and not a.dry:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (one rank per GPU; HIP_VISIBLE_DEVICES=%r)"
                             % (a.gpus, have, os.environ.get("HIP_VISIBLE_DEVICES")))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush(); sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def model_args(wl):
    return argparse.Namespace(layers=wl["layers"], hidden=wl["hidden"], with_random=True,
                              without_node_attention=False, without_edge_attention=False,
                              fc_num="222", cat_or_add="add", c=0.5, o=1.0, co=0.5)


def make_graphs(wl, n, seed):
    from cal_amd import spmotif, synth
    if wl["data"] == "spmotif":
        return spmotif.train_mix(n, bias=0.9, node_num=wl["node_num"], seed=seed)
    if wl["data"] == "ba":
        return synth.ba_graphs(n, n=wl["node_num"], seed=seed)
    return synth.tu_like(n, kind=wl["data"], seed=seed)


def make_batches(wl, nb, seed):
    from cal_amd.data import Batch
    gs = make_graphs(wl, nb * wl["batch"], seed)
    # (batches of small graphs are ordered for the engine's 64-node tiles, as DeviceLoader orders them: a mini-batch is a set)
    return [Batch.from_data_list(gs[i * wl["batch"]:(i + 1) * wl["batch"]], pack=True) for i in range(nb)]


def cpu_baseline(wl, batches_cpu, seconds):
    """Restated reference CPU path (oracle/cal_oracle.py, kind 'port') on this host's cores.
    torch's intra-op thread count is picked by a calibration (median of 2 x 10 timed steps per candidate): the
    unfused op sequence on ~7k-row tensors does not scale to hundreds of threads, and the best
    setting is what a user of the reference would run."""
    from oracle import cal_oracle as O
    cores = os.cpu_count() or 1
    torch.manual_seed(666)
    sd = O.init_state(wl["model"], wl["nfeat"], wl["ncls"], hidden=wl["hidden"], layers=wl["layers"], heads=4)
    tr = O.CpuTrainer(wl["model"], sd, wl["ncls"], lr=1e-3, layers=wl["layers"], heads=4)
    nb = len(batches_cpu)

    def one(i):
        b = batches_cpu[i % nb]
        perm = torch.tensor(O.intervention_perm(b.num_graphs, True, True, wl["model"]))
        tr.step((b.x if b.x is not None else b.feat), b.edge_index, b.batch, b.y, perm=perm)
        return b.num_graphs

    # Calibration (round-5 review item 6: the baseline moved 40 % between identical boxes because a candidate was judged on 3
    # steps): every candidate thread count is timed over >= 10 steps per pass (one untimed step first; fewer only when a step
    # is so slow that the per-candidate budget runs out), in TWO passes over the candidates, and judged by the MEDIAN step
    # time of both passes together.  The candidates start at min(cores, 16) threads and walk down (8, 4, 1) and up (32, 64,
    # 128) while a direction stays within 25 % of the best median so far.
    mid = min(cores, 16)
    samples, calib = {}, {}
    budget = max(2.0, 0.15 * seconds)         # per candidate and pass: cap on the calibration's own cost

    def measure(t):
        torch.set_num_threads(t)
        one(0)
        t0 = time.perf_counter()
        k = 0
        while k < 10:
            t1 = time.perf_counter()
            one(k + 1)
            samples.setdefault(t, []).append(time.perf_counter() - t1)
            k += 1
            if k >= 2 and time.perf_counter() - t0 > budget:
                break
        return float(np.median(samples[t]))

    best_t, best_dt = mid, measure(mid)
    tried = [mid]
    for direction in ([t for t in (8, 4, 1) if t < mid], [t for t in (32, 64, 128) if mid < t <= cores]):
        for t in direction:
            dt = measure(t)
            tried.append(t)
            if dt < best_dt:
                best_t, best_dt = t, dt
            elif dt > 1.25 * best_dt:
                break
    for t in tried:                            # second pass over the same candidates, other order of cache / clock state
        measure(t)
    med = {t: float(np.median(samples[t])) for t in tried}
    best_t = min(med, key=med.get)
    calib = {t: round(1e3 * med[t], 1) for t in tried}
    calib = dict(sorted(calib.items()))
    torch.set_num_threads(best_t)
    t0 = time.perf_counter()
    n, steps = 0, 0
    while (time.perf_counter() - t0 < seconds and steps < 2000) or steps < 3:
        n += one(steps)
        steps += 1
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="graphs/s", cores=best_t, kind="port", host_cores=cores,
                sample="%d train steps of batch %d (same synthetic batches, %.1f s) through "
                       "oracle/cal_oracle.py (unfused restatement of the PyG path); torch threads=%d "
                       "chosen by calibration (median step time of 2 passes x <= 10 steps per candidate) %s ms/step" % (steps, wl["batch"], dt, best_t, calib),
                ms_per_step=1e3 * dt / steps)


def spmm_roofline(trainer, batches, wl, iters=20):
    """Live HIP-event timing of the dominant graph kernel (k_spmm, the CSR aggregation of
    gcn_conv.py:92-104) as launched inside the step: an eager replica of the step is run with an
    event pair around every cal_spmm_fwd launch on the launch stream."""
    from cal_amd import _lib
    events = []
    orig = _lib.call

    def timed(name, *a):
        if name == "cal_spmm_fwd":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            orig(name, *a)
            e1.record()
            events.append((e0, e1, a[10], a[11]))      # N, H
        else:
            orig(name, *a)

    import cal_amd.ops as ops_mod
    import cal_amd.plan as plan_mod
    ops_mod._lib.call = timed
    try:
        algo = []
        for i in range(iters):
            b = batches[i % len(batches)]
            perm = torch.arange(b.num_graphs, device="cuda")
            n_before = len(events)
            trainer._fwd_bwd(b, perm, trainer.stats)
            E_nsl = int((b.edge_index[0] != b.edge_index[1]).sum().item())
            for (_, _, N, H) in events[n_before:]:
                algo.append(2 * N * H * 4 + (E_nsl + N) * 8 + (N + 1) * 4)
        torch.cuda.synchronize()
    finally:
        ops_mod._lib.call = orig
    durs = np.array([e0.elapsed_time(e1) * 1e-3 for e0, e1, _, _ in events])   # seconds
    algo = np.array(algo, dtype=np.float64)
    # drop the first step's launches (cold instruction cache)
    k = len(events) // iters
    durs, algo = durs[k:], algo[k:]
    avg_dur = float(durs.mean())
    achieved = float(algo.mean() / avg_dur / 1e9)
    peak = 8000.0
    return dict(bound="hbm", kernel="k_spmm (cal_spmm_fwd)", achieved=achieved, peak=peak, unit="GB/s",
                frac=achieved / peak, traffic=None, avg_launch_us=avg_dur * 1e6,
                algorithmic_bytes_per_launch=float(algo.mean()), launches_per_step=k,
                note="event pairs include ~launch gap; working set is cache-resident at this config "
                     "(launch/latency-bound by construction, SURVEY.md 8d)")


def engine_roofline(trainer, batches, workload, iters=20):
    """Live HIP-event timing (on the launch stream) of the two kernels the north star names, as
    launched inside the native step: the dense [N,H]x[H,H] MFMA GEMM of every backbone layer
    (dominant by time -> primary roofline, bound = mfma) and the CSR aggregation k_espmm (bound =
    hbm).  Algorithmic work per launch: 2*N*H*H flops, resp. 2*N*H*4 + E'*8 + (N+1)*4 bytes
    (SURVEY.md section 8d).  The step runs eagerly with the engine's event hooks enabled."""
    import ctypes
    from cal_amd import _lib
    h = _lib.lib()
    eng = trainer.engine
    for i in range(3):
        eng.train_step(batches[i % len(batches)], None, adam=True)
    torch.cuda.synchronize()
    h.cal_engine_profile(1)
    for i in range(iters):
        eng.train_step(batches[i % len(batches)], None, adam=True)
    torch.cuda.synchronize()
    h.cal_engine_profile(0)
    cap = 64 * iters
    buf = (ctypes.c_double * (3 * cap))()
    n = int(h.cal_engine_profile_read(buf, cap))
    rec = np.array(buf[:3 * n], dtype=np.float64).reshape(n, 3)
    out = {}
    for cls, key in ((0, "gemm"), (1, "spmm"), (2, "gconv"), (3, "dual"), (4, "gconv_bwd"), (5, "gat_fwd"), (6, "gat_bwd"), (7, "ggat"), (8, "ggat_bwd"),
                     (9, "ggin"), (10, "ggin_bwd")):
        r = rec[(rec[:, 0] == cls) & (rec[:, 1] > 0)]
        if len(r) == 0:
            continue
        dur = r[:, 1].mean() * 1e-3       # s
        work = r[:, 2].mean()
        out[key] = (dur, work, len(r) / iters)
    roof = {}
    # HBM-side traffic per launch from the PMC passes (separate rocprofv3 --pmc runs of this command;
    # committed under profiles/): null when the summary is absent
    from cal_amd.build import kernel_source_sha
    src_sha = kernel_source_sha()
    pmc, pmc_file, pmc_stale = {}, None, None
    for cand in ("pmc_traffic_%s.json" % workload, "pmc_traffic.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", cand)))
            if pmc.get("workload", HEADLINE) == workload:
                pmc_file = "profiles/" + cand
                break
            pmc = {}                                       # the counters were collected on another workload
        except Exception:
            pmc = {}
    if pmc and pmc.get("kernel_source_sha") != src_sha:
        # the PMC passes are separate rocprofv3 runs (scripts/profile_round.sh): a summary collected on other kernel
        # sources says nothing about this run -- traffic is null until the passes are repeated
        pmc_stale = pmc.get("kernel_source_sha", "unstamped")
        pmc = {}
    traffic_src = {"file": pmc_file, "kernel_source_sha": src_sha, "stale_summary_sha": pmc_stale}
    if workload == HEADLINE:
        note = ("config-2 working set (3.7 MB activations) is cache-resident and every launch is a few hundred "
                "workgroups of one ~10 us dependent chain: the step is latency-bound by construction (SURVEY.md 8d); "
                "events are attached to the dispatch (hipExtLaunchKernelGGL start/stop) in an eager replica of the step")
    else:
        note = ("single-kernel classes: events attached to the dispatch (hipExtLaunchKernelGGL start/stop), eager step" if pmc else
                "single-kernel classes: events attached to the dispatch, eager step; no PMC pass was collected for this workload (traffic null)")
    mfma = {
        "gemm": ("k_gemm", "k_gemm / k_wres<NN, BN prologue> [N,H]x[H,H] fp32 MFMA 32x32x2 (backbone layers, unfused path; the weight-resident kernel of gemm_wres.hip from 16k rows at H = 128 / 256)", "k_gemm_backbone"),
        "gconv": ("k_gconv_fwd", "k_gconv_fwd: per-graph fused BN + [n,H]x[H,64] MFMA GEMM + dense-block aggregation MFMA + "
                                 "bias/ReLU/BN statistics (backbone layers, forward)", "k_gconv_fwd"),
        "gconv_bwd": ("k_gconv_bwd", "k_gconv_bwd: per-graph fused backward -- dense-block transposed aggregation + dX' = dz W^T (BN-backward "
                                     "sums) + dW = x'^T dz, all on MFMA (backbone layers, backward)", "k_gconv_bwd"),
        "ggat": ("k_ggat_fwd", "k_ggat_fwd: per-graph fused GATConv forward -- BN + [n,H]x[H,64] MFMA GEMM + scores + edge softmax (dropout) + "
                               "dense attention-block aggregation MFMA + bias/ReLU/BN statistics (backbone layers, forward)", "k_ggat_fwd"),
        "ggat_bwd": ("k_ggat_bwd", "k_ggat_bwd: per-graph fused GATConv backward -- attention backward on two dense blocks (MFMA) + dX' = dz W^T "
                                   "(BN-backward sums) + dW = x'^T dz + d att (backbone layers, backward)", "k_ggat_bwd"),
        "ggin": ("k_ggin_fwd", "k_ggin_fwd<1>: per-graph fused GINConv first half -- [n,H]x[H,64] MFMA GEMM (W1^T slice) + unit dense-block aggregation "
                               "MFMA + bias + BN statistics (backbone layers, forward)", "k_ggin_fwd"),
        "ggin_bwd": ("k_ggin_bwd", "k_ggin_bwd<1>: per-graph fused GINConv backward, first half -- BatchNorm backward + transposed unit aggregation + "
                                   "d h = dz W1 + dW1 = dz^T h, all on MFMA (backbone layers, backward)", "k_ggin_bwd"),
        "dual": ("k_gemm_dual", "k_gemm_dual (one grid) / k_wres<NT, dot sums> + k_tn (two launches, timed as one class): dX = dZ W^T with the BN-backward sums "
                                "+ dW = BN(h)^T dZ over node ranges (backbone layers, backward)", "k_gemm_dual"),
    }
    if WORKLOADS.get(workload, {}).get("node_num") == 15:
        # graphs of 129-256 nodes (the reference's default SPMotif shape): the same two classes are the WIDE per-graph kernels
        mfma["gconv"] = ("k_gconv_fwd", "k_gw_fwd (engine_gwide.hpp): per-graph fused GCNConv for 129-256-node graphs -- BN + [n,H]x[H,64|32] MFMA product "
                                        "with the node operand straight from global memory + SPARSE aggregation from the LDS-resident z slice + "
                                        "bias/ReLU/BN statistics (backbone layers, forward)", "k_gw_fwd")
        mfma["gconv_bwd"] = ("k_gconv_bwd", "k_gw_bwd (engine_gwide.hpp): its backward -- sparse transposed aggregation from LDS, dX' = dz W^T (BN-backward sums) "
                                            "and dW = x'^T dz on MFMA, split over two workgroups per (graph, slice) (backbone layers, backward)", "k_gw_bwd")
    # primary roofline = the MFMA kernel class that takes the most time per step
    best = None
    for key in mfma:
        if key in out and (best is None or out[key][0] * out[key][2] > out[best][0] * out[best][2]):
            best = key
    for key in mfma:
        if key not in out:
            continue
        dur, work, per_step = out[key]
        ach = work / dur / 1e12
        # the PMC summary names the instantiation that ran: 64x64 tiles, 128x128 tiles or the weight-resident kernels (by size)
        alts = {"k_gemm_backbone": (("k_wres_fwd",), ("k_gemm_big",), ("k_gemm_backbone",)),
                "k_gemm_dual": (("k_wres_nt", "k_tn"), ("k_gemm_big_dual",), ("k_gemm_dual",))}.get(mfma[key][2], ((mfma[key][2],),))
        traffic = None
        for names in alts:
            vals = [pmc.get(nm, {}).get("bytes_per_launch") for nm in names]
            if all(v is not None for v in vals):
                traffic = int(sum(vals))
                break
        d = dict(bound="mfma", kernel=mfma[key][1], achieved=ach, peak=157.3, unit="TFLOP/s", frac=ach / 157.3,
                 traffic=traffic,
                 avg_launch_us=dur * 1e6,
                 algorithmic_flops_per_launch=work, timed_launches_per_step=per_step, note=note, traffic_source=traffic_src)
        roof["roofline" if key == best else "roofline_" + mfma[key][0]] = d
    if "spmm" in out:
        dur, work, per_step = out["spmm"]
        ach = work / dur / 1e9
        roof["roofline_aggregation"] = dict(bound="hbm", kernel="k_espmm (CSR aggregation; transposed, backward of the backbone layers)",
                                            achieved=ach, peak=8000.0, unit="GB/s", frac=ach / 8000.0,
                                            traffic=pmc.get("k_espmm", {}).get("bytes_per_launch"),
                                            avg_launch_us=dur * 1e6, algorithmic_bytes_per_launch=work,
                                            timed_launches_per_step=per_step)
    # GATConv layers (CausalGAT): scores + edge softmax + aggregation (forward), the five backward kernels
    # algorithmic bytes: z and the output once each, CSR slot + 3 logits-sized [E',K] passes (+3*E'*K*4)
    def _sum_traffic(keys):
        vals = [pmc.get(k, {}).get("bytes_per_launch") for k in keys]
        return int(sum(vals)) if vals and all(v is not None for v in vals) else None
    gat_traffic = {"gat_fwd": _sum_traffic(["k_gat_fwd"]), "gat_bwd": _sum_traffic(["k_gat_bwd_dst", "k_gat_bwd_src"])}
    for key, label in (("gat_fwd", "GATConv forward: k_gat_fwd_w at H = 256 / 4 heads (one wave per row, slots in lanes), else k_gat_fwd_fused -- scores + online edge softmax + dropout + aggregation + bias + ReLU, one pass"),
                       ("gat_bwd", "GATConv backward: k_gat_bwd_dst_w / k_gat_bwd_src_w at H = 256 / 4 heads (else k_gat_bwd_dst_c / k_gat_bwd_src) + k_gat_datt_part (alpha recomputed)")):
        if key in out:
            dur, work, per_step = out[key]
            ach = work / dur / 1e9
            roof["roofline_" + key] = dict(bound="hbm", kernel=label, achieved=ach, peak=8000.0, unit="GB/s", frac=ach / 8000.0,
                                           traffic=gat_traffic.get(key), avg_launch_us=dur * 1e6, algorithmic_bytes_per_launch=work,
                                           timed_launches_per_step=per_step)
    return roof


def end_to_end(wl, margs, steps=160):
    """Secondary figure: graphs/s of real epochs INCLUDING batch assembly --
    shuffled permutation, on-device collate of a device-resident dataset (cal_collate), eager
    engine step (shapes change every step, so no graph replay) -- next to the same loop fed by the
    host-side Python collate the reference's DataLoader does."""
    from cal_amd import model as M
    from cal_amd.data import DataLoader
    from cal_amd.device_data import DeviceDataset, DeviceLoader
    from cal_amd.trainer import CausalTrainer
    gs = make_graphs(wl, 16 * wl["batch"], seed=4242)
    res = {}
    for kind in ("device_collate", "host_collate"):
        torch.manual_seed(1)
        model = getattr(M, wl["model"])(wl["nfeat"], wl["ncls"], margs).cuda()
        tr = CausalTrainer(model, margs, lr=1e-3, use_graph=False)
        ds = DeviceDataset(gs) if kind == "device_collate" else None

        # ONE loader per leg, iterated epoch after epoch (train_causal.py:13-15 builds its loaders once, before the epoch loop);
        # the cyclic garbage collector is off inside the timed region as in reference_loop below and in Python's own timeit: a
        # full collection of this process is ~80 ms, i.e. MORE than the 60 steps this leg used to time -- that, plus a per-graph
        # cache validation for every new loader object, was the "regression" of the round-5 line (308 k -> 71 k graphs/s)
        g = torch.Generator().manual_seed(7)
        loader = DeviceLoader(ds, wl["batch"], shuffle=True, generator=g) if ds is not None else DataLoader(gs, wl["batch"], shuffle=True, generator=g)
        n_steps, n_graphs = 0, 0
        for b in loader:                         # warm-up epoch (workspace sizing, code paths, pinned staging ring)
            tr.step(b if ds is not None else b.to("cuda"))
        torch.cuda.synchronize()
        import gc
        gc.collect()
        gc.disable()
        try:
            t0 = time.perf_counter()
            while n_steps < steps:
                for b in loader:
                    tr.step(b if ds is not None else b.to("cuda"))
                    n_steps += 1
                    n_graphs += b.num_graphs
                    if n_steps >= steps:
                        break
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        res[kind] = {"graphs_per_s": n_graphs / dt, "ms_per_step": 1e3 * dt / n_steps, "steps": n_steps}
    res["note"] = "epochs over a %d-graph dataset, one shuffling loader per leg, eager engine step; includes batch assembly; gc off inside the timed regions" % len(gs)
    gs = list(gs)
    try:
        # dataset size of the reference-shaped loop: the reference's SPMotif b = 0.9 train split is 4 x (1260 + 139) = 5596
        # graphs (utils.py:133-150), 44 mini-batches of 128 per epoch -- train_causal_epoch returns host floats, i.e. every
        # epoch ends with a pipeline drain, so the epoch length is part of the loop's throughput (16 mini-batches per epoch,
        # the figure of the earlier rounds' lines, cost 3 %); the stand-in workloads keep 16 mini-batches per epoch
        if wl["data"] == "spmotif" and wl["batch"] == 128:
            gs = gs + make_graphs(wl, 5596 - len(gs), seed=4243)
        res["reference_loop"] = reference_loop(wl, margs, gs, max(steps, 160))
    except Exception as exc:
        res["reference_loop"] = {"error": repr(exc)}
    return res


def reference_loop(wl, margs, gs, steps=60):
    """The loop the reference's entry scripts run, through the reference-named surface: ``train_causal_epoch(model, optimizer,
    loader, device, args)`` (train_causal.py:162-200) with an ``Adam`` object and a cosine schedule stepped once per epoch
    (train_causal.py:21-29), shuffled epochs over the same dataset as end_to_end.  Four feeds:
      fused_device_loader  what ``train_causal_syn`` builds on a GPU: device-resident dataset + one-call fused step
      fused_device_loader_device_perm   the same with ``args.device_perm`` (the permutation of model.py:147-152 drawn inside the
                           step's first kernel instead of by Python's RNG + a pinned H2D copy)
      fused_host_loader    a caller's own host DataLoader (Python collate + H2D per batch) in front of the fused step
      module_surface       ``--no_fused_step``: model(data) -> torch loss -> backward -> optimizer.step(), statement by
                           statement on the nn.Module surface (what a foreign loop gets), with a host read-back of the
                           loss every iteration like train_causal.py:186-191"""
    import copy
    from torch.optim.lr_scheduler import CosineAnnealingLR
    from cal_amd import model as M
    from cal_amd.data import DataLoader
    from cal_amd.device_data import DeviceDataset, DeviceLoader
    from cal_amd.optim import EngineAdam
    from cal_amd.train_causal import causal_loss, train_causal_epoch
    dev = torch.device("cuda")
    out = {}
    for kind in ("fused_device_loader", "fused_device_loader_device_perm", "fused_host_loader", "module_surface"):
        args = copy.copy(margs)
        args.no_fused_step = kind == "module_surface"
        args.device_perm = kind.endswith("device_perm")     # (not a reference flag: the intervention permutation drawn on the GPU)
        torch.manual_seed(1)
        random.seed(1)
        model = getattr(M, wl["model"])(wl["nfeat"], wl["ncls"], args).cuda()
        opt = EngineAdam(model.parameters(), lr=1e-3)
        sched = CosineAnnealingLR(opt, T_max=100, eta_min=1e-6)
        if kind.startswith("fused_device_loader"):
            loader = DeviceLoader(DeviceDataset(gs), wl["batch"], shuffle=True)
        else:
            loader = DataLoader(gs, wl["batch"], shuffle=True)
        per_epoch = len(loader)
        n_epochs = max(1, -(-steps // per_epoch))
        if kind == "module_surface":
            def epoch():
                model.train()
                tot = 0.0
                for data in loader:
                    opt.zero_grad()
                    data = data.to(dev)
                    c, o, co = model(data, eval_random=args.with_random)
                    loss, lc, lo, lco = causal_loss(c, o, co, data.y, model.num_classes, args)
                    loss.backward()
                    tot += loss.item() * data.num_graphs            # the reference's per-iteration host sync
                    opt.step()
                return tot / len(gs)
        else:
            def epoch():
                return train_causal_epoch(model, opt, loader, dev, args)[0]
        epoch()                                                   # warm-up epoch (workspace sizing, engine creation)
        sched.step()
        torch.cuda.synchronize()
        # (the cyclic garbage collector is off inside the timed region, as in Python's own timeit: one full collection of a
        #  torch process is ~80 ms -- several hundred steps -- and when it fires depends on allocation counts, not on the loop)
        import gc
        gc.collect()
        gc.disable()
        try:
            t0 = time.perf_counter()
            last = None
            for _ in range(n_epochs):
                last = epoch()
                sched.step()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        finally:
            gc.enable()
        n_steps = n_epochs * per_epoch
        out[kind] = {"graphs_per_s": n_epochs * len(gs) / dt, "ms_per_step": 1e3 * dt / n_steps, "steps": n_steps,
                     "epoch_loss": float(last), "fused": bool(getattr(opt, "_cal_binding", None) is not None and kind != "module_surface")}
    out["note"] = ("train_causal_epoch (train_causal.py:162-200) over shuffled epochs of a %d-graph dataset, Adam object + " % len(gs) +
                   "cosine schedule; intervention permutation from Python's RNG on the host as in model.py:147-152; host loader = "
                   "DataLoader over a list (vectorised collate into pinned staging, cal_collate_host); gc off inside the timed regions")
    return out


def dry_run(a, wl, world, rank, local_rank):
    """--dry: everything the launch needs and nothing the measurement does -- process group up (RCCL when GPUs are there, else
    CAL_BENCH_BACKEND=gloo), one all-reduce so that every rank is seen, rank 0 prints the line with value null."""
    import torch.distributed as dist
    backend = os.environ.get("CAL_BENCH_BACKEND", "nccl")
    dev = torch.device("cpu")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dev = torch.device("cuda", local_rank % torch.cuda.device_count())
    seen = 1
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        t = torch.ones(1, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        seen = int(t.item())
    out = {"metric": "graphs/sec (train step) on SPMotif b=0.9 batch=128" if wl["data"] == "spmotif" else
                     "graphs/sec (train step) on %s batch=%d" % (a.workload, wl["batch"]),
           "value": None, "unit": "graphs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": None,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "dry": True,
           "config": {"workload": a.workload, "model": wl["model"], "batch_per_gpu": wl["batch"], "global_batch": wl["batch"] * world,
                      "parallelism": "dp%d" % world},
           "data_parallel": {"backend": backend if world > 1 else None, "rccl_ranks_seen": seen}}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    a = parse()
    wl = WORKLOADS[a.workload]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if "WORLD_SIZE" not in os.environ and a.gpus > 1:
            self_launch(a)              # re-exec under torch.distributed.run; does not return
        raise SystemExit("bench.py --gpus %d inside a launcher with WORLD_SIZE=%d: the two must agree" % (a.gpus, world))
    if a.dry:
        return dry_run(a, wl, world, rank, local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # CAL_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with fewer GPUs than
    # ranks (ranks share devices; test aid only -- the driver's runs use nccl = RCCL, one GPU per rank)
    backend = os.environ.get("CAL_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from cal_amd import _lib
    from cal_amd import model as M
    from cal_amd.trainer import CausalTrainer
    _lib.lib()      # fail loudly if the HIP extension is missing

    seed = 666
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed + rank)
    margs = model_args(wl)
    model = getattr(M, wl["model"])(wl["nfeat"], wl["ncls"], margs).cuda()
    if world > 1:   # identical replicas
        for p in model.parameters():
            dist.broadcast(p.data, 0)
        for bname, buf in model.named_buffers():
            if buf.dtype.is_floating_point:
                dist.broadcast(buf, 0)
    batches_cpu = make_batches(wl, a.batches, seed=seed + 1000 * rank)
    batches = [b.to("cuda") for b in make_batches(wl, a.batches, seed=seed + 1000 * rank)]
    mode = a.mode
    use_engine = False if a.no_engine else None
    trainer = CausalTrainer(model, margs, lr=1e-3, use_graph=(mode == "graph"), world_size=world, use_engine=use_engine)
    trainer.reserve_for(batches)
    if mode == "graph":
        try:
            for b in batches:
                trainer.prepare(b)
        except Exception as exc:   # capture unsupported -> measured eagerly, and said so
            sys.stderr.write("graph capture failed (%r); falling back to eager launches\n" % (exc,))
            mode = "eager"
            model = getattr(M, wl["model"])(wl["nfeat"], wl["ncls"], margs).cuda()
            trainer = CausalTrainer(model, margs, lr=1e-3, use_graph=False, world_size=world, use_engine=use_engine)
            trainer.reserve_for(batches)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    nb = len(batches)
    # Every timed region (and the warm-up) walks the resident batches cyclically FROM batch 0.  On one GPU a pass over the nb
    # batches shares one hipGraph launch (CausalTrainer.step_sequence); the remainder of a region -- K mod nb steps -- is one
    # more sequence graph over the first K mod nb batches, captured before anything is timed, so a region of ANY length runs
    # whole sequence graphs (round-5 review item 6: with --steps 20 --warmup 5 four of every 20 steps used to fall off the
    # 8-step graph).  K steps are K full train steps either way.
    seq = mode == "graph" and not a.no_sequence and trainer.can_sequence() and nb > 1

    def run(count):
        stats, i = None, 0
        while count > 0:
            k = min(nb, count) if seq else 1
            if k > 1:
                stats = trainer.step_sequence(batches[:k])
            else:
                stats = trainer.step(batches[i % nb])
            i += k; count -= k
        return stats

    if seq:
        for k in sorted({nb, a.steps % nb, a.warmup % nb}, reverse=True):
            if k > 1:
                trainer.step_sequence(batches[:k])          # capture (and one run) before anything is timed
    run(a.warmup)
    # --repeats timed regions of EXACTLY --steps steps each, every one bracketed by barrier + synchronize and reduced
    # with MAX over ranks; value / ms_per_step come from the MEDIAN region (median of 5 repeats).
    # A HIP event pair on the launch stream brackets the same steps from the inside: ms_per_step_device is the region
    # without the host's barrier + synchronize + first-launch latency (reported next to, never instead of, ms_per_step).
    regions, local_regions, dev_regions = [], [], []
    for _ in range(max(1, a.repeats)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        t0 = time.perf_counter()
        e0.record()
        stats = run(a.steps)
        e1.record()
        barrier()
        dt = time.perf_counter() - t0
        local_regions.append(dt)
        dev_regions.append(e0.elapsed_time(e1) * 1e-3)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        regions.append(dt)
    dt = float(np.median(regions))
    # N > 1 diagnostics (the driver's 8-GPU run is the only multi-GPU measurement there is): per-rank step time of the
    # median region's policy, and the gradient exchange alone -- 20 eager all-reduces of the flat bucket between events
    dp_diag = None
    if world > 1:
        local = torch.tensor([float(np.median(local_regions))], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(allr, local)
        for _ in range(3):
            dist.all_reduce(trainer.flat_g)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        saved = trainer.flat_g.clone()
        e0.record()
        for _ in range(20):
            dist.all_reduce(trainer.flat_g)
        e1.record()
        torch.cuda.synchronize()
        trainer.flat_g.copy_(saved)
        dp_diag = {"exchange": "one-shot peer-memory kernel (CAL_AMD_P2P_EXCHANGE=1)" if getattr(trainer, "p2p", None) is not None else "all-reduce (%s)" % dist.get_backend(),
                   "exchange_in_graph": bool(trainer.exchange_in_graph), "fused_opt": bool(trainer.fused_opt),
                   "sequence_graph": bool(seq), "backend": dist.get_backend(), "rccl_ranks_seen": dist.get_world_size(),
                   "bucket_bytes": int(trainer.flat_g.numel() * 4),
                   "allreduce_us_eager": 1e3 * e0.elapsed_time(e1) / 20,
                   "ms_per_step_by_rank": [round(1e3 * float(t.item()) / a.steps, 5) for t in allr]}
    trainer.check_status()
    final = stats.tolist()
    graphs = wl["batch"] * a.steps * world
    nodes = float(np.mean([b.batch.numel() for b in batches]))
    edges = float(np.mean([b.edge_index.size(1) for b in batches]))

    out = {
        "metric": "graphs/sec (train step) on SPMotif b=0.9 batch=128" if wl["data"] == "spmotif" else
                  "graphs/sec (train step) on %s batch=%d" % (a.workload, wl["batch"]),
        "value": graphs / dt, "unit": "graphs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "timed_region_ms": 1e3 * dt,
        "ms_per_step_device": 1e3 * float(np.median(dev_regions)) / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": a.workload, "model": wl["model"], "batch_per_gpu": wl["batch"],
                   "global_batch": wl["batch"] * world, "hidden": wl["hidden"], "layers": wl["layers"],
                   "node_num": wl["node_num"], "mean_nodes_per_batch": nodes, "mean_edges_per_batch": edges,
                   "launch": mode + ("(%d steps per graph launch, remainder %d)" % (nb, a.steps % nb) if seq else ""), "path": "native step engine" if trainer.engine is not None else "operator-level autograd", "parallelism": "dp%d" % world, "resident_batches": nb,
                   "final_loss": final[0],
                   "repeats": {"n": len(regions), "ms_per_step": [round(1e3 * r / a.steps, 5) for r in regions], "pick": "median"}},
    }
    if dp_diag is not None:
        out["data_parallel"] = dp_diag
    # rank 0 also at N > 1 (after the timed regions; the other ranks wait at the barrier below): a SCALE line then carries
    # the same roofline / cpu_baseline blocks as the N = 1 line and can be cross-checked against it (round-4 review)
    if rank == 0:
        if not a.no_roofline:
            try:
                if trainer.engine is not None:
                    out.update(engine_roofline(trainer, batches, a.workload))
                else:
                    out["roofline"] = spmm_roofline(trainer, batches, wl)
            except Exception as exc:
                out["roofline"] = {"error": repr(exc)}
        if not a.no_e2e and world == 1:
            try:
                out["end_to_end"] = end_to_end(wl, margs)
            except Exception as exc:
                out["end_to_end"] = {"error": repr(exc)}
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl, batches_cpu, a.cpu_seconds)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]      # (N > 1: the whole job against ONE host's cores)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
