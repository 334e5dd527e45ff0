"""profiles/pmc_traffic.json from the two per-kernel PMC summaries of scripts/profile_round.sh:
    python scripts/pmc_traffic.py gpurun_out/<tag>/pmc_FETCH_SIZE.json gpurun_out/<tag>/pmc_WRITE_SIZE.json <workload> > profiles/pmc_traffic.json
HBM-side bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950: wide coalesced reads are tallied at half their bytes; WRITE_SIZE as reported).  bench.py copies
`bytes_per_launch` of the kernel classes it times into `roofline.traffic`."""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cal_amd.build import kernel_source_sha
fetch, write, workload = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3]
# bench.py roofline class -> substring (or, with a leading "re:", regular expression) of the kernel name in the PMC summaries.
# One class = ONE template instantiation wherever bench.py times one (round-4 review: the k_espmm mean mixed the single-branch
# and the two-branch launches); "by_kernel" below lists every kernel of the pass by its full template name as well.
classes = {
    "k_gconv_fwd": "k_gconv_fwd<false", "k_gconv_fwd_co": "k_gconv_fwd<true", "k_gconv_bwd": "k_gconv_bwd<false, 1",
    "k_gw_fwd": "k_gw_fwd<false", "k_gw_fwd_co": "k_gw_fwd<true", "k_gw_bwd": "k_gw_bwd<false, 1", "k_gw_bwd_top": "k_gw_bwd<false, 0",
    "k_gw_bwd_co": "k_gw_bwd<true, 2", "k_ro_step": "k_ro_step", "k_plan_graph": "k_plan_graph",
    "k_gconv_bwd_top": "k_gconv_bwd<false, 0", "k_gconv_bwd_co": "k_gconv_bwd<true, 2", "k_att_fwd_graph": "k_att_fwd_graph",
    "k_ggin_fwd": "k_ggin_fwd<1>", "k_ggin_bwd": "k_ggin_bwd<1>", "k_feat_bwd": "k_feat_bwd",
    "k_att_bwd_graph": "k_att_bwd_graph", "k_finish": "k_finish", "k_espmm_all": "k_espmm", "k_gemm_backbone": "k_gemm<",
    # the instantiation bench.py's aggregation class times: single-branch, no edge weights / statistics / SDDMM (transposed backbone)
    "k_espmm": "re:k_espmm<4, \\d+, false, false, false, false>", "k_espmm_fwd_stats": "re:k_espmm<4, \\d+, false, true, false, false>",
    "k_espmm_co": "re:k_espmm<4, \\d+, true, false, false, false>", "k_espmm_co_T": "re:k_espmm<4, \\d+, true, false, true, true>",
    "k_wres_fwd": "re:k_wres<false, 1,", "k_wres_fwd_co": "re:k_wres<false, 2,", "k_wres_nt": "re:k_wres<true, 0, \\d+, 2>", "k_wres_nt_co": "re:k_wres<true, 0, \\d+, 3>",
    "k_tn": "k_tn<1>", "k_tn_co": "k_tn<2>",
    "k_gemm_dual": "k_gemm_dual", "k_ggat_fwd": "k_ggat_fwd", "k_ggat_bwd": "k_ggat_bwd", "k_gat_fwd": "k_gat_fwd",
    "k_gat_bwd_dst": "k_gat_bwd_dst", "k_gat_bwd_src": "k_gat_bwd_src", "k_gemm_big": "k_gemm_big<", "k_gemm_big_dual": "k_gemm_big_dual",
}
def mean(summary, sub):
    tot, n = 0.0, 0
    for k, v in summary.items():
        if (re.search(sub[3:], k) is not None) if sub.startswith("re:") else (sub in k):
            tot += v["mean_KB"] * v["calls"]; n += v["calls"]
    return (tot / n, n) if n else (None, 0)
out = {"workload": workload,
       "kernel_source_sha": kernel_source_sha(),      # bench.py nulls roofline.traffic when the kernels have changed since
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two separate passes of the eager step "
                 "(scripts/profile_round.sh); bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 "
                 "(gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md); per-kernel means"}
for cls, sub in classes.items():
    f, nf = mean(fetch, sub)
    w, nw = mean(write, sub)
    if f is None or w is None:
        continue
    out[cls] = {"fetch_KB_raw": round(f, 1), "write_KB": round(w, 1), "launches_sampled": nf,
                "bytes_per_launch": int((2 * f + w) * 1024)}
by = {}
for k, v in fetch.items():
    if k in write and "cal::" in k:
        name = k.split("(")[0].replace("void ", "")
        by[name] = {"launches_sampled": v["calls"], "bytes_per_launch": int((2 * v["mean_KB"] + write[k]["mean_KB"]) * 1024)}
out["by_kernel"] = by
json.dump(out, sys.stdout, indent=1)
