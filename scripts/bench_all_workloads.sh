#!/bin/bash
# usage (GPU box, via gpurun): scripts/bench_all_workloads.sh <tag>   -> gpurun_out/<tag>/bench_<workload>.json for every
# non-headline workload of bench.py (SURVEY.md 8d configs 1, 3-5 on synthetic stand-ins) + rocprof kernel stats of config 5
tag=$1
mkdir -p gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# PMC traffic of every non-headline workload first (two separate passes each), so that their bench lines carry roofline.traffic
for w in ba5000_causalgat_h256_l3_bs32 ba5000_causalgcn_h256_l3_bs32 spmotif_b0.9_causalgcn_nodenum15_bs32 spmotif_b0.9_causalgcn_nodenum15_bs128 mutaglike_causalgat_h128_l3_bs64 nci1like_causalgcn_h128_l3_bs512 spmotif_b0.9_causalgat_h128_l3_bs128 spmotif_b0.9_causalgin_h128_l3_bs128; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/prof_$tag/${w}_$ctr -o b -- python bench.py --workload $w --steps 6 --warmup 2 --batches 2 --no-e2e --no-cpu-baseline --no-roofline --mode eager --repeats 1 > /tmp/pmc_$w.log 2>&1
        python scripts/pmc_summary.py $(find /tmp/prof_$tag/${w}_$ctr -name "*counter_collection.csv" | head -1) $ctr > gpurun_out/$tag/pmc_${ctr}_$w.json
    done
    python scripts/pmc_traffic.py gpurun_out/$tag/pmc_FETCH_SIZE_$w.json gpurun_out/$tag/pmc_WRITE_SIZE_$w.json $w > gpurun_out/$tag/pmc_traffic_$w.json
    cp gpurun_out/$tag/pmc_traffic_$w.json profiles/pmc_traffic_$w.json
done
for w in spmotif_b0.9_causalgcn_nodenum15_bs32 spmotif_b0.9_causalgcn_nodenum15_bs128 spmotif_b0.9_causalgat_h128_l3_bs128 spmotif_b0.9_causalgin_h128_l3_bs128 mutaglike_causalgat_h128_l3_bs64 nci1like_causalgcn_h128_l3_bs512 ba5000_causalgcn_h256_l3_bs32 ba5000_causalgat_h256_l3_bs32; do
    python bench.py --workload $w --steps 100 --warmup 10 --batches 4 --no-e2e --cpu-seconds 8 > gpurun_out/$tag/bench_$w.json 2> gpurun_out/$tag/bench_$w.err
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for w in ba5000_causalgat_h256_l3_bs32 ba5000_causalgcn_h256_l3_bs32 spmotif_b0.9_causalgcn_nodenum15_bs32 spmotif_b0.9_causalgcn_nodenum15_bs128 mutaglike_causalgat_h128_l3_bs64 nci1like_causalgcn_h128_l3_bs512 spmotif_b0.9_causalgin_h128_l3_bs128 spmotif_b0.9_causalgat_h128_l3_bs128; do
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/$w -o b -- python bench.py --workload $w --steps 20 --warmup 3 --batches 2 --no-e2e --no-cpu-baseline --no-roofline --mode eager --repeats 1 > /tmp/prof_$w.log 2>&1
    cp $(find /tmp/prof_$tag/$w -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/rocprof_kernel_stats_$w.csv
done
