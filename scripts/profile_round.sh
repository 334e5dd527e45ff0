#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/profile_round.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the bench command (hipGraph replay, like the bench line)
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of an eager run, --kernel-trace only
# Outputs land in gpurun_out/<tag>/ (copy what should be judged into profiles/).
tag=$1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/stats -o bench -- \
    python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-e2e > gpurun_out/$tag/bench_under_rocprof.log 2>&1
cp $(find /tmp/prof_$tag/stats -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/rocprof_kernel_stats_engine.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/prof_$tag/$ctr -o bench -- \
        python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-e2e --mode eager --batches 2 > gpurun_out/$tag/pmc_$ctr.log 2>&1
    f=$(find /tmp/prof_$tag/$ctr -name "*counter_collection.csv" | head -1)
    python scripts/pmc_summary.py $f $ctr > gpurun_out/$tag/pmc_$ctr.json
done
python scripts/pmc_traffic.py gpurun_out/$tag/pmc_FETCH_SIZE.json gpurun_out/$tag/pmc_WRITE_SIZE.json spmotif_b0.9_causalgcn_h128_l3_bs128 > gpurun_out/$tag/pmc_traffic.json
cp gpurun_out/$tag/pmc_traffic.json profiles/pmc_traffic.json       # so that the bench line below carries roofline.traffic
python bench.py > gpurun_out/$tag/bench_default.log 2>&1
tail -1 gpurun_out/$tag/bench_default.log > gpurun_out/$tag/bench_engine_graph.json
head -12 gpurun_out/$tag/rocprof_kernel_stats_engine.csv | cut -c1-160
