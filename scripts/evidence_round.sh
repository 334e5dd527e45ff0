#!/bin/bash
# usage (GPU box, via gpurun): scripts/evidence_round.sh <tag>   -- one evidence pass at the current kernels:
#   profile_round.sh (headline: rocprof stats, FETCH / WRITE PMC, the default bench line), bench_all_workloads.sh (every other
#   workload: PMC traffic, bench line, rocprof stats), MFMA-busy and issue PMC passes (headline + both config-5 models), issue-order
#   traces, and the config-5 steps with the weight-resident GEMMs off / on (CAL_AMD_WRES) in the same build.
tag=$1
bash scripts/profile_round.sh $tag > gpurun_out/profile_round.log 2>&1
bash scripts/bench_all_workloads.sh $tag > gpurun_out/bench_all.log 2>&1
for w in spmotif_b0.9_causalgcn_h128_l3_bs128 ba5000_causalgcn_h256_l3_bs32 ba5000_causalgat_h256_l3_bs32; do
    bash scripts/pmc_mfma_util.sh $w $tag > /dev/null 2>&1
    bash scripts/pmc_issue.sh $w $tag > /dev/null 2>&1
done
for w in ba5000_causalgcn_h256_l3_bs32 ba5000_causalgat_h256_l3_bs32 spmotif_b0.9_causalgcn_h128_l3_bs128 spmotif_b0.9_causalgcn_nodenum15_bs32 spmotif_b0.9_causalgcn_nodenum15_bs128 mutaglike_causalgat_h128_l3_bs64 nci1like_causalgcn_h128_l3_bs512; do
    bash scripts/trace_step.sh $w $tag/traces > /dev/null 2>&1
done
for v in 0 1; do
    for w in ba5000_causalgcn_h256_l3_bs32 ba5000_causalgat_h256_l3_bs32; do
        CAL_AMD_WRES=$v python bench.py --workload $w --steps 50 --warmup 5 --batches 4 --no-e2e --no-cpu-baseline --no-roofline 2>/dev/null | \
            python -c "import json,sys; d=json.loads(sys.stdin.read()); print('CAL_AMD_WRES=$v', d['config']['workload'], round(d['ms_per_step'],4), 'ms/step')"
    done
done | tee gpurun_out/$tag/ab_wres.txt
rm -f gpurun_out/$tag/*.log
python -c "import json; d=json.load(open('gpurun_out/$tag/bench_engine_graph.json')); print(d['value'], d['ms_per_step'])"
