#!/bin/bash
# usage (GPU box, via gpurun): scripts/pmc_issue.sh <workload> <tag>
# Three rocprofv3 PMC passes (kernel-trace only) of the eager step: where do a kernel's wave-cycles go?  Per kernel and launch:
#   SQ_WAVE_CYCLES (wave-resident cycles, summed over waves), SQ_BUSY_CYCLES, SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU (cycles a VALU instruction is
#   executing), SQ_INSTS_LDS, SQ_ACTIVE_INST_LDS, SQ_WAIT_INST_ANY (waves waiting on any s_waitcnt), SQ_INSTS_SALU
#   third pass: SQ_ACTIVE_INST_ANY, SQ_INSTS_VMEM_RD / _WR (SQ_ACTIVE_INST_VMEM reads 0 on gfx950 with ROCm 7.2)
# -> gpurun_out/<tag>/pmc_issue_<workload>.json  (valu_active / wave_cycles = share of a wave's life spent executing VALU instructions)
w=$1; tag=$2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/pmc_issue_$w/$i -o bench -- \
        python bench.py --workload $w --steps 6 --warmup 2 --batches 2 --mode eager --no-cpu-baseline --no-roofline --no-e2e --repeats 1 > gpurun_out/$tag/pmc_issue_$w.$i.log 2>&1
done
python - $(find /tmp/pmc_issue_$w -name "*counter_collection.csv") > gpurun_out/$tag/pmc_issue_$w.json <<'PY'
import csv, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(collections.Counter)
for f in sys.argv[1:]:
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:64]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); calls[k][row["Counter_Name"]] += 1
out = {}
for k, c in acc.items():
    per = {n: v / calls[k][n] for n, v in c.items()}
    wc = per.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0: continue
    per["valu_active_over_wave_cycles"] = per.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
    per["lds_active_over_wave_cycles"] = per.get("SQ_ACTIVE_INST_LDS", 0.0) / wc
    per["wait_any_over_wave_cycles"] = per.get("SQ_WAIT_INST_ANY", 0.0) / wc
    per["vmem_active_over_wave_cycles"] = per.get("SQ_ACTIVE_INST_VMEM", 0.0) / wc
    per["any_active_over_wave_cycles"] = per.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    out[k] = {n: (round(v, 4) if v < 10 else round(v)) for n, v in per.items()}
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))), sys.stdout, indent=1)
PY
head -c 2500 gpurun_out/$tag/pmc_issue_$w.json
