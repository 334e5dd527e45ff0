#!/bin/bash
# usage (GPU box): scripts/prof_workload.sh <workload> <tag> [steps]  -> gpurun_out/<tag>/rocprof_kernel_stats_<workload>.csv
w=$1; tag=$2; steps=${3:-20}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag/$w -o bench -- \
    python bench.py --workload $w --steps $steps --warmup 3 --repeats 1 --no-cpu-baseline --no-e2e --no-roofline > gpurun_out/$tag/bench_under_rocprof_$w.log 2>&1
cp $(find /tmp/prof_$tag/$w -name "*kernel_stats.csv" | head -1) gpurun_out/$tag/rocprof_kernel_stats_$w.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/$tag/rocprof_kernel_stats_$w.csv")))
steps=$steps+3  # (+ capture warm-ups: per-kernel averages are exact, per-step sums approximate)
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total per step %.1f us" % (tot/steps/1e3))
for r in rows[:28]:
    print("%-60s calls/step %5.1f  avg %8.1f us  per step %8.1f us" % (r["Name"][:60], int(r["Calls"])/steps, float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/steps/1e3))
PY
