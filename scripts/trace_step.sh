#!/bin/bash
# usage (GPU box): scripts/trace_step.sh <workload> <tag>  -> gpurun_out/<tag>/trace_<workload>.txt
# the launches of ONE eager step in issue order with their durations (rocprofv3 --kernel-trace, last of 6 steps)
w=$1; tag=$2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$tag/$w -o t -- \
    python bench.py --workload $w --steps 4 --warmup 2 --batches 1 --repeats 1 --mode eager --no-cpu-baseline --no-e2e --no-roofline > /tmp/trace_$w.log 2>&1
python - <<PY > gpurun_out/$tag/trace_$w.txt
import csv, glob
f = glob.glob("/tmp/trace_$tag/$w/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# one step = from the last k_zero_f64 (first launch of a step) to the end
starts = [i for i, r in enumerate(rows) if "k_zero_f64" in r["Kernel_Name"]]
lo = starts[-2] if len(starts) > 1 else 0
hi = starts[-1] if len(starts) > 1 else len(rows)
t0 = int(rows[lo]["Start_Timestamp"]); prev_end = t0; tot = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += e - s
    print("%9.1f us  +gap %6.1f  dur %8.1f us  grid %-12s wg %-5s %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3,
          r["Grid_Size_X"] + "x" + r["Grid_Size_Y"], r["Workgroup_Size_X"], r["Kernel_Name"][:110]))
    prev_end = e
print("launches %d, sum of durations %.1f us, span %.1f us" % (hi - lo, tot / 1e3, (prev_end - t0) / 1e3))
PY
tail -3 gpurun_out/$tag/trace_$w.txt
