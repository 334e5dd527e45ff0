cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -o bench -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-e2e --batches 1 > /tmp/seq.log 2>&1
f=$(find /tmp/prof_seq -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"][:60] for r in rows]
# find last occurrence of k_zero_f64 and print the sequence of the last full step
idx=[i for i,n in enumerate(names) if "k_zero_f64" in n]
a,b=idx[-2],idx[-1]
t0=int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%8.1f us  %6.1f us  %s" % ((int(r["Start_Timestamp"])-t0)/1000.0,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000.0,r["Kernel_Name"][:70]))
print("copyBuffer total in run:", sum(1 for n in names if "copyBuffer" in n), "steps kernels", b-a)
PY
