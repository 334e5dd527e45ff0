"""Per-kernel mean of one rocprofv3 --pmc counter: pmc_summary.py <counter_collection.csv> <COUNTER> -> JSON."""
import csv, json, sys, collections
path, ctr = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != ctr:
            continue
        k = row.get("Kernel_Name", "")[:90]
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
out = {k: {"calls": n, "mean_KB": s / n} for k, (s, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])}
json.dump(out, sys.stdout, indent=1)
