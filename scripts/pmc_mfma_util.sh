#!/bin/bash
# usage (GPU box, via gpurun): scripts/pmc_mfma_util.sh <workload> <tag>
# One rocprofv3 PMC pass (kernel-trace only) with the matrix-core busy counter next to the GPU-active counter:
# per kernel  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)   (gfx94x formula; gfx950 has no
# derived-counter section in ROCm 7.2).  Output: gpurun_out/<tag>/pmc_mfma_util_<workload>.json
w=$1; tag=$2
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_mfma_$w -o bench -- \
    python bench.py --workload $w --steps 6 --warmup 2 --batches 2 --mode eager --no-cpu-baseline --no-roofline --no-e2e > gpurun_out/$tag/pmc_mfma_$w.log 2>&1
f=$(find /tmp/pmc_mfma_$w -name "*counter_collection.csv" | head -1)
python - "$f" > gpurun_out/$tag/pmc_mfma_util_$w.json <<'PY'
import csv, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"][:70]
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "GRBM_GUI_ACTIVE": calls[k] += 1
out = {}
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0 or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) <= 0: continue
    out[k] = {"calls": calls[k], "mfma_busy_cycles_per_call": c["SQ_VALU_MFMA_BUSY_CYCLES"] / calls[k], "gui_active_per_call": gui / calls[k],
              "sq_busy_cycles_per_call": c.get("SQ_BUSY_CYCLES", 0.0) / calls[k], "mfma_util": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 256 * 4)}   # GUI_ACTIVE is summed over the 8 XCDs
json.dump(dict(sorted(out.items(), key=lambda kv: -kv[1]["mfma_busy_cycles_per_call"] * kv[1]["calls"])), sys.stdout, indent=1)
PY
head -c 1500 gpurun_out/$tag/pmc_mfma_util_$w.json
