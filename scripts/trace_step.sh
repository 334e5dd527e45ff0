#!/bin/bash
# usage: scripts/trace_step.sh <outdir-name> [bench args...]   (run on the GPU box via gpurun)
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$out -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --batches 1 --mode eager "$@" > /tmp/prof_$out.log 2>&1
mkdir -p gpurun_out/$out
cp /tmp/prof_$out/bench_kernel_trace.csv /tmp/prof_$out/bench_kernel_stats.csv gpurun_out/$out/
tail -1 /tmp/prof_$out.log | cut -c1-150
