#!/bin/bash
# Timeline of ONE captured training step (run on the GPU box from the repo root): bash scripts/trace_round.sh r02 [bench args]
#   rocprofv3 --kernel-trace of a short bench run -> gpurun_out/<tag>_trace_step.txt (per-dispatch start, duration, gap to the
#   previous kernel's end, and the per-kernel totals of the last replayed step)
set -o pipefail
tag=${1:-r02}; shift
R=$(pwd); out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/trace_$tag
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/trace_$tag -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra --no-roofline --min-seconds 0 "$@" > $out/${tag}_trace_bench.json 2> $out/${tag}_trace.err || { tail -5 $out/${tag}_trace.err; exit 1; }
cd $R
csv=$(find $out/trace_$tag -name '*kernel_trace.csv' | head -1)
python scripts/trace_step.py $csv --all > $out/${tag}_trace_step.txt
tail -45 $out/${tag}_trace_step.txt
rm -rf $out/trace_$tag
