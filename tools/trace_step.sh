#!/bin/bash
# kernel trace of one steady-state train step: tools/trace_step.sh <tag>  ->  gpurun_out/<tag>_step_breakdown.txt
tag=${1:-x}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strict > $root/gpurun_out/${tag}_bench_under_rocprof.json 2>/dev/null
cd $root
python3 tools/step_breakdown.py $(ls /tmp/prof_kt/*/*kernel_trace.csv | head -1) 300 > gpurun_out/${tag}_step_breakdown.txt
head -5 gpurun_out/${tag}_step_breakdown.txt
