#!/bin/bash
# Everything profiles/ holds for one revision, in one gpurun call (from the repo root on the GPU box):
#   tools/collect_evidence.sh <tag>
set -u
tag=${1:-r3}
root=$(pwd)
out=$root/gpurun_out
python3 bench.py > $out/${tag}_bench_n1.json 2> $out/${tag}_bench_n1.err
ms=$(python3 -c "import json;print(json.load(open('$out/${tag}_bench_n1.json'))['ms_per_step'])")
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strict > $out/${tag}_bench_under_rocprof.json 2>/dev/null
cd $root
python3 tools/step_breakdown.py $(ls /tmp/prof_kt/*/*kernel_trace.csv | head -1) 80 > $out/${tag}_step_breakdown.txt
cp $(ls /tmp/prof_kt/*/*kernel_stats.csv | head -1) $out/${tag}_rocprofv3_kernel_stats.csv
PMC_MS=$ms tools/collect_pmc.sh $tag > $out/${tag}_collect.log 2>&1
SLN_PROFILE_SHAPES=1 python3 bench.py --no-cpu-baseline --no-strict > /dev/null 2> $out/${tag}_shapes.txt
head -c 400 $out/${tag}_bench_n1.json; echo; head -12 $out/${tag}_step_breakdown.txt; cat $out/${tag}_pmc_passes.txt
