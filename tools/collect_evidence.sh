#!/bin/bash
# Everything profiles/ holds for one revision, in one gpurun call (from the repo root on the GPU box):
#   tools/collect_evidence.sh <tag>
# The GPU suite runs FIRST, without -x, into <tag>_gpu_suite.log; on a red run nothing else is produced (no
# *_bench_n1.json: a number measured on a revision whose tests fail is not evidence).  SKIP_SUITE=1 only for a
# revision whose suite log of the SAME tree already exists next to it.
set -u
tag=${1:-r4}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
if [ "${SKIP_SUITE:-0}" != "1" ]; then
    python3 -m pytest tests -m gpu -q > $out/${tag}_gpu_suite.log 2>&1
    rc=$?
    tail -3 $out/${tag}_gpu_suite.log
    if [ $rc -ne 0 ]; then
        echo "collect_evidence: GPU suite rc=$rc -- refusing to write ${tag}_bench_n1.json"
        grep -E "^(FAILED|ERROR)" $out/${tag}_gpu_suite.log | head -20
        exit 1
    fi
fi
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
