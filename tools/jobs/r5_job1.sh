#!/bin/bash
# round 5, first GPU call: the suite (no -x), the headline line, the 'heads' stage line + its kernel table,
# the file-fed line (loader inside the timed region).
set -u
tag=${1:-r5_a}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
python3 -m pytest tests -m gpu -q --maxfail=12 > $out/${tag}_gpu_suite.log 2>&1
echo "suite rc=$?"; tail -3 $out/${tag}_gpu_suite.log; grep -E "^(FAILED|ERROR)" $out/${tag}_gpu_suite.log | head -20
python3 bench.py --steps 20 --warmup 3 > $out/${tag}_bench_n1.json 2> $out/${tag}_bench_n1.err
head -c 600 $out/${tag}_bench_n1.json; echo
python3 bench.py --steps 20 --warmup 3 --stage heads --no-cpu-baseline > $out/${tag}_bench_heads.json 2> $out/${tag}_bench_heads.err
head -c 400 $out/${tag}_bench_heads.json; echo
python3 bench.py --steps 20 --warmup 3 --data files --no-cpu-baseline --no-strict > $out/${tag}_bench_files.json 2> $out/${tag}_bench_files.err
head -c 400 $out/${tag}_bench_files.json; echo; tail -5 $out/${tag}_bench_files.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_heads
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_heads -- python3 $root/bench.py --stage heads --steps 4 --warmup 1 --settle 2 --no-cpu-baseline --no-strict > $out/${tag}_heads_under_rocprof.json 2>/dev/null
cd $root
cp $(ls /tmp/prof_heads/*/*kernel_stats.csv | head -1) $out/${tag}_heads_rocprofv3_kernel_stats.csv
python3 tools/step_breakdown.py $(ls /tmp/prof_heads/*/*kernel_trace.csv | head -1) 60 > $out/${tag}_heads_step_breakdown.txt 2>&1
head -14 $out/${tag}_heads_step_breakdown.txt
