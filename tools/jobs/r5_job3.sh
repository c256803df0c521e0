#!/bin/bash
set -u
tag=${1:-r5_c}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_f16_gpu.py tests/test_multistep_gpu.py tests/test_resnext_gpu.py "tests/test_zz_dynamics_gpu.py::test_config5_resnext101_msc_train_step_full_depth" -q --maxfail=30 > $out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -3 $out/${tag}_tests.log; grep -E "^(FAILED|ERROR)" $out/${tag}_tests.log | head -30
python3 bench.py --config resnext --parts 2 --steps 10 --warmup 3 > $out/${tag}_resnext_p2.json 2> $out/${tag}_resnext_p2.err
head -c 400 $out/${tag}_resnext_p2.json; echo; tail -3 $out/${tag}_resnext_p2.err
python3 bench.py --config resnext --parts 1 --steps 10 --warmup 3 > $out/${tag}_resnext_p1.json 2> $out/${tag}_resnext_p1.err
head -c 400 $out/${tag}_resnext_p1.json; echo; tail -3 $out/${tag}_resnext_p1.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_rx
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rx -- python3 $root/bench.py --config resnext --parts 1 --steps 4 --warmup 2 > $out/${tag}_resnext_p1_under_rocprof.json 2>/dev/null
cd $root
cp $(ls /tmp/prof_rx/*/*kernel_stats.csv | head -1) $out/${tag}_resnext_p1_rocprofv3_kernel_stats.csv
python3 tools/step_breakdown.py $(ls /tmp/prof_rx/*/*kernel_trace.csv | head -1) 40 > $out/${tag}_resnext_p1_step_breakdown.txt 2>&1
head -30 $out/${tag}_resnext_p1_step_breakdown.txt
