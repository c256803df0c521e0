#!/bin/bash
# round 5, second GPU call: the new fp16 tests first (fast feedback), then the whole suite, then configs[4] lines.
set -u
tag=${1:-r5_b}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_f16_gpu.py tests/test_loader_gpu.py -q --maxfail=30 > $out/${tag}_f16_tests.log 2>&1
echo "f16+loader rc=$?"; tail -3 $out/${tag}_f16_tests.log; grep -E "^(FAILED|ERROR)" $out/${tag}_f16_tests.log | head -30
python3 -m pytest tests -m gpu -q --maxfail=20 --deselect tests/test_f16_gpu.py --deselect tests/test_loader_gpu.py > $out/${tag}_gpu_suite.log 2>&1
echo "suite rc=$?"; tail -3 $out/${tag}_gpu_suite.log; grep -E "^(FAILED|ERROR)" $out/${tag}_gpu_suite.log | head -20
python3 bench.py --config resnext --parts 2 --steps 10 --warmup 3 > $out/${tag}_resnext_p2.json 2> $out/${tag}_resnext_p2.err
head -c 500 $out/${tag}_resnext_p2.json; echo; tail -3 $out/${tag}_resnext_p2.err
python3 bench.py --config resnext --parts 1 --steps 10 --warmup 3 > $out/${tag}_resnext_p1.json 2> $out/${tag}_resnext_p1.err
head -c 500 $out/${tag}_resnext_p1.json; echo; tail -3 $out/${tag}_resnext_p1.err
