#!/bin/bash
set -u
tag=${1:-r5_al}
out=$(pwd)/gpurun_out
mkdir -p $out
for i in 1 2 3 4; do
  python3 -m pytest tests/test_conv_gpu.py -q -m gpu -k "rpn_heads or row3 or head_room or linked or two_reader" > $out/${tag}_conv_run$i.log 2>&1; echo "conv run $i rc=$?"; tail -1 $out/${tag}_conv_run$i.log
done
python3 -m pytest tests -x -q -m gpu > $out/${tag}_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -1 $out/${tag}_gpu_suite.log; grep -E "^(FAILED|ERROR)" $out/${tag}_gpu_suite.log | head
