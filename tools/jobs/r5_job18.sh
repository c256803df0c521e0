#!/bin/bash
# three full -x suite runs at the head (flake hunt)
set -u
tag=${1:-r5_t}
out=$(pwd)/gpurun_out
mkdir -p $out
for i in 1 2 3; do
  python3 -m pytest tests -x -q -m gpu > $out/${tag}_gpu_suite_run$i.log 2>&1
  echo "suite run $i rc=$?"; tail -1 $out/${tag}_gpu_suite_run$i.log; grep -E "^(FAILED|ERROR)" $out/${tag}_gpu_suite_run$i.log | head
done
