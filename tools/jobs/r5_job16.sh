#!/bin/bash
# full GPU suite twice (-x, as the driver runs it) at the ROW3 head
set -u
tag=${1:-r5_r}
out=$(pwd)/gpurun_out
mkdir -p $out
for i in 1 2; do
  python3 -m pytest tests -x -q -m gpu > $out/${tag}_gpu_suite_run$i.log 2>&1
  echo "suite run $i rc=$?"; tail -2 $out/${tag}_gpu_suite_run$i.log; grep -E "^(FAILED|ERROR)" $out/${tag}_gpu_suite_run$i.log | head
done
