#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
for i in 2 3; do
  python3 -m pytest tests -m gpu -x -q > $out/r6_o_gpu_suite_head_run$i.log 2>&1; echo "run $i rc=$?"; tail -2 $out/r6_o_gpu_suite_head_run$i.log
done
