#!/bin/bash
# flake hunt: the tests that live in the amplifying regime, six times over
set -u
tag=${1:-r5_s}
out=$(pwd)/gpurun_out
mkdir -p $out
for i in 1 2 3 4 5 6; do
  python3 -m pytest tests/test_multistep_gpu.py tests/test_model_gpu.py tests/test_f16_gpu.py tests/test_zz_dynamics_gpu.py -q -m gpu --durations=8 \
     -k "multistep or steps or positive_slots or dynamics or learn or learning or block or bottleneck or full_depth" > $out/${tag}_flake_run$i.log 2>&1
  echo "run $i rc=$?"; tail -1 $out/${tag}_flake_run$i.log; grep -E "^(FAILED|ERROR)" $out/${tag}_flake_run$i.log | head
done
grep -A 10 "slowest" $out/${tag}_flake_run1.log | head -12
