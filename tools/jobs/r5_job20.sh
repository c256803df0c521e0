#!/bin/bash
# margins of the assertions that live in the amplifying regime: the tests' own reports, twelve runs (-s)
set -u
tag=${1:-r5_w}
out=$(pwd)/gpurun_out
mkdir -p $out
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  python3 -m pytest tests/test_multistep_gpu.py tests/test_model_gpu.py tests/test_zz_dynamics_gpu.py -q -m gpu -s \
     -k "multistep or steps or positive_slots or learns_one_fixed" > $out/${tag}_margins_run$i.log 2>&1
  echo "run $i rc=$?"; tail -1 $out/${tag}_margins_run$i.log; grep -E "^(FAILED|ERROR)" $out/${tag}_margins_run$i.log | head
done
