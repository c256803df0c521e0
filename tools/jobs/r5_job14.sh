#!/bin/bash
# in-suite context of the lr-0.01 dynamics test: which preceding files move its start state / trajectory?
set -u
out=$(pwd)/gpurun_out
mkdir -p $out
K="reference_learning_rate"
run() {  # name, files...
  name=$1; shift
  python3 -m pytest "$@" tests/test_zz_dynamics_gpu.py -q -m gpu -s -k "$K or not zz_dynamics" -p no:cacheprovider > $out/r5_p_ctx_$name.log 2>&1
  echo "== $name rc=$?"; grep -A 4 "^HIP" $out/r5_p_ctx_$name.log | grep "step': 0\|step': 79" | head -2
  grep -A 4 "^aten" $out/r5_p_ctx_$name.log | grep "step': 79" | head -1
}
run alone
run f16 tests/test_f16_gpu.py
run loader tests/test_loader_gpu.py
run multistep tests/test_multistep_gpu.py
run resnext tests/test_resnext_gpu.py
run parallel tests/test_parallel_gpu.py
