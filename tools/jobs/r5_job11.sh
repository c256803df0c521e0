#!/bin/bash
set -u
tag=${1:-r5_l}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_model_gpu.py -q --maxfail=10 -k "positive_slots" -s > $out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -2 $out/${tag}_tests.log; grep -n "^E \|worst tensors\|all slots vs" $out/${tag}_tests.log | head -12
