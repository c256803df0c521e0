#!/bin/bash
set -u
tag=${1:-r5_af}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_conv_gpu.py -q -x -k "rpn_heads or two_reader" > $out/${tag}_tests0.log 2>&1; echo "unit rc=$?"; tail -1 $out/${tag}_tests0.log; grep -n "^E " $out/${tag}_tests0.log | head -20
python3 -m pytest tests/test_e2e_gpu.py tests/test_model_gpu.py tests/test_parallel_gpu.py tests/test_multistep_gpu.py -q -m gpu -x > $out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -1 $out/${tag}_tests.log; grep -n "^E " $out/${tag}_tests.log | head
for i in 1 2 3; do
  for v in 0 1; do
    SLN_FUSE_RPN_HEADS=$v python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-strict > $out/${tag}_ab_fuse${v}_$i.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$out/${tag}_ab_fuse${v}_$i.json'));print('FUSE_RPN_HEADS=$v run $i', d['value'], d['ms_per_step'], d['config']['conv_saturated_blocks'])"
  done
done
