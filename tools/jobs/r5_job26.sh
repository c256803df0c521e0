#!/bin/bash
set -u
tag=${1:-r5_ag}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_conv_gpu.py -q -x -k "rpn_heads or two_reader or head_room" > $out/${tag}_tests0.log 2>&1; echo "unit rc=$?"; tail -1 $out/${tag}_tests0.log; grep -n "^E " $out/${tag}_tests0.log | head -20
python3 -m pytest tests -x -q -m gpu > $out/${tag}_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -1 $out/${tag}_gpu_suite.log; grep -E "^(FAILED|ERROR)" $out/${tag}_gpu_suite.log | head
for i in 1 2 3 4 5; do
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strict > $out/${tag}_bench_$i.json 2>/dev/null
  python3 -c "import json;d=json.load(open('$out/${tag}_bench_$i.json'));print('run $i', d['value'], d['ms_per_step'], d['config']['conv_saturated_blocks'], d['config']['conv_saturated_events_timed'])"
done
