#!/bin/bash
set -u
tag=${1:-r5_x}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_conv_gpu.py -q -x -k "head_room or two_part or chained or bit_reproducible or row3_weight" > $out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -3 $out/${tag}_tests.log; grep -n "^E " $out/${tag}_tests.log | head
python3 -m pytest tests -x -q -m gpu > $out/${tag}_gpu_suite.log 2>&1
echo "suite rc=$?"; tail -2 $out/${tag}_gpu_suite.log; grep -E "^(FAILED|ERROR)" $out/${tag}_gpu_suite.log | head
for i in 1 2 3; do
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strict > $out/${tag}_bench_$i.json 2>/dev/null
python3 -c "import json;d=json.load(open('$out/${tag}_bench_$i.json'));print(d['value'], d['ms_per_step'], d['config']['conv_saturated_blocks'], d['config']['conv_saturated_events_timed'])"
done
