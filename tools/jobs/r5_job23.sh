#!/bin/bash
set -u
tag=${1:-r5_ac}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_f16_gpu.py tests/test_resnext_gpu.py -q -m gpu > $out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -1 $out/${tag}_tests.log
for i in 1 2 3; do
python3 bench.py --config resnext --parts 1 --steps 10 --warmup 3 > $out/${tag}_resnext_p1_$i.json 2>/dev/null
python3 -c "import json;d=json.load(open('$out/${tag}_resnext_p1_$i.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'])"
done
