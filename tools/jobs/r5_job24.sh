#!/bin/bash
set -u
tag=${1:-r5_ad}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_f16_gpu.py tests/test_resnext_gpu.py tests/test_e2e_gpu.py tests/test_model_gpu.py -q -m gpu -x > $out/${tag}_tests.log 2>&1; echo "tests rc=$?"; tail -1 $out/${tag}_tests.log; grep -n "^E " $out/${tag}_tests.log | head
export SLN_DEBUG_KNOBS=1
for i in 1 2; do
  for v in 0 1; do
    SLN_CONV_SMALLGRID64=$v python3 bench.py --config resnext --parts 1 --steps 10 --warmup 3 > $out/${tag}_resnext_sg${v}_$i.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$out/${tag}_resnext_sg${v}_$i.json'));print('resnext SMALLGRID64=$v', d['value'], d['ms_per_step'])"
  done
done
for i in 1 2; do
  for v in 0 1; do
    SLN_CONV_SMALLGRID64=$v python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-strict > $out/${tag}_sln_sg${v}_$i.json 2>/dev/null
    python3 -c "import json;d=json.load(open('$out/${tag}_sln_sg${v}_$i.json'));print('sln SMALLGRID64=$v', d['value'], d['ms_per_step'])"
  done
done
