#!/bin/bash
set -u
tag=${1:-r5_k}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_model_gpu.py -q --maxfail=10 > $out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -2 $out/${tag}_tests.log; grep -E "^(FAILED|ERROR)" $out/${tag}_tests.log | head
grep -n "^E " $out/${tag}_tests.log | head -20
python3 bench.py --steps 20 --warmup 3 > $out/${tag}_bench_n1.json 2> $out/${tag}_bench_n1.err
python3 - <<PY
import json
d=json.load(open("$out/${tag}_bench_n1.json"))
print(d["value"], d["ms_per_step"], "strict", d.get("strict_bf16x3_images_per_sec"), "pos-slots", d.get("mask_head_positive_slots_images_per_sec"), "sat", d["config"]["conv_saturated_blocks"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["warm_step_seconds"])
PY
tail -3 $out/${tag}_bench_n1.err
