#!/bin/bash
set -u
tag=${1:-r5_j}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -m pytest tests/test_model_gpu.py -q --maxfail=10 -k "detect or config" > $out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -2 $out/${tag}_tests.log; grep -E "^(FAILED|ERROR)" $out/${tag}_tests.log | head
python3 bench.py --config detect --steps 20 --warmup 3 > $out/${tag}_detect.json 2> $out/${tag}_detect.err
head -c 400 $out/${tag}_detect.json; echo; tail -3 $out/${tag}_detect.err
python3 bench.py --config detect --steps 20 --warmup 3 --tail > $out/${tag}_detect_tail.json 2> $out/${tag}_detect_tail.err
head -c 400 $out/${tag}_detect_tail.json; echo; tail -3 $out/${tag}_detect_tail.err
