#!/bin/bash
# flake hunt at the head: the driver's own commands, three times over
set -u
tag=${1:-r5_h}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/${tag}_smoke.log
for i in 1 2 3; do
  python3 -m pytest tests -x -q -m gpu > $out/${tag}_suite_x_$i.log 2>&1
  echo "run $i rc=$?"; tail -1 $out/${tag}_suite_x_$i.log; grep -E "^(FAILED|ERROR)" $out/${tag}_suite_x_$i.log | head -5
done
python3 bench.py --gpus 1 --steps 20 --warmup 3 > $out/${tag}_bench_n1.json 2> $out/${tag}_bench_n1.err
head -c 300 $out/${tag}_bench_n1.json; echo
