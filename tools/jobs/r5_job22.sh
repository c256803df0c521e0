#!/bin/bash
# the driver's own commands at the head: smoke, the suite with -x, the default bench line
set -u
tag=${1:-r5_z}
out=$(pwd)/gpurun_out
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/${tag}_smoke.log
python3 -m pytest tests -x -q -m gpu > $out/${tag}_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -1 $out/${tag}_gpu_suite.log
python3 bench.py --gpus 1 --steps 20 --warmup 3 > $out/${tag}_bench_n1.json 2> $out/${tag}_bench_n1.err; echo "bench rc=$?"
python3 -c "import json;d=json.load(open('$out/${tag}_bench_n1.json'));r=d['roofline'];print(d['value'], d['ms_per_step'], r['frac'], r['traffic_source'], d['config']['conv_saturated_blocks'], d['cpu_baseline']['value'])"
