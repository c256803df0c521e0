#!/bin/bash
# the driver's three commands at the head
set -u
out=gpurun_out; mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/r6_n_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -2 $out/r6_n_gpu_suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/r6_n_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/r6_n_smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 3 > $out/r6_n_bench_n1.json 2> $out/r6_n_bench_n1.err; echo "bench rc=$?"; head -c 420 $out/r6_n_bench_n1.json; echo
python3 bench.py --config detect --steps 12 --warmup 3 > $out/r6_n_detect.json 2>/dev/null; python3 bench.py --config detect --tail --steps 12 --warmup 3 > $out/r6_n_detect_tail.json 2>/dev/null
python3 -c "
import json
for f in ('detect','detect_tail'):
    d=json.load(open('$out/r6_n_%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['config']['hip_graph'])
d=json.load(open('$out/r6_n_bench_n1.json')); print({k:d['roofline'][k] for k in d['roofline'] if k.startswith('roialign')})"
