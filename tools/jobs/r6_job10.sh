#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
python3 -m pytest tests/test_loader_gpu.py tests/test_loader_cpu.py -x -q > $out/r6_j_loader_tests.log 2>&1; echo "loader tests rc=$?"; tail -2 $out/r6_j_loader_tests.log
for r in 1 2; do
for v in 0 1; do
  SLN_LOADER_ASSEMBLE_STREAM=$v python3 bench.py --steps 20 --warmup 3 --data files --no-cpu-baseline --no-strict > $out/r6_j_files_as${v}_run$r.json 2> $out/r6_j_files_as${v}_run$r.err
  python3 -c "
import json; d=json.load(open('$out/r6_j_files_as${v}_run$r.json')); print('assemble_stream=$v run $r', d['value'], d['ms_per_step'], d['loader']['consumer_wait_ms_per_batch'], d['config']['clamped_and_skipped_steps'], d['roofline']['frac'])"
done; done
