#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
for r in 1 2; do for v in 0 2 1; do
  SLN_GLM_STREAM=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-strict > $out/r6_p_glmstream${v}_run$r.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('$out/r6_p_glmstream${v}_run$r.json')); print('SLN_GLM_STREAM=$v run $r', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('avg_launch_us'))"
done; done
