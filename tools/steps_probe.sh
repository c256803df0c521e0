#!/bin/bash
# does the rate depend on how long the bench runs?  same box, 8 / 40 / 8 / 80 timed steps
for k in 8 40 8 80; do
  python bench.py --steps $k --warmup 3 --no-cpu-baseline --no-strict 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps $k', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
