#!/bin/bash
# same-box A/B of bench.py under debug knobs: tools/ab.sh "KNOB=a" "KNOB=b" ...   (each run: no strict leg, no cpu baseline)
export SLN_DEBUG_KNOBS=1
for kv in "$@"; do
  env $kv python bench.py --no-strict --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$kv', d['value'], d['ms_per_step'], 'dom', r['frac'], 'mfma', r['by_bound']['mfma_bound_launches']['frac_of_mfma_roofline'], 'hbm', r['by_bound']['hbm_bound_launches']['algorithmic_tb_per_s'])"
done
