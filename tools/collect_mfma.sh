#!/bin/bash
# only the two matrix-pipe counter passes of tools/collect_pmc.sh:  tools/collect_mfma.sh <tag>
set -u
tag=${1:-r3}
root=$(pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in MfmaUtil MfmaFlopsF16; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strict > $out/${tag}_pmc_${c}_bench.json 2> /tmp/pmc_$c.err
    echo "pass $c rc=$?"
done
cd $root/tools && python3 pmc_mfma.py /tmp/pmc_MfmaUtil /tmp/pmc_MfmaFlopsF16 ${PMC_MS:-180.4} > $out/${tag}_pmc_mfma.json; cd $root
python3 -c "
import json; m=json.load(open('gpurun_out/${tag}_pmc_mfma.json'))
for k,v in m.items():
    if isinstance(v,dict) and 'MfmaUtil_mean_percent' in v: print(k, v['launches'], v['MfmaUtil_mean_percent'], v['MfmaUtil_time_weighted_percent'])
print(m['whole_step'])"
