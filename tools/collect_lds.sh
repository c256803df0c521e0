#!/bin/bash
# LDS bank-conflict counters of one revision (run on the GPU box through gpurun, from the repo root):
#   tools/collect_lds.sh <tag>   ->  gpurun_out/<tag>_pmc_lds.json     (one counter per pass, --kernel-trace only next to --pmc)
set -u
tag=${1:-r4}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-strict > /tmp/pmc_$c.json 2> /tmp/pmc_$c.err
    echo "pass $c rc=$?"
done
cd $root/tools && python3 pmc_lds.py /tmp/pmc_SQ_LDS_BANK_CONFLICT /tmp/pmc_SQ_LDS_IDX_ACTIVE > $out/${tag}_pmc_lds.json
