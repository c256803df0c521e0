#!/bin/bash
# Counter evidence of one revision (run on the GPU box through gpurun, from the repo root):
#   tools/collect_pmc.sh <tag>     ->  gpurun_out/<tag>_pmc_traffic.json, <tag>_pmc_mfma.json, <tag>_pmc_roialign.json
# One counter per pass (MI355X_MICROARCH.md, HBM / rocprofv3 section), --kernel-trace only next to --pmc, the
# program itself after "--" (python3, no env / shell hop).
set -u
tag=${1:-r3}
root=$(pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE MfmaUtil MfmaFlopsF16; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $root/bench.py --steps 2 --warmup 1 --settle 2 --no-cpu-baseline --no-strict > $out/${tag}_pmc_${c}_bench.json 2> /tmp/pmc_$c.err
    echo "pass $c rc=$?" >> $out/${tag}_pmc_passes.txt
done
cd $root
python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE > $out/${tag}_pmc_traffic.json 2>> $out/${tag}_pmc_passes.txt
ms=$(python3 -c "import json;print(json.load(open('$out/${tag}_pmc_MfmaUtil_bench.json'))['ms_per_step'])" 2>/dev/null || echo 200)
cd tools && python3 pmc_mfma.py /tmp/pmc_MfmaUtil /tmp/pmc_MfmaFlopsF16 ${PMC_MS:-$ms} > $out/${tag}_pmc_mfma.json 2>> $out/${tag}_pmc_passes.txt; cd $root
# RoIAlign / label decode with measured traffic next to the algorithmic figure
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcr_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcr_$c -- python3 $root/tools/roialign_bench.py > /tmp/pmcr_$c.log 2>&1
    echo "roialign pass $c rc=$?" >> $out/${tag}_pmc_passes.txt
done
cd $root
python3 tools/pmc_roialign.py /tmp/pmcr_FETCH_SIZE /tmp/pmcr_WRITE_SIZE > $out/${tag}_pmc_roialign.json 2>> $out/${tag}_pmc_passes.txt
cat $out/${tag}_pmc_passes.txt
