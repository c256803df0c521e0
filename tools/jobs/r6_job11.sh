#!/bin/bash
# the five-step replay's second scene under the same policies (scene 0: r6_a_precision_subsets_0.txt)
set -u
out=gpurun_out; mkdir -p $out
timeout 1500 python3 tools/precision_subsets.py --scene 1 --repeats 1 --only 0,2,4,6 --json $out/r6_k_precision_subsets_1.json > $out/r6_k_precision_subsets_1.txt 2>&1; echo "rc=$?"
grep -v "^/opt" $out/r6_k_precision_subsets_1.txt | cut -c1-330
