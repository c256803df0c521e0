#!/bin/bash
# round 6, first GPU call: the suite at the groundwork commit, then the fp32-class control tables
set -u
out=gpurun_out; mkdir -p $out
python3 -m pytest tests -m gpu -x -q > $out/r6_a_gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 $out/r6_a_gpu_suite.log
grep -E "^(FAILED|ERROR)" $out/r6_a_gpu_suite.log | head
timeout 900 python3 tools/precision_control.py --json $out/r6_a_precision_control.json > $out/r6_a_precision_control.txt 2>&1; echo "control rc=$?"
timeout 600 python3 tools/precision_control.py --headroom 0 --json $out/r6_a_precision_control_h0.json > $out/r6_a_precision_control_h0.txt 2>&1; echo "control h0 rc=$?"
timeout 1500 python3 tools/precision_subsets.py --scene 0 --repeats 2 --json $out/r6_a_precision_subsets_0.json > $out/r6_a_precision_subsets_0.txt 2>&1; echo "subsets rc=$?"
tail -20 $out/r6_a_precision_subsets_0.txt
