#!/bin/bash
# round 6, third GPU call: operand probe, new tests (precision, tail worker, world-8 bench), file-fed run without settle steps
set -u
out=gpurun_out; mkdir -p $out
timeout 300 python3 tools/operand_probe.py > $out/r6_c_operand_probe.txt 2>&1; echo "probe rc=$?"; grep -v "^/opt" $out/r6_c_operand_probe.txt | head -130
timeout 900 python3 tools/precision_control.py --json $out/r6_c_precision_control.json > $out/r6_c_precision_control.txt 2>&1; echo "control rc=$?"; grep -A60 "^median" $out/r6_c_precision_control.txt
python3 -m pytest tests/test_precision_gpu.py tests/test_tail_gpu.py tests/test_parallel_gpu.py -m gpu -x -q > $out/r6_c_new_tests.log 2>&1; echo "new tests rc=$?"; tail -8 $out/r6_c_new_tests.log
python3 bench.py --data files --settle 0 --steps 16 --warmup 4 --no-strict --no-cpu-baseline > $out/r6_c_files_settle0.json 2> $out/r6_c_files_settle0.err; echo "files rc=$?"; tail -3 $out/r6_c_files_settle0.err
python3 - <<'PY'
import json
for f in ("r6_c_files_settle0",):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, d["value"], {k:c.get(k) for k in ("conv_saturated_blocks","clamped_and_skipped_steps","clamped_and_applied_blocks","timed_region_reruns","cold_start","loss_trace")})
    except Exception as e: print(f, "ERR", e)
PY
