#!/bin/bash
# round 6, second GPU call: new tests (clamp veto, batched tail), which steady-state fusion carries the 3.7x, detect tail, cold start
set -u
out=gpurun_out; mkdir -p $out
python3 -m pytest tests/test_optim_gpu.py tests/test_tail_gpu.py "tests/test_model_gpu.py::test_a_train_step_with_a_clamped_operand_block_is_skipped_not_applied" tests/test_model_gpu.py::test_train_step_runs_updates_and_stays_finite -m gpu -x -q > $out/r6_b_new_tests.log 2>&1; echo "new tests rc=$?"; tail -15 $out/r6_b_new_tests.log
timeout 1200 python3 tools/precision_subsets.py --scene 0 --repeats 1 --only 0,7,8,9,10,11 --json $out/r6_b_precision_fusions_0.json > $out/r6_b_precision_fusions_0.txt 2>&1; echo "fusions rc=$?"
grep -v "^/opt" $out/r6_b_precision_fusions_0.txt | cut -c1-400
python3 bench.py --config detect --steps 12 --warmup 3 > $out/r6_b_detect.json 2> $out/r6_b_detect.err; echo "detect rc=$?"; head -c 300 $out/r6_b_detect.json; echo
python3 bench.py --config detect --tail --steps 12 --warmup 3 > $out/r6_b_detect_tail.json 2> $out/r6_b_detect_tail.err; echo "detect tail rc=$?"; head -c 300 $out/r6_b_detect_tail.json; echo; tail -5 $out/r6_b_detect_tail.err
python3 bench.py --data files --settle 0 --cold-start --steps 12 --warmup 4 --no-strict --no-cpu-baseline > $out/r6_b_cold_files.json 2> $out/r6_b_cold_files.err; echo "cold rc=$?"; head -c 300 $out/r6_b_cold_files.json; echo; tail -5 $out/r6_b_cold_files.err
python3 - <<'PY'
import json
for f in ("r6_b_cold_files",):
    try:
        d=json.load(open("gpurun_out/%s.json"%f)); c=d["config"]
        print(f, d["value"], {k:c.get(k) for k in ("conv_saturated_blocks","clamped_and_skipped_steps","clamped_and_applied_blocks","timed_region_reruns","cold_start","loss_trace")})
    except Exception as e: print(f, "ERR", e)
PY
