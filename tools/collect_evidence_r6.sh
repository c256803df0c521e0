#!/bin/bash
# Everything profiles/ holds for one round-6 revision, in one gpurun call (from the repo root on the GPU box):
#   tools/collect_evidence_r6.sh <tag>
# The GPU suite runs FIRST (the driver's own command: -x), into <tag>_gpu_suite.log; on a red run nothing else is produced.
set -u
tag=${1:-r6}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
if [ "${SKIP_SUITE:-0}" != "1" ]; then
    python3 -m pytest tests -m gpu -x -q > $out/${tag}_gpu_suite.log 2>&1
    rc=$?
    tail -3 $out/${tag}_gpu_suite.log
    if [ $rc -ne 0 ]; then
        echo "collect_evidence: GPU suite rc=$rc -- refusing to write bench lines"
        grep -E "^(FAILED|ERROR)" $out/${tag}_gpu_suite.log | head -20
        exit 1
    fi
fi
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/${tag}_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $out/${tag}_smoke.log
python3 bench.py --gpus 1 --steps 20 --warmup 3 > $out/${tag}_bench_n1.json 2> $out/${tag}_bench_n1.err
head -c 400 $out/${tag}_bench_n1.json; echo
python3 bench.py --steps 20 --warmup 3 --stage heads --no-cpu-baseline > $out/${tag}_bench_heads.json 2> $out/${tag}_bench_heads.err
head -c 300 $out/${tag}_bench_heads.json; echo
python3 bench.py --steps 20 --warmup 3 --data files --no-cpu-baseline --no-strict > $out/${tag}_bench_files.json 2> $out/${tag}_bench_files.err
head -c 300 $out/${tag}_bench_files.json; echo
python3 bench.py --steps 16 --warmup 4 --data files --settle 0 --no-cpu-baseline --no-strict > $out/${tag}_bench_files_settle0.json 2> $out/${tag}_bench_files_settle0.err
python3 bench.py --steps 16 --warmup 4 --data files --settle 0 --cold-start --no-cpu-baseline --no-strict > $out/${tag}_bench_files_cold_start.json 2> $out/${tag}_bench_files_cold_start.err
python3 bench.py --config resnext --parts 1 --steps 10 --warmup 3 > $out/${tag}_resnext_p1.json 2> $out/${tag}_resnext_p1.err
head -c 300 $out/${tag}_resnext_p1.json; echo
python3 bench.py --config resnext --parts 1 --steps 10 --warmup 3 --no-graph > $out/${tag}_resnext_p1_eager.json 2> $out/${tag}_resnext_p1_eager.err
python3 bench.py --config resnext --parts 2 --steps 10 --warmup 3 > $out/${tag}_resnext_p2.json 2> $out/${tag}_resnext_p2.err
python3 bench.py --config detect --steps 12 --warmup 3 > $out/${tag}_detect.json 2> $out/${tag}_detect.err
python3 bench.py --config detect --tail --steps 12 --warmup 3 > $out/${tag}_detect_tail.json 2> $out/${tag}_detect_tail.err
python3 - <<PY
import json
for f in ("bench_n1", "bench_heads", "bench_files", "bench_files_settle0", "bench_files_cold_start", "resnext_p1", "resnext_p1_eager", "resnext_p2", "detect", "detect_tail"):
    try:
        d = json.load(open("$out/${tag}_%s.json" % f)); c = d["config"]
        print("%-24s %9.2f %s  %8.2f ms  roofline %s  sat %s skipped %s reruns %s graph %s" % (
            f, d["value"], d["unit"], d["ms_per_step"], d.get("roofline", {}).get("frac"), c.get("conv_saturated_blocks"),
            c.get("clamped_and_skipped_steps"), c.get("timed_region_reruns"), c.get("hip_graph")))
    except Exception as e:
        print(f, "ERR", e)
PY
ms=$(python3 -c "import json;print(json.load(open('$out/${tag}_bench_n1.json'))['ms_per_step'])")
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_kt /tmp/prof_heads /tmp/prof_rx
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -- python3 $root/bench.py --steps 4 --warmup 1 --settle 4 --no-cpu-baseline --no-strict > $out/${tag}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_heads -- python3 $root/bench.py --stage heads --steps 4 --warmup 1 --settle 4 --no-cpu-baseline --no-strict > $out/${tag}_heads_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rx -- python3 $root/bench.py --config resnext --parts 1 --steps 4 --warmup 2 > $out/${tag}_resnext_p1_under_rocprof.json 2>/dev/null
cd $root
python3 tools/step_breakdown.py $(ls /tmp/prof_kt/*/*kernel_trace.csv | head -1) 80 > $out/${tag}_step_breakdown.txt
cp $(ls /tmp/prof_kt/*/*kernel_stats.csv | head -1) $out/${tag}_rocprofv3_kernel_stats.csv
python3 tools/step_breakdown.py $(ls /tmp/prof_heads/*/*kernel_trace.csv | head -1) 60 > $out/${tag}_heads_step_breakdown.txt
cp $(ls /tmp/prof_heads/*/*kernel_stats.csv | head -1) $out/${tag}_heads_rocprofv3_kernel_stats.csv
python3 tools/step_breakdown.py $(ls /tmp/prof_rx/*/*kernel_trace.csv | head -1) 40 > $out/${tag}_resnext_p1_step_breakdown.txt
cp $(ls /tmp/prof_rx/*/*kernel_stats.csv | head -1) $out/${tag}_resnext_p1_rocprofv3_kernel_stats.csv
PMC_MS=$ms tools/collect_pmc.sh $tag > $out/${tag}_collect.log 2>&1
SLN_PROFILE_SHAPES=1 python3 bench.py --no-cpu-baseline --no-strict > /dev/null 2> $out/${tag}_shapes.txt
head -12 $out/${tag}_step_breakdown.txt; head -8 $out/${tag}_resnext_p1_step_breakdown.txt; cat $out/${tag}_pmc_passes.txt
