#!/bin/bash
# per-kernel table of the configs[4] forward / train steps: tools/trace_resnext.sh <tag>
tag=${1:-x}
root=$(pwd)
python3 tools/resnext_bench.py 8 2>/dev/null | tail -1 > gpurun_out/${tag}_resnext_bench.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_rx
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rx -- python3 $root/tools/resnext_bench.py 8 > /dev/null 2>&1
cd $root
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/prof_rx/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/${tag}_resnext_kernel_stats.txt", "w") as o:
    o.write("rocprofv3 --kernel-trace --stats -- python3 tools/resnext_bench.py 8  (whole process: 7 forwards at 8 x 513^2, 7 train steps at 8 x 321^2)\n")
    for r in rows[:28]:
        o.write("%6.2f%% %8d calls %10.1f us avg  %s\n" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
print(open("gpurun_out/${tag}_resnext_kernel_stats.txt").read()[:1800])
PY
cat gpurun_out/${tag}_resnext_bench.json
