#!/bin/bash
# (scratch driver of one gpurun call)
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5; do
  python3 -m pytest tests/test_multistep_gpu.py tests/test_zz_dynamics_gpu.py -m gpu -k "not config" -q > gpurun_out/r4_c_dyn_$i.log 2>&1
  tail -1 gpurun_out/r4_c_dyn_$i.log
done
python3 -m pytest tests/test_parallel_gpu.py -m gpu -q -s > gpurun_out/r4_c_parallel.log 2>&1; tail -2 gpurun_out/r4_c_parallel.log
timeout 600 python3 tools/graph_probe.py resnet101 1024 16 > gpurun_out/r4_c_graph_probe.log 2>&1; tail -4 gpurun_out/r4_c_graph_probe.log
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict > gpurun_out/r4_c_bench_base.json 2> gpurun_out/r4_c_bench_base.err; head -c 300 gpurun_out/r4_c_bench_base.json; echo
SLN_GLM_STREAM=1 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict > gpurun_out/r4_c_bench_glmstream.json 2> gpurun_out/r4_c_bench_glmstream.err; head -c 300 gpurun_out/r4_c_bench_glmstream.json; echo
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-strict > gpurun_out/r4_c_bench_base2.json 2>/dev/null; head -c 300 gpurun_out/r4_c_bench_base2.json; echo
