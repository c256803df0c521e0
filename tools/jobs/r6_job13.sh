#!/bin/bash
set -u
out=gpurun_out; mkdir -p $out
python3 -m pytest "tests/test_model_gpu.py::test_bench_config_detect_prints_the_contract_line" "tests/test_resnext_gpu.py::test_bench_config_resnext_prints_the_contract_line" "tests/test_configs_gpu.py::test_bench_config_resnext_fp16_at_its_default_size_prints_the_contract_line" tests/test_loader_gpu.py -m gpu -x -q > $out/r6_m_contract_tests.log 2>&1; echo "rc=$?"; tail -3 $out/r6_m_contract_tests.log
