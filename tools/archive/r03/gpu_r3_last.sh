#!/bin/bash
# round 3, last pass: long fuzz on the final build; the GPU tier + fuzz on the -DMEMO_EXEC_CHECK build; bench on the driver's command line
TAG=${1:-r3last}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 800 python tests/fuzz_gpu.py --seconds ${2:-600} > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-600
timeout 300 python tests/fuzz_dap_gpu.py --seconds 60 > $OUT/fuzz_dap.txt 2>&1; tail -1 $OUT/fuzz_dap.txt | cut -c1-300
MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_execcheck_ab.so timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $OUT/execcheck_pytest.txt | cut -c1-300
MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_execcheck_ab.so timeout 300 python tests/fuzz_gpu.py --seconds 150 > $OUT/execcheck_fuzz.txt 2>&1; tail -2 $OUT/execcheck_fuzz.txt | cut -c1-300
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $OUT/pytest_gpu.txt | cut -c1-300
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-260 $OUT/bench_driver.json
