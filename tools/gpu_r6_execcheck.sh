#!/bin/bash
# round 6: the sweeping tests and the fuzzer on the -DMEMO_EXEC_CHECK build of the A/B library (every branch-free row block tests EXEC on entry)
TAG=${1:-r6exec}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_execcheck_ab.so timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants or views or level_arrays or row_order or random or tile or scatter or packed_k or 120 or prepare or cycling or memb or planes or bucket_widths" 2>&1 | tail -4 | tee $OUT/pytest_execcheck.txt | cut -c1-300
MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_execcheck_ab.so timeout 300 python tests/fuzz_gpu.py --seconds 120 > $OUT/fuzz_execcheck.txt 2>&1; tail -2 $OUT/fuzz_execcheck.txt | cut -c1-400
