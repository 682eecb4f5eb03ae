#!/bin/bash
# round 4 validation: the sidecar cache tests (v3: k-class view in the file), the differential fuzzer on the product build, then the
# sweeping tests and the fuzzer on the -DMEMO_EXEC_CHECK build of the A/B library (every branch-free row block tests EXEC on entry)
TAG=${1:-r4valid}; SECS=${2:-300}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu -k "sidecar or cache or cli or fast" 2>&1 | tail -4 | tee $OUT/pytest_cache.txt
timeout $((SECS + 120)) python tests/fuzz_gpu.py --seconds $SECS > $OUT/fuzz.txt 2>&1; tail -2 $OUT/fuzz.txt | cut -c1-400
MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_execcheck_ab.so timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "variants or views or level_arrays or row_order or random or tile or scatter or packed_k or 120 or prepare or cycling" 2>&1 | tail -4 | tee $OUT/pytest_execcheck.txt
MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_execcheck_ab.so timeout 300 python tests/fuzz_gpu.py --seconds 120 > $OUT/fuzz_execcheck.txt 2>&1; tail -2 $OUT/fuzz_execcheck.txt | cut -c1-400
