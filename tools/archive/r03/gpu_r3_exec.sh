#!/bin/bash
# round 3: the GPU tier's sweeping tests + 60 s of fuzz on the -DMEMO_EXEC_CHECK build of the AB library (every branch-free
# row block tests that EXEC is all ones on entry and reports through the index's status word), then the tests of this step
TAG=${1:-r3e}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu -k "dense or sidecar or realistic or golden_one_shot or goldens_on" 2>&1 | tail -4 | cut -c1-400
export MEMO_AMD_AB_LIB=$PWD/memo_amd/libmemo_amd_execcheck_ab.so
timeout 2400 python -m pytest tests -x -q -m gpu -k "resident_index_windows or packed_rows_equal or dense or bucket_widths or level_arrays or randomized or config5 or ragged" 2>&1 | tail -4 | tee $OUT/execcheck_pytest.txt | cut -c1-400
timeout 200 python tests/fuzz_gpu.py --seconds 90 > $OUT/execcheck_fuzz.txt 2>&1; tail -2 $OUT/execcheck_fuzz.txt | cut -c1-300
