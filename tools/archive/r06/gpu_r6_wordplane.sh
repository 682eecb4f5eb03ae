#!/bin/bash
# round 6, VERDICT r05 item 5: the membership planes with a word plane (long runs: first word, last word, one 64-bit run of word bits)
# against the plain stores of whole words -- parity first, then the A/B in one process (tools/ab.py: scatter 6 = never, 7 = wherever it fits)
TAG=${1:-r6wp}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "membership or planes or word_boundaries or many_genomes or randomized or config3_full or packed_k_class or ragged or golden_one_shot or resident_index" 2>&1 | tail -6 | tee $OUT/pytest.txt
for k in 101 128 80 66 65 48; do
  echo "== c4 k=$k" >> $OUT/ab.txt
  timeout 600 python tools/ab.py --workload c4 --k $k --pack only --prepare --rounds 40 "1024,4,0,0,6" "1024,4,0,0,7" >> $OUT/ab.txt 2>> $OUT/ab.err
done
python3 - <<PY
import json
for l in open("$OUT/ab.txt"):
    if l.startswith("=="): print(l.strip()[:150]); continue
    j = json.loads(l); print("   %-22s %.4f ms (min %.4f)  frac %.3f  sweep %s rows %d" % (j["variant"], j["ms_median"], j["ms_min"], j["frac_of_8TBs"], j["last_sweep"], j["last_rows_read"]))
PY
grep -v amdgpu.ids $OUT/ab.err | tail -5
