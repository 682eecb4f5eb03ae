#!/bin/bash
# round 3: dense rows on every way in (builder, one-shot, cache v2, CLI), parity additions
TAG=${1:-r3b}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 2400 python -m pytest tests -x -q -m gpu -k "dense or builder or sidecar or golden or cli or multi_device or region_index or integration" 2>&1 | tail -15 | tee $OUT/pytest.txt | cut -c1-400
