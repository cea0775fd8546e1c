#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3m}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops.py -x -q -m gpu -k "stem or conv3x3_halo" 2>&1 | tail -3 | tee $OUT/pytest.txt
bash tools/gpu_r3g.sh ${1:-r3m}/prof 2>&1 | grep -E "stem|conv3x3|^[0-9]|total"
