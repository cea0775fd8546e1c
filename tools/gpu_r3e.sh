#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r3e}
mkdir -p $OUT
echo "== warm parity + local consistency"; timeout 1400 python -m pytest tests/test_warm_parity.py tests/test_step_local_consistency.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v "Warning\|warn\|detach\|total +=\|INFO\|^$" | tail -80 | tee $OUT/pytest_parity.txt
echo "== done"; date
