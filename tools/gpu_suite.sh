#!/bin/bash
# One gpurun call: the whole GPU test suite, smoke(), then an optional interleaved A/B.  tools/gpu_suite.sh TAG [ROUNDS specs...]
export TMPDIR=/tmp
TAG=${1:-suite}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
echo "== gpu tests"; timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
if [ -n "$1" ]; then R=$1; shift; echo "== A/B"; AB_STEPS=${AB_STEPS:-30} timeout 1800 python tools/gpu_ab.py $TAG $R "$@" 2>&1 | tee $OUT/ab.txt; fi
