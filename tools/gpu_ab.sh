#!/bin/bash
# One gpurun call: optional op tests, then an interleaved whole-step A/B (tools/gpu_ab.py), then one profiled bench run.
#   tools/gpu_ab.sh TAG "pytest -k expression or -" ROUNDS "tagA ENV=v ..." "tagB ENV=v ..." ...
export TMPDIR=/tmp
TAG=$1; KEXPR=$2; ROUNDS=$3; shift 3
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$KEXPR" != "-" ]; then
  echo "== tests -k '$KEXPR'"; timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$KEXPR" 2>&1 | tail -6 | tee $OUT/tests.txt
fi
echo "== A/B"; AB_STEPS=${AB_STEPS:-30} timeout 2400 python tools/gpu_ab.py $TAG $ROUNDS "$@" 2>&1 | tee $OUT/ab.txt
if [ -n "$DETAIL_ENV" ]; then
  echo "== detail ($DETAIL_ENV)"; env $DETAIL_ENV timeout 600 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --detail-out $OUT/bench_detail.json | tail -c 300
fi
