#!/bin/bash
# Round-4 first call: the contract-line test, the default bench line, environment.
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r4a}
mkdir -p $OUT
{ date; python -c "import torch;print('torch',torch.__version__,'devices',torch.cuda.device_count(),torch.cuda.get_device_name(0))"; nproc; } > $OUT/env.txt 2>&1
echo "== contract test"; timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/contract.txt
echo "== bench"; timeout 900 python bench.py --steps 20 --warmup 5 --detail-out $OUT/bench_detail.json > $OUT/bench.out 2> $OUT/bench.err; tail -c 6000 $OUT/bench.out; echo; wc -c $OUT/bench.out; tail -3 $OUT/bench.err
echo "== lazy-dy / trajectory spot tests"; timeout 900 python -m pytest tests/test_trajectory.py tests/test_step_local_consistency.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -5 | tee $OUT/traj.txt
echo "== done"; date
