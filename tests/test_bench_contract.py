"""The driver reads bench.py's LAST stdout line as the contract object (a 30 KB line left BENCH_r03 unparsed).

GPU test: the default-shaped run (ResNet-50 bf16 b=256, kernel profile + CPU baseline on) prints exactly one line that
starts with {"metric", it is the last line, it is compact, and it carries `roofline` (a single kernel, with `alone` beside
it) and `cpu_baseline`; the per-kernel / per-layer tables live in the detail file instead.
Reference metric: /root/reference/trainer.py:232 (B * world / step.avg)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_contract_line_is_last_compact_and_complete(tmp_path):
    env = dict(os.environ, CONVNET_AMD_EMULATE='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    detail = tmp_path / 'detail.json'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '3', '--cpu-steps', '1',
                        '--detail-out', str(detail)], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.split('\n') if l.strip()]
    last = lines[-1]
    assert last.startswith('{"metric"') and len(last) < 8192, (len(last), last[:300])
    assert sum(l.startswith('{"metric"') for l in lines) == 1
    rec = json.loads(last)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'transport_fallback'):
        assert k in rec, k
    assert rec['steps'] == 4 and rec['warmup'] == 3 and rec['n_gpus'] == 1 and rec['dtype'] == 'bf16'
    assert rec['transport_fallback'] is False
    assert abs(rec['value'] - 256 * 1e3 / rec['ms_per_step']) / rec['value'] < 1e-3
    roof = rec['roofline']
    assert 0.0 < roof['frac'] <= 1.0 and roof['bound'] in ('hbm', 'mfma') and roof['peak'] > 0
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-3
    # one kernel, not a composite call label
    base = roof['kernel'].split(' (')[0].split(' [')[0]
    assert '+' not in base.split('<')[0] and ' + ' not in base, roof['kernel']
    assert roof['alone']['avg_us_per_launch'] > 0
    assert rec['cpu_baseline']['value'] > 0 and rec['cpu_baseline']['cores'] >= 1
    assert rec['cpu_baseline']['kind'] in ('port', 'reference')
    for k in ('kernels', 'kernels_overlapped', 'conv_layers'):
        assert k not in rec
    d = json.load(open(detail))
    assert d['kernels'] and d['kernels_overlapped'] and d['conv_layers'] and d['value'] == rec['value']


@pytest.mark.gpu
def test_contract_line_is_last_in_an_rccl_run_too(tmp_path):
    """A distributed run (here: a 1-rank RCCL group, the only one a 1-GPU box can build) brings RCCL's version banner into
    the process - five lines through C stdio that a piped stdout used to deliver BEHIND the contract line.  The line must
    still be the last one, and the only {"metric" line."""
    env = dict(os.environ, CONVNET_AMD_EMULATE='0', BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29571')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2', '--no-cpu-baseline',
                        '--no-kernel-profile'], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.split('\n') if l.strip()]
    assert lines[-1].startswith('{"metric"'), lines[-3:]
    assert sum(l.startswith('{"metric"') for l in lines) == 1
    rec = json.loads(lines[-1])
    assert rec['n_gpus'] == 1 and rec['transport_fallback'] is False and 'RCCL' in (rec['config']['transport'] or '')
