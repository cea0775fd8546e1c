"""The main.py CLI surface + checkpoint round trip, and the data-parallel path on 2 ranks against an
oracle-side restatement of DistributedDataParallel semantics (per-rank BN statistics, gradients averaged
over ranks, SURVEY.md section 8e / level T4).

Every test exists in two modes: 'emul' (no GPU: kernel sources through the TEST-ONLY emulator, gloo) and
'gpu' (`-m gpu`: libconvnet_hip.so on cuda:0; the two ranks share the one device of the test box, so their
process group is gloo - RCCL refuses two ranks on one device - while the world-1 RCCL tests in
tests/test_rccl_gpu.py cover the direct-RCCL transport)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import HAS_GPU
from helpers import ROOT, rel_l2, check_dp_against_oracle, check_syncbn_against_oracle, DP_WORKER

MODES = [pytest.param('emul'), pytest.param('gpu', marks=pytest.mark.gpu)]


def _mode(mode):
    if mode == 'emul' and HAS_GPU:
        pytest.skip('emulator mode is for GPU-less hosts')
    if mode == 'gpu' and not HAS_GPU:
        pytest.skip('no GPU')
    return 'cuda' if mode == 'gpu' else 'cpu'

SMALL = "{'depth': 18, 'width': [8, 16, 32, 64], 'inplanes': 8, 'num_classes': 16}"


@pytest.mark.parametrize('mode', MODES)
def test_cli_train_checkpoint_resume(mode, tmp_path):
    dev = _mode(mode)
    import convnet_amd as ca
    from convnet_amd.main import main
    common = ['--model', 'resnet', '--model-config', SMALL, '--input-size', '32', '-b', '4', '--device', dev,
              '--steps-per-epoch', '2', '--val-steps', '1', '--results-dir', str(tmp_path), '--print-freq', '1']
    out = main(common + ['--save', 'run', '--epochs', '1'])
    run = tmp_path / 'run'
    for f in ('config.json', 'log.txt', 'results.csv', 'checkpoint.pth.tar'):   # model_best only when prec1 improves
        assert (run / f).exists(), f
    ck = torch.load(run / 'checkpoint.pth.tar', map_location='cpu')
    assert set(ck) == {'epoch', 'model', 'config', 'state_dict', 'optim_state_dict', 'best_prec1'}
    assert ck['epoch'] == 1 and ck['model'] == 'resnet'
    assert ck['state_dict']['conv1.weight'].shape == (8, 3, 7, 7)          # reference OIHW shape
    assert ck['state_dict']['layer1.0.conv1.weight'].shape == (8, 8, 3, 3)
    assert set(out['train']) >= {'step', 'data', 'loss', 'prec1', 'prec5', 'error1', 'error5'}
    # the checkpoint loads into the ORACLE model (reference state_dict layout) and evaluates equally
    from oracle import convnet_oracle as O
    oracle = O.OracleResNet(18, 16, 8, (8, 16, 32, 64))
    oracle.load_state_dict(ck['state_dict'])
    val = main(common + ['--save', 'ev', '-e', str(run / 'checkpoint.pth.tar')])
    from convnet_amd.main import SyntheticLoader
    data = list(SyntheticLoader(1, 4, 32, 16, 3, 123 + 10000))
    ref = O.oracle_validate(oracle, data)
    assert val['loss'] == pytest.approx(ref['loss'], rel=1e-4) and val['prec1'] == ref['prec1']
    # resume continues from the saved epoch with the optimizer state
    out2 = main(common + ['--save', 'run2', '--epochs', '2', '--resume', str(run / 'checkpoint.pth.tar')])
    ck2 = torch.load(tmp_path / 'run2' / 'checkpoint.pth.tar', map_location='cpu')
    assert ck2['epoch'] == 2




def _worker_env(dev, port):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', OMP_NUM_THREADS='2')
    env['CONVNET_AMD_EMULATE'] = '1' if dev == 'cpu' else '0'
    return env


@pytest.mark.parametrize('mode', MODES)
def test_data_parallel_two_ranks_gloo(mode, tmp_path):
    dev = _mode(mode)
    script = tmp_path / 'dp_worker.py'
    out_pat = str(tmp_path / 'rank%d.pt')
    script.write_text(DP_WORKER % {'root': ROOT, 'out': out_pat, 'dev': dev if dev == 'cpu' else 'cuda:0', 'backend': 'gloo'})
    env = _worker_env(dev, 29531)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    outs = [torch.load(out_pat % r) for r in range(2)]
    assert outs[0]['nbuckets'] > 1, 'the tiny bucket size must exercise multi-bucket overlap'
    # replicas stay bit-identical (same broadcast weights, same reduced gradients)
    for k in outs[0]['sd']:
        if 'running' in k or 'num_batches' in k:
            continue
        assert torch.equal(outs[0]['sd'][k], outs[1]['sd'][k]), k
    check_dp_against_oracle(outs)


@pytest.mark.parametrize('mode', MODES)
def test_sync_batchnorm_two_ranks_equals_global_batch(mode, tmp_path):
    """--sync-bn (main.py:190-191, nn.SyncBatchNorm): 2 ranks x 4 samples with synchronised batch
    statistics train exactly like one process on the 8-sample batch (statistics, input gradients and,
    after the data-parallel averaging, parameter gradients).  Oracle = the plain-PyTorch model on the
    full batch."""
    dev = _mode(mode)
    script = tmp_path / 'sync_worker.py'
    out_pat = str(tmp_path / 'sync_rank%d.pt')
    worker = DP_WORKER.replace("tr = ca.Trainer(", "ca.nn.convert_sync_batchnorm(model)\ntr = ca.Trainer(", 1)
    assert worker != DP_WORKER
    script.write_text(worker % {'root': ROOT, 'out': out_pat, 'dev': dev if dev == 'cpu' else 'cuda:0', 'backend': 'gloo'})
    env = _worker_env(dev, 29533)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    outs = [torch.load(out_pat % r) for r in range(2)]
    for k in outs[0]['sd']:   # with synchronised statistics even the running stats agree across ranks
        if 'num_batches' in k:
            continue
        assert torch.equal(outs[0]['sd'][k], outs[1]['sd'][k]), k
    check_syncbn_against_oracle(outs)
