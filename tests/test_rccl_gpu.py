"""The data-parallel transport on real hardware (`-m gpu`).

* world-1 direct RCCL (cn_comm_* behind engine.BucketReducer): a 1-rank process group must train
  bit-identically to the non-distributed Trainer - same kernels, the bucket all-reduces are identities,
  only the stream choreography (communicator stream ordered behind the wgrad side stream and the main
  stream by events, joined before the optimizer step) differs.  Also bit-identical: the
  torch.distributed transport (CONVNET_AMD_COMM=torch) on the same group.
* cn_comm_allreduce (fp64, in-stream) and cn_comm_broadcast on the 1-rank communicator.
* `bench.py --gpus 2` with no launcher env starts its own ranks (here both on the one device of the
  test box, BENCH_SHARE_GPU=1 -> gloo) and reports n_gpus = 2.
* With >= 2 devices visible (`skipif` otherwise: the 1-GPU test boxes; the driver's 8-GPU node runs them):
  the SAME 2-rank bodies as tests/test_cli_and_dp.py, but one GPU per rank on the direct-RCCL transport -
  training vs the oracle's restatement of DistributedDataParallel semantics, SyncBatchNorm over
  cn_comm_allreduce vs the oracle on the global batch, and `bench.py --gpus 2` reporting
  "direct RCCL ... 2 ranks".  These make the first multi-rank execution of cn_comm_allreduce_bucket a test,
  not the scaling bench.
* A failing set-up stops the job: there is no silent second transport (CONVNET_AMD_COMM=torch is explicit).
  bench.py alone recovers - loudly: stderr + the JSON line's transport carry the reason - so that a scaling run is
  not left without a number.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from helpers import ROOT, DP_WORKER, check_dp_against_oracle, check_syncbn_against_oracle

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import convnet_amd as ca
torch.cuda.set_device(0)
assert not ca._lib.is_emulated()
kw = dict(depth=50, width=(16, 32, 64, 128), inplanes=16, num_classes=32)
g = torch.Generator().manual_seed(5)
data = [(torch.randn(16, 3, 64, 64, generator=g), torch.randint(0, 32, (16,), generator=g)) for _ in range(3)]

def run(distributed, dtype):
    torch.manual_seed(123)
    model = ca.models.resnet(**kw)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device='cuda:0',
                    dtype=dtype, distributed=distributed, local_rank=0, grad_clip=5.0, print_freq=10**9,
                    bucket_mb=0.25)
    recs = [tr.train([b]) for b in data]
    torch.cuda.synchronize()
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    return recs, sd, tr

out = {}
for dtype in (torch.float32, torch.bfloat16):
    base_recs, base_sd, _ = run(False, dtype)
    if not dist.is_initialized():
        dist.init_process_group('nccl', init_method='env://', world_size=1, rank=0)
    recs, sd, tr = run(True, dtype)
    want_direct = os.environ.get('CONVNET_AMD_COMM', 'rccl') != 'torch'
    assert (tr.reducer.comm is not None) == want_direct, tr.reducer.describe()
    assert len(tr.arena.buckets) > 2
    for a, b in zip(base_recs, recs):
        assert a['loss'] == b['loss'] and a['grad'] == b['grad'] and a['prec1'] == b['prec1'], (a, b)
    for k in base_sd:
        assert torch.equal(base_sd[k], sd[k]), k
    out[str(dtype)] = tr.reducer.describe()
    if want_direct:
        c = tr.reducer.comm
        t = torch.arange(7, dtype=torch.float64, device='cuda:0') * 0.5
        ref = t.clone()
        c.allreduce_(t)                       # 1 rank: identity, through ncclAllReduce on the compute stream
        c.broadcast_(t, 0)
        torch.cuda.synchronize()
        assert torch.equal(t, ref)
        assert c.rccl_version >= 20000, c.rccl_version
print('TRANSPORT', out)
ca.comm.destroy_default()
dist.destroy_process_group()
'''


def _run_worker(tmp_path, extra_env, port):
    script = tmp_path / 'rccl_worker.py'
    script.write_text(WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CONVNET_AMD_EMULATE='0', **extra_env)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_world1_direct_rccl_is_bit_identical_to_single_process(tmp_path):
    out = _run_worker(tmp_path, {}, 29541)
    assert 'direct RCCL' in out, out


def test_world1_torch_distributed_transport_is_bit_identical_too(tmp_path):
    out = _run_worker(tmp_path, {'CONVNET_AMD_COMM': 'torch'}, 29543)
    assert 'torch.distributed (nccl)' in out, out


def test_bench_self_launches_its_ranks(tmp_path):
    env = dict(os.environ, BENCH_SHARE_GPU='1', CONVNET_AMD_EMULATE='0')
    env.pop('WORLD_SIZE', None)
    env.pop('RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--batch', '8', '--steps', '2',
                        '--warmup', '1', '--no-cpu-baseline', '--no-kernel-profile'], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.split('\n') if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['global_batch'] == 16 and rec['scaling'] == 'weak'
    assert rec['config']['parallelism'] == 'dp2' and 'gloo' in rec['config']['transport']
    # a launcher environment that disagrees with --gpus is refused instead of silently measuring 1 GPU
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '1'],
                        env=dict(env, WORLD_SIZE='1', RANK='0'), capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and 'WORLD_SIZE' in (r2.stdout + r2.stderr)


def test_bench_eight_rank_protocol_under_the_drivers_launcher(tmp_path):
    """the 8-GPU run is the driver's, on a node this build never sees - so its PROTOCOL is exercised
    here with eight ranks sharing the one device (BENCH_SHARE_GPU=1: gloo collectives, RCCL refuses two ranks per device),
    launched exactly as the driver launches it (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 --steps K --warmup W`): rendezvous, WORLD_SIZE / RANK /
    LOCAL_RANK handling, the barrier-bracketed timed region with the MAX over ranks, and ONE contract line, from rank 0
    only, that says n_gpus = 8, global batch = 8 x per-GPU batch, weak scaling, `transport_fallback` false (gloo here is
    asked for, not a recovery) and which device every rank used.  Reference: /root/reference/main.py:145-156,
    trainer.py:79-82."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, BENCH_SHARE_GPU='1', CONVNET_AMD_EMULATE='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '8', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '3',
           '--warmup', '2', '--batch', '4', '--no-cpu-baseline', '--no-kernel-profile']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.split('\n') if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 only
    assert [l for l in r.stdout.split('\n') if l.strip()][-1] == lines[0]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 8 and rec['steps'] == 3 and rec['warmup'] == 2 and rec['scaling'] == 'weak'
    assert rec['config']['global_batch'] == 32 and rec['config']['parallelism'] == 'dp8'
    assert rec['config']['rank_devices'] == [0] * 8          # (a full node: [0, 1, ..., 7])
    assert rec['config']['params_in_sync_across_ranks'] is True     # eight shards, one set of weights after 5 steps
    assert 'gloo' in rec['config']['transport'] and rec['transport_fallback'] is False
    assert abs(rec['value'] - 32 * 1e3 / rec['ms_per_step']) / rec['value'] < 1e-3
    assert len(lines[0]) < 8192


# ------------------------------------------------------------------------------------------------------------------
# >= 2 devices: the direct-RCCL transport with more than one rank
NDEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
two_gpus = pytest.mark.skipif(NDEV < 2, reason='needs >= 2 HIP devices (RCCL refuses two ranks on one device)')


def _run_two_ranks(tmp_path, worker, port, name):
    script = tmp_path / (name + '.py')
    out_pat = str(tmp_path / (name + '_rank%d.pt'))
    script.write_text(worker % {'root': ROOT, 'out': out_pat, 'dev': 'cuda:%d', 'backend': 'nccl'})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', OMP_NUM_THREADS='2',
               CONVNET_AMD_EMULATE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('CONVNET_AMD_COMM', None)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    return [torch.load(out_pat % r) for r in range(2)]


@two_gpus
def test_two_ranks_direct_rccl_training_matches_oracle_ddp(tmp_path):
    outs = _run_two_ranks(tmp_path, DP_WORKER, 29551, 'dp_rccl')
    for o in outs:
        assert 'direct RCCL' in o['transport'] and '2 ranks' in o['transport'], o['transport']
    assert outs[0]['nbuckets'] > 1, 'the tiny bucket size must exercise multi-bucket overlap'
    for k in outs[0]['sd']:   # replicas stay bit-identical (same broadcast weights, same reduced gradients)
        if 'running' in k or 'num_batches' in k:
            continue
        assert torch.equal(outs[0]['sd'][k], outs[1]['sd'][k]), k
    check_dp_against_oracle(outs)


@two_gpus
def test_two_ranks_sync_batchnorm_over_cn_comm_allreduce(tmp_path):
    worker = DP_WORKER.replace("tr = ca.Trainer(", "ca.nn.convert_sync_batchnorm(model)\ntr = ca.Trainer(", 1)
    assert worker != DP_WORKER
    outs = _run_two_ranks(tmp_path, worker, 29553, 'sync_rccl')
    for o in outs:
        assert 'direct RCCL' in o['transport'], o['transport']
    for k in outs[0]['sd']:
        if 'num_batches' in k:
            continue
        assert torch.equal(outs[0]['sd'][k], outs[1]['sd'][k]), k
    check_syncbn_against_oracle(outs)


@two_gpus
def test_bench_two_gpus_runs_on_direct_rccl(tmp_path):
    env = dict(os.environ, CONVNET_AMD_EMULATE='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'BENCH_SHARE_GPU', 'CONVNET_AMD_COMM'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--batch', '32', '--steps', '3',
                        '--warmup', '2', '--no-cpu-baseline', '--no-kernel-profile'], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([l for l in r.stdout.split('\n') if l.startswith('{"metric"')][0])
    assert rec['n_gpus'] == 2 and rec['config']['global_batch'] == 64 and rec['config']['parallelism'] == 'dp2'
    assert 'direct RCCL' in rec['config']['transport'] and '2 ranks' in rec['config']['transport'], rec['config']


def test_failed_rccl_setup_raises_instead_of_falling_back(tmp_path):
    """No silent second transport: with the RCCL library unresolvable the communicator set-up raises on the rank
    (and, through the agreement step, on every rank).  CN_RCCL_LIB points the loader at a file that is not there."""
    code = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
import convnet_amd as ca
torch.cuda.set_device(0)
dist.init_process_group('nccl', init_method='env://', world_size=1, rank=0)
try:
    ca.comm.create_default(torch.device('cuda', 0))
except ca._lib.ConvNetHipError as e:
    print('RAISED', e)
else:
    print('NO ERROR')
dist.destroy_process_group()
""" % ROOT
    script = tmp_path / 'fail_worker.py'
    script.write_text(code)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29557', CONVNET_AMD_EMULATE='0',
               CN_RCCL_LIB=str(tmp_path / 'no_such_librccl.so'))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert 'RAISED' in r.stdout and 'load librccl' in r.stdout, r.stdout[-2000:]


def test_bench_recovers_loudly_when_the_direct_communicator_cannot_be_built(tmp_path):
    """bench.py ONLY (the product stops, see the test above): when the direct RCCL communicator cannot be built - here
    the loader is pointed at a file that is not there - the bench continues on the torch.distributed collectives, says so
    on stderr and labels the JSON line's transport with the reason, instead of leaving a scaling run without a number."""
    env = dict(os.environ, BENCH_FORCE_DIST='1', CONVNET_AMD_EMULATE='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29561',
               CN_RCCL_LIB=str(tmp_path / 'no_such_librccl.so'))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'CONVNET_AMD_COMM'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '2',
                        '--no-cpu-baseline', '--no-kernel-profile'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.split('\n') if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    rec = json.loads(lines[0])
    tr = rec['config']['transport']
    assert rec['n_gpus'] == 1 and rec['value'] > 0
    assert 'torch.distributed' in tr and 'direct RCCL communicator failed' in tr and 'load librccl' in tr, tr
    assert 'bench.py[rank 0]' in r.stderr and 'direct RCCL communicator failed' in r.stderr
