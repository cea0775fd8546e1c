"""Whole-step HIP graph (Trainer._graph_step; `-m gpu`): after two eager warm-up steps the device side of
the training step is captured once and replayed.  Replays must be bit-identical to the eager launches -
same kernels, same order, same streams - including across a learning-rate change (lr / momentum live in
device memory, so the schedule moves without a re-capture) and with the gradient all-reduce of a 1-rank
direct-RCCL group inside the graph."""
import os
import subprocess
import sys

import pytest
import torch

from helpers import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import convnet_amd as ca
torch.cuda.set_device(0)
kw = dict(depth=50, width=(16, 32, 64, 128), inplanes=16, num_classes=32)
g = torch.Generator().manual_seed(9)
data = [(torch.randn(16, 3, 64, 64, generator=g).cuda(), torch.randint(0, 32, (16,), generator=g).cuda())
        for _ in range(7)]
DIST = os.environ.get('TEST_DIST') == '1'
if DIST:
    dist.init_process_group('nccl', init_method='env://', world_size=1, rank=0)

def run(graph, dtype, clip):
    torch.manual_seed(123)
    model = ca.models.resnet(**kw)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(smooth_eps=0.1), ca.OptimRegime(model, model.regime),
                    device='cuda:0', dtype=dtype, distributed=DIST, local_rank=0, grad_clip=clip, loss_scale=4.0,
                    print_freq=10**9, bucket_mb=0.25)
    tr._use_graph = graph
    tr._graph_mode = '1' if graph else '0'     # force the capture (the default 'auto' decides by host vs device time)
    recs = []
    for i, b in enumerate(data):
        if i == 4:
            tr.epoch = 30            # models/resnet.py:253: lr 0.1 -> 0.01 at epoch 30
        r = tr.train([b])
        recs.append((r['loss'], r['prec1'], r.get('grad')))
    val = tr.validate(data[:2])      # eager evaluation right after replays must see the updated weights
    torch.cuda.synchronize()
    sd = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    return recs, val, sd, tr

for dtype in (torch.float32, torch.bfloat16):
    for clip in (-1, 5.0):
        e_recs, e_val, e_sd, _ = run(False, dtype, clip)
        g_recs, g_val, g_sd, tr = run(True, dtype, clip)
        assert any(g['graph'] is not None for g in tr._gstates.values()), 'the step was never captured'
        sts = [g['graph'] for g in tr._gstates.values() if g['graph'] is not None]
        if os.environ.get('TEST_EXPECT_PLAN') == '1':
            # the launch plan is what ran: this library's launches on TWO streams (the weight gradients kept their side
            # stream), torch's own kernels of the step (loss scaling: mul + its backward) imported, RCCL buckets as calls
            assert all(st.get('plan') is not None for st in sts)
            info = sts[0]['plan'].info()
            assert info[1] > 100 and info[6] == 2 and info[7] >= 3, info
            assert info[2] >= 1, info
            assert (info[5] > 0) == DIST, info
        else:
            assert all(st.get('plan') is None for st in sts)
        assert tr.optimizer.hyper['lr'] == 0.01
        assert e_recs == g_recs, (e_recs, g_recs)
        assert e_val['loss'] == g_val['loss'] and e_val['prec1'] == g_val['prec1']
        for k in e_sd:
            assert torch.equal(e_sd[k], g_sd[k]), k
print('GRAPH_OK', 'dist' if DIST else 'single')
if DIST:
    ca.comm.destroy_default()
    dist.destroy_process_group()
'''


def _run(tmp_path, env_extra, port):
    script = tmp_path / 'graph_worker.py'
    script.write_text(WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), CONVNET_AMD_EMULATE='0', **env_extra)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_graph_replay_is_bit_identical_to_eager(tmp_path):
    assert 'GRAPH_OK single' in _run(tmp_path, {'CONVNET_AMD_FLAGS': 'plan=0'}, 29551)


def test_graph_with_world1_rccl_buckets_inside(tmp_path):
    assert 'GRAPH_OK dist' in _run(tmp_path, {'TEST_DIST': '1', 'CONVNET_AMD_FLAGS': 'graph_dp=1,plan=0'}, 29553)


def test_graph_replay_with_lazy_dy_is_bit_identical_to_eager(tmp_path):
    """The same with the junction BatchNorms' backward apply left to the consumers (ops.LAZY_DY forced on for this
    small model): the placeholder gradients and the finalize-only BatchNorm calls are capture-safe."""
    assert 'GRAPH_OK single' in _run(tmp_path, {'CONVNET_AMD_FLAGS': 'lazy_min_mb=0,plan=0'}, 29555)


# ---- launch plan (csrc/plan.hip; the default): the same bit-identity, with the two-stream schedule kept ------------
def test_plan_replay_is_bit_identical_to_eager(tmp_path):
    assert 'GRAPH_OK single' in _run(tmp_path, {'TEST_EXPECT_PLAN': '1'}, 29561)


def test_plan_with_world1_rccl_buckets_issued_live(tmp_path):
    """distributed=True on the direct-RCCL communicator: the bucket all-reduces and the join are plan entries that call
    RCCL in every replay (never captured), default flags."""
    assert 'GRAPH_OK dist' in _run(tmp_path, {'TEST_DIST': '1', 'TEST_EXPECT_PLAN': '1'}, 29563)


def test_plan_replay_with_lazy_dy_is_bit_identical_to_eager(tmp_path):
    assert 'GRAPH_OK single' in _run(tmp_path, {'CONVNET_AMD_FLAGS': 'lazy_min_mb=0', 'TEST_EXPECT_PLAN': '1'}, 29565)


WATCH_WORKER = r'''
import sys, torch
sys.path.insert(0, %(root)r)
import convnet_amd as ca
torch.cuda.set_device(0)
kw = dict(depth=50, width=(16, 32, 64, 128), inplanes=16, num_classes=32)
g = torch.Generator().manual_seed(9)
data = [(torch.randn(16, 3, 64, 64, generator=g).cuda(), torch.randint(0, 32, (16,), generator=g).cuda())
        for _ in range(24)]

def run(mode):
    torch.manual_seed(123)
    model = ca.models.resnet(**kw)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device='cuda:0',
                    dtype=torch.bfloat16, print_freq=10**9)
    tr._graph_mode = mode
    tr._use_graph = mode != '0'
    key = None
    recs, states = [], []
    for i, b in enumerate(data):
        if mode == 'auto' and i == 5:
            (key,) = list(tr._gstates.keys())      # the one configuration this loop runs (shapes + step options)
            # Whatever auto decided for this small (host-bound) model: install the state a DEVICE-bound configuration
            # is in after its fourth step - eager verdict, watch armed - with a reference time no real step can meet.
            tr._gstates[key] = {'seen': {'n': 4, 'use': False, 'eager_ms': 1e-3}, 'graph': None}
            tr._graph_eager_for.add(key)
            tr._watch[key] = ca.trainer.EagerWatch(1e-3)
        recs.append(tr.train([b])['loss'])
        states.append((key in tr._graph_eager_for, key in tr._watch,
                       tr._gstates.get(key, {}).get('graph') is not None))
    torch.cuda.synchronize()
    return recs, states, tr

e_recs, _, _ = run('0')
a_recs, states, tr = run('auto')
assert e_recs == a_recs, (e_recs, a_recs)              # eager -> watched eager -> capture -> replays: the same numbers
assert states[5][0] and states[5][1] and not states[5][2]          # eager verdict in force, watched
fired = [i for i, s in enumerate(states) if i > 5 and not s[0]]
assert fired and 13 <= fired[0] <= 16, states                      # nine periods after the watch was armed
assert states[-1] == (False, False, True), states[-1]              # the graph was captured and kept (it IS faster here)
print('WATCH_OK', fired[0])
'''


def test_auto_mode_withdraws_an_eager_verdict_when_the_step_slows_down(tmp_path):
    """graph = auto: a configuration that was found device-bound keeps being watched (trainer.EagerWatch); when its step
    period stays above 1.2 x the time the verdict was based on, the step is captured after all, and the numbers do not
    change.  (The window logic itself: tests/test_flags.py.)"""
    script = tmp_path / 'watch_worker.py'
    script.write_text(WATCH_WORKER % {'root': ROOT})
    env = dict(os.environ, CONVNET_AMD_EMULATE='0')
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'WATCH_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


TWO_KEYS_WORKER = r'''
import sys, torch
sys.path.insert(0, %(root)r)
import convnet_amd as ca
torch.cuda.set_device(0)
kw = dict(depth=50, width=(16, 32, 64, 128), inplanes=16, num_classes=32)
g = torch.Generator().manual_seed(9)
big = [(torch.randn(16, 3, 64, 64, generator=g).cuda(), torch.randint(0, 32, (16,), generator=g).cuda()) for _ in range(4)]
small = [(torch.randn(8, 3, 64, 64, generator=g).cuda(), torch.randint(0, 32, (8,), generator=g).cuda()) for _ in range(4)]
torch.manual_seed(123)
model = ca.models.resnet(**kw)
tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device='cuda:0', dtype=torch.bfloat16,
                print_freq=10**9)
tr._graph_mode, tr._use_graph = 'auto', True
# two configurations, each with an 'eager' verdict under watch against a reference no step can exceed (1e9 ms)
keys = []
for b in (big[0], small[0]):
    tr.train([b])
    (k,) = [k for k in tr._gstates if k not in keys]
    keys.append(k)
    tr._gstates[k] = {'seen': {'n': 4, 'use': False, 'eager_ms': 1e9}, 'graph': None}
    tr._graph_eager_for.add(k)
    tr._watch[k] = ca.trainer.EagerWatch(1e9)
wa, wb = tr._watch[keys[0]], tr._watch[keys[1]]
# strictly alternating configurations: every period lies between marks of DIFFERENT watches -> nobody collects one
for i in range(12):
    tr.train([big[i %% 4] if i %% 2 == 0 else small[i %% 4]])
torch.cuda.synchronize()
tr.train([big[0]]); tr.train([small[0]])          # (polls what completed)
assert wa.periods == [] and wb.periods == [], (wa.periods, wb.periods)
# runs of one configuration: its watch collects periods (len - 1 per run of consecutive steps), the other stays empty
tr.train(big * 2)
torch.cuda.synchronize()
tr.train([big[0]])
assert len(wa.periods) >= 6 and wb.periods == [], (len(wa.periods), wb.periods)
assert all(0.0 < p < 1e4 for p in wa.periods), wa.periods
n_a = len(wa.periods)
c = ca.trainer.EagerWatch(1e9)                     # a third watch resets nobody
assert len(wa.periods) == n_a
print('TWO_KEYS_OK', n_a)
'''


def test_step_periods_are_attributed_to_the_configuration_they_belong_to(tmp_path):
    """the library's ring of timing marks is shared by every EagerWatch of the process.  Marks carry the watch's
    tag; a period whose two marks belong to different configurations (an odd-shaped last batch between full ones) is
    nobody's, a new watch discards nothing."""
    script = tmp_path / 'two_keys_worker.py'
    script.write_text(TWO_KEYS_WORKER % {'root': ROOT})
    env = dict(os.environ, CONVNET_AMD_EMULATE='0')
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'TWO_KEYS_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


FALLBACK_WORKER = r'''
import sys, torch
sys.path.insert(0, %(root)r)
import convnet_amd as ca
torch.cuda.set_device(0)
kw = dict(depth=50, width=(16, 32, 64, 128), inplanes=16, num_classes=32)
g = torch.Generator().manual_seed(9)
data = [(torch.randn(16, 3, 64, 64, generator=g).cuda(), torch.randint(0, 32, (16,), generator=g).cuda()) for _ in range(8)]

def run(break_capture):
    torch.manual_seed(123)
    model = ca.models.resnet(**kw)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device='cuda:0',
                    dtype=torch.bfloat16, print_freq=10**9)
    assert tr._graph_mode == 'auto' and tr._plan
    if break_capture:
        real = ca.trainer.LaunchPlan.end
        def boom(self):
            real(self)
            raise RuntimeError('injected: the recording cannot be finished')
        ca.trainer.LaunchPlan.end = boom
    try:
        recs = [tr.train([b])['loss'] for b in data]
    finally:
        if break_capture:
            ca.trainer.LaunchPlan.end = real
    torch.cuda.synchronize()
    return recs, tr

ok, tr_ok = run(False)
assert any(g['graph'] is not None and g['graph'].get('plan') is not None for g in tr_ok._gstates.values())
bad, tr_bad = run(True)
assert all(g['graph'] is None for g in tr_bad._gstates.values()) and len(tr_bad._graph_eager_for) == 1
assert ok == bad, (ok, bad)          # the job went on with eager launches: the same numbers
print('FALLBACK_OK')
'''


def test_a_failed_plan_capture_leaves_the_job_on_eager_launches(tmp_path):
    """graph = auto (the default): the launch plan is an optimisation of a step that already ran eagerly - if its recording
    fails for whatever reason the configuration stays on eager launches (with a warning and the watch armed) and trains to
    the same numbers; nothing is left in the per-step mailboxes of the aborted capture."""
    script = tmp_path / 'fallback_worker.py'
    script.write_text(FALLBACK_WORKER % {'root': ROOT})
    env = dict(os.environ, CONVNET_AMD_EMULATE='0')
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'FALLBACK_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
