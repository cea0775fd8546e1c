"""Launch plans (csrc/plan.hip), host-side logic on the TEST-ONLY emulator: a recorded sequence of library calls is
re-issued by cn_plan_replay with the recorded arguments (private copies: by-value structs, scalars, pointers) and
reads the CURRENT contents of the buffers.  The Trainer-level use (recording under HIP stream capture, two streams,
imported torch kernels, RCCL buckets) needs a GPU: tests/test_graph_gpu.py."""
import ctypes

import pytest
import torch

import convnet_amd as ca
from convnet_amd import ops

pytestmark = pytest.mark.skipif(torch.cuda.is_available(), reason='emulator-side test (the GPU form is test_graph_gpu.py)')


def _plan():
    L = ca._lib.load()
    return L, ca.trainer.LaunchPlan(L, None)


def test_plan_replays_recorded_launches_on_current_buffer_contents():
    L, plan = _plan()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 6, 8, generator=g)
    w = torch.randn(16, 3, 3, 8, generator=g) * 0.1
    acc = torch.zeros(2 * 6 * 6 * 16)
    # recorded: y = conv(x, w);  acc = 0;  acc += 1.5 (fill of a second buffer + add)
    y = ops.conv2d_fwd(x, w, None, 16, 3, 3, (1, 1), (1, 1))
    one = torch.empty(acc.numel())
    ops.fill_f32_(acc, 0.0)
    ops.fill_f32_(one, 1.5)
    ops.add_(acc, one)
    plan.end()
    info = plan.info()
    assert info[1] >= 4 and info[2] == 0 and info[7] == 0, info     # own launches, nothing imported, no replay yet
    assert 'kernel' in plan.describe()
    ref0 = y.clone()
    # new inputs in the SAME buffers: a replay computes on them
    x.copy_(torch.randn(2, 6, 6, 8, generator=g))
    acc.fill_(7.0)
    ca._lib.check(L.cn_plan_replay(plan.handle), 'cn_plan_replay')
    y_replayed = y.clone()
    y_eager = ops.conv2d_fwd(x, w, None, 16, 3, 3, (1, 1), (1, 1))
    assert not torch.equal(ref0, y_replayed)
    assert torch.equal(y_replayed, y_eager)
    assert torch.equal(acc, torch.full_like(acc, 1.5))
    assert plan.info()[7] == 1
    plan.destroy()


def test_plan_recording_is_exclusive_and_replay_needs_a_finished_plan():
    L, plan = _plan()
    other = ctypes.c_void_p()
    assert L.cn_plan_begin(ctypes.byref(other), None) != 0           # one recording at a time, process-wide
    assert L.cn_plan_replay(plan.handle) != 0                          # still recording
    plan.end()
    assert L.cn_plan_replay(plan.handle) == 0                          # empty plan: nothing to issue
    assert L.cn_plan_import_graph(plan.handle, None) < 0               # no HIP graphs on the emulator
    plan.destroy()
    L2, again = _plan()                                                # the slot is free again
    again.destroy()
    assert L.cn_plan_begin(ctypes.byref(other), None) == 0             # destroying a recording plan ends the recording
    L.cn_plan_destroy(other)


def test_plan_hand_offs_are_logged_in_call_order():
    L, plan = _plan()
    a = torch.zeros(64)
    ops.fill_f32_(a, 1.0)
    ca._lib.check(L.cn_stream_fork(None, None), 'cn_stream_fork')
    ops.fill_f32_(a, 2.0)
    plan.end()
    lines = [l for l in plan.describe().split('\n') if l]
    kinds = [l.split()[1] for l in lines]
    assert kinds == ['kernel', 'fork', 'kernel'], lines
    assert plan.info()[4] == 1
    a.zero_()
    ca._lib.check(L.cn_plan_replay(plan.handle), 'cn_plan_replay')
    assert torch.equal(a, torch.full_like(a, 2.0))
    plan.destroy()
