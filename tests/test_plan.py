"""Launch plans (csrc/plan.hip), host-side logic on the TEST-ONLY emulator: a recorded sequence of library calls is
re-issued by cn_plan_replay with the recorded arguments (private copies: by-value structs, scalars, pointers) and
reads the CURRENT contents of the buffers.  The Trainer-level use (recording under HIP stream capture, two streams,
imported torch kernels, RCCL buckets) needs a GPU: tests/test_graph_gpu.py."""
import ctypes

import pytest
import torch

import convnet_amd as ca
from convnet_amd import ops

DEV = torch.device('cuda', 0) if torch.cuda.is_available() else torch.device('cpu')
# (the same bodies run on a GPU through tests/test_plan_gpu.py: recording outside stream capture executes AND logs)


def _plan():
    L = ca._lib.load()
    stream = torch.cuda.current_stream(DEV).cuda_stream if DEV.type == 'cuda' else None
    return L, ca.trainer.LaunchPlan(L, stream)


def _sync():
    if DEV.type == 'cuda':
        torch.cuda.synchronize()


def test_plan_replays_recorded_launches_on_current_buffer_contents():
    L, plan = _plan()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 6, 8, generator=g).to(DEV)
    w = (torch.randn(16, 3, 3, 8, generator=g) * 0.1).to(DEV)
    acc = torch.zeros(2 * 6 * 6 * 16, device=DEV)
    # recorded: y = conv(x, w);  acc = 0;  acc += 1.5 (fill of a second buffer + add)
    y = ops.conv2d_fwd(x, w, None, 16, 3, 3, (1, 1), (1, 1))
    one = torch.empty(acc.numel(), device=DEV)
    ops.fill_f32_(acc, 0.0)
    ops.fill_f32_(one, 1.5)
    ops.add_(acc, one)
    plan.end()
    info = plan.info()
    assert info[1] >= 4 and info[2] == 0 and info[7] == 0, info     # own launches, nothing imported, no replay yet
    assert 'kernel' in plan.describe()
    _sync()
    ref0 = y.clone()
    # new inputs in the SAME buffers: a replay computes on them
    x.copy_(torch.randn(2, 6, 6, 8, generator=g))
    acc.fill_(7.0)
    ca._lib.check(L.cn_plan_replay(plan.handle), 'cn_plan_replay')
    _sync()
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
    assert L.cn_plan_import_graph(plan.handle, None) < 0               # no graph to import from
    plan.destroy()
    L2, again = _plan()                                                # the slot is free again
    again.destroy()
    assert L.cn_plan_begin(ctypes.byref(other), None) == 0             # destroying a recording plan ends the recording
    L.cn_plan_destroy(other)


def test_plan_hand_offs_are_logged_in_call_order():
    L, plan = _plan()
    a = torch.zeros(64, device=DEV)
    ops.fill_f32_(a, 1.0)
    st = torch.cuda.current_stream(DEV).cuda_stream if DEV.type == 'cuda' else None
    ca._lib.check(L.cn_stream_fork(st, st), 'cn_stream_fork')
    ops.fill_f32_(a, 2.0)
    plan.end()
    lines = [l for l in plan.describe().split('\n') if l]
    kinds = [l.split()[1] for l in lines]
    assert kinds == ['kernel', 'fork', 'kernel'], lines
    assert plan.info()[4] == 1
    a.zero_()
    ca._lib.check(L.cn_plan_replay(plan.handle), 'cn_plan_replay')
    _sync()
    assert torch.equal(a, torch.full_like(a, 2.0))
    plan.destroy()


@pytest.mark.skipif(DEV.type != 'cuda', reason='the emulator keeps closures, not argument blocks')
def test_plan_input_can_be_rebound_to_another_buffer():
    """cn_plan_bind_input / cn_plan_set_input: the launches that read a batch tensor are re-pointed at another buffer of
    the same layout (Trainer feeds the caller's batch to a recorded step without copying it)."""
    L, plan = _plan()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 6, 8, generator=g).to(DEV)
    w = (torch.randn(16, 1, 1, 8, generator=g) * 0.1).to(DEV)
    y = ops.conv2d_fwd(x, w, None, 16, 1, 1, (1, 1), (0, 0))
    plan.end()
    nbytes = x.numel() * 4
    sites = L.cn_plan_bind_input(plan.handle, 0, ctypes.c_void_p(x.data_ptr()), nbytes)
    assert sites >= 1
    assert L.cn_plan_bind_input(plan.handle, 1, ctypes.c_void_p(y.data_ptr() + y.numel() * 4 + 4096), 64) == 0   # nobody reads there
    x2 = torch.randn(2, 6, 6, 8, generator=g).to(DEV)
    x.fill_(float('nan'))                                   # the recorded buffer is no longer what the plan reads
    ca._lib.check(L.cn_plan_set_input(plan.handle, 0, ctypes.c_void_p(x2.data_ptr())), 'cn_plan_set_input')
    ca._lib.check(L.cn_plan_replay(plan.handle), 'cn_plan_replay')
    _sync()
    assert torch.equal(y, ops.conv2d_fwd(x2, w, None, 16, 1, 1, (1, 1), (0, 0)))
    plan.destroy()
