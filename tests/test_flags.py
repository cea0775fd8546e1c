"""convnet.pytorch_amd/flags.py: one table of engine switches (round 4).  CPU test."""
import os

import pytest
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = ("import sys; sys.path.insert(0, %r); import convnet_amd as ca; "
         "print('VAL', ca.ops.LAZY_DY, ca.ops.LAZY_DY_MIN_MB, ca.ops.JPAIR, ca.quant.FUSE_QUANT)" % ROOT)


def _run(flags):
    env = dict(os.environ, CONVNET_AMD_EMULATE='1')
    env.pop('CONVNET_AMD_FLAGS', None)
    if flags is not None:
        env['CONVNET_AMD_FLAGS'] = flags
    return subprocess.run([sys.executable, '-c', PROBE], env=env, capture_output=True, text=True, timeout=300)


def test_defaults_overrides_and_unknown_names():
    r = _run(None)
    assert r.returncode == 0 and 'VAL True 153.6 True True' in r.stdout, r.stdout + r.stderr
    r = _run('lazy_dy=0,lazy_min_mb=0,quant_fuse=0')
    assert r.returncode == 0 and 'VAL False 0.0 True False' in r.stdout, r.stdout + r.stderr
    # a stale script must not silently measure the default
    r = _run('lazy_dy=0,wgrad_streams=2')
    assert r.returncode != 0 and 'unknown flag' in r.stderr and 'wgrad_streams' in r.stderr, r.stderr[-500:]


def test_no_other_environment_switches_in_the_package():
    """The only environment variables the package reads: the flag table, the library's option table, the library path,
    the test-only emulator switch and the explicit transport override."""
    import re
    allowed = {'CONVNET_AMD_FLAGS', 'CONVNET_AMD_OPTIONS', 'CONVNET_AMD_HIP_LIB', 'CONVNET_AMD_EMULATE', 'CONVNET_AMD_COMM'}
    seen = set()
    pkg = os.path.join(ROOT, 'convnet.pytorch_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                seen |= set(re.findall(r"CONVNET_AMD_[A-Z0-9_]+", open(os.path.join(d, f)).read()))
    assert seen <= allowed, sorted(seen - allowed)


def test_eager_verdict_watch_window_logic():
    """graph = auto: the 'eager launches' verdict is withdrawn when the MEDIAN step period of the last nine steps exceeds
    1.2 x the step time the verdict was based on - not by single long periods (the pause between two train() calls, a
    validation pass), not by a window that is still filling."""
    import convnet_amd as ca
    W = ca.trainer.EagerWatch
    w = W(17.0)
    assert not any(w.add_period(17.1) for _ in range(50))                       # healthy
    w = W(17.0)
    assert not any(w.add_period(p) for p in [17.0] * 4 + [300.0] + [17.0] * 20)   # one pause
    w = W(17.0)
    assert not any(w.add_period(p) for p in [17.0, 90.0, 17.0, 17.0] * 10)        # bursts in a quarter of the steps
    w = W(17.0)
    fired = [w.add_period(34.0) for _ in range(12)]                              # a host-bound step, twice the time
    assert fired.index(True) == 8 and abs(w.recent_ms() - 34.0) < 1e-9            # as soon as the window is full
    w = W(17.0)
    seq = [17.0] * 9 + [21.0] * 9
    fired = [w.add_period(p) for p in seq]
    assert fired.index(True) == 13                                               # five of the last nine are slow
    w = W(17.0)
    assert not any(w.add_period(19.5) for _ in range(30))                        # 15 % slower: a replay would not help


def test_eager_watch_ignores_loader_wait_and_foreign_marks():
    """(i) the step period on the device timeline includes the time the loop waited for the loader; that
    wait (measured on the host, handed to the watch) is subtracted, so a loader-bound run is not taken for a
    launch-bound one; (ii) the library's ring of marks is shared: a period is routed to the watch BOTH of whose marks
    bound it, whichever watch polled it, and periods between two configurations' marks are dropped."""
    import convnet_amd as ca
    W = ca.trainer.EagerWatch
    w = W(17.0)
    assert not any(w.add_period(60.0, wait_ms=45.0) for _ in range(30))          # loader-bound: 15 ms of step
    w = W(17.0)
    fired = [w.add_period(60.0, wait_ms=20.0) for _ in range(12)]                # 40 ms of launches: host-bound
    assert fired.index(True) == 8 and abs(w.recent_ms() - 40.0) < 1e-9
    a, b = W(17.0), W(17.0)
    tag = lambda w_, seq: (w_.id << 32) | seq
    a._waits.update({2: 0.0, 3: 30.0})
    assert W.route(34.0, tag(a, 1), tag(a, 2)) is a and a.periods == [34.0]      # polled by anyone: lands in a
    assert W.route(64.0, tag(a, 2), tag(a, 3)) is a and a.periods == [34.0, 34.0]   # ... minus ITS step's loader wait
    assert W.route(500.0, tag(a, 3), tag(b, 1)) is None and b.periods == []      # a step of a, then one of b: nobody's
    assert W.route(500.0, tag(b, 1), tag(b, 3)) is None and b.periods == []      # a mark of b was dropped in between
    assert W.route(-1.0, tag(b, 3), tag(b, 4)) is b and b.periods == []          # a pair the runtime could not time
    nb = len(a.periods)
    c = W(17.0)                                                                  # a new watch resets nobody
    assert len(a.periods) == nb and c.id not in (a.id, b.id)


def test_graph_flag_spellings(monkeypatch):
    """graph=off / False used to mean 'always capture' (compared as text against '0')."""
    import convnet_amd as ca
    for v, want in (('auto', 'auto'), ('0', '0'), ('off', '0'), ('False', '0'), ('', '0'), ('1', '1'), ('on', '1')):
        monkeypatch.setitem(ca.flags._VALUES, 'graph', v)
        assert ca.flags.graph_mode() == want, v
    monkeypatch.setitem(ca.flags._VALUES, 'graph', 'sometimes')
    with pytest.raises(ValueError):
        ca.flags.graph_mode()



def test_rccl_topology_summary_parses_info_lines(tmp_path):
    """comm.topology_summary: what bench.py prints about the RCCL communicators of an N > 1 run (channels, rings / trees,
    transports) is read back from RCCL's own INFO log; unreadable or empty logs give {} (never an exception)."""
    import convnet_amd as ca
    log = tmp_path / 'rccl.log'
    log.write_text('h:1:1 [0] NCCL INFO Channel 00/08 :    0   1   2   3\n'
                   'h:1:1 [0] NCCL INFO Channel 07/08 :    0   1\n'
                   'h:1:1 [0] NCCL INFO Trees [0] 1/-1/-1->0->-1\n'
                   'h:1:1 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via P2P/IPC\n'
                   'h:1:1 [0] NCCL INFO Connected all rings\n'
                   'h:1:1 [0] NCCL INFO 8 coll channels, 8 nvls channels, 8 p2p channels\n')
    t = ca.comm.topology_summary(str(log))
    assert t['channels'] == 8 and t['coll_channels'] == 8 and t['rings'] and t['trees'] and t['via'] == ['P2P/IPC']
    assert ca.comm.topology_summary(str(tmp_path / 'missing.log')) == {}
    (tmp_path / 'empty.log').write_text('')
    assert ca.comm.topology_summary(str(tmp_path / 'empty.log')) == {}
