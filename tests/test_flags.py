"""convnet.pytorch_amd/flags.py: one table of engine switches (round 4).  CPU test."""
import os
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
