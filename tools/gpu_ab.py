"""Interleaved bench.py A/B on one box: tools/gpu_ab.py OUTTAG ROUNDS "tagA ENV=v ENV=v" "tagB ENV=v" ..."""
import json, os, subprocess, sys
out = os.path.join('gpurun_out', sys.argv[1])
os.makedirs(out, exist_ok=True)
rounds = int(sys.argv[2])
specs = [s.split() for s in sys.argv[3:]]
cmd = [sys.executable, 'bench.py', '--steps', os.environ.get('AB_STEPS', '30'), '--warmup', '5', '--no-cpu-baseline', '--no-kernel-profile'] + \
    os.environ.get('AB_ARGS', '').split()
res = {}
for r in range(rounds):
    for spec in specs:
        env = dict(os.environ)
        env.update(kv.split('=', 1) for kv in spec[1:])
        p = subprocess.run(cmd, env=env, capture_output=True, text=True)
        try:
            v = json.loads([l for l in p.stdout.strip().splitlines() if l.startswith('{"metric"')][-1])['value']
        except Exception:
            v = None
            open(os.path.join(out, 'err_%s.txt' % spec[0]), 'w').write(p.stdout[-2000:] + p.stderr[-4000:])
        res.setdefault(spec[0], []).append(v)
        print(spec[0], v, flush=True)
with open(os.path.join(out, 'ab.json'), 'w') as f:
    json.dump(res, f)
print({k: (sum(x for x in v if x) / max(1, len([x for x in v if x]))) for k, v in res.items()})
