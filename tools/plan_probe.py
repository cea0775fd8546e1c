#!/usr/bin/env python
"""Launch plan (csrc/plan.hip) against the eager step and the replayed HIP graph, one process, interleaved rounds.

    python tools/plan_probe.py [--batch 256] [--steps 20] [--rounds 3] [--burners 0] [--describe] [--modes eager,plan,graph]

Per mode: device step period (events around `steps` steps) and the HOST time the loop spends per step.  --burners N
starts N busy-loop processes on the same cores first (the "loaded host" case of VERDICT r5 item 1)."""
import argparse
import json
import multiprocessing as mp
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def burn(stop):
    x = 1.0
    while not stop.value:
        for _ in range(100000):
            x = x * 1.0000001 + 1e-9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--burners', type=int, default=0)
    ap.add_argument('--pin', type=int, default=0, help='confine this process AND the burners to the first N allowed cores')
    ap.add_argument('--depth', type=int, default=50)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--describe', action='store_true')
    ap.add_argument('--modes', default='eager,plan,graph')
    ap.add_argument('--out', default='')
    args = ap.parse_args()

    allowed = sorted(os.sched_getaffinity(0))
    if args.pin > 0:
        os.sched_setaffinity(0, set(allowed[:args.pin]))      # (inherited by the burners)
    stop = mp.Value('i', 0)
    procs = [mp.Process(target=burn, args=(stop,), daemon=True) for _ in range(args.burners)]
    for p in procs:
        p.start()

    import torch
    import convnet_amd as ca
    dev = torch.device('cuda', 0)
    dt = {'bf16': torch.bfloat16, 'fp32': torch.float32, 'fp16': torch.float16}[args.dtype]
    g = torch.Generator().manual_seed(123)
    pool = [(torch.randn(args.batch, 3, 224, 224, generator=g).to(dev),
             torch.randint(0, 1000, (args.batch,), generator=g).to(dev)) for _ in range(2)]

    def make(mode):
        mode = mode.rstrip('0123456789')       # (eager2, plan3: further instances of a mode - instance-to-instance spread)
        torch.manual_seed(123)
        model = ca.models.resnet(dataset='imagenet', depth=args.depth)
        tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=str(dev), dtype=dt,
                        print_freq=10 ** 9)
        eager = mode in ('eager', 'eagerpool')
        tr._graph_mode = '0' if eager else '1'
        tr._use_graph = not eager
        tr._plan = mode in ('plan', 'planthrottle')
        if mode == 'eagerpool':
            # the eager step with its tensors in a private allocator pool (what a capture gives the plan): address layout A/B
            mp_ = torch.cuda.MemPool()
            orig_body = tr._body

            def body(*a, **k):
                with torch.cuda.use_mem_pool(mp_):
                    return orig_body(*a, **k)
            tr._body = body
            tr._keep = mp_
        if mode == 'planthrottle':
            # the plan with the host held one step behind the device's previous step (queue-depth A/B)
            orig_replay = tr._replay
            ev = [None]

            def replay(st):
                if ev[0] is not None:
                    ev[0].synchronize()
                orig_replay(st)
                e = torch.cuda.Event()
                e.record(torch.cuda.current_stream(dev))
                ev[0] = e
            tr._replay = replay
        tr.train([pool[i % 2] for i in range(6)])      # warm-up + capture + first replays
        torch.cuda.synchronize()
        if mode == 'plan':       # host time INSIDE every replay call in the steady state (a blocked host shows up here)
            inner = tr._replay
            tr._replay_times = []

            def timed(st):
                t0 = time.perf_counter()
                inner(st)
                tr._replay_times.append((time.perf_counter() - t0) * 1e3)
            tr._replay = timed
        return tr

    modes = args.modes.split(',')
    trainers = {}
    for m in modes:        # device memory each mode's trainer holds after its warm-up (arenas + activation pools)
        torch.cuda.synchronize()
        r0 = torch.cuda.memory_reserved(dev)
        trainers[m] = make(m)
        print('%s: +%.1f GB reserved (allocated now %.1f GB)' % (m, (torch.cuda.memory_reserved(dev) - r0) / 2 ** 30,
                                                                 torch.cuda.memory_allocated(dev) / 2 ** 30))
    for m, tr in trainers.items():
        for gs in tr._gstates.values():
            st = gs.get('graph')
            if st is not None and st.get('plan') is not None and not m[-1].isdigit():
                info = st['plan'].info()
                print('plan[%s]: ops %d, own launches %d, imported %d, events %d, hand-offs %d, comm %d, streams %d; input sites %d' %
                      (m, info[0], info[1], info[2], info[3], info[4], info[5], info[6], st.get('x_sites', -1)))
                if args.describe:
                    text = st['plan'].describe()
                    print('\n'.join(l for l in text.split('\n') if ' foreign ' in l or ' memset ' in l)[:6000])
                    if args.out:
                        with open(args.out + '.plan.txt', 'w') as f:
                            f.write(text)
    res = {m: {'dev_ms': [], 'host_ms': []} for m in modes}
    for r in range(args.rounds):
        for m in modes:
            tr = trainers[m]
            data = [pool[i % 2] for i in range(args.steps)]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ms = tr._main_stream if tr._main_stream is not None else torch.cuda.current_stream(dev)
            t0 = time.perf_counter()
            e0.record(ms)
            tr.train(data)
            t1 = time.perf_counter()          # (train() ends with one D2H of the meters: host time includes the drain)
            e1.record(ms)
            torch.cuda.synchronize()
            res[m]['dev_ms'].append(e0.elapsed_time(e1) / args.steps)
            res[m]['host_ms'].append((t1 - t0) * 1e3 / args.steps)
    # host time proper: time to ISSUE the steps, device left to drain afterwards
    for m in modes:
        tr = trainers[m]
        tr.print_freq = 10 ** 9
        x, t = pool[0]
        tr.model.train()
        torch.cuda.synchronize()
        with torch.cuda.stream(tr._main_stream) if tr._main_stream is not None else torch.cuda.stream(torch.cuda.current_stream(dev)):
            t0 = time.perf_counter()
            for _ in range(5):
                tr._step(x, t, training=True)
            t1 = time.perf_counter()
        torch.cuda.synchronize()
        res[m]['issue_ms'] = (t1 - t0) * 1e3 / 5
    for m in modes:
        ts = getattr(trainers[m], '_replay_times', None)
        if ts:
            srt = sorted(ts)
            print('%s: host time inside cn_plan_replay over %d steady-state calls: median %.2f ms, p90 %.2f, max %.2f' %
                  (m, len(ts), srt[len(srt) // 2], srt[int(len(srt) * 0.9)], srt[-1]))
    stop.value = 1
    out = {'batch': args.batch, 'burners': args.burners, 'cores_allowed': len(allowed),
           'cores_used': len(os.sched_getaffinity(0))}
    for m in modes:
        d = res[m]
        out[m] = {'dev_ms_med': statistics.median(d['dev_ms']), 'dev_ms_all': [round(v, 3) for v in d['dev_ms']],
                  'loop_ms_med': statistics.median(d['host_ms']), 'issue_ms': round(d['issue_ms'], 3),
                  'img_s': args.batch / statistics.median(d['dev_ms']) * 1e3}
        print('%-6s step %.3f ms (%s)  %.0f img/s   host issue %.2f ms/step' %
              (m, out[m]['dev_ms_med'], ' '.join('%.2f' % v for v in d['dev_ms']), out[m]['img_s'], d['issue_ms']))
    print(json.dumps(out))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
