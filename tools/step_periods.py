#!/usr/bin/env python
"""Per-step period of the training loop on the device timeline and host time inside Trainer._step, step by step
(measurement aid for the graph = auto policy: is a slow run host-bound all along, in bursts, or slow on the device?).
    python tools/step_periods.py [steps]      (flags through CONVNET_AMD_FLAGS as everywhere)"""
import os
import sys
import time

os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T0 = time.time()
import torch          # noqa: E402
import convnet_amd as ca   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    dev = torch.device('cuda', 0)
    t_import = time.time() - T0
    torch.manual_seed(123)
    model = ca.models.resnet(dataset='imagenet', depth=50)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=str(dev),
                    dtype=torch.bfloat16, print_freq=10 ** 9)
    g = torch.Generator().manual_seed(123)
    pool = [(torch.randn(256, 3, 224, 224, generator=g).to(dev), torch.randint(0, 1000, (256,), generator=g).to(dev))
            for _ in range(4)]
    host, evs = [], []
    inner = tr._step

    def wrapped(*a, **k):
        t0 = time.perf_counter()
        r = inner(*a, **k)
        host.append((time.perf_counter() - t0) * 1e3)
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(dev))
        evs.append(e)
        return r
    tr._step = wrapped
    t0 = time.perf_counter()
    tr.train([pool[i % 4] for i in range(n)])
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    per = [evs[i - 1].elapsed_time(evs[i]) for i in range(1, len(evs))]
    tail = per[len(per) // 2:]
    print('flags=%s import %.1f s; %d steps in %.2f s; period ms: median(second half) %.2f max %.2f | host ms in _step: median %.2f max %.2f'
          % (os.environ.get('CONVNET_AMD_FLAGS', ''), t_import, n, wall, sorted(tail)[len(tail) // 2], max(tail),
             sorted(host)[len(host) // 2], max(host)))
    print(' period:', ' '.join('%.1f' % p for p in per))
    print(' host  :', ' '.join('%.1f' % h for h in host))


if __name__ == '__main__':
    main()
