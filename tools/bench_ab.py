#!/usr/bin/env python
"""Within-process interleaved A/B of kernel-variant knobs on ResNet-50 conv layers at B=256 bf16.

    python tools/bench_ab.py --knob igemm_8w --values 0,16 [--only 16,22] [--dirs fwd,dgrad,wgrad] [--rounds 5]

Every (layer, direction) is timed under each knob value in turn, `rounds` times (interleaved, one process), and the
median / min per value is printed with the kernel the dispatcher picked.  GPU only."""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convnet_amd as ca  # noqa: E402
from bench_layers import R50  # noqa: E402


def timeit(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--knob', required=True)
    ap.add_argument('--values', required=True)
    ap.add_argument('--fixed', default='', help='other knobs held fixed: name=v,name=v')
    ap.add_argument('--only', default='')
    ap.add_argument('--dirs', default='fwd,dgrad')
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--iters', type=int, default=4)
    ap.add_argument('--batch', type=int, default=256)
    args = ap.parse_args()
    L = ca._lib.load()
    for kv in [x for x in args.fixed.split(',') if x]:
        k, v = kv.split('=')
        L.cn_set_option(k.encode(), int(v))
    values = [int(v) for v in args.values.split(',')]
    dev, dt = torch.device('cuda', 0), torch.bfloat16
    sel = [int(i) for i in args.only.split(',')] if args.only else range(len(R50))
    dirs = args.dirs.split(',')
    totals = {(d, v): 0.0 for d in dirs for v in values}
    print('%-28s %-6s | ' % ('layer', 'dir') + ' | '.join('%s=%d  med    min  TF/s' % (args.knob, v) for v in values))
    for idx in sel:
        cnt, C, H, K, R, st, pad = R50[idx]
        N = args.batch
        P = (H + 2 * pad - R) // st + 1
        x = torch.randn(N, H, H, C, device=dev).to(dt)
        w = (torch.randn(K, R, R, C, device=dev) * 0.05).to(dt)
        wc = w.permute(3, 1, 2, 0).contiguous()
        dy = torch.randn(N, P, P, K, device=dev).to(dt)
        dw = torch.zeros(K, R, R, C, device=dev)
        gf = 2.0 * N * P * P * K * C * R * R / 1e9
        fns = {'fwd': lambda: ca.ops.conv2d_fwd(x, w, None, K, R, R, (st, st), (pad, pad), bn_stats=True),
               'dgrad': lambda: ca.ops.conv2d_dgrad(dy, wc, x.shape, K, R, R, (st, st), (pad, pad)),
               'wgrad': lambda: ca.ops.conv2d_wgrad(x, dy, dw, C, K, R, R, (st, st), (pad, pad), beta=0.0)}
        for d in dirs:
            if d == 'dgrad' and C == 8:
                continue
            ts = {v: [] for v in values}
            names = {}
            for v in values:      # warm
                L.cn_set_option(args.knob.encode(), v)
                fns[d]()
                names[v] = L.cn_last_kernel_name().decode()
            torch.cuda.synchronize()
            for _ in range(args.rounds):
                for v in values:
                    L.cn_set_option(args.knob.encode(), v)
                    ts[v].append(timeit(fns[d], args.iters))
            cols = []
            for v in values:
                med, mn = statistics.median(ts[v]), min(ts[v])
                totals[(d, v)] += med * cnt
                cols.append('%7.1f %6.1f %5.0f' % (med * 1e3, mn * 1e3, gf / med))
            print('%dx %4d,%3d->%4d %dx%d/%d %6s %-6s | ' % (cnt, C, H, K, R, R, st, '', d) + ' | '.join(cols) +
                  '   ' + ' / '.join(sorted(set(n.replace('igemm_kernel', 'ig').replace('bf16_t, ', '') for n in names.values()))))
        del x, w, wc, dy, dw
    print('TOTAL ms/step: ' + '  '.join('%s[%d]=%.3f' % (d, v, totals[(d, v)]) for d in dirs for v in values))


if __name__ == '__main__':
    main()
