#!/usr/bin/env python
"""Per-layer microbenchmark of the conv kernels on the ResNet-50 layer inventory at
B=256 bf16: fwd / dgrad / wgrad time, algorithmic TFLOP/s and GB/s, with kernel-variant A/B
(cn_set_option).  GPU only; writes a table to stdout."""
import argparse
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convnet_amd as ca  # noqa: E402

# (count, C, H, K, R, stride, pad)
R50 = [(1, 8, 224, 64, 7, 2, 3), (1, 64, 56, 64, 1, 1, 0), (3, 64, 56, 64, 3, 1, 1), (4, 64, 56, 256, 1, 1, 0),
       (2, 256, 56, 64, 1, 1, 0), (1, 256, 56, 128, 1, 1, 0), (1, 128, 56, 128, 3, 2, 1), (4, 128, 28, 512, 1, 1, 0),
       (1, 256, 56, 512, 1, 2, 0), (3, 512, 28, 128, 1, 1, 0), (3, 128, 28, 128, 3, 1, 1), (1, 512, 28, 256, 1, 1, 0),
       (1, 256, 28, 256, 3, 2, 1), (6, 256, 14, 1024, 1, 1, 0), (1, 512, 28, 1024, 1, 2, 0),
       (5, 1024, 14, 256, 1, 1, 0), (5, 256, 14, 256, 3, 1, 1), (1, 1024, 14, 512, 1, 1, 0),
       (1, 512, 14, 512, 3, 2, 1), (3, 512, 7, 2048, 1, 1, 0), (1, 1024, 14, 2048, 1, 2, 0),
       (2, 2048, 7, 512, 1, 1, 0), (2, 512, 7, 512, 3, 1, 1)]


def timeit(fn, iters):
    for _ in range(3):      # (one warm call left the very first row of a cold box at clock-ramp speed: 10 ms for the stem)
        fn()
    torch.cuda.synchronize()
    ms = None
    for _ in range(2):      # the first timed round of a process absorbs a one-time ~40 ms event (first event pair): discarded
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--variants', default='0,1,3', help='igemm_variant values to A/B (0 = heuristic, 1 = register-staged, 3 = LDS-DMA)')
    ap.add_argument('--fragdb', type=int, default=0)
    ap.add_argument('--only', default='', help='comma-separated layer indices (default: all)')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    dt = torch.bfloat16
    L = ca._lib.load()
    variants = [int(v) for v in args.variants.split(',')]
    tot = {('fwd', v): 0.0 for v in variants}
    tot.update({('dgrad', v): 0.0 for v in variants})
    tot['wgrad'] = 0.0
    print('%-34s %5s | %s | %s | %s' % ('layer (n x C,H -> K, RxR/s)', 'GF',
                                        ' '.join('fwd[v=%d] ms  TF/s  GB/s' % v for v in variants),
                                        ' '.join('dgrad[v=%d] ms TF/s' % v for v in variants), 'wgrad ms TF/s GB/s'))
    sel = [int(i) for i in args.only.split(',')] if args.only else range(len(R50))
    for cnt, C, H, K, R, st, pad in [R50[i] for i in sel]:
        N = args.batch
        P = (H + 2 * pad - R) // st + 1
        x = torch.randn(N, H, H, C, device=dev).to(dt)
        w = (torch.randn(K, R, R, C, device=dev) * 0.05).to(dt)
        wc = w.permute(3, 1, 2, 0).contiguous()
        dy = torch.randn(N, P, P, K, device=dev).to(dt)
        dw = torch.zeros(K, R, R, C, device=dev)
        gf = 2.0 * N * P * P * K * C * R * R / 1e9
        by_f = (x.numel() + N * P * P * K + w.numel()) * 2 / 1e9
        cols = []
        for v in variants:
            L.cn_set_option(b'igemm_variant', v)
            ms = timeit(lambda: ca.ops.conv2d_fwd(x, w, None, K, R, R, (st, st), (pad, pad)), args.iters)
            tot[('fwd', v)] += ms * cnt
            cols.append('%8.3f %6.0f %5.0f' % (ms, gf / ms, by_f / ms * 1e3))
        cols_d = []
        for v in variants:
            L.cn_set_option(b'igemm_variant', v)
            if C == 8:
                cols_d.append('%8s %6s' % ('-', '-'))
                continue
            ms = timeit(lambda: ca.ops.conv2d_dgrad(dy, wc, x.shape, K, R, R, (st, st), (pad, pad)), args.iters)
            tot[('dgrad', v)] += ms * cnt
            cols_d.append('%8.3f %6.0f' % (ms, gf / ms))
        L.cn_set_option(b'igemm_variant', 0)
        msw = timeit(lambda: ca.ops.conv2d_wgrad(x, dy, dw, C, K, R, R, (st, st), (pad, pad), beta=0.0), args.iters)
        tot['wgrad'] += msw * cnt
        print('%dx %4d,%3d -> %4d, %dx%d/%d %12s %5.0f | %s | %s | %8.3f %6.0f %5.0f' % (
            cnt, C, H, K, R, R, st, '', gf, ' '.join(cols), ' '.join(cols_d), msw, gf / msw,
            (x.numel() + dy.numel()) * 2 / 1e9 / msw * 1e3))
        del x, w, wc, dy, dw
    print('TOTAL per step (ms): ' + '  '.join('%s=%.2f' % (str(k), v) for k, v in tot.items()))


if __name__ == '__main__':
    main()
