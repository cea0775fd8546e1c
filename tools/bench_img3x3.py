#!/usr/bin/env python
"""The image-resident 3x3 kernel (csrc/conv3x3_img.hip) against the tiled implicit-GEMM kernel on its three layer shapes at
B = 256 bf16: bit-equality of forward / data gradient, then time, TFLOP/s and fraction of the dense MFMA peak of both,
interleaved.  GPU only.  Options: --wgs N (cn_set_option conv3x3_img_wgs)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convnet_amd as ca  # noqa: E402


def slab_of(wsrc):
    """[rows C][9][k C] -> [tap][k / 16][row / 32][k half][row % 32][8]: cn_weight_prep_tiled's slab layout."""
    C = wsrc.shape[0]
    return wsrc.reshape(C // 32, 32, 9, C // 16, 2, 8).permute(2, 3, 0, 4, 1, 5).contiguous().view(-1)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        best = ms if best is None else min(best, ms)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--wgs', default='256')
    ap.add_argument('--dbg', default='0')
    args = ap.parse_args()
    dev, dt = torch.device('cuda', 0), torch.bfloat16
    ops, L = ca.ops, ca._lib.load()
    N = args.batch
    for (C, H) in ((128, 28), (256, 14), (512, 7)):
        g = torch.Generator().manual_seed(C)
        x = torch.randn(N, H, H, C, generator=g).to(dt).to(dev)
        w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5).to(dt)
        dy = torch.randn(N, H, H, C, generator=g).to(dt).to(dev)
        wk = w.permute(0, 2, 3, 1).contiguous()
        wc = w.permute(1, 2, 3, 0).contiguous()
        sf, sb = slab_of(wk.reshape(C, 9, C)).to(dev), slab_of(wc.reshape(C, 9, C)).to(dev)
        wk, wc = wk.view(-1).to(dev), wc.view(-1).to(dev)
        args4 = (C, 3, 3, (1, 1), (1, 1))
        y0 = ops.conv2d_fwd(x, wk, None, *args4, bn_stats=True)
        p0 = ops.take_pending_stats(y0)
        d0 = ops.conv2d_dgrad(dy, wc, (N, H, H, C), *args4)
        gf = 2.0 * N * H * H * C * C * 9 / 1e9
        for wgs in [int(v) for v in args.wgs.split(',')]:
            L.cn_set_option(b'conv3x3_img_wgs', wgs)
            for dbg in [int(v) for v in args.dbg.split(',')]:
                L.cn_set_option(b'dbg_img', dbg)
                t = timeit(lambda: ops.conv2d_fwd(x, wk, None, *args4, bn_stats=True, w_slab=sf))
                ops.take_pending_stats(y0) if False else None
                td = timeit(lambda: ops.conv2d_dgrad(dy, wc, (N, H, H, C), *args4, w_slab_t=sb))
                print('   dbg=%2d  fwd %6.1f us  dgrad %6.1f us' % (dbg, t * 1e3, td * 1e3), flush=True)
            L.cn_set_option(b'dbg_img', 0)
            y1 = ops.conv2d_fwd(x, wk, None, *args4, bn_stats=True, w_slab=sf)
            assert 'conv3x3_img' in L.cn_last_kernel_name().decode()
            p1 = ops.take_pending_stats(y1)
            d1 = ops.conv2d_dgrad(dy, wc, (N, H, H, C), *args4, w_slab_t=sb)
            eq = (torch.equal(y0, y1), torch.equal(d0, d1))
            st = float((p1.partial.double().sum(0) - p0.partial.double().sum(0)).norm() / p0.partial.double().sum(0).norm())
            t_tf = timeit(lambda: ops.conv2d_fwd(x, wk, None, *args4, bn_stats=True))
            ops.take_pending_stats(y0)
            t_if = timeit(lambda: ops.conv2d_fwd(x, wk, None, *args4, bn_stats=True, w_slab=sf))
            t_td = timeit(lambda: ops.conv2d_dgrad(dy, wc, (N, H, H, C), *args4))
            t_id = timeit(lambda: ops.conv2d_dgrad(dy, wc, (N, H, H, C), *args4, w_slab_t=sb))
            print('C=%3d %2dx%2d wgs=%4d  equal fwd/dgrad %s stats %.1e | fwd tiled %6.1f us %5.0f TF/s (%.2f)  img %6.1f us %5.0f TF/s (%.2f) | '
                  'dgrad tiled %6.1f us %5.0f TF/s  img %6.1f us %5.0f TF/s (%.2f)' % (
                      C, H, H, wgs, eq, st, t_tf * 1e3, gf / t_tf, gf / t_tf / 2500, t_if * 1e3, gf / t_if, gf / t_if / 2500,
                      t_td * 1e3, gf / t_td, t_id * 1e3, gf / t_id, gf / t_id / 2500), flush=True)
        L.cn_set_option(b'conv3x3_img_wgs', 256)


if __name__ == '__main__':
    main()
