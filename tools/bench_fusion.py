#!/usr/bin/env python
"""Per-layer cost of the two epilogue fusions on the ResNet-50 layer inventory at B=256 bf16
(measurement aid): conv fwd with / without the BatchNorm statistics epilogue (vs the separate
bn_stats pass it replaces), and dgrad with / without the BatchNorm-backward reduction epilogue (vs the
separate bn_bwd_reduce pass it replaces)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convnet_amd as ca  # noqa: E402
from bench_layers import R50, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=5)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    dev, dt = torch.device('cuda', 0), torch.bfloat16
    L = ca._lib.load()
    ops, lib = ca.ops, ca._lib
    sel = [int(i) for i in args.only.split(',')] if args.only else range(len(R50))
    print('%-30s | %9s %9s %9s | %9s %9s %9s %9s  (ms; sep = the standalone BN pass the fusion removes)' % (
        'layer', 'fwd', 'fwd+stats', 'sep stats', 'dgrad', 'dgrad+bn', 'dg+bn+add', 'sep reduce'))
    tot = [0.0] * 7
    for cnt, C, H, K, R, st, pad in [R50[i] for i in sel]:
        N = args.batch
        P = (H + 2 * pad - R) // st + 1
        x = torch.randn(N, H, H, C, device=dev).to(dt)
        w = (torch.randn(K, R, R, C, device=dev) * 0.05).to(dt)
        wc = w.permute(3, 1, 2, 0).contiguous()
        dy = torch.randn(N, P, P, K, device=dev).to(dt)
        y = torch.empty(N, P, P, K, device=dev, dtype=dt)
        t_f = timeit(lambda: ops.conv2d_fwd(x, w, None, K, R, R, (st, st), (pad, pad)), args.iters)
        t_fs = timeit(lambda: ops.conv2d_fwd(x, w, None, K, R, R, (st, st), (pad, pad), bn_stats=True), args.iters)
        # standalone statistics pass over the conv output (bn_stats + finalize + apply minus ... : time the
        # whole fwd BN with and without partials)
        M = N * P * P
        code = lib.dtype_code(dt)
        ws = ops.workspace(L.cn_bn_workspace(M, K, code), dev)
        yy = ops.conv2d_fwd(x, w, None, K, R, R, (st, st), (pad, pad), bn_stats=True)
        ps = ops.take_pending_stats(yy)
        z = torch.empty_like(yy)
        gam, bet = torch.ones(K, device=dev), torch.zeros(K, device=dev)
        stats = torch.empty(4 * K, device=dev)
        t_bn = timeit(lambda: lib.check(L.cn_bn_fwd_train(lib.ptr(yy), None, lib.ptr(z), None, lib.ptr(gam), lib.ptr(bet),
                                                          None, None, None, 0.1, 1e-5, lib.ptr(stats), M, K, 1, code,
                                                          lib.ptr(ws), ws.numel() * 4, lib.stream_of(yy))), args.iters)
        t_bnp = timeit(lambda: lib.check(L.cn_bn_fwd_train_partials(lib.ptr(yy), None, lib.ptr(z), None, lib.ptr(gam),
                                                                    lib.ptr(bet), None, None, None, 0.1, 1e-5,
                                                                    lib.ptr(stats), M, K, 1, code, lib.ptr(ps.partial),
                                                                    ps.rows, lib.ptr(ws), ws.numel() * 4,
                                                                    lib.stream_of(yy))), args.iters)
        cols = [t_f, t_fs, t_bn - t_bnp]
        if C != 8:
            bn_y = torch.randn(N, H, H, C, device=dev).to(dt)
            add = torch.randn(N, H, H, C, device=dev).to(dt)
            st4 = torch.cat([torch.zeros(C), torch.ones(C), torch.ones(C), torch.zeros(C)]).to(dev)
            Mi = N * H * H
            wsi = ops.workspace(L.cn_bn_workspace(Mi, C, code), dev)
            t_d = timeit(lambda: ops.conv2d_dgrad(dy, wc, x.shape, K, R, R, (st, st), (pad, pad)), args.iters)
            t_db = timeit(lambda: ops.conv2d_dgrad(dy, wc, x.shape, K, R, R, (st, st), (pad, pad),
                                                   bn=(bn_y, None, st4, True)), args.iters)
            t_dba = timeit(lambda: ops.conv2d_dgrad(dy, wc, x.shape, K, R, R, (st, st), (pad, pad), addend=add,
                                                    bn=(bn_y, None, st4, True)), args.iters)
            dx = ops.conv2d_dgrad(dy, wc, x.shape, K, R, R, (st, st), (pad, pad))
            g, partial, rows = ops.conv2d_dgrad(dy, wc, x.shape, K, R, R, (st, st), (pad, pad), bn=(bn_y, None, st4, True))
            dyo = torch.empty_like(dx)
            dg, db, coef = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.empty(3 * C, device=dev)
            t_b = timeit(lambda: lib.check(L.cn_bn_bwd(lib.ptr(dx), lib.ptr(bn_y), None, lib.ptr(st4[:C]), lib.ptr(st4),
                                                       lib.ptr(dyo), None, lib.ptr(dg), lib.ptr(db), 0.0, 1.0,
                                                       lib.ptr(coef), Mi, C, 1, code, lib.ptr(wsi), wsi.numel() * 4,
                                                       lib.stream_of(dx))), args.iters)
            t_bp = timeit(lambda: lib.check(L.cn_bn_bwd_partials(lib.ptr(g), lib.ptr(bn_y), lib.ptr(st4[:C]), lib.ptr(st4),
                                                                 lib.ptr(dyo), lib.ptr(dg), lib.ptr(db), 0.0, 1.0,
                                                                 lib.ptr(coef), Mi, C, code, lib.ptr(partial), rows,
                                                                 lib.ptr(wsi), wsi.numel() * 4, lib.stream_of(g))),
                          args.iters)
            cols += [t_d, t_db, t_dba, t_b - t_bp]
        else:
            cols += [0.0, 0.0, 0.0, 0.0]
        for i, c in enumerate(cols):
            tot[i] += c * cnt
        print('%dx %4d,%3d -> %4d, %dx%d/%d %6s | %9.3f %9.3f %9.3f | %9.3f %9.3f %9.3f %9.3f' % (
            (cnt, C, H, K, R, R, st, '') + tuple(cols)))
        del x, w, wc, dy, y
    print('TOTAL per step (ms): ' + ' '.join('%.2f' % t for t in tot))


if __name__ == '__main__':
    main()
