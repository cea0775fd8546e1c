#!/usr/bin/env python
"""Where does the time of a short-reduction igemm launch go?  Times conv forward on a few ResNet-50 layers with
the measurement knob igemm_dbg: 0 = normal, 1 = reduction loop skipped (prologue + epilogue only), 2 = global
stores skipped, 3 = both.  GPU only (measurement aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convnet_amd as ca  # noqa: E402
from bench_layers import timeit  # noqa: E402

LAYERS = [(256, 14, 1024, 1, 1, 0), (1024, 14, 256, 1, 1, 0), (512, 7, 2048, 1, 1, 0), (64, 56, 256, 1, 1, 0),
          (256, 56, 64, 1, 1, 0), (128, 28, 512, 1, 1, 0), (256, 14, 256, 3, 1, 1), (128, 28, 128, 3, 1, 1)]


def main():
    dev = torch.device('cuda', 0)
    L = ca._lib.load()
    N = 256
    print('%-26s %s' % ('layer', '  '.join('dbg=%d us' % d for d in range(4))))
    for C, H, K, R, st, pad in LAYERS:
        x = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
        w = (torch.randn(K, R, R, C, device=dev) * 0.05).to(torch.bfloat16)
        cols = []
        for dbg in range(4):
            L.cn_set_option(b'igemm_dbg', dbg)
            ms = timeit(lambda: ca.ops.conv2d_fwd(x, w, None, K, R, R, (st, st), (pad, pad)), 20)
            cols.append('%9.1f' % (ms * 1e3))
        L.cn_set_option(b'igemm_dbg', 0)
        print('%4d,%3d -> %4d %dx%d/%d      %s' % (C, H, K, R, R, st, '  '.join(cols)))
    # launch floor: an (almost) empty launch of the same kernel
    x = torch.randn(1, 8, 8, 64, device=dev).to(torch.bfloat16)
    w = (torch.randn(128, 1, 1, 64, device=dev) * 0.05).to(torch.bfloat16)
    ms = timeit(lambda: ca.ops.conv2d_fwd(x, w, None, 128, 1, 1, (1, 1), (0, 0)), 50)
    print('one-tile launch: %.1f us' % (ms * 1e3))


if __name__ == '__main__':
    main()
