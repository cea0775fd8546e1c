#!/usr/bin/env python
"""Inner-BatchNorm apply folded into the consumer convolution's operand load (cn_conv2d_fwd_xf) against the
separate passes it replaces, per ResNet-50 layer at B=256 bf16: conv(z) + an apply-sized streaming pass (read n,
write n) vs conv_xf(y).  GPU only (measurement aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convnet_amd as ca  # noqa: E402
from bench_layers import timeit  # noqa: E402

# consumers of an inner BN output: (count, C, H, K, R, stride, pad)
LAYERS = [(3, 64, 56, 64, 3, 1, 1), (3, 64, 56, 256, 1, 1, 0), (1, 128, 56, 128, 3, 2, 1), (3, 128, 28, 128, 3, 1, 1),
          (4, 128, 28, 512, 1, 1, 0), (1, 256, 28, 256, 3, 2, 1), (5, 256, 14, 256, 3, 1, 1), (6, 256, 14, 1024, 1, 1, 0),
          (1, 512, 14, 512, 3, 2, 1), (2, 512, 7, 512, 3, 1, 1), (3, 512, 7, 2048, 1, 1, 0)]


def main():
    dev = torch.device('cuda', 0)
    L = ca._lib.load()
    N, dt = 256, torch.bfloat16
    tot = [0.0, 0.0, 0.0, 0.0]
    print('%-28s %9s %9s %9s | %9s %9s' % ('layer', 'conv us', 'apply us', 'sum', 'conv_xf', 'xf(v1 base)'))
    for cnt, C, H, K, R, st, pad in LAYERS:
        y = torch.randn(N, H, H, C, device=dev).to(dt)
        z = torch.empty_like(y)
        w = (torch.randn(K, R, R, C, device=dev) * 0.05).to(dt)
        xf = torch.cat([torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1]).contiguous()
        t_conv = timeit(lambda: ca.ops.conv2d_fwd(y, w, None, K, R, R, (st, st), (pad, pad), bn_stats=True), 10)
        ca.ops._PENDING.clear()
        t_apply = timeit(lambda: ca._lib.check(L.cn_eltwise(1, z.data_ptr(), y.data_ptr(), None, y.numel(), 1,
                                                            ca._lib.stream_of(y))), 10)
        t_xf = timeit(lambda: ca.ops.conv2d_fwd_xf(y, xf, True, w, K, R, R, (st, st), (pad, pad), bn_stats=True), 10)
        ca.ops._PENDING.clear()
        L.cn_set_option(b'igemm_variant', 1)
        t_v1 = timeit(lambda: ca.ops.conv2d_fwd(y, w, None, K, R, R, (st, st), (pad, pad), bn_stats=True), 10)
        L.cn_set_option(b'igemm_variant', 0)
        ca.ops._PENDING.clear()
        print('%dx %4d,%3d -> %4d %dx%d/%d   %9.1f %9.1f %9.1f | %9.1f %9.1f' % (
            cnt, C, H, K, R, R, st, t_conv * 1e3, t_apply * 1e3, (t_conv + t_apply) * 1e3, t_xf * 1e3, t_v1 * 1e3))
        tot[0] += cnt * t_conv; tot[1] += cnt * t_apply; tot[2] += cnt * t_xf; tot[3] += cnt * t_v1
    print('per step (ms): conv %.3f + apply %.3f = %.3f   vs conv_xf %.3f   (conv forced to variant 1: %.3f)' % (
        tot[0], tot[1], tot[0] + tot[1], tot[2], tot[3]))


if __name__ == '__main__':
    main()
