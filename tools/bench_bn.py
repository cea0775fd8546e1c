#!/usr/bin/env python
"""BatchNorm kernel sweep on the ResNet-50 BN shapes (B=256 bf16): forward (stats+finalize+apply) and
backward (reduce+finalize+apply) time per step for combinations of the grid-size knobs."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convnet_amd as ca  # noqa: E402

# (count, C, H, has_residual_and_relu)   -- ResNet-50: bn1/bn2 of each bottleneck, bn3 (+res), downsample BNs, stem
SHAPES = [(1, 64, 112, 0), (6, 64, 56, 0), (3, 256, 56, 1), (1, 256, 56, 0), (1, 128, 56, 0), (7, 128, 28, 0),
          (4, 512, 28, 1), (1, 512, 28, 0), (1, 256, 28, 0), (11, 256, 14, 0), (6, 1024, 14, 1), (1, 1024, 14, 0),
          (1, 512, 14, 0), (5, 512, 7, 0), (3, 2048, 7, 1), (1, 2048, 7, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--reduce', default='256,512,1024,2048')
    ap.add_argument('--apply', default='1024,2048,4096')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    L = ca._lib.load()
    mods = []
    for cnt, C, H, res in SHAPES:
        bn = ca.nn.BatchNorm2d(C)
        m = torch.nn.Sequential(bn)
        ca.engine.prepare(m, dev, torch.bfloat16)
        y = torch.randn(args.batch, H, H, C, device=dev).to(torch.bfloat16).requires_grad_(True)
        r = torch.randn(args.batch, H, H, C, device=dev).to(torch.bfloat16).requires_grad_(True) if res else None
        dz = torch.randn(args.batch, H, H, C, device=dev).to(torch.bfloat16)
        mods.append((cnt, bn, y, r, dz, res))

    def run(which):
        tot = 0.0
        for cnt, bn, y, r, dz, res in mods:
            z = bn(y, residual=r, relu=True)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if which == 'fwd':
                s.record()
                for _ in range(args.iters):
                    z = bn(y, residual=r, relu=True)
                e.record()
            else:
                zs = [bn(y, residual=r, relu=True) for _ in range(args.iters)]
                torch.cuda.synchronize()
                s.record()
                for zz in zs:
                    zz.backward(dz)
                e.record()
            torch.cuda.synchronize()
            tot += s.elapsed_time(e) / args.iters * cnt
        return tot

    for rb in [int(v) for v in args.reduce.split(',')]:
        for ab in [int(v) for v in args.apply.split(',')]:
            L.cn_set_option(b'bn_reduce_blocks', rb)
            L.cn_set_option(b'bn_apply_blocks', ab)
            print('reduce_blocks=%5d apply_blocks=%5d : fwd %.3f ms/step  bwd %.3f ms/step' % (rb, ab, run('fwd'), run('bwd')))


if __name__ == '__main__':
    main()
