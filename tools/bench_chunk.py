#!/usr/bin/env python
"""Producer -> consumer chains through the 256 MB Infinity Cache: an elementwise pass that writes a tensor followed by
a 1x1 convolution that reads it, over the whole batch vs in batch chunks small enough to stay cache-resident. (GPU)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convnet_amd as ca
dev, dt = torch.device('cuda', 0), torch.bfloat16
def run(N, H, C, K, chunks, iters=6):
    y = torch.randn(N, H, H, C, device=dev).to(dt)
    w = (torch.randn(K, 1, 1, C, device=dev) * 0.05).to(dt)
    bn = ca.nn.BatchNorm2d(C); ca.engine.prepare(torch.nn.Sequential(bn), dev, dt); bn.eval()
    z = torch.empty_like(y)
    def step():
        for yc in y.chunk(chunks, 0):
            zc = ca.ops.batch_norm_infer(yc, None, bn, True)          # reads y chunk, writes z chunk
            ca.ops.conv2d_fwd(zc, w, None, K, 1, 1, (1, 1), (0, 0))    # reads z chunk
    step(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(iters):
        s.record(); step(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return min(ts) * 1e3
for (N, H, C, K) in ((256, 56, 256, 64), (256, 28, 512, 128), (256, 56, 64, 256)):
    mb = N * H * H * C * 2 / 2**20
    print('tensor %4.0f MB (%d,%d,%d)->%d:' % (mb, N, H, C, K), '  '.join('%d chunk(s) %.1f us' % (c, run(N, H, C, K, c)) for c in (1, 2, 4, 8)))
