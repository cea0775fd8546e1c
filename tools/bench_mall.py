#!/usr/bin/env python
"""Does a consumer kernel find a tensor its producer just wrote in the 256 MB Infinity Cache?  (GPU)
For tensor sizes S: time `c = relu-like pass over b` (our eltwise-class kernel: torch add as a stand-in reads b, writes c)
right after b was produced (warm) vs after a 1 GB buffer was streamed in between (cold)."""
import torch, sys
dev = torch.device('cuda', 0)
def t(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(n):
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return min(ts)
big = torch.empty(1 << 29, dtype=torch.bfloat16, device=dev)   # 1 GiB
for mb in (25, 50, 100, 150, 200, 300, 400, 800):
    n = mb * (1 << 20) // 2
    a = torch.randn(n, device=dev).to(torch.bfloat16) if mb <= 400 else torch.zeros(n, dtype=torch.bfloat16, device=dev)
    b = torch.empty_like(a); c = torch.empty_like(a)
    def warm():
        torch.add(a, 1.0, out=b)      # producer: reads a, writes b
    def consumer():
        torch.add(b, 1.0, out=c)      # consumer: reads b, writes c
    res = {}
    for mode in ('warm', 'cold'):
        ts = []
        for _ in range(5):
            warm()
            if mode == 'cold':
                big.add_(1.0)          # stream 2 GiB through the caches
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); consumer(); e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        res[mode] = min(ts)
    print('S = %4d MB: consumer right after producer %.1f us (%.0f GB/s of read+write), cold %.1f us (%.0f GB/s)' % (
        mb, res['warm'] * 1e3, 2 * mb * 1.048576 / res['warm'], res['cold'] * 1e3, 2 * mb * 1.048576 / res['cold']))
