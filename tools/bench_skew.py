#!/usr/bin/env python
"""Does the relative placement of the streams of a multi-stream pass matter?  a = relu(b + c) (cn_eltwise op 4:
two reads, one write, 16-byte accesses) over 392 MB bf16 tensors carved out of one buffer at different relative
offsets (tensors from the caching allocator are all 2 MiB aligned, i.e. skew 0).  GPU only (measurement aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convnet_amd as ca  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    L = ca._lib.load()
    n = 256 * 56 * 56 * 256          # elements of a layer1 junction tensor
    nbytes = n * 2
    big = torch.empty(4 * nbytes + (64 << 20), dtype=torch.uint8, device=dev)
    base = big.data_ptr()
    base = (base + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    off0 = base - big.data_ptr()
    st = torch.cuda.current_stream().cuda_stream

    def view(off):
        return big[off0 + off: off0 + off + nbytes].view(torch.bfloat16)

    span = (nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    for skew in (0, 256, 4096, 16384, 65536, 256 * 1024, 1 << 20, (1 << 20) + 4096, 3 * 4096 + 256):
        a, b, c = view(0), view(span + skew), view(2 * span + 2 * skew)
        b.fill_(1.0)
        c.fill_(-0.5)
        for _ in range(2):
            ca._lib.check(L.cn_eltwise(4, a.data_ptr(), b.data_ptr(), c.data_ptr(), n, 1, st))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            ca._lib.check(L.cn_eltwise(4, a.data_ptr(), b.data_ptr(), c.data_ptr(), n, 1, st))
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        print('skew %8d B : %.3f ms  %.0f GB/s' % (skew, ms, 3 * nbytes / ms / 1e6))


if __name__ == '__main__':
    main()
