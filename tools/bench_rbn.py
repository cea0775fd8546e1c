#!/usr/bin/env python
"""RangeBN forward / backward on config-5 shapes (B=256 bf16): separate passes vs the round-4 fused producers (GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import convnet_amd as ca
Q, L, lib = ca.quant, ca._lib.load(), ca._lib
ptr = lib.ptr
dev = torch.device('cuda', 0)


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


for (N, H, C) in [(256, 56, 64), (256, 56, 256), (256, 28, 128), (256, 28, 512), (256, 14, 256), (256, 14, 1024), (256, 7, 512), (256, 7, 2048)]:
    M, chunks = N * H * H, 16
    y = (torch.randn(N, H, H, C, device=dev) * 1.3).to(torch.bfloat16)
    g = (torch.randn(N, H, H, C, device=dev) * 0.1).to(torch.bfloat16)
    w, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    stats = torch.empty(2 * C, device=dev)
    arg = torch.empty(C * 2 * chunks, dtype=torch.int32, device=dev)
    ws = ca.ops.workspace(L.cn_rangebn_workspace(M, C, chunks), dev, 'quant')
    z, qy, dx = torch.empty_like(y), torch.empty_like(y), torch.empty_like(y)
    dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    fix = Q._scale_fix(M // chunks)
    qp = Q.qparams(Q.minmax_rows(y, N), N, 0)
    mm = torch.empty(N * 2, device=dev)
    code = 1
    st = lib.stream_of(y)
    f_sep = lambda: (L.cn_quantize_s(ptr(y), ptr(qy), y.numel(), code, ptr(qp[0:1]), ptr(qp[1:2]), 8, None, 0, 0, None, st),
                     L.cn_rangebn_fwd(ptr(qy), None, ptr(z), ptr(w), ptr(b), ptr(rm), ptr(rv), 0.1, 1e-5, chunks, fix, ptr(stats), ptr(arg), M, C, 1, 1, code, ptr(ws), ws.numel() * 4, st),
                     Q.minmax_rows(z, N))
    f_fus = lambda: L.cn_rangebn_fwd_q(ptr(y), ptr(qp), 8, ptr(qy), None, ptr(z), ptr(w), ptr(b), ptr(rm), ptr(rv), 0.1, 1e-5, chunks, fix, ptr(stats), ptr(arg), M, C, 1, code, N, ptr(mm), ptr(ws), ws.numel() * 4, st)
    b_sep = lambda: (L.cn_rangebn_bwd(ptr(g), ptr(qy), ptr(w), ptr(stats), ptr(arg), ptr(dx), ptr(dw), ptr(db), M, C, chunks, fix, code, ptr(ws), ws.numel() * 4, st),
                     Q.minmax_rows(dx, N))
    b_fus = lambda: L.cn_rangebn_bwd_mm(ptr(g), ptr(qy), ptr(w), ptr(stats), ptr(arg), ptr(dx), ptr(dw), ptr(db), M, C, chunks, fix, code, N, ptr(mm), ptr(ws), ws.numel() * 4, st)
    f_sep(); f_fus()
    mb = y.numel() * 2 / 1e6
    print('N=%d H=%d C=%4d (%6.1f MB): fwd separate %7.1f us fused %7.1f | bwd separate %7.1f fused %7.1f' % (
        N, H, C, mb, timeit(f_sep), timeit(f_fus), timeit(b_sep), timeit(b_fus)), flush=True)
