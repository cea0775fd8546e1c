"""Per-operator parity: every HIP kernel against the PyTorch CPU fp32 op the reference executes
(level T1).  Each case runs in two modes:
  * emul : the same kernel source through the TEST-ONLY SIMT emulator, tiny shapes (build container)
  * gpu  : libconvnet_hip.so on a real MI355X, shapes up to the ResNet-50 layer inventory
Tolerances (rel-L2 unless noted): fp32 <= 1e-5 forward / 1e-4 gradients; bf16 storage with fp32
accumulation <= 1e-2 (inputs are pre-rounded to bf16 so only output rounding + summation order
differ)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import HAS_GPU
from helpers import rel_l2

MODES = [pytest.param('emul'), pytest.param('gpu', marks=pytest.mark.gpu)]
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def _dev(mode):
    if mode == 'emul' and HAS_GPU:
        pytest.skip('emulator mode is for GPU-less hosts')
    if mode == 'gpu' and not HAS_GPU:
        pytest.skip('no GPU')
    import convnet_amd as ca
    assert ca._lib.is_emulated() == (mode == 'emul')
    return torch.device('cuda', 0) if mode == 'gpu' else torch.device('cpu')


@pytest.fixture
def request_cleanup():
    """MonkeyPatch objects registered by a test are undone when it ends (pass or fail)."""
    items = []
    yield items
    for m in items:
        m.undo()


def _tol(dtype, grad=False):
    if dtype == torch.bfloat16:
        return 1e-2
    if dtype == torch.float16:      # 11 significant bits against bf16's 8
        return 2e-3
    return 1e-4 if grad else 1e-5


def _f16_emul_subset(mode, dtype, keep=False):
    """fp16 shares every kernel template with bf16: on the emulated (CPU) suite only the conv and BatchNorm tests
    run it, the GPU suite runs all of them."""
    if mode == 'emul' and dtype == torch.float16 and not keep:
        pytest.skip('fp16 on the emulator: conv + BatchNorm tests only')


def _q(t, dtype):
    """round test data to the compute dtype (so the fp32 reference sees the same values)"""
    return t.to(dtype).float()


def _nhwc(t, dtype, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)


# (N, H, W, C, K, R, stride, pad)
CONV_EMUL = [(2, 8, 8, 16, 64, 3, 1, 1), (1, 9, 7, 16, 72, 3, 2, 1), (2, 6, 6, 8, 64, 7, 2, 3),
             (3, 5, 5, 64, 136, 1, 1, 0), (1, 8, 8, 32, 40, 1, 2, 0)]
# the 23 distinct ResNet-50 conv configs at N=2 plus ResNet-18's extra ones
CONV_GPU = [(2, 224, 224, 8, 64, 7, 2, 3), (2, 56, 56, 64, 64, 1, 1, 0), (2, 56, 56, 64, 64, 3, 1, 1),
            (2, 56, 56, 64, 256, 1, 1, 0), (2, 56, 56, 256, 64, 1, 1, 0), (2, 56, 56, 256, 128, 1, 1, 0),
            (2, 56, 56, 128, 128, 3, 2, 1), (2, 28, 28, 128, 512, 1, 1, 0), (2, 56, 56, 256, 512, 1, 2, 0),
            (2, 28, 28, 512, 128, 1, 1, 0), (2, 28, 28, 128, 128, 3, 1, 1), (2, 28, 28, 512, 256, 1, 1, 0),
            (2, 28, 28, 256, 256, 3, 2, 1), (2, 14, 14, 256, 1024, 1, 1, 0), (2, 28, 28, 512, 1024, 1, 2, 0),
            (2, 14, 14, 1024, 256, 1, 1, 0), (2, 14, 14, 256, 256, 3, 1, 1), (2, 14, 14, 1024, 512, 1, 1, 0),
            (2, 14, 14, 512, 512, 3, 2, 1), (2, 7, 7, 512, 2048, 1, 1, 0), (2, 14, 14, 1024, 2048, 1, 2, 0),
            (2, 7, 7, 2048, 512, 1, 1, 0), (2, 7, 7, 512, 512, 3, 1, 1),
            (2, 56, 56, 64, 128, 3, 2, 1), (2, 56, 56, 64, 128, 1, 2, 0), (3, 17, 13, 64, 72, 3, 2, 1)]


def _conv_case(cfg, dtype, dev, seed=0):
    import convnet_amd as ca
    ops = ca.ops
    N, H, W, C, K, R, st, pad = cfg
    g = torch.Generator().manual_seed(seed)
    x = _q(torch.randn(N, C, H, W, generator=g), dtype)
    w = _q(torch.randn(K, C, R, R, generator=g) * (2.0 / (C * R * R)) ** 0.5, dtype)
    x.requires_grad_(True)
    w.requires_grad_(True)
    y_ref = F.conv2d(x, w, stride=st, padding=pad)
    dy = _q(torch.randn(y_ref.shape, generator=g), dtype)
    y_ref.backward(dy)
    xh = _nhwc(x.detach(), dtype, dev)
    wk = w.detach().permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)
    wc = w.detach().permute(1, 2, 3, 0).contiguous().to(dtype).to(dev)
    dyh = _nhwc(dy, dtype, dev)
    y = ops.conv2d_fwd(xh, wk, None, K, R, R, (st, st), (pad, pad))
    dx = ops.conv2d_dgrad(dyh, wc, xh.shape, K, R, R, (st, st), (pad, pad))
    dw = torch.zeros(K, R, R, C, dtype=torch.float32, device=dev)
    ops.conv2d_wgrad(xh, dyh, dw, C, K, R, R, (st, st), (pad, pad), beta=0.0)
    return (rel_l2(y.float().cpu().permute(0, 3, 1, 2), y_ref.detach()),
            rel_l2(dx.float().cpu().permute(0, 3, 1, 2), x.grad),
            rel_l2(dw.cpu().permute(0, 3, 1, 2), w.grad))


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
def test_conv2d_fwd_dgrad_wgrad(mode, dtype):
    dev = _dev(mode)
    _f16_emul_subset(mode, dtype, keep=True)
    cases = CONV_EMUL if mode == 'emul' else CONV_GPU
    bad = []
    for cfg in cases:
        if dtype != torch.float32 and cfg[3] % 8:
            continue
        ef, ed, ew = _conv_case(cfg, dtype, dev)
        if ef > _tol(dtype) or ed > _tol(dtype, True) or ew > _tol(dtype, True):
            bad.append((cfg, ef, ed, ew))
    assert not bad, 'conv mismatches (cfg, fwd, dgrad, wgrad rel-L2): %s' % bad


# (N, H, W, C, K, R, stride, pad): ragged pixel tiles (M % 128 != 0), K below / above one channel tile,
# > 512 partial rows (the compress pass) on the GPU
STATS_EMUL = [(2, 9, 9, 16, 64, 3, 1, 1), (3, 7, 5, 16, 72, 1, 1, 0), (1, 20, 20, 8, 136, 3, 2, 1)]
STATS_GPU = [(8, 56, 56, 64, 64, 1, 1, 0), (32, 56, 56, 64, 256, 1, 1, 0), (4, 28, 28, 128, 128, 3, 1, 1),
             (5, 14, 14, 1024, 256, 1, 1, 0), (3, 17, 13, 64, 72, 3, 2, 1), (2, 224, 224, 8, 64, 7, 2, 3)]


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
def test_conv_epilogue_bn_statistics(mode, dtype, request_cleanup):
    """cn_conv2d_fwd_bnstats: same y as cn_conv2d_fwd (bit for bit), partial rows that sum to the
    per-channel sum / sum of squares of the stored y, and a BatchNorm fed from those partials that
    matches the BatchNorm that re-reads y (models/resnet.py:141-165 conv -> bn pairs)."""
    dev = _dev(mode)
    _f16_emul_subset(mode, dtype)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    # this test pins the TILED kernel's epilogue (one partial row per 128-pixel tile); the streaming / halo kernels that
    # serve some of these shapes in production emit one row per workgroup and have their own tests
    monkey = pytest.MonkeyPatch()
    monkey.setattr(ops, 'CONV1X1_STREAM', False)
    monkey.setattr(ops, 'CONV3X3_HALO', False)
    request_cleanup.append(monkey)
    for (N, H, W, C, K, R, st, pad) in (STATS_EMUL if mode == 'emul' else STATS_GPU):
        if dtype != torch.float32 and C % 8:
            continue
        g = torch.Generator().manual_seed(K + H)
        xh = _nhwc(torch.randn(N, C, H, W, generator=g), dtype, dev)
        wk = (torch.randn(K, R, R, C, generator=g) * (2.0 / (C * R * R)) ** 0.5).to(dtype).to(dev)
        y_plain = ops.conv2d_fwd(xh, wk, None, K, R, R, (st, st), (pad, pad))
        y = ops.conv2d_fwd(xh, wk, None, K, R, R, (st, st), (pad, pad), bn_stats=True)
        assert torch.equal(y.cpu(), y_plain.cpu())
        ps = ops.take_pending_stats(y)
        assert ps is not None and ps.rows == (y.numel() // K + 127) // 128 and ops.take_pending_stats(y) is None
        y2 = y.float().cpu().double().reshape(-1, K)
        part = ps.partial.cpu().double()
        tol = 1e-5 if dtype == torch.float32 else 1e-5   # fp32 sums of the *stored* values in both cases
        assert rel_l2(part[:, :K].sum(0), y2.sum(0)) < tol
        assert rel_l2(part[:, K:].sum(0), (y2 * y2).sum(0)) < tol
        # per-tile rows, not just their total
        r0 = y2[:128]
        assert rel_l2(part[0, :K], r0.sum(0)) < tol and rel_l2(part[0, K:], (r0 * r0).sum(0)) < tol

        # BatchNorm from the partials == BatchNorm that re-reads y
        outs = []
        for fused in (True, False):
            bn = ca.nn.BatchNorm2d(K)
            ca.engine.prepare(torch.nn.Sequential(bn), dev, dtype)
            bn.train()
            with torch.no_grad():
                yy = ops.conv2d_fwd(xh, wk, None, K, R, R, (st, st), (pad, pad), bn_stats=fused)
                z = bn(yy, relu=True)
            assert (yy is not None) and getattr(yy, '_cn_stats', None) is None   # taken by the BatchNorm
            outs.append((z.float().cpu(), bn.running_mean.cpu().clone(), bn.running_var.cpu().clone()))
        (zf, mf, vf), (zp, mp, vp) = outs
        assert rel_l2(mf, mp) < 1e-5 and rel_l2(vf, vp) < 1e-5
        assert rel_l2(zf, zp) < (1e-5 if dtype == torch.float32 else 4e-3)


# (N, H, W, C, K, R, stride, pad, use_bits, use_addend)
BNBWD_EMUL = [(2, 9, 9, 16, 64, 3, 1, 1, False, False), (3, 7, 5, 72, 16, 1, 1, 0, True, True),
              (1, 12, 12, 16, 24, 3, 2, 1, False, True), (2, 8, 8, 64, 32, 1, 2, 0, True, False)]
BNBWD_GPU = [(8, 56, 56, 64, 64, 3, 1, 1, False, False), (16, 56, 56, 256, 64, 1, 1, 0, True, True),
             (4, 56, 56, 128, 128, 3, 2, 1, False, False), (4, 56, 56, 256, 512, 1, 2, 0, True, True),
             (5, 14, 14, 1024, 256, 1, 1, 0, True, True), (3, 17, 13, 72, 64, 3, 2, 1, False, True)]


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dgrad_with_subsampled_addend_equals_dense_addend(mode, dtype):
    """addend_sub = 2 (cn_conv2d_dgrad_sa / cn_conv2d_dgrad_bnbwd_sa): the addend holds only the even (h, w) pixels of
    a gradient that is zero elsewhere (the stride-2 1x1 projection shortcut's input gradient); same bits as adding the
    dense tensor, for a 1x1 stride-1 dgrad (Bottleneck conv1) and a 3x3 stride-2 one (BasicBlock conv1), odd sizes too."""
    dev = _dev(mode)
    import convnet_amd as ca
    ops = ca.ops
    ch = ca._lib.chunk_elems(dtype)
    cases = [(2, 9, 7, 16, 24, 1, 1, 0), (1, 10, 12, 16, 16, 3, 2, 1)] if mode == 'emul' else \
        [(8, 56, 56, 256, 128, 1, 1, 0), (4, 56, 56, 64, 128, 3, 2, 1), (3, 17, 13, 64, 72, 1, 1, 0)]
    for (N, H, W, C, K, R, st, pad) in cases:
        if C % ch:
            continue
        g_ = torch.Generator().manual_seed(H + C)
        P, Q = ops.conv_out_hw(H, W, R, R, (st, st), (pad, pad))
        dyh = _nhwc(torch.randn(N, K, P, Q, generator=g_), dtype, dev)
        wc = (torch.randn(C, R, R, K, generator=g_) * 0.1).to(dtype).to(dev)
        compact = _nhwc(torch.randn(N, C, (H + 1) // 2, (W + 1) // 2, generator=g_), dtype, dev)
        dense = torch.zeros(N, H, W, C, dtype=dtype, device=dev)
        dense[:, ::2, ::2, :] = compact
        a = ops.conv2d_dgrad(dyh, wc, (N, H, W, C), K, R, R, (st, st), (pad, pad), addend=dense)
        b = ops.conv2d_dgrad(dyh, wc, (N, H, W, C), K, R, R, (st, st), (pad, pad), addend=compact, addend_sub=2)
        assert torch.equal(a.cpu(), b.cpu()), (N, H, W, C, K, R, st)
        # ... and with the fused BatchNorm-backward reduction
        bn_y = _nhwc(torch.randn(N, C, H, W, generator=g_), dtype, dev)
        yf = bn_y.float().reshape(-1, C)
        mean, invstd = yf.mean(0), 1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-5)
        stats = torch.cat([mean, invstd, invstd, -mean * invstd]).contiguous()
        ga, pa, ra = ops.conv2d_dgrad(dyh, wc, (N, H, W, C), K, R, R, (st, st), (pad, pad), addend=dense,
                                      bn=(bn_y, None, stats, True))
        gb, pb, rb = ops.conv2d_dgrad(dyh, wc, (N, H, W, C), K, R, R, (st, st), (pad, pad), addend=compact,
                                      bn=(bn_y, None, stats, True), addend_sub=2)
        assert torch.equal(ga.cpu(), gb.cpu()) and ra == rb and torch.equal(pa.cpu(), pb.cpu())


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
def test_dgrad_epilogue_bn_backward_reduction(mode, dtype):
    """cn_conv2d_dgrad_bnbwd + cn_bn_bwd_partials == cn_conv2d_dgrad followed by cn_bn_bwd: the masked
    gradient g, the per-tile partial sums, and the BatchNorm input/parameter gradients."""
    dev = _dev(mode)
    _f16_emul_subset(mode, dtype)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    ch = ca._lib.chunk_elems(dtype)
    for (N, H, W, C, K, R, st, pad, use_bits, use_addend) in (BNBWD_EMUL if mode == 'emul' else BNBWD_GPU):
        if C % ch:
            continue
        g_ = torch.Generator().manual_seed(C + K + H)
        P, Q = ops.conv_out_hw(H, W, R, R, (st, st), (pad, pad))
        dyh = _nhwc(torch.randn(N, K, P, Q, generator=g_), dtype, dev)
        wc = (torch.randn(C, R, R, K, generator=g_) * (2.0 / (K * R * R)) ** 0.5).to(dtype).to(dev)
        bn_y = _nhwc(torch.randn(N, C, H, W, generator=g_) * 1.5 + 0.3, dtype, dev)
        addend = _nhwc(torch.randn(N, C, H, W, generator=g_), dtype, dev) if use_addend else None
        M = N * H * W
        # forward state of the BN that produced the conv input: stats = [mean | invstd | scale | shift]
        yf = bn_y.float().reshape(M, C)
        mean, var = yf.mean(0), yf.var(0, unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        gamma = (torch.rand(C, generator=g_) + 0.5).to(dev)
        beta = (torch.randn(C, generator=g_) * 0.2).to(dev)
        stats = torch.cat([mean, invstd, gamma * invstd, beta - mean * gamma * invstd]).contiguous()
        pre = yf * stats[2 * C:3 * C] + stats[3 * C:]
        if use_bits:   # ReLU after a residual add: positive-output bits are an input, not recomputable
            on = torch.rand(M, C, generator=g_).to(dev) > 0.4
            w8 = (2 ** torch.arange(ch, device=dev)).view(1, 1, ch)
            bits = (on.view(M, C // ch, ch).long() * w8).sum(-1).to(torch.uint8).contiguous()
        else:
            on, bits = pre > 0, None
        dx_plain = ops.conv2d_dgrad(dyh, wc, (N, H, W, C), K, R, R, (st, st), (pad, pad), addend=addend)
        g, partial, rows = ops.conv2d_dgrad(dyh, wc, (N, H, W, C), K, R, R, (st, st), (pad, pad), addend=addend,
                                            bn=(bn_y, bits, stats, True))
        g_ref = torch.where(on.view(N, H, W, C), dx_plain, torch.zeros_like(dx_plain))
        assert torch.equal(g.cpu(), g_ref.cpu()), (N, H, W, C, K, R, st)
        if 'jdgrad_kernel' in L.cn_last_kernel_name().decode():    # the large 1x1 junctions run on the streaming kernel
            assert rows == L.cn_conv2d_dgrad_junction_rows(N, H, W, C)
        else:
            assert rows == L.cn_conv2d_dgrad_bnbwd_rows(N, H, W, C, st, st)
        assert tuple(partial.shape) == (rows, 2 * C)
        gd = g.float().reshape(M, C).double()
        xhat = ((yf - mean) * invstd).double()
        s1, s2 = partial[:, :C].double().sum(0), partial[:, C:].double().sum(0)
        assert rel_l2(s1.cpu(), gd.sum(0).cpu()) < 1e-5
        assert rel_l2(s2.cpu(), (gd * xhat).sum(0).cpu()) < 2e-5

        # BatchNorm backward from the partials == the plain three-kernel backward on (dx_plain, mask)
        code = ca._lib.dtype_code(dtype)
        ws = ops.workspace(L.cn_bn_workspace(M, C, code), dev)
        outs = []
        for fused in (True, False):
            dy = torch.empty_like(bn_y)
            dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            coef = torch.empty(3 * C, device=dev)
            if fused:
                ca._lib.check(L.cn_bn_bwd_partials(ca._lib.ptr(g), ca._lib.ptr(bn_y), ca._lib.ptr(gamma),
                                                   ca._lib.ptr(stats), ca._lib.ptr(dy), ca._lib.ptr(dgam),
                                                   ca._lib.ptr(dbet), 0.0, 1.0, ca._lib.ptr(coef), M, C, code,
                                                   ca._lib.ptr(partial), rows, ca._lib.ptr(ws), ws.numel() * 4,
                                                   ca._lib.stream_of(g)), 'cn_bn_bwd_partials')
            else:
                ca._lib.check(L.cn_bn_bwd(ca._lib.ptr(dx_plain), ca._lib.ptr(bn_y), ca._lib.ptr(bits),
                                          ca._lib.ptr(gamma), ca._lib.ptr(stats), ca._lib.ptr(dy), None,
                                          ca._lib.ptr(dgam), ca._lib.ptr(dbet), 0.0, 1.0, ca._lib.ptr(coef), M, C, 1,
                                          code, ca._lib.ptr(ws), ws.numel() * 4, ca._lib.stream_of(g)), 'cn_bn_bwd')
            outs.append((dy.float().cpu(), dgam.cpu(), dbet.cpu()))
        (dyf, gf, bf), (dyp, gp, bp) = outs
        assert rel_l2(gf, gp) < 1e-4 and rel_l2(bf, bp) < 1e-4
        assert rel_l2(dyf, dyp) < (1e-5 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
def test_sync_batchnorm_building_blocks(mode, dtype):
    """The --sync-bn kernels (local double-precision sums -> caller's all-reduce -> forward / backward
    from global sums) reproduce the single-rank BatchNorm when the 'global' sums are the local ones, and
    a two-shard split (sums added on the host, as the all-reduce would) reproduces BatchNorm over the
    concatenated batch."""
    dev = _dev(mode)
    _f16_emul_subset(mode, dtype)
    import convnet_amd as ca
    lib, L, ops = ca._lib, ca._lib.load(), ca.ops
    code = lib.dtype_code(dtype)
    for (N, C, H, W) in ([(4, 16, 5, 5), (2, 72, 3, 7)] if mode == 'emul' else [(8, 64, 56, 56), (6, 136, 9, 5)]):
        g = torch.Generator().manual_seed(C)
        y = _nhwc(torch.randn(N, C, H, W, generator=g) * 2 + 0.5, dtype, dev)
        dz = _nhwc(torch.randn(N, C, H, W, generator=g), dtype, dev)
        gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
        beta = (torch.randn(C, generator=g) * 0.1).to(dev)
        M = N * H * W
        ws = ops.workspace(L.cn_bn_workspace(M, C, code), dev)

        def plain(yy, dzz, m):
            z, dy = torch.empty_like(yy), torch.empty_like(yy)
            st, coef = torch.empty(4 * C, device=dev), torch.empty(3 * C, device=dev)
            rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            lib.check(L.cn_bn_fwd_train(lib.ptr(yy), None, lib.ptr(z), None, lib.ptr(gamma), lib.ptr(beta), lib.ptr(rm),
                                        lib.ptr(rv), None, 0.1, 1e-5, lib.ptr(st), m, C, 1, code, lib.ptr(ws),
                                        ws.numel() * 4, lib.stream_of(yy)))
            lib.check(L.cn_bn_bwd(lib.ptr(dzz), lib.ptr(yy), None, lib.ptr(gamma), lib.ptr(st), lib.ptr(dy), None,
                                  lib.ptr(dg), lib.ptr(db), 0.0, 1.0, lib.ptr(coef), m, C, 1, code, lib.ptr(ws),
                                  ws.numel() * 4, lib.stream_of(yy)))
            return z.float().cpu(), dy.float().cpu(), dg.cpu(), db.cpu(), rm.cpu(), rv.cpu()

        def synced(shards):
            """shards: list of (y, dz) per 'rank'; returns per-rank outputs with host-summed sums."""
            m_tot = sum(yy.numel() // C for yy, _ in shards)
            fs = []
            for yy, _ in shards:
                sm = torch.empty(2 * C, dtype=torch.float64, device=dev)
                lib.check(L.cn_bn_local_sums(lib.ptr(yy), yy.numel() // C, C, code, None, 0, lib.ptr(sm), lib.ptr(ws),
                                             ws.numel() * 4, lib.stream_of(yy)))
                fs.append(sm)
            gsum = torch.stack(fs).sum(0).contiguous()
            outs, states = [], []
            for yy, _ in shards:
                m = yy.numel() // C
                z = torch.empty_like(yy)
                st = torch.empty(4 * C, device=dev)
                rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
                lib.check(L.cn_bn_fwd_train_sums(lib.ptr(yy), None, lib.ptr(z), None, lib.ptr(gamma), lib.ptr(beta),
                                                 lib.ptr(rm), lib.ptr(rv), None, 0.1, 1e-5, lib.ptr(st), m, C, 1, code,
                                                 lib.ptr(gsum), m_tot, lib.stream_of(yy)))
                states.append((z, st, rm, rv))
            ls = []
            for (yy, dzz), (z, st, rm, rv) in zip(shards, states):
                sm = torch.empty(2 * C, dtype=torch.float64, device=dev)
                lib.check(L.cn_bn_bwd_local_sums(lib.ptr(dzz), lib.ptr(yy), None, lib.ptr(st), yy.numel() // C, C, 1,
                                                 code, None, 0, lib.ptr(sm), lib.ptr(ws), ws.numel() * 4,
                                                 lib.stream_of(yy)))
                ls.append(sm)
            gl = torch.stack(ls).sum(0).contiguous()
            for (yy, dzz), (z, st, rm, rv), loc in zip(shards, states, ls):
                m = yy.numel() // C
                dy, coef = torch.empty_like(yy), torch.empty(3 * C, device=dev)
                dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
                lib.check(L.cn_bn_bwd_sums(lib.ptr(dzz), lib.ptr(yy), None, lib.ptr(gamma), lib.ptr(st), lib.ptr(dy), None,
                                           lib.ptr(dg), lib.ptr(db), 0.0, 1.0, lib.ptr(coef), m, C, 1, 0, code,
                                           lib.ptr(loc), lib.ptr(gl), m_tot, lib.stream_of(yy)))
                outs.append((z.float().cpu(), dy.float().cpu(), dg.cpu(), db.cpu(), rm.cpu(), rv.cpu()))
            return outs

        ref = plain(y, dz, M)
        one = synced([(y, dz)])[0]
        tol = 1e-5 if dtype == torch.float32 else 4e-3
        for a, b in zip(one, ref):
            assert rel_l2(a, b) < max(tol, 2e-5)
        h = N // 2
        two = synced([(y[:h].contiguous(), dz[:h].contiguous()), (y[h:].contiguous(), dz[h:].contiguous())])
        assert rel_l2(torch.cat([two[0][0], two[1][0]]), ref[0]) < tol
        assert rel_l2(torch.cat([two[0][1], two[1][1]]), ref[1]) < tol
        assert rel_l2(two[0][2] + two[1][2], ref[2]) < 1e-4 and rel_l2(two[0][3] + two[1][3], ref[3]) < 1e-4
        assert rel_l2(two[0][4], ref[4]) < 1e-5 and rel_l2(two[1][5], ref[5]) < 1e-5


@pytest.mark.parametrize('mode', MODES)
def test_stem_pixel_pair_convolution(mode):
    """The stride-2 stem in pixel-pair form (cn_nchw_to_pairs + cn_weight_prep_pairs + igemm with
    ceil(S/2) taps per row + cn_wgrad_unpack_pairs) against F.conv2d (models/resnet.py:226: 7x7/2 pad 3,
    3 input channels), plus an even kernel and a 1-channel input."""
    dev = _dev(mode)
    import convnet_amd as ca
    cases = [(2, 3, 20, 20, 64, 7, 3), (1, 1, 12, 18, 16, 4, 1), (2, 4, 9, 14, 72, 3, 1)]
    if mode == 'gpu':
        cases += [(4, 3, 224, 224, 64, 7, 3)]
    for (N, C, H, W, K, R, pad) in cases:
        g = torch.Generator().manual_seed(H + K)
        x = _q(torch.randn(N, C, H, W, generator=g), torch.bfloat16)
        conv = ca.nn.Conv2d(C, K, kernel_size=R, stride=2, padding=pad, bias=False)
        conv.needs_dgrad = False
        model = torch.nn.Sequential(conv)
        ca.engine.prepare(model, dev, torch.bfloat16)
        w0 = _q(torch.randn(K, C, R, R, generator=g) * (2.0 / (C * R * R)) ** 0.5, torch.bfloat16)
        conv.weight.data.copy_(w0.to(dev))
        model._cn_arena.bump_version()
        assert conv.pair_eligible(x)
        xr, wr = x.clone().requires_grad_(False), w0.clone().requires_grad_(True)
        y_ref = F.conv2d(xr, wr, stride=2, padding=pad)
        dy = _q(torch.randn(y_ref.shape, generator=g), torch.bfloat16)
        y_ref.backward(dy)
        model._cn_arena.zero_grad()
        y = conv.forward_from_nchw(x.to(dev))
        assert tuple(y.shape) == (N, y_ref.shape[2], y_ref.shape[3], K)
        assert rel_l2(y.float().cpu().permute(0, 3, 1, 2), y_ref.detach()) < 1e-2
        y.backward(_nhwc(dy, torch.bfloat16, dev))
        ca.ops.SIDE.join(dev) if dev.type == 'cuda' else None
        assert rel_l2(conv.weight.grad.float().cpu(), wr.grad) < 1e-2
        # and it agrees with the channel-padded 49-tap form of the same conv
        model._cn_arena.zero_grad()
        y2 = conv(ca.nn.to_nhwc(x.to(dev), torch.bfloat16, conv.padded_in_channels()))
        assert rel_l2(y.float().cpu(), y2.float().cpu()) < 4e-3


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
def test_stem_bn_relu_maxpool_fused_equals_chain(mode, dtype):
    """ops.BnReluMaxPoolFunction (bn1 -> relu -> maxpool of models/resnet.py:228-230 in one pooling pass,
    backward with the pool's gather folded into the BatchNorm backward) against the separate BatchNorm2d
    and MaxPool2d modules: same pooled map bit for bit, same gradients / statistics."""
    dev = _dev(mode)
    _f16_emul_subset(mode, dtype)
    import convnet_amd as ca
    shapes = [(2, 16, 9, 11), (3, 8, 12, 12)] if mode == 'emul' else [(4, 64, 112, 112), (3, 72, 17, 13)]
    for (N, C, H, W) in shapes:
        g = torch.Generator().manual_seed(C + H)
        y0 = _nhwc(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.2, dtype, dev)
        outs = []
        for fused in (True, False):
            bn, pool = ca.nn.BatchNorm2d(C), ca.nn.MaxPool2d(3, 2, 1)
            model = torch.nn.Sequential(bn)
            ca.engine.prepare(model, dev, dtype)
            gw = torch.Generator().manual_seed(5)
            bn.weight.data.copy_((torch.rand(C, generator=gw) + 0.5).to(dev))
            bn.bias.data.copy_((torch.randn(C, generator=gw) * 0.3).to(dev))
            bn.train()
            y = y0.clone().requires_grad_(True)
            ca.ops.FUSE_STEM_POOL = fused
            try:
                out = ca.nn.bn_relu_maxpool(bn, pool, y)
            finally:
                ca.ops.FUSE_STEM_POOL = True
            model._cn_arena.zero_grad()
            gd = torch.Generator().manual_seed(9)
            dout = _q(torch.randn(out.shape, generator=gd), dtype).to(dtype).to(dev)
            out.backward(dout)
            outs.append((out.detach().float().cpu(), y.grad.float().cpu(), bn.weight.grad.cpu().clone(),
                         bn.bias.grad.cpu().clone(), bn.running_mean.cpu().clone(), bn.running_var.cpu().clone()))
        f, u = outs
        assert torch.equal(f[0], u[0])
        tol = 1e-5 if dtype == torch.float32 else 4e-3
        assert rel_l2(f[1], u[1]) < tol
        # dgamma / dbeta: the fused form sums the pooled gradients unrounded (over the pooled map, ops.STEM_XMAX); the
        # chain rounds the max-pool backward's output to the storage type first - one storage epsilon on a few dozen
        # terms per channel at these sizes
        ptol = 1e-4 if dtype == torch.float32 else 4e-3
        assert rel_l2(f[2], u[2]) < ptol and rel_l2(f[3], u[3]) < ptol
        assert torch.equal(f[4], u[4]) and torch.equal(f[5], u[5])


@pytest.mark.parametrize('mode', MODES)
def test_batchnorm_statistics_are_centred_on_the_running_mean(mode):
    """|mean| = 1000 sigma: E[x^2] - E[x]^2 in fp32 partials loses the variance; with the sums taken of
    (y - running_mean) - standalone statistics pass and convolution epilogue alike - mean and invstd come out to
    fp32 accuracy once the running mean is near the batch mean (bn.hip variance)."""
    dev = _dev(mode)
    import convnet_amd as ca
    lib, L, ops = ca._lib, ca._lib.load(), ca.ops
    code = lib.F32
    N, H, W, C = (2, 12, 12, 16) if mode == 'emul' else (8, 56, 56, 64)
    M = N * H * W
    g = torch.Generator().manual_seed(3)
    mu = (torch.rand(C, generator=g) + 0.5) * 1000.0 * torch.where(torch.rand(C, generator=g) > 0.5, 1.0, -1.0)
    x = (mu.view(1, 1, 1, C) + torch.randn(N, H, W, C, generator=g)).to(dev)
    xd = x.double().cpu().reshape(M, C)
    mean_ref, var_ref = xd.mean(0), xd.var(0, unbiased=False)
    invstd_ref = 1.0 / torch.sqrt(var_ref + 1e-5)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ws = ops.workspace(L.cn_bn_workspace(M, C, code), dev)
    w_eye = torch.eye(C, device=dev).view(C, 1, 1, C).contiguous()            # 1x1 identity convolution: y == x
    for fused in (False, True):
        rm = (mu * 0.999).to(dev).contiguous()      # a running mean that has converged to within 0.1 % (= 1 sigma)
        rv = torch.ones(C, device=dev)
        st = torch.empty(4 * C, device=dev)
        z = torch.empty_like(x)
        if fused:
            y = ops.conv2d_fwd(x, w_eye, None, C, 1, 1, (1, 1), (0, 0), bn_stats=True, pivot=rm)
            assert torch.equal(y, x)
            ps = ops.take_pending_stats(y)
            assert ps is not None and ps.pivot == rm.data_ptr()
            lib.check(L.cn_bn_fwd_train_partials_centered(lib.ptr(y), None, lib.ptr(z), None, lib.ptr(gamma), lib.ptr(beta),
                                                          lib.ptr(rm), lib.ptr(rv), None, 0.1, 1e-5, lib.ptr(st), M, C, 0,
                                                          code, lib.ptr(ps.partial), ps.rows, lib.ptr(ws), ws.numel() * 4,
                                                          lib.stream_of(y)))
        else:
            lib.check(L.cn_bn_fwd_train(lib.ptr(x), None, lib.ptr(z), None, lib.ptr(gamma), lib.ptr(beta), lib.ptr(rm),
                                        lib.ptr(rv), None, 0.1, 1e-5, lib.ptr(st), M, C, 0, code, lib.ptr(ws),
                                        ws.numel() * 4, lib.stream_of(x)))
        mean, invstd = st[:C].double().cpu(), st[C:2 * C].double().cpu()
        assert float(((mean - mean_ref).abs() / mean_ref.abs()).max()) < 1e-6, fused
        assert float(((invstd - invstd_ref).abs() / invstd_ref).max()) < 1e-3, fused      # plain sums: off by 10-100 %
        zr = ((xd - mean_ref) * invstd_ref).float()
        assert float((z.cpu().reshape(M, C) - zr).abs().max()) < 2e-2, fused    # x itself carries 6e-5 of rounding at |x| = 1000
        # running statistics moved from the pivot towards the batch statistics
        assert rel_l2(rm.cpu(), (0.9 * mu * 0.999 + 0.1 * mean_ref.float())) < 1e-6


@pytest.mark.parametrize('mode', MODES)
def test_batchnorm_from_many_partial_rows(mode):
    """cn_bn_fwd_train_partials / cn_bn_bwd_partials with more than 512 partial rows (the 1024-thread
    finalize that replaces a separate row-compression launch) against the standalone passes."""
    dev = _dev(mode)
    import convnet_amd as ca
    lib, L, ops = ca._lib, ca._lib.load(), ca.ops
    dtype, code = torch.float32, ca._lib.F32
    C, rows, per = 16, 1300, 2
    M = rows * per
    g = torch.Generator().manual_seed(1)
    y = (torch.randn(M, C, generator=g) * 1.7 + 0.4).to(dev)
    dz = torch.randn(M, C, generator=g).to(dev)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.1).to(dev)
    ws = ops.workspace(L.cn_bn_workspace(M, C, code), dev)
    yr = y.view(rows, per, C)
    partial = torch.cat([yr.sum(1), (yr * yr).sum(1)], dim=1).contiguous()       # [rows][2C]
    outs = []
    for use_partials in (True, False):
        z = torch.empty_like(y)
        st = torch.empty(4 * C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        if use_partials:
            lib.check(L.cn_bn_fwd_train_partials(lib.ptr(y), None, lib.ptr(z), None, lib.ptr(gamma), lib.ptr(beta),
                                                 lib.ptr(rm), lib.ptr(rv), None, 0.1, 1e-5, lib.ptr(st), M, C, 1, code,
                                                 lib.ptr(partial), rows, lib.ptr(ws), ws.numel() * 4, lib.stream_of(y)))
        else:
            lib.check(L.cn_bn_fwd_train(lib.ptr(y), None, lib.ptr(z), None, lib.ptr(gamma), lib.ptr(beta), lib.ptr(rm),
                                        lib.ptr(rv), None, 0.1, 1e-5, lib.ptr(st), M, C, 1, code, lib.ptr(ws),
                                        ws.numel() * 4, lib.stream_of(y)))
        # backward: g = dz * relu mask, partial rows of [sum g | sum g*xhat]
        mean, invstd = st[:C], st[C:2 * C]
        gm = torch.where(z > 0, dz, torch.zeros_like(dz))
        dy = torch.empty_like(y)
        dg, db, coef = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.empty(3 * C, device=dev)
        if use_partials:
            xhat = (y - mean) * invstd
            bp = torch.cat([gm.view(rows, per, C).sum(1), (gm * xhat).view(rows, per, C).sum(1)], dim=1).contiguous()
            lib.check(L.cn_bn_bwd_partials(lib.ptr(gm), lib.ptr(y), lib.ptr(gamma), lib.ptr(st), lib.ptr(dy), lib.ptr(dg),
                                           lib.ptr(db), 0.0, 1.0, lib.ptr(coef), M, C, code, lib.ptr(bp), rows,
                                           lib.ptr(ws), ws.numel() * 4, lib.stream_of(y)))
        else:
            lib.check(L.cn_bn_bwd(lib.ptr(dz), lib.ptr(y), None, lib.ptr(gamma), lib.ptr(st), lib.ptr(dy), None,
                                  lib.ptr(dg), lib.ptr(db), 0.0, 1.0, lib.ptr(coef), M, C, 1, code, lib.ptr(ws),
                                  ws.numel() * 4, lib.stream_of(y)))
        outs.append([t.float().cpu() for t in (z, st, rm, rv, dy, dg, db)])
    for a, b in zip(*outs):
        assert rel_l2(a, b) < 2e-5


@pytest.mark.parametrize('mode', MODES)
def test_wgrad_accumulates_and_scales(mode):
    dev = _dev(mode)
    import convnet_amd as ca
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 6, 6, 8, generator=g).to(dev)
    dy = torch.randn(2, 6, 6, 64, generator=g).to(dev)
    a = torch.zeros(64, 3, 3, 8, device=dev)
    ca.ops.conv2d_wgrad(x, dy, a, 8, 64, 3, 3, (1, 1), (1, 1), beta=0.0)
    b = a.clone()
    ca.ops.conv2d_wgrad(x, dy, b, 8, 64, 3, 3, (1, 1), (1, 1), beta=1.0, scale=0.5)
    assert rel_l2(b.cpu(), 1.5 * a.cpu()) < 1e-6
    c = torch.zeros_like(a)
    ca.ops.conv2d_wgrad(x, dy, c, 8, 64, 3, 3, (1, 1), (1, 1), beta=0.0)
    assert torch.equal(a, c), 'split reduction must be run-to-run deterministic'


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
def test_linear_with_bias_fp32_logits(mode, dtype):
    dev = _dev(mode)
    _f16_emul_subset(mode, dtype)
    import convnet_amd as ca
    g = torch.Generator().manual_seed(1)
    B, C, K = (5, 64, 1000) if mode == 'emul' else (256, 2048, 1000)
    x = _q(torch.randn(B, C, generator=g), dtype)
    w = _q(torch.randn(K, C, generator=g) * 0.05, dtype)
    b = torch.randn(K, generator=g)
    y_ref = F.linear(x, w, b)
    y = ca.ops.conv2d_fwd(x.view(B, 1, 1, C).to(dtype).to(dev), w.to(dtype).to(dev), b.to(dev), K, 1, 1, (1, 1),
                          (0, 0), out_f32=True)
    assert y.dtype == torch.float32
    assert rel_l2(y.cpu().view(B, K), y_ref) < (1e-5 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize('mode', MODES)
def test_classifier_runs_on_the_small_batch_dense_kernel(mode):
    """Round 6 (csrc/dense.hip): nn.Linear on a batch - forward and data gradient - runs as 32 x 32 output tiles with a
    four-way split of the reduction instead of 16 workgroups of the 128 x 128 tile; ragged batch (rows beyond M), ragged
    width (1000 = 31 tiles + 8 columns) and a reduction length that is not a multiple of 16 (the data gradient reduces over
    the 1000 outputs) are all in this one shape."""
    dev = _dev(mode)
    import convnet_amd as ca
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(2)
    B, C, K = (37, 96, 1000) if mode == 'emul' else (256, 2048, 1000)
    x = _q(torch.randn(B, C, generator=g), dtype)
    w = _q(torch.randn(K, C, generator=g) * 0.05, dtype)
    b = torch.randn(K, generator=g)
    dy = _q(torch.randn(B, K, generator=g), dtype)
    L = ca._lib.load()
    y = ca.ops.conv2d_fwd(x.view(B, 1, 1, C).to(dtype).to(dev), w.to(dtype).to(dev), b.to(dev), K, 1, 1, (1, 1), (0, 0),
                          out_f32=True)
    assert 'dense_smallm_kernel' in L.cn_last_kernel_name().decode()
    assert rel_l2(y.cpu().view(B, K), F.linear(x, w, b)) < 2e-3
    w_crsk = w.t().contiguous()          # [C][K]: the copy the data gradient reads
    dx = ca.ops.conv2d_dgrad(dy.view(B, 1, 1, K).to(dtype).to(dev), w_crsk.to(dtype).to(dev), (B, 1, 1, C), K, 1, 1,
                             (1, 1), (0, 0))
    assert 'dense_smallm_kernel' in L.cn_last_kernel_name().decode()
    assert dx.dtype == dtype
    assert rel_l2(dx.float().cpu().view(B, C), dy @ w) < 5e-3
    # the tiled kernel (option dense_smallm = 0) gives the same numbers up to the summation order
    L.cn_set_option(b'dense_smallm', 0)
    try:
        y0 = ca.ops.conv2d_fwd(x.view(B, 1, 1, C).to(dtype).to(dev), w.to(dtype).to(dev), b.to(dev), K, 1, 1, (1, 1),
                               (0, 0), out_f32=True)
        assert 'igemm' in L.cn_last_kernel_name().decode()
    finally:
        L.cn_set_option(b'dense_smallm', 1)
    assert rel_l2(y.cpu(), y0.cpu()) < 1e-5


def _bn_ref(y, gamma, beta, res, relu, eps=1e-5, momentum=0.1):
    y = y.clone().requires_grad_(True)
    gamma = gamma.clone().requires_grad_(True)
    beta = beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(y.shape[1]), torch.ones(y.shape[1])
    z = F.batch_norm(y, rm, rv, gamma, beta, True, momentum, eps)
    r = None
    if res is not None:
        r = res.clone().requires_grad_(True)
        z = z + r
    if relu:
        z = F.relu(z)
    return y, gamma, beta, r, z, rm, rv


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('relu,use_res', [(False, False), (True, False), (True, True)])
def test_batchnorm_train_fwd_bwd(mode, dtype, relu, use_res):
    dev = _dev(mode)
    _f16_emul_subset(mode, dtype, keep=True)
    import convnet_amd as ca
    shapes = [(4, 16, 5, 5), (2, 72, 3, 7)] if mode == 'emul' else [(8, 64, 56, 56), (4, 2048, 7, 7), (3, 136, 9, 5)]
    for (N, C, H, W) in shapes:
        g = torch.Generator().manual_seed(N + C)
        y0 = _q(torch.randn(N, C, H, W, generator=g) * 2 + 0.5, dtype)
        res0 = _q(torch.randn(N, C, H, W, generator=g), dtype) if use_res else None
        gamma0, beta0 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
        dz = _q(torch.randn(N, C, H, W, generator=g), dtype)
        y, gamma, beta, r, z, rm, rv = _bn_ref(y0, gamma0, beta0, res0, relu)
        z.backward(dz)

        bn = ca.nn.BatchNorm2d(C)
        model = torch.nn.Sequential(bn)
        ca.engine.prepare(model, dev, dtype)
        bn.weight.data.copy_(gamma0.to(dev))
        bn.bias.data.copy_(beta0.to(dev))
        bn.train()
        yh = _nhwc(y0, dtype, dev).requires_grad_(True)
        rh = _nhwc(res0, dtype, dev).requires_grad_(True) if use_res else None
        zh = bn(yh, residual=rh, relu=relu)
        model._cn_arena.zero_grad()
        zh.backward(_nhwc(dz, dtype, dev))
        t = _tol(dtype)
        assert rel_l2(zh.detach().float().cpu().permute(0, 3, 1, 2), z.detach()) < t
        assert rel_l2(bn.running_mean.cpu(), rm) < 1e-4 and rel_l2(bn.running_var.cpu(), rv) < 1e-4
        assert int(bn.num_batches_tracked) == 1
        tg = _tol(dtype, True) if dtype == torch.float32 else 1.5e-2
        assert rel_l2(yh.grad.float().cpu().permute(0, 3, 1, 2), y.grad) < tg
        assert rel_l2(bn.weight.grad.cpu(), gamma.grad) < max(tg, 2e-4)
        assert rel_l2(bn.bias.grad.cpu(), beta.grad) < max(tg, 2e-4)
        if use_res:
            assert rel_l2(rh.grad.float().cpu().permute(0, 3, 1, 2), r.grad) < tg
        # inference path from the running statistics
        bn.eval()
        with torch.no_grad():
            ze = bn(yh.detach(), residual=rh.detach() if use_res else None, relu=relu)
        z_ref = F.batch_norm(y0, rm, rv, gamma0, beta0, False, 0.1, 1e-5)
        if use_res:
            z_ref = z_ref + res0
        if relu:
            z_ref = F.relu(z_ref)
        assert rel_l2(ze.float().cpu().permute(0, 3, 1, 2), z_ref) < t


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
def test_maxpool_fwd_bwd_with_ties(mode, dtype):
    dev = _dev(mode)
    _f16_emul_subset(mode, dtype)
    import convnet_amd as ca
    shapes = [(2, 8, 9, 9, 3, 2, 1), (1, 16, 8, 8, 2, 2, 0)] if mode == 'emul' else \
        [(4, 64, 112, 112, 3, 2, 1), (2, 32, 26, 26, 2, 2, 0), (3, 8, 13, 13, 2, 2, 0)]
    for (N, C, H, W, k, st, pad) in shapes:
        g = torch.Generator().manual_seed(C)
        x0 = F.relu(_q(torch.randn(N, C, H, W, generator=g), dtype))   # post-ReLU: many exact ties at 0
        x = x0.clone().requires_grad_(True)
        y = F.max_pool2d(x, k, st, pad)
        dy = _q(torch.randn(y.shape, generator=g), dtype)
        y.backward(dy)
        xh = _nhwc(x0, dtype, dev).requires_grad_(True)
        yh = ca.ops.MaxPool2dFunction.apply(xh, k, st, pad)
        yh.backward(_nhwc(dy, dtype, dev))
        assert torch.equal(yh.detach().float().cpu().permute(0, 3, 1, 2), y.detach())
        assert rel_l2(xh.grad.float().cpu().permute(0, 3, 1, 2), x.grad) < (1e-6 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
def test_global_avgpool_and_fork_and_relu(mode, dtype):
    dev = _dev(mode)
    _f16_emul_subset(mode, dtype)
    import convnet_amd as ca
    N, C, H, W = (2, 16, 7, 7) if mode == 'emul' else (8, 2048, 7, 7)
    g = torch.Generator().manual_seed(5)
    x0 = _q(torch.randn(N, C, H, W, generator=g), dtype)
    x = x0.clone().requires_grad_(True)
    a, b = x, x
    out = F.adaptive_avg_pool2d(F.relu(a), 1).flatten(1).sum() * 2 + (b * b).sum()
    out.backward()
    xh = _nhwc(x0, dtype, dev).requires_grad_(True)
    xa, xb = ca.nn.fork(xh)
    p = ca.ops.GlobalAvgPoolFunction.apply(ca.ops.ReLUFunction.apply(xa))
    assert rel_l2(p.detach().float().cpu().view(N, C), F.adaptive_avg_pool2d(F.relu(x0), 1).flatten(1)) < _tol(dtype)
    # hand the two branch gradients to autograd: d/dxa = relu'(x) * 2/(HW), d/dxb = 2x
    gp = torch.full(p.shape, 2.0, dtype=dtype, device=dev)
    torch.autograd.backward([p, xb], [gp, (2 * xh.detach().float()).to(dtype)])
    assert rel_l2(xh.grad.float().cpu().permute(0, 3, 1, 2), x.grad) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('smooth', [0.0, 0.1])
def test_softmax_cross_entropy_and_accuracy(mode, smooth):
    dev = _dev(mode)
    import convnet_amd as ca
    from oracle import convnet_oracle as O
    B, K = (8, 16) if mode == 'emul' else (256, 1000)
    g = torch.Generator().manual_seed(9)
    logits0 = torch.randn(B, K, generator=g) * 3
    target = torch.randint(0, K, (B,), generator=g)
    logits0[0, target[0]] = logits0[0].max() + 1   # at least one top-1 hit
    lr = logits0.clone().requires_grad_(True)
    loss_ref = O.oracle_cross_entropy(lr, target, smooth)
    (loss_ref * 3.0).backward()
    crit = ca.CrossEntropyLoss(smooth_eps=smooth)
    meters = torch.zeros(8, device=dev)
    crit.meters = meters
    lh = logits0.to(dev).requires_grad_(True)
    loss = crit(lh, target.to(dev))
    (loss * 3.0).backward()
    assert float(loss) == pytest.approx(float(loss_ref), rel=2e-6)
    assert rel_l2(lh.grad.cpu(), lr.grad) < 1e-5
    p1, p5 = O.oracle_accuracy(logits0, target, (1, 5))
    m = meters.cpu().tolist()
    assert m[3] == B and m[1] / B == pytest.approx(p1) and m[2] / B == pytest.approx(p5)
    assert m[0] / B == pytest.approx(float(loss_ref), rel=2e-6)
    a1, a5 = ca.accuracy(logits0.to(dev), target.to(dev), (1, 5))
    assert float(a1) == pytest.approx(p1) and float(a5) == pytest.approx(p5)


@pytest.mark.parametrize('mode', MODES)
def test_sgd_momentum_weight_decay_clip(mode):
    dev = _dev(mode)
    import convnet_amd as ca
    from convnet_amd._lib import check, load, ptr
    n = 1000 + 3 if mode == 'emul' else 1_000_003   # odd tail exercised
    g = torch.Generator().manual_seed(2)
    p0, g0 = torch.randn(n, generator=g), torch.randn(n, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.1, momentum=0.9)
    npad = (n + 3) // 4 * 4
    p = torch.zeros(npad, device=dev); p[:n] = p0.to(dev)
    gr = torch.zeros(npad, device=dev)
    buf = torch.zeros(npad, device=dev)
    norm_out = torch.zeros(2, device=dev)
    ws = torch.zeros(load().cn_grad_norm_workspace() // 4, device=dev)
    for step in range(3):
        gs = g0 * (step + 1)
        pr.grad = (gs / 8.0).clone()                       # reference: p.grad.div_(loss_scale)
        total = torch.nn.utils.clip_grad_norm_([pr], 5.0)   # then clip (trainer.py:171-172)
        pr.grad.add_(pr.detach(), alpha=1e-4)               # WeightDecay regulariser, then SGD
        opt.step()
        gr[:n] = gs.to(dev)
        check(load().cn_grad_norm_clip(ptr(gr), npad, 1.0 / 8.0, 5.0, ptr(norm_out), None, 0.0, ptr(ws), None
                                       if dev.type == 'cpu' else torch.cuda.current_stream().cuda_stream))
        # device-resident (lr, momentum) on odd steps: same arithmetic
        hyp = torch.tensor([0.1, 0.9], device=dev) if step % 2 else None
        check(load().cn_sgd_momentum(ptr(p), ptr(gr), ptr(buf), n, 0.1 if hyp is None else 7.0, 0.9 if hyp is None else 0.0,
                                     1e-4, 1.0 / 8.0, ptr(norm_out[1:]), ptr(hyp),
                                     None if dev.type == 'cpu' else torch.cuda.current_stream().cuda_stream))
        assert float(norm_out[0]) == pytest.approx(float(total), rel=1e-4)   # fp32 vs double summation
        assert rel_l2(p[:n].cpu(), pr.detach()) < 1e-6


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
def test_layout_conversion_roundtrip(mode, dtype):
    dev = _dev(mode)
    _f16_emul_subset(mode, dtype)
    import convnet_amd as ca
    N, C, H, W = (2, 3, 6, 5) if mode == 'emul' else (16, 3, 224, 224)
    x = torch.randn(N, C, H, W)
    xh = ca.ops.nchw_to_nhwc(x.to(dev), dtype)
    cp = xh.shape[-1]
    assert cp % (4 if dtype == torch.float32 else 8) == 0 and cp >= C
    assert torch.equal(xh[..., :C].float().cpu(), x.permute(0, 2, 3, 1).to(dtype).float())
    assert float(xh[..., C:].float().abs().sum()) == 0.0
    back = ca.ops.nhwc_to_nchw(xh, C)
    assert torch.equal(back.cpu(), x.to(dtype).float())


@pytest.mark.parametrize('mode', MODES)
def test_igemm_256x256_tile_is_bit_identical_to_128x128(mode):
    """The 8-wave 256x256 tile (long reductions on >= 256-channel layers; forced here with the igemm_256sq knob)
    accumulates every output in the same order as the 128x128 tile: identical y / dx bits, and statistics
    partial rows that carry each 256-pixel tile's sums in the first of its two 128-pixel rows."""
    dev = _dev(mode)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    dtype = torch.bfloat16
    cfgs = [(2, 12, 12, 16, 256, 3, 1, 1), (1, 18, 17, 64, 384, 1, 1, 0)] if mode == 'emul' else \
        [(8, 14, 14, 256, 256, 3, 1, 1), (6, 14, 14, 1024, 256, 1, 1, 0), (4, 28, 28, 256, 256, 3, 2, 1),
         (3, 14, 15, 512, 384, 1, 1, 0)]
    try:
        for (N, H, W, C, K, R, st, pad) in cfgs:
            g = torch.Generator().manual_seed(K + H)
            xh = _nhwc(torch.randn(N, C, H, W, generator=g), dtype, dev)
            wk = (torch.randn(K, R, R, C, generator=g) * (2.0 / (C * R * R)) ** 0.5).to(dtype).to(dev)
            res = {}
            for big in (0, 1):
                L.cn_set_option(b'igemm_256sq', big)
                y = ops.conv2d_fwd(xh, wk, None, K, R, R, (st, st), (pad, pad))
                name = L.cn_last_kernel_name().decode()
                assert ('4, 2, 2, 4' in name) == bool(big), name
                ys = ops.conv2d_fwd(xh, wk, None, K, R, R, (st, st), (pad, pad), bn_stats=True)
                ps = ops.take_pending_stats(ys)
                # dgrad of the transposed problem: gradient w.r.t. a K-channel input from a C-channel dy
                dy = _nhwc(torch.randn(N, C, H, W, generator=torch.Generator().manual_seed(5)), dtype, dev)
                wt = (torch.randn(C, R, R, K, generator=torch.Generator().manual_seed(6)) * 0.05).to(dtype).to(dev)
                Hin, Win = (H - 1) * st + R - 2 * pad, (W - 1) * st + R - 2 * pad
                dx = ops.conv2d_dgrad(dy, wt.permute(3, 1, 2, 0).contiguous(), (N, Hin, Win, K), C, R, R, (st, st),
                                      (pad, pad))
                res[big] = (y.cpu(), ys.cpu(), ps.partial.cpu().double(), ps.rows, dx.cpu())
            y0, ys0, p0, r0, dx0 = res[0]
            y1, ys1, p1, r1, dx1 = res[1]
            assert torch.equal(y0, y1) and torch.equal(ys0, ys1) and torch.equal(ys1, y1) and torch.equal(dx0, dx1)
            assert r0 == r1 and rel_l2(p1.sum(0), p0.sum(0)) < 1e-6
            y2 = ys1.float().double().reshape(-1, K)
            assert rel_l2(p1[0, :K], y2[:256].sum(0)) < 1e-5 and rel_l2(p1[0, K:], (y2[:256] ** 2).sum(0)) < 1e-5
            if r1 > 1:
                assert float(p1[1].abs().max()) == 0.0
    finally:
        L.cn_set_option(b'igemm_256sq', -1)


@pytest.mark.parametrize('mode', MODES)
def test_igemm_interleaved_dma_tiles_equal_the_register_staged_tile(mode):
    """The LDS-DMA tiles with interleaved DMA issue (round 3: the DMA instructions of K tile kt+1 are issued between the
    MFMAs of tile kt) - the 256x256 tile and the 128x128 tile - against the register-staged 128x128 tile: same loads,
    same accumulation order, so the stored outputs are identical bit for bit (forward with the statistics epilogue, and
    dgrad), including reductions that end in a partial K tile and a single-tile reduction (where every interleaved DMA is
    an out-of-range no-op).  (The block-issue LDS-DMA forms this test compared them with in round 3 were removed in
    round 4.)"""
    dev = _dev(mode)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    dtype = torch.bfloat16
    # (N, H, W, C, K, R, stride, pad, knobs): igemm_256sq forces the big tile; igemm_variant=3 the 128x128 LDS-DMA tile
    cfgs = [(2, 12, 12, 16, 256, 3, 1, 1, {'igemm_256sq': 1}), (1, 9, 10, 64, 384, 1, 1, 0, {'igemm_256sq': 1}),
            (2, 8, 9, 24, 128, 3, 1, 1, {'igemm_variant': 3}), (1, 6, 6, 40, 192, 3, 2, 1, {'igemm_variant': 3})] \
        if mode == 'emul' else \
        [(8, 14, 14, 256, 256, 3, 1, 1, {'igemm_256sq': 1}), (6, 14, 14, 1024, 256, 1, 1, 0, {'igemm_256sq': 1}),
         (4, 28, 28, 256, 256, 3, 2, 1, {'igemm_256sq': 1}), (5, 7, 7, 512, 512, 3, 1, 1, {'igemm_variant': 3}),
         (4, 7, 7, 2048, 512, 1, 1, 0, {'igemm_variant': 3}), (3, 14, 15, 72, 384, 1, 1, 0, {'igemm_256sq': 1})]
    for (N, H, W, C, K, R, st, pad, knobs) in cfgs:
        g = torch.Generator().manual_seed(K + H)
        xh = _nhwc(torch.randn(N, C, H, W, generator=g), dtype, dev)
        wk = (torch.randn(K, R, R, C, generator=g) * (2.0 / (C * R * R)) ** 0.5).to(dtype).to(dev)
        dy = _nhwc(torch.randn(N, C, H, W, generator=torch.Generator().manual_seed(5)), dtype, dev)
        wt = (torch.randn(C, R, R, K, generator=torch.Generator().manual_seed(6)) * 0.05).to(dtype).to(dev)
        Hin, Win = (H - 1) * st + R - 2 * pad, (W - 1) * st + R - 2 * pad
        res = {}
        try:
            for tag, kn in (('staged', {'igemm_variant': 1, 'igemm_256sq': 0, 'igemm_8w': 0}), ('ilv', knobs)):
                for k, v in kn.items():
                    L.cn_set_option(k.encode(), v)
                ys = ops.conv2d_fwd(xh, wk, None, K, R, R, (st, st), (pad, pad), bn_stats=True)
                name = L.cn_last_kernel_name().decode()
                assert name.endswith(', true>') == (tag == 'ilv'), name
                ps = ops.take_pending_stats(ys)
                dx = ops.conv2d_dgrad(dy, wt.permute(3, 1, 2, 0).contiguous(), (N, Hin, Win, K), C, R, R, (st, st),
                                      (pad, pad))
                res[tag] = (ys.cpu(), ps.partial.double().sum(0).cpu(), dx.cpu())
                for k in kn:
                    L.cn_set_option(k.encode(), {'igemm_256sq': -1, 'igemm_8w': 16}.get(k, 0))
        finally:
            for k in ('igemm_variant', 'igemm_256sq', 'igemm_8w'):
                L.cn_set_option(k.encode(), {'igemm_256sq': -1, 'igemm_8w': 16}.get(k, 0))
        assert torch.equal(res['staged'][0], res['ilv'][0]), (C, K, R)
        assert torch.equal(res['staged'][2], res['ilv'][2]), (C, K, R)
        assert rel_l2(res['ilv'][1], res['staged'][1]) < 1e-6, (C, K, R)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_wgrad_3x3_band_kernel(mode, dtype):
    """wgrad3x3_kernel (round 3): 3x3 / stride 1 / pad 1 weight gradients with the activation staged once per band of
    image rows as a zero-padded 2-D LDS image (a tap = a uniform shift; all nine taps per tile).  Against the CPU
    weight gradient and the tile kernels (knob wgrad_3x3=0) on: bands that cross image boundaries (H smaller than the
    band), H != W, a last band / last split that is ragged, 64-channel outputs (two k-parity waves summed through
    LDS) and 128-multiple outputs (four co waves), 32 / 64 / 96 input channels, accumulation (beta) and scaling."""
    dev = _dev(mode)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    if mode == 'emul' and dtype == torch.float16:
        cfgs = [(2, 5, 6, 32, 64)]
    elif mode == 'emul':
        cfgs = [(3, 5, 6, 32, 64), (2, 9, 7, 64, 128), (5, 3, 4, 96, 64), (1, 13, 11, 32, 128), (7, 2, 3, 32, 64)]
    else:
        cfgs = [(256, 56, 56, 64, 64), (64, 28, 28, 128, 128), (37, 14, 14, 256, 256), (61, 7, 7, 512, 512),
                (5, 28, 20, 96, 192), (3, 9, 31, 32, 64), (16, 56, 56, 64, 128)]
    for (N, H, W, C, K) in cfgs:
        g = torch.Generator().manual_seed(K + H + C)
        x = torch.randn(N, C, H, W, generator=g)
        dy = torch.randn(N, K, H, W, generator=g)
        xh, dyh = _nhwc(x, dtype, dev), _nhwc(dy, dtype, dev)
        ref = None
        if N * H * W <= 200000:
            ref = torch.nn.grad.conv2d_weight(xh.float().cpu().permute(0, 3, 1, 2), (K, C, 3, 3),
                                              dyh.float().cpu().permute(0, 3, 1, 2), 1, 1)
        outs = {}
        try:
            for band in (1, 0):
                L.cn_set_option(b'wgrad_3x3', band)
                dw = torch.zeros(K, 3, 3, C, device=dev)
                ops.conv2d_wgrad(xh, dyh, dw, C, K, 3, 3, (1, 1), (1, 1), beta=0.0)
                name = L.cn_last_kernel_name().decode()
                assert ('wgrad3x3_kernel' in name) == bool(band), name
                if band:
                    assert ('128>' in name) == (K % 128 == 0), name
                outs[band] = dw.cpu().permute(0, 3, 1, 2)
        finally:
            L.cn_set_option(b'wgrad_3x3', 1)
        # fp32 accumulation of bf16 products in a different (fixed) order: the two kernels agree to fp32 rounding
        assert rel_l2(outs[1], outs[0]) < 2e-5, (N, H, W, C, K, rel_l2(outs[1], outs[0]))
        if ref is not None:
            assert rel_l2(outs[1], ref) < 2e-5, (N, H, W, C, K, rel_l2(outs[1], ref))
        # accumulate + scale: dw = beta*dw + scale*wgrad
        dw2 = torch.full((K, 3, 3, C), 0.5, device=dev)
        ops.conv2d_wgrad(xh, dyh, dw2, C, K, 3, 3, (1, 1), (1, 1), beta=1.0, scale=0.25)
        assert rel_l2(dw2.cpu().permute(0, 3, 1, 2), 0.5 + 0.25 * outs[1]) < 2e-5


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float32])
def test_lazy_dy_dgrad_and_wgrad_are_bit_identical_to_apply_then_conv(mode, dtype):
    """Round 3 ("lazy dy"): cn_bn_bwd_partials(dy = NULL) + cn_conv2d_dgrad_lazy / cn_conv2d_wgrad_lazy form
    dy = c1*g + c2*y + c3 on the operand loads.  Same operation order and rounding as the apply kernel: the data
    gradient carries the same bits as apply -> dgrad, the weight gradient the same bits as apply -> the register-staged
    weight-gradient kernel (and agrees with the LDS-DMA kernel to fp32 rounding)."""
    dev = _dev(mode)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    cfgs = [(3, 6, 5, 16, 64, 1), (2, 8, 8, 32, 136, 2)] if mode == 'emul' else \
        [(32, 56, 56, 64, 256, 1), (16, 56, 56, 256, 512, 2), (8, 28, 28, 128, 512, 1)]
    for (N, H, W, Cin, K, st) in cfgs:
        g_ = torch.Generator().manual_seed(K + H)
        P, Q = (H - 1) // st + 1, (W - 1) // st + 1
        x = _nhwc(torch.randn(N, Cin, H, W, generator=g_), dtype, dev)
        gz = _nhwc(torch.randn(N, K, P, Q, generator=g_), dtype, dev)           # masked gradient w.r.t. the BN output
        y = _nhwc(torch.randn(N, K, P, Q, generator=g_) * 1.5 + 0.3, dtype, dev)  # BN input
        w = (torch.randn(K, 1, 1, Cin, generator=g_) * 0.1).to(dtype).to(dev)
        wt = w.permute(3, 1, 2, 0).contiguous()
        M = N * P * Q
        gamma = (torch.rand(K, generator=g_) + 0.5).to(dev)
        mean, var = y.float().view(M, K).mean(0), y.float().view(M, K).var(0, unbiased=False)
        invstd = (var + 1e-5).rsqrt()
        stats = torch.cat([mean, invstd, gamma * invstd, -mean * gamma * invstd]).contiguous()
        # partial rows as a dgrad epilogue would emit them: one row with the full sums
        gf = gz.float().view(M, K)
        xhat = (y.float().view(M, K) - mean) * invstd
        partial = torch.cat([gf.sum(0), (gf * xhat).sum(0)]).view(1, 2 * K).contiguous()
        code = ca._lib.dtype_code(dtype)
        ws = torch.empty(L.cn_bn_workspace(M, K, code) // 4 + 16, dtype=torch.float32, device=dev)
        dgam, dbet = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
        ptr, chk, so = ca._lib.ptr, ca._lib.check, ca._lib.stream_of
        outs = {}
        for lazy in (0, 1):
            coef = torch.empty(3 * K, dtype=torch.float32, device=dev)
            dy = torch.empty_like(y)
            chk(L.cn_bn_bwd_partials(ptr(gz), ptr(y), ptr(gamma), ptr(stats), None if lazy else ptr(dy), ptr(dgam), ptr(dbet),
                                     0.0, 1.0, ptr(coef), M, K, code, ptr(partial), 1, ptr(ws), ws.numel() * 4, so(y)),
                'cn_bn_bwd_partials')
            dw = torch.zeros(K, 1, 1, Cin, device=dev)
            if lazy:
                dx = ops.conv2d_dgrad_lazy(gz, y, coef, wt, x.shape, K, 1, 1, (st, st), (0, 0))
                ops.conv2d_wgrad_lazy(x, gz, y, coef, dw, Cin, K, 1, 1, (st, st), (0, 0), beta=0.0)
                assert 'wgrad_kernel' in L.cn_last_kernel_name().decode()
            else:
                dx = ops.conv2d_dgrad(dy, wt, x.shape, K, 1, 1, (st, st), (0, 0))
                L.cn_set_option(b'wgrad_variant', 1)          # the register-staged kernel, as the lazy form uses
                try:
                    ops.conv2d_wgrad(x, dy, dw, Cin, K, 1, 1, (st, st), (0, 0), beta=0.0)
                finally:
                    L.cn_set_option(b'wgrad_variant', 0)
                dw_dma = torch.zeros_like(dw)
                ops.conv2d_wgrad(x, dy, dw_dma, Cin, K, 1, 1, (st, st), (0, 0), beta=0.0)
                outs['dma'] = dw_dma.cpu()
            outs[lazy] = (dx.cpu(), dw.cpu(), coef.cpu())
        assert torch.equal(outs[0][2], outs[1][2])
        assert torch.equal(outs[0][0], outs[1][0]), ('dgrad', N, H, Cin, K, st)
        assert torch.equal(outs[0][1], outs[1][1]), ('wgrad', N, H, Cin, K, st)
        assert rel_l2(outs[1][1], outs['dma']) < 2e-5


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
def test_bn_apply_dual_equals_two_apply_passes(mode, dtype):
    """cn_bn_apply_dual (junction behind a projection shortcut): z and the ReLU bits equal, bit for bit, the shortcut
    BatchNorm's own apply followed by the junction's apply with that tensor as the residual."""
    _f16_emul_subset(mode, dtype, keep=True)
    dev = _dev(mode)
    import convnet_amd as ca
    from convnet_amd._lib import ptr, dtype_code, stream_of, check
    L = ca._lib.load()
    N, H, W, C = (2, 5, 7, 40) if mode == 'emul' else (8, 28, 28, 512)
    g = torch.Generator().manual_seed(3)
    y = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.3).to(dtype).to(dev)
    yd = (torch.randn(N, H, W, C, generator=g) * 0.7 - 0.2).to(dtype).to(dev)
    M = N * H * W
    code = dtype_code(dtype)

    def finalize(t, seed):
        gg = torch.Generator().manual_seed(seed)
        gamma = (torch.rand(C, generator=gg) + 0.5).to(dev)
        beta = (torch.randn(C, generator=gg) * 0.1).to(dev)
        stats = torch.empty(4 * C, dtype=torch.float32, device=dev)
        ws = ca.ops.workspace(L.cn_bn_workspace(M, C, code), dev)
        check(L.cn_bn_fwd_train(ptr(t), None, None, None, ptr(gamma), ptr(beta), None, None, None, 0.1, 1e-5, ptr(stats),
                                M, C, 0, code, ptr(ws), ws.numel() * 4, stream_of(t)), 'cn_bn_fwd_train')
        return gamma, beta, stats

    g3, b3, st3 = finalize(y, 11)
    gd, bd, std = finalize(yd, 12)
    # reference: two apply passes (inference-form apply with the same scale / shift is not exposed; the training call
    # with z recomputes identical statistics, so use it)
    ws = ca.ops.workspace(L.cn_bn_workspace(M, C, code), dev)
    res = torch.empty_like(yd)
    st_tmp = torch.empty(4 * C, dtype=torch.float32, device=dev)
    check(L.cn_bn_fwd_train(ptr(yd), None, ptr(res), None, ptr(gd), ptr(bd), None, None, None, 0.1, 1e-5, ptr(st_tmp),
                            M, C, 0, code, ptr(ws), ws.numel() * 4, stream_of(yd)), 'cn_bn_fwd_train')
    assert torch.equal(st_tmp, std)
    z_ref = torch.empty_like(y)
    CH = 16 // y.element_size()
    m_ref = torch.zeros(M * (C // CH), dtype=torch.uint8, device=dev)
    check(L.cn_bn_fwd_train(ptr(y), ptr(res), ptr(z_ref), ptr(m_ref), ptr(g3), ptr(b3), None, None, None, 0.1, 1e-5,
                            ptr(st_tmp), M, C, 1, code, ptr(ws), ws.numel() * 4, stream_of(y)), 'cn_bn_fwd_train')
    assert torch.equal(st_tmp, st3)
    z = torch.empty_like(y)
    m = torch.zeros_like(m_ref)
    check(L.cn_bn_apply_dual(ptr(y), ptr(yd), ptr(z), ptr(m), ptr(st3), ptr(std), M, C, 1, code, stream_of(y)),
          'cn_bn_apply_dual')
    assert torch.equal(z.cpu().view(torch.uint8 if False else z.dtype), z_ref.cpu())
    assert torch.equal(m.cpu(), m_ref.cpu())
    assert float(z.float().abs().sum()) > 0


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('dual', [False, True])
def test_lazy_z_conv_equals_apply_then_conv(mode, dtype, dual):
    """cn_conv2d_fwd_lazyz: the junction output z (and its ReLU bits) written by the convolution, the convolution's
    output and its statistics partials equal, bit for bit, the junction's apply pass (plain or with the projection
    shortcut's BatchNorm folded in) followed by cn_conv2d_fwd_bnstats."""
    _f16_emul_subset(mode, dtype, keep=True)
    dev = _dev(mode)
    import convnet_amd as ca
    from convnet_amd._lib import ptr, dtype_code, stream_of, check
    L = ca._lib.load()
    ops = ca.ops
    cases = [(2, 5, 7, 40, 24), (1, 6, 6, 72, 128)] if mode == 'emul' else [(4, 56, 56, 256, 64), (4, 28, 28, 512, 128)]
    code = dtype_code(dtype)
    for (N, H, W, C, K) in cases:
        g = torch.Generator().manual_seed(5 + C)
        y = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.3).to(dtype).to(dev)
        r = (torch.randn(N, H, W, C, generator=g) * 0.7 - 0.2).to(dtype).to(dev)
        w = (torch.randn(K, 1, 1, C, generator=g) * (2.0 / C) ** 0.5).to(dtype).to(dev)
        M = N * H * W

        def finalize(t, seed):
            gg = torch.Generator().manual_seed(seed)
            gamma = (torch.rand(C, generator=gg) + 0.5).to(dev)
            beta = (torch.randn(C, generator=gg) * 0.1).to(dev)
            stats = torch.empty(4 * C, dtype=torch.float32, device=dev)
            ws = ops.workspace(L.cn_bn_workspace(M, C, code), dev)
            check(L.cn_bn_fwd_train(ptr(t), None, None, None, ptr(gamma), ptr(beta), None, None, None, 0.1, 1e-5,
                                    ptr(stats), M, C, 0, code, ptr(ws), ws.numel() * 4, stream_of(t)), 'cn_bn_fwd_train')
            return gamma, beta, stats

        g3, b3, st3 = finalize(y, 21)
        gd, bd, std = finalize(r, 22)
        CH = 16 // y.element_size()
        ws = ops.workspace(L.cn_bn_workspace(M, C, code), dev)
        tmp = torch.empty(4 * C, dtype=torch.float32, device=dev)
        z_ref = torch.empty_like(y)
        m_ref = torch.zeros(M * (C // CH), dtype=torch.uint8, device=dev)
        if dual:
            check(L.cn_bn_apply_dual(ptr(y), ptr(r), ptr(z_ref), ptr(m_ref), ptr(st3), ptr(std), M, C, 1, code,
                                     stream_of(y)), 'cn_bn_apply_dual')
        else:
            check(L.cn_bn_fwd_train(ptr(y), ptr(r), ptr(z_ref), ptr(m_ref), ptr(g3), ptr(b3), None, None, None, 0.1,
                                    1e-5, ptr(tmp), M, C, 1, code, ptr(ws), ws.numel() * 4, stream_of(y)),
                  'cn_bn_fwd_train')
        o_ref = ops.conv2d_fwd(z_ref, w, None, K, 1, 1, (1, 1), (0, 0), bn_stats=True)
        ps_ref = ops.take_pending_stats(o_ref)
        z = torch.zeros_like(y)
        m = torch.zeros_like(m_ref)
        o = ops.conv2d_fwd_lazyz((y, r, st3, std if dual else None, z, m, True), w, K, bn_stats=True)
        ps = ops.take_pending_stats(o)
        assert 'lazy z' in L.cn_last_kernel_name().decode() or ', 3, false>' in L.cn_last_kernel_name().decode()
        assert torch.equal(z.cpu(), z_ref.cpu()), (N, H, W, C, K)
        assert torch.equal(m.cpu(), m_ref.cpu())
        assert torch.equal(o.cpu(), o_ref.cpu())
        assert ps.rows == ps_ref.rows and torch.equal(ps.partial.cpu(), ps_ref.partial.cpu())
        assert float(o.float().abs().sum()) > 0


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_junction_pair_equals_lazy_dgrad_and_wgrad(mode, dtype):
    """cn_conv2d_bwd1x1_lazy (data + weight gradient of a 64 -> 256 channel 1x1 convolution in one pass over g and y): dx
    has the bits of cn_conv2d_dgrad_lazy, dW equals cn_conv2d_wgrad_lazy up to fp32 summation order (other pixel
    ranges), and both agree with the CPU products of the dy the apply kernel would have stored."""
    _f16_emul_subset(mode, dtype, keep=True)
    dev = _dev(mode)
    import convnet_amd as ca
    ops = ca.ops
    L = ca._lib.load()
    C, K = 64, 256
    shapes = [(1, 10, 13), (2, 16, 16)] if mode == 'emul' else [(3, 56, 56), (256, 56, 56), (5, 17, 9)]
    for (N, H, W) in shapes:
        g_ = torch.Generator().manual_seed(N * H + W)
        gq = (torch.randn(N, H, W, K, generator=g_) * 0.5).to(dtype).to(dev)
        yq = (torch.randn(N, H, W, K, generator=g_) * 1.2 + 0.1).to(dtype).to(dev)
        x = torch.randn(N, H, W, C, generator=g_).to(dtype).to(dev)
        coef = torch.cat([torch.rand(K, generator=g_) + 0.5, torch.randn(K, generator=g_) * 0.05,
                          torch.randn(K, generator=g_) * 0.01]).to(dev)
        w = (torch.randn(K, 1, 1, C, generator=g_) * (2.0 / C) ** 0.5).to(dtype).to(dev)
        wc = w.permute(3, 1, 2, 0).contiguous()
        L.cn_set_option(b'jbwd_splits', 3 if mode == 'emul' else 256)
        try:
            dw = torch.zeros(K, 1, 1, C, dtype=torch.float32, device=dev)
            dx = ops.conv2d_bwd1x1_lazy(x, gq, yq, coef, wc, dw, K, beta=0.0)
        finally:
            L.cn_set_option(b'jbwd_splits', 256)
        assert 'jbwd_kernel' in L.cn_last_kernel_name().decode() or 'reduce' in L.cn_last_kernel_name().decode()
        dx_ref = ops.conv2d_dgrad_lazy(gq, yq, coef, wc, x.shape, K, 1, 1, (1, 1), (0, 0))
        dw_ref = torch.zeros_like(dw)
        ops.conv2d_wgrad_lazy(x, gq, yq, coef, dw_ref, C, K, 1, 1, (1, 1), (0, 0), beta=0.0)
        if dtype == torch.bfloat16 or mode == 'emul':
            assert torch.equal(dx.cpu(), dx_ref.cpu()), (N, H, W)
        else:   # fp16 on the GPU: hipcc may fold the dy rounding into a mixed-precision FMA in one kernel and not in the
            # other (single vs double rounding of a few elements in 10^4): equal to a few fp16 ulps, not bit for bit
            assert rel_l2(dx.float().cpu(), dx_ref.float().cpu()) < 2e-4, (N, H, W)
        assert rel_l2(dw.cpu(), dw_ref.cpu()) < (2e-6 if dtype == torch.bfloat16 or mode == 'emul' else 2e-4), (N, H, W, rel_l2(dw.cpu(), dw_ref.cpu()))
        if N * H * W <= 12000:     # CPU products of the rounded dy
            dy = (coef[:K].cpu() * gq.float().cpu() + (coef[K:2 * K].cpu() * yq.float().cpu() + coef[2 * K:].cpu())).to(dtype).float()
            dw_cpu = dy.reshape(-1, K).t() @ x.float().cpu().reshape(-1, C)
            dx_cpu = dy.reshape(-1, K) @ w.float().cpu().reshape(K, C)
            assert rel_l2(dw.cpu().reshape(K, C), dw_cpu) < 1e-3
            assert rel_l2(dx.float().cpu().reshape(-1, C), dx_cpu) < _tol(dtype)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_streaming_junction_dgrad_equals_tiled_epilogue_kernel(mode, dtype):
    """cn_conv2d_dgrad_junction (csrc/junction.hip) against cn_conv2d_dgrad_bnbwd_sa on every instantiated shape, with a
    dense and with a subsampled addend, pixel counts that are not whole stages, few and many workgroups: g bit for bit,
    the partial rows' column sums to fp32 association, and both against the definition on the CPU."""
    _f16_emul_subset(mode, dtype, keep=True)
    dev = _dev(mode)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    if mode == 'emul':
        cases = [(1, 6, 10, 256, 64, 1, 3), (2, 6, 6, 256, 128, 2, 2), (1, 5, 9, 512, 128, 1, 5), (1, 4, 6, 512, 256, 2, 2),
                 (1, 3, 5, 1024, 256, 1, 3)]
    else:
        # (the last three: the bench's own sizes, N = 256)
        cases = [(8, 56, 56, 256, 64, 1, 256), (8, 56, 56, 256, 128, 2, 256), (16, 28, 28, 512, 128, 1, 256),
                 (3, 17, 13, 256, 64, 2, 7), (2, 9, 11, 512, 128, 1, 256), (16, 28, 28, 512, 256, 2, 256),
                 (32, 14, 14, 1024, 256, 1, 256), (256, 56, 56, 256, 64, 1, 256), (256, 28, 28, 512, 128, 1, 256),
                 (256, 14, 14, 1024, 256, 1, 256)]
    for (N, H, W, C, K, sub, splits) in cases:
        g_ = torch.Generator().manual_seed(C + K + H)
        M = N * H * W
        dyh = (torch.randn(N, H, W, K, generator=g_)).to(dtype).to(dev)
        wc = (torch.randn(C, 1, 1, K, generator=g_) * (2.0 / K) ** 0.5).to(dtype).to(dev)
        bn_y = (torch.randn(N, H, W, C, generator=g_) * 1.5 + 0.3).to(dtype).to(dev)
        if sub == 2:
            addend = torch.randn(N, (H + 1) // 2, (W + 1) // 2, C, generator=g_).to(dtype).to(dev)
        else:
            addend = torch.randn(N, H, W, C, generator=g_).to(dtype).to(dev)
        yf = bn_y.float().reshape(M, C)
        mean, var = yf.mean(0), yf.var(0, unbiased=False)
        invstd = 1.0 / torch.sqrt(var + 1e-5)
        gamma = (torch.rand(C, generator=g_) + 0.5).to(dev)
        beta = (torch.randn(C, generator=g_) * 0.2).to(dev)
        stats = torch.cat([mean, invstd, gamma * invstd, beta - mean * gamma * invstd]).contiguous()
        on = torch.rand(M, C, generator=g_).to(dev) > 0.4
        w8 = (2 ** torch.arange(8, device=dev)).view(1, 1, 8)
        bits = (on.view(M, C // 8, 8).long() * w8).sum(-1).to(torch.uint8).contiguous()
        saved = ops.JDGRAD
        try:
            ops.JDGRAD = False
            g0, p0, r0 = ops.conv2d_dgrad(dyh, wc, (N, H, W, C), K, 1, 1, (1, 1), (0, 0), addend=addend,
                                          bn=(bn_y, bits, stats, True), addend_sub=sub)
            assert 'igemm_kernel' in L.cn_last_kernel_name().decode()
            ops.JDGRAD = True
            L.cn_set_option(b'jdgrad_splits', splits)
            g1, p1, r1 = ops.conv2d_dgrad(dyh, wc, (N, H, W, C), K, 1, 1, (1, 1), (0, 0), addend=addend,
                                          bn=(bn_y, bits, stats, True), addend_sub=sub)
            assert ('jdgrad_w32_kernel' if K >= 256 else 'jdgrad_kernel') in L.cn_last_kernel_name().decode()
            assert r1 == L.cn_conv2d_dgrad_junction_rows_k(N, H, W, C, K)
        finally:
            ops.JDGRAD = saved
            L.cn_set_option(b'jdgrad_splits', 256)
        assert tuple(p1.shape) == (r1, 2 * C) and r1 <= max(splits, 1)
        assert torch.equal(g1.cpu(), g0.cpu()), (N, H, W, C, K, sub)
        s0, s1 = p0.double().sum(0), p1.double().sum(0)
        assert rel_l2(s1[:C].cpu(), s0[:C].cpu()) < 1e-5 and rel_l2(s1[C:].cpu(), s0[C:].cpu()) < 5e-5
        gd = g1.float().reshape(M, C).double()
        xhat = ((yf - mean) * invstd).double()
        assert rel_l2(s1[:C].cpu(), gd.sum(0).cpu()) < 1e-5
        assert rel_l2(s1[C:].cpu(), (gd * xhat).sum(0).cpu()) < 5e-5
        assert float(gd.abs().sum()) > 0


@pytest.mark.parametrize('mode', MODES)
def test_stem_halo_kernel_equals_tiled_kernel_on_the_pair_image(mode):
    """cn_stem_fwd (csrc/stem.hip: input rows staged in LDS once, MFMA fragments read straight out of the halo) against
    cn_conv2d_fwd_bnstats on the same pixel-pair image: output bit for bit, the statistics partials' column sums to fp32
    association, and the output against F.conv2d.  Image heights that are not whole bands of 8 output rows included."""
    dev = _dev(mode)
    import convnet_amd as ca
    from convnet_amd._lib import ptr, dtype_code, stream_of, check
    ops, L = ca.ops, ca._lib.load()
    cases = [(2, 20, 24), (1, 38, 18)] if mode == 'emul' else [(4, 224, 224), (3, 70, 50), (256, 224, 224)]
    K, R, S, C, pad = 64, 7, 7, 3, 3
    S2 = 4
    for (N, H, W) in cases:
        g = torch.Generator().manual_seed(H + W)
        x = _q(torch.randn(N, C, H, W, generator=g), torch.bfloat16).to(dev)
        w = _q(torch.randn(K, R, S, C, generator=g) * (2.0 / (C * R * S)) ** 0.5, torch.bfloat16).to(dev)
        xp = ops.nchw_to_pairs(x, (pad, pad))
        Hp, Jp = xp.shape[1], xp.shape[2]
        wp = torch.empty(K * R * S2 * 8, dtype=torch.bfloat16, device=dev)
        check(L.cn_weight_prep_pairs(ptr(w.float().contiguous()), ptr(wp), K, R, S, C, stream_of(xp)), 'cn_weight_prep_pairs')
        y0 = ops.conv2d_fwd(xp, wp, None, K, R, S2, (2, 1), (0, 0), bn_stats=True)
        ps0 = ops.take_pending_stats(y0)
        assert L.cn_stem_fwd_ok(K, R, S2, Jp, dtype_code(torch.bfloat16))
        P, Q = (Hp - R) // 2 + 1, Jp - S2 + 1
        assert tuple(y0.shape) == (N, P, Q, K)
        y1 = torch.zeros_like(y0)
        rows = L.cn_stem_fwd_rows(N, P)
        part = torch.empty((rows, 2 * K), dtype=torch.float32, device=dev)
        check(L.cn_stem_fwd(ptr(xp), ptr(wp), ptr(y1), N, Hp, Jp, dtype_code(torch.bfloat16), ptr(part), rows,
                            stream_of(xp)), 'cn_stem_fwd')
        assert 'stem_fwd_kernel' in L.cn_last_kernel_name().decode()
        assert torch.equal(y1.cpu(), y0.cpu()), (N, H, W)
        s0, s1 = ps0.partial.double().sum(0), part.double().sum(0)
        assert rel_l2(s1.cpu(), s0.cpu()) < 1e-5
        if N * H * W < 300000:
            y_ref = F.conv2d(x.float().cpu(), w.float().cpu().permute(0, 3, 1, 2), stride=2, padding=pad)
            assert rel_l2(y1.float().cpu().permute(0, 3, 1, 2), y_ref) < 1e-2


@pytest.mark.parametrize('mode', MODES)
def test_stem_halo_weight_gradient_equals_tiled_kernel(mode):
    """cn_stem_wgrad (both MFMA operands as LDS transpose reads, the activation straight out of the band's halo) against
    cn_conv2d_wgrad on the same pixel-pair image and against the CPU weight gradient of the 7x7/2 convolution."""
    dev = _dev(mode)
    import convnet_amd as ca
    from convnet_amd._lib import ptr, dtype_code, stream_of, check
    ops, L = ca.ops, ca._lib.load()
    cases = [(2, 26, 32, 3), (1, 42, 64, 2)] if mode == 'emul' else [(4, 224, 224, 512), (3, 70, 96, 5), (64, 224, 224, 512)]
    K, R, S, C, pad, S2 = 64, 7, 7, 3, 3, 4
    code = dtype_code(torch.bfloat16)
    for (N, H, W, wgs) in cases:
        g = torch.Generator().manual_seed(H + W)
        x = _q(torch.randn(N, C, H, W, generator=g), torch.bfloat16)
        xp = ops.nchw_to_pairs(x.to(dev), (pad, pad))
        Hp, Jp = xp.shape[1], xp.shape[2]
        assert L.cn_stem_wgrad_ok(K, R, S2, Jp, code), Jp
        P, Q = (Hp - R) // 2 + 1, Jp - S2 + 1
        dy = _q(torch.randn(N, K, P, Q, generator=g), torch.bfloat16)
        dyh = _nhwc(dy, torch.bfloat16, dev)
        t0 = torch.zeros(K * R * S2 * 8, dtype=torch.float32, device=dev)
        ops.conv2d_wgrad(xp, dyh, t0, 8, K, R, S2, (2, 1), (0, 0), beta=0.0)
        t1 = torch.full_like(t0, 7.0)
        L.cn_set_option(b'stem_wgrad_wgs', wgs)
        try:
            ws = ops.workspace(L.cn_stem_wgrad_workspace(N, Hp), dev, 'main')
            check(L.cn_stem_wgrad(ptr(xp), ptr(dyh), ptr(t1), N, Hp, Jp, code, 0.0, 1.0, ptr(ws), ws.numel() * 4,
                                  stream_of(xp)), 'cn_stem_wgrad')
        finally:
            L.cn_set_option(b'stem_wgrad_wgs', 256)
        assert rel_l2(t1.cpu(), t0.cpu()) < 1e-5, (N, H, W, rel_l2(t1.cpu(), t0.cpu()))
        if N * H * W < 300000:
            xr = x.clone()
            wr = torch.zeros(K, C, R, S, requires_grad=True)
            F.conv2d(xr, wr, stride=2, padding=pad).backward(dy)
            dw = torch.zeros(K, R, S, C, device=dev)
            check(L.cn_wgrad_unpack_pairs(ptr(t1), ptr(dw), K, R, S, C, 0.0, stream_of(t1)), 'cn_wgrad_unpack_pairs')
            assert rel_l2(dw.cpu().permute(0, 3, 1, 2), wr.grad) < 2e-3


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_conv3x3_halo_kernel_equals_tiled_kernel(mode, dtype):
    """cn_conv3x3_c64 (csrc/conv3x3.hip: the band's input rows staged in LDS once, all nine taps' MFMA fragments read
    straight out of the halo) against the tiled implicit-GEMM kernel, forward (with the statistics partials) and data
    gradient: outputs bit for bit, column sums of the partials to fp32 association, and both against F.conv2d."""
    _f16_emul_subset(mode, dtype, keep=True)
    dev = _dev(mode)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    C = K = 64
    cases = [(1, 9, 7, 2), (2, 5, 12, 256)] if mode == 'emul' else [(4, 56, 56, 256), (3, 13, 21, 5), (256, 56, 56, 256)]
    for (N, H, W, wgs) in cases:
        g = torch.Generator().manual_seed(H * W)
        x = _q(torch.randn(N, C, H, W, generator=g), dtype)
        w = _q(torch.randn(K, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5, dtype)
        dy = _q(torch.randn(N, K, H, W, generator=g), dtype)
        xh, dyh = _nhwc(x, dtype, dev), _nhwc(dy, dtype, dev)
        wk = w.permute(0, 2, 3, 1).contiguous().to(dtype).to(dev)     # KRSC
        wc = w.permute(1, 2, 3, 0).contiguous().to(dtype).to(dev)     # CRSK
        saved = ops.CONV3X3_HALO
        try:
            ops.CONV3X3_HALO = False
            y0 = ops.conv2d_fwd(xh, wk, None, K, 3, 3, (1, 1), (1, 1), bn_stats=True)
            ps0 = ops.take_pending_stats(y0)
            d0 = ops.conv2d_dgrad(dyh, wc, (N, H, W, C), K, 3, 3, (1, 1), (1, 1))
            assert 'igemm_kernel' in L.cn_last_kernel_name().decode()
            ops.CONV3X3_HALO = True
            L.cn_set_option(b'conv3x3_wgs', wgs)
            y1 = ops.conv2d_fwd(xh, wk, None, K, 3, 3, (1, 1), (1, 1), bn_stats=True)
            assert 'conv3x3_c64_kernel' in L.cn_last_kernel_name().decode()
            ps1 = ops.take_pending_stats(y1)
            d1 = ops.conv2d_dgrad(dyh, wc, (N, H, W, C), K, 3, 3, (1, 1), (1, 1))
            assert 'conv3x3_c64_kernel' in L.cn_last_kernel_name().decode()
        finally:
            ops.CONV3X3_HALO = saved
            L.cn_set_option(b"conv3x3_wgs", 512)
        assert torch.equal(y1.cpu(), y0.cpu()), (N, H, W)
        assert torch.equal(d1.cpu(), d0.cpu()), (N, H, W)
        assert rel_l2(ps1.partial.double().sum(0).cpu(), ps0.partial.double().sum(0).cpu()) < 1e-5
        if N * H * W < 20000:
            xr = x.clone().requires_grad_(True)
            yr = F.conv2d(xr, w, padding=1)
            yr.backward(dy)
            assert rel_l2(y1.float().cpu().permute(0, 3, 1, 2), yr.detach()) < _tol(dtype)
            assert rel_l2(d1.float().cpu().permute(0, 3, 1, 2), xr.grad) < _tol(dtype)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_streaming_conv1x1_forward_equals_tiled_kernel(mode, dtype):
    """cn_conv1x1_stream_fwd (conv3 / the stride-1 projection as a persistent streaming kernel, statistics in registers)
    against the tiled kernel on every instantiated shape: output bit for bit, partial column sums to fp32 association."""
    _f16_emul_subset(mode, dtype, keep=True)
    dev = _dev(mode)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    cases = [(1, 6, 10, 64, 256), (2, 6, 6, 128, 256), (1, 5, 9, 128, 512), (1, 5, 7, 256, 1024)] if mode == 'emul' else \
        [(8, 56, 56, 64, 256), (8, 56, 56, 128, 256), (16, 28, 28, 128, 512), (3, 17, 13, 64, 256), (64, 14, 14, 256, 1024)]
    for (N, H, W, C, K) in cases:
        g = torch.Generator().manual_seed(C + K + H)
        x = torch.randn(N, H, W, C, generator=g).to(dtype).to(dev)
        w = (torch.randn(K, 1, 1, C, generator=g) * (2.0 / C) ** 0.5).to(dtype).to(dev)
        saved = ops.CONV1X1_STREAM
        try:
            ops.CONV1X1_STREAM = False
            y0 = ops.conv2d_fwd(x, w, None, K, 1, 1, (1, 1), (0, 0), bn_stats=True)
            assert 'igemm_kernel' in L.cn_last_kernel_name().decode()
            p0 = ops.take_pending_stats(y0)
            ops.CONV1X1_STREAM = True
            y1 = ops.conv2d_fwd(x, w, None, K, 1, 1, (1, 1), (0, 0), bn_stats=True)
            assert 'jfwd_kernel' in L.cn_last_kernel_name().decode()
            p1 = ops.take_pending_stats(y1)
            y2 = ops.conv2d_fwd(x, w, None, K, 1, 1, (1, 1), (0, 0))     # without the statistics
        finally:
            ops.CONV1X1_STREAM = saved
        assert torch.equal(y1.cpu(), y0.cpu()) and torch.equal(y2.cpu(), y0.cpu()), (N, H, W, C, K)
        assert rel_l2(p1.partial.double().sum(0).cpu(), p0.partial.double().sum(0).cpu()) < 1e-5
        # ... and against the definition on the CPU (fp32 product of the same 16-bit operands; sums of the stored values)
        y_cpu = x.float().cpu().reshape(-1, C) @ w.float().cpu().reshape(K, C).t()
        assert rel_l2(y1.float().cpu().reshape(-1, K), y_cpu) < _tol(dtype), (N, H, W, C, K)
        yd = y1.double().cpu().reshape(-1, K)
        ps = p1.partial.double().sum(0).cpu()
        assert rel_l2(ps[:K], yd.sum(0)) < 1e-5 and rel_l2(ps[K:], (yd * yd).sum(0)) < 1e-5, (N, H, W, C, K)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_lazy_a_conv_equals_apply_then_conv(mode, dtype):
    """cn_conv1x1_stream_fwd_lazya (an inner BatchNorm's apply + ReLU formed on the streaming 1x1 kernel's operand path,
    a written as a side output) against cn_bn_fwd_train (statistics + apply) followed by cn_conv1x1_stream_fwd: a, y and the statistics partials
    bit for bit, on every instantiated shape, pixel counts that are not whole stages included."""
    _f16_emul_subset(mode, dtype, keep=True)
    dev = _dev(mode)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    cases = [(1, 6, 10, 64, 256), (2, 6, 6, 128, 256), (1, 5, 9, 128, 512), (1, 5, 7, 256, 1024)] if mode == 'emul' else \
        [(8, 56, 56, 64, 256), (8, 56, 56, 128, 256), (16, 28, 28, 128, 512), (3, 17, 13, 64, 256), (64, 14, 14, 256, 1024)]
    for relu in (True, False):
        for (N, H, W, C, K) in cases:
            g = torch.Generator().manual_seed(C + K + H + int(relu))
            y2 = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.3).to(dtype).to(dev)
            w = (torch.randn(K, 1, 1, C, generator=g) * (2.0 / C) ** 0.5).to(dtype).to(dev)
            gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
            beta = (torch.randn(C, generator=g) * 0.3).to(dev)
            stats = torch.empty(4 * C, dtype=torch.float32, device=dev)
            M = N * H * W
            a0 = torch.empty_like(y2)
            code = ca._lib.dtype_code(dtype)
            ws = ops.workspace(L.cn_bn_workspace(M, C, code), dev)
            ca._lib.check(L.cn_bn_fwd_train(ca._lib.ptr(y2), None, ca._lib.ptr(a0), None, ca._lib.ptr(gamma),
                                            ca._lib.ptr(beta), None, None, None, 0.1, 1e-5, ca._lib.ptr(stats), M, C,
                                            int(relu), code, ca._lib.ptr(ws), ws.numel() * 4, ca._lib.stream_of(y2)),
                          'cn_bn_fwd_train')
            out0 = ops.conv2d_fwd(a0, w, None, K, 1, 1, (1, 1), (0, 0), bn_stats=True)
            assert 'jfwd_kernel' in L.cn_last_kernel_name().decode()
            p0 = ops.take_pending_stats(out0)
            a1 = torch.full_like(y2, float('nan'))
            out1 = ops.conv2d_fwd_lazya((y2, stats, a1, relu), w, K, bn_stats=True)
            assert L.cn_last_kernel_name().decode().endswith(', true>')
            p1 = ops.take_pending_stats(out1)
            assert torch.equal(a1.cpu().view(torch.int16), a0.cpu().view(torch.int16)), (N, H, W, C, K, relu)
            assert torch.equal(out1.cpu(), out0.cpu()), (N, H, W, C, K, relu)
            if p0 is not None or p1 is not None:
                assert p0.rows == p1.rows and torch.equal(p1.partial.cpu(), p0.partial.cpu())
            # ... and against the definition on the CPU: batch statistics of y2, a = relu?(bn(y2)), out = conv1x1(a)
            yc = y2.float().cpu().reshape(-1, C)
            mean, var = yc.double().mean(0), yc.double().var(0, unbiased=False)
            a_cpu = ((yc.double() - mean) / (var + 1e-5).sqrt() * gamma.double().cpu() + beta.double().cpu()).float()
            a_cpu = a_cpu.clamp_min(0) if relu else a_cpu
            assert rel_l2(a1.float().cpu().reshape(-1, C), a_cpu) < _tol(dtype), (N, H, W, C, K, relu)
            out_cpu = a1.float().cpu().reshape(-1, C) @ w.float().cpu().reshape(K, C).t()
            assert rel_l2(out1.float().cpu().reshape(-1, K), out_cpu) < _tol(dtype), (N, H, W, C, K, relu)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_lazy_a_conv3x3_halo_equals_apply_then_conv(mode, dtype):
    """cn_conv3x3_c64_lazya (bn1 + ReLU formed on the way into the halo of the 64-channel 3x3 kernel; zero padding pads
    a, not the BatchNorm input) against cn_bn_fwd_train followed by cn_conv3x3_c64: a, y and the statistics partials bit
    for bit; heights that are not whole bands included."""
    _f16_emul_subset(mode, dtype, keep=True)
    dev = _dev(mode)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    C = K = 64
    for relu in (True, False):
        for (N, H, W) in ([(1, 6, 7), (2, 5, 4)] if mode == 'emul' else [(8, 56, 56), (3, 17, 13), (2, 9, 56)]):
            g = torch.Generator().manual_seed(N * H + W + int(relu))
            y1 = (torch.randn(N, H, W, C, generator=g) * 1.5 + 0.3).to(dtype).to(dev)
            w = (torch.randn(K, 3, 3, C, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dtype).to(dev)
            gamma = (torch.rand(C, generator=g) + 0.5).to(dev)
            beta = (torch.randn(C, generator=g) * 0.3 + 0.2).to(dev)   # relu(shift) != 0: a transformed pad would show
            stats = torch.empty(4 * C, dtype=torch.float32, device=dev)
            M = N * H * W
            a0 = torch.empty_like(y1)
            code = ca._lib.dtype_code(dtype)
            ws = ops.workspace(L.cn_bn_workspace(M, C, code), dev)
            ca._lib.check(L.cn_bn_fwd_train(ca._lib.ptr(y1), None, ca._lib.ptr(a0), None, ca._lib.ptr(gamma),
                                            ca._lib.ptr(beta), None, None, None, 0.1, 1e-5, ca._lib.ptr(stats), M, C,
                                            int(relu), code, ca._lib.ptr(ws), ws.numel() * 4, ca._lib.stream_of(y1)),
                          'cn_bn_fwd_train')
            out0 = ops.conv2d_fwd(a0, w, None, K, 3, 3, (1, 1), (1, 1), bn_stats=True)
            assert 'conv3x3_c64_kernel' in L.cn_last_kernel_name().decode()
            p0 = ops.take_pending_stats(out0)
            a1 = torch.full_like(y1, float('nan'))
            out1 = ops.conv2d_fwd_lazya((y1, stats, a1, relu), w, K, bn_stats=True, kernel=(3, 3))
            assert ', true>' in L.cn_last_kernel_name().decode()
            p1 = ops.take_pending_stats(out1)
            assert torch.equal(a1.cpu().view(torch.int16), a0.cpu().view(torch.int16)), (N, H, W, relu)
            assert torch.equal(out1.cpu(), out0.cpu()), (N, H, W, relu)
            if p0 is not None or p1 is not None:
                assert p0.rows == p1.rows and torch.equal(p1.partial.cpu(), p0.partial.cpu())
            # ... and against the definition on the CPU: a = relu?(bn(y1)) with batch statistics, out = conv3x3(a), pad 1
            yc = y1.float().cpu().reshape(-1, C)
            mean, var = yc.double().mean(0), yc.double().var(0, unbiased=False)
            a_cpu = ((yc.double() - mean) / (var + 1e-5).sqrt() * gamma.double().cpu() + beta.double().cpu()).float()
            a_cpu = a_cpu.clamp_min(0) if relu else a_cpu
            assert rel_l2(a1.float().cpu().reshape(-1, C), a_cpu) < _tol(dtype), (N, H, W, relu)
            out_cpu = F.conv2d(a1.float().cpu().permute(0, 3, 1, 2), w.float().cpu().permute(0, 3, 1, 2), padding=1)
            assert rel_l2(out1.float().cpu().permute(0, 3, 1, 2), out_cpu) < _tol(dtype), (N, H, W, relu)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_streaming_lazy_dgrad_equals_tiled_kernel(mode, dtype):
    """cn_conv2d_dgrad_lazy_stream (512 -> 128 / 256 channels, csrc/junction.hip: jdlazy_kernel) against the tiled lazy data
    gradient: bit for bit (bf16; fp16 to a few ulps on the GPU, see the junction-pair test), pixel counts that are not
    whole stages included."""
    _f16_emul_subset(mode, dtype, keep=True)
    dev = _dev(mode)
    import convnet_amd as ca
    ops, L = ca.ops, ca._lib.load()
    K = 512
    for (N, H, W, C) in ([(1, 5, 9, 128), (2, 4, 8, 128), (1, 6, 7, 256)] if mode == 'emul' else
                         [(16, 28, 28, 128), (3, 17, 13, 128), (256, 28, 28, 128), (3, 17, 13, 256), (64, 28, 28, 256)]):
        g_ = torch.Generator().manual_seed(N * H + W + C)
        gq = (torch.randn(N, H, W, K, generator=g_) * 0.5).to(dtype).to(dev)
        yq = (torch.randn(N, H, W, K, generator=g_) * 1.2 + 0.1).to(dtype).to(dev)
        coef = torch.cat([torch.rand(K, generator=g_) + 0.5, torch.randn(K, generator=g_) * 0.05,
                          torch.randn(K, generator=g_) * 0.01]).to(dev)
        wc = (torch.randn(C, 1, 1, K, generator=g_) * (2.0 / K) ** 0.5).to(dtype).to(dev)
        # the tiled kernel through its own entry point, the streaming kernel through the dispatching wrapper
        d0 = torch.empty((N, H, W, C), dtype=dtype, device=dev)
        P = ca._lib.ptr
        ca._lib.check(L.cn_conv2d_dgrad_lazy(P(gq), P(yq), P(coef), P(wc), P(d0), N, H, W, C, K, 1, 1, 1, 1, 0, 0,
                                             ca._lib.dtype_code(dtype), ca._lib.stream_of(gq)), 'cn_conv2d_dgrad_lazy')
        assert 'igemm_kernel' in L.cn_last_kernel_name().decode()
        d1 = ops.conv2d_dgrad_lazy(gq, yq, coef, wc, (N, H, W, C), K, 1, 1, (1, 1), (0, 0))
        assert 'jdlazy_kernel' in L.cn_last_kernel_name().decode()
        if dtype == torch.bfloat16 or mode == 'emul':
            assert torch.equal(d1.cpu(), d0.cpu()), (N, H, W, C)
        else:
            assert rel_l2(d1.float().cpu(), d0.float().cpu()) < 2e-4
        assert float(d1.float().abs().sum()) > 0
