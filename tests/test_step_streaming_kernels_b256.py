"""The round-3 streaming / halo kernels pinned AT BENCH SIZE against CPU formulas (`-m gpu`).

tests/test_ops.py proves these kernels bit-identical to the tiled kernels at N <= 32, and tests/test_headline_parity.py
holds the end-to-end bf16 step to loose bounds.  What was missing: the kernels the bench really times - the junction data
gradient (`jdgrad_kernel`, `jdgrad_w32_kernel`), the junction pair (`jbwd_kernel`), the lazy-dy streaming dgrad
(`jdlazy_kernel`), the streaming 1x1 forward with and without the folded BatchNorm apply (`jfwd_kernel`), lazy z
(`igemm_kernel<..., 3, ...>`), the 64-channel 3x3 halo kernel forward (lazy a) and data gradient (`conv3x3_c64_kernel`)
and the stem halo kernels (`stem_fwd_kernel`, `stem_wgrad_kernel`) - compared with a CPU value on the tensors of a
ResNet-50 bf16 b=256 step.

One warm-start training step (non-trivial BatchNorm state: tests/helpers.warm_bn_state) runs through the Trainer with
the DEFAULT dispatch (lazy dy / z / a on).  The Python wrappers of the C ABI calls (convnet.pytorch_amd/ops.py) are
wrapped so that the first call per (entry point, shape) keeps its operands and results, with the kernel name the library
reports (cn_kernel_log).  Afterwards every record is recomputed ON THE CPU in fp32 from the same 16-bit operands:

  * junction data gradient   g  = relubits * (dy W + shortcut gradient)            (models/resnet.py:141-165 backward)
        stored g  vs CPU on sampled pixel rows (first / last 4096 + 8192 random; 1x1 convolutions are per pixel) <= 5e-3
        partial rows: column sums vs sum(g), sum(g * xhat) of the STORED g over the whole tensor (fp64 on the GPU)  <= 1e-5
  * junction pair / lazy dgrad   dy = bf16(c1 g + c2 y + c3); dx = dy W (sampled rows <= 5e-3); dW = dy^T x over ALL
        pixels (fp32 on the CPU) <= 1e-4
  * lazy z     z = relu(y scale + shift + residual [or bn(residual)]) (sampled rows <= 5e-3, ReLU bits exact where
        |z| is not within rounding of 0), out = conv1x1(stored z) <= 5e-3
  * lazy a / streaming forward / halo forward   a = relu(y scale + shift) <= 5e-3; out = conv(stored a) on the CPU
        (1x1: sampled rows; 3x3: images 0, 1, 254, 255) <= 5e-3; statistics partials vs sums of the stored output <= 1e-5
  * stem   y = conv7x7/2(bf16 x) on images 0, 255 <= 5e-3; dW = conv2d_weight over the whole batch <= 1e-4

and the set of kernel names that served a checked call must contain every kernel listed above.
Tolerances: 5e-3 rel-L2 for tensors stored in bf16 (one rounding is 1.7e-3 rms - what every row measures,
profiles/r04_streaming_kernels_b256_vs_cpu.txt -, plus the fp32 summation order over <= 4608 products), 1e-5 for the
fused reductions against fp64 sums of the stored values (measured <= 4e-7), 1e-4 for weight gradients over up to
802 816 pixels (measured 1e-5)."""
import pytest
import torch
import torch.nn.functional as F

from helpers import load_warm, rel_l2, warm_bn_state

pytestmark = pytest.mark.gpu

WANT = ('jdgrad_kernel', 'jdgrad_w32_kernel', 'jbwd_kernel', 'jdlazy_kernel', 'jfwd_kernel', 'conv3x3_c64_kernel',
        'stem_fwd_kernel', 'stem_wgrad_kernel')


def _rows(M, seed):
    """Pixel rows a per-pixel (1x1) product is recomputed on: the ends (large offsets) + a random sample."""
    if M <= 20000:
        return torch.arange(M)
    g = torch.Generator().manual_seed(seed)
    mid = torch.randint(4096, M - 4096, (8192,), generator=g)
    return torch.cat([torch.arange(4096), mid, torch.arange(M - 4096, M)]).unique()


def _f(t):
    return t.detach().float().cpu()


def _bits(mask_rows, C):
    """bn_apply's ReLU bits (one byte per 8-channel chunk, bit e = channel e of the chunk) -> bool [rows, C]."""
    m = mask_rows.cpu().to(torch.int32).reshape(mask_rows.shape[0], C // 8, 1)
    return ((m >> torch.arange(8).view(1, 1, 8)) & 1).bool().reshape(mask_rows.shape[0], C)


def _colsum64(t2d):
    return t2d.double().sum(0).cpu()


def test_streaming_and_halo_kernels_on_the_tensors_of_a_bf16_b256_step(monkeypatch):
    import convnet_amd as ca
    ops = ca.ops
    L = ca._lib.load()
    dev = torch.device('cuda', 0)
    meta, _ = load_warm('r50_b256_warm')
    torch.manual_seed(123)
    model = ca.models.resnet(dataset='imagenet', depth=50)
    warm_bn_state(model, meta['warm_seed'], last_gamma=tuple(meta['warm_last_gamma']))
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device='cuda:0',
                    dtype=torch.bfloat16, grad_clip=1e9, print_freq=10 ** 9)
    tr._use_graph = False
    B = 256
    g0 = torch.Generator().manual_seed(meta['seed'])
    x0 = torch.randn(B, 3, 224, 224, generator=g0)
    t0 = torch.randint(0, 1000, (B,), generator=g0)

    rec = {}

    def logged(fn):
        L.cn_kernel_log(1)
        out = fn()
        return out, L.cn_kernel_log(0).decode()

    orig = {n: getattr(ops, n) for n in ('conv2d_dgrad', 'conv2d_bwd1x1_lazy', 'conv2d_dgrad_lazy', 'conv2d_fwd_lazyz',
                                         'conv2d_fwd_lazya', 'conv2d_fwd', '_park_stats')}
    parked = {}

    def park(y, partial, rows, pivot=None):
        parked[y.data_ptr()] = (partial, rows, pivot)
        return orig['_park_stats'](y, partial, rows, pivot)

    def w_dgrad(dy, w_crsk, x_shape, K, R, S, stride, pad, addend=None, bn=None, addend_sub=1):
        out, names = logged(lambda: orig['conv2d_dgrad'](dy, w_crsk, x_shape, K, R, S, stride, pad, addend=addend, bn=bn,
                                                          addend_sub=addend_sub))
        key = ('dgrad', tuple(x_shape), K, R, tuple(stride), bn is not None, addend_sub)
        if key not in rec and (bn is not None or 'conv3x3_c64' in names):
            rec[key] = dict(dy=dy, w=w_crsk, x_shape=tuple(x_shape), K=K, R=R, stride=tuple(stride), pad=tuple(pad),
                            addend=addend, bn=bn, sub=addend_sub, out=out, names=names)
        return out

    def w_pair(x, g, bn_y, coef, w_crsk, dw_krsc, K, beta=1.0, scale=1.0):
        before = dw_krsc.detach().clone()
        out, names = logged(lambda: orig['conv2d_bwd1x1_lazy'](x, g, bn_y, coef, w_crsk, dw_krsc, K, beta=beta, scale=scale))
        key = ('pair', tuple(x.shape), K)
        if key not in rec:
            rec[key] = dict(x=x, g=g, y=bn_y, coef=coef.clone(), w=w_crsk, dw=dw_krsc, dw_before=before, beta=beta,
                            scale=scale, K=K, out=out, names=names)
        return out

    def w_dlazy(g, bn_y, coef, w_crsk, x_shape, K, R, S, stride, pad):
        out, names = logged(lambda: orig['conv2d_dgrad_lazy'](g, bn_y, coef, w_crsk, x_shape, K, R, S, stride, pad))
        key = ('dlazy', tuple(x_shape), K, tuple(stride))
        if key not in rec:
            rec[key] = dict(g=g, y=bn_y, coef=coef.clone(), w=w_crsk, x_shape=tuple(x_shape), K=K, stride=tuple(stride),
                            out=out, names=names)
        return out

    def w_lazyz(lz, w_krsc, K, bn_stats=False, pivot=None):
        out, names = logged(lambda: orig['conv2d_fwd_lazyz'](lz, w_krsc, K, bn_stats=bn_stats, pivot=pivot))
        key = ('lazyz', tuple(lz[0].shape), K, lz[3] is not None)
        if key not in rec:
            rec[key] = dict(lz=lz, w=w_krsc, K=K, out=out, names=names, parked=parked.get(out.data_ptr()))
        return out

    def w_lazya(la, w_krsc, K, bn_stats=False, kernel=(1, 1)):
        out, names = logged(lambda: orig['conv2d_fwd_lazya'](la, w_krsc, K, bn_stats=bn_stats, kernel=kernel))
        key = ('lazya', tuple(la[0].shape), K, tuple(kernel))
        if key not in rec:
            rec[key] = dict(la=la, w=w_krsc, K=K, kernel=tuple(kernel), out=out, names=names, parked=parked.get(out.data_ptr()))
        return out

    def w_fwd(x, w_krsc, bias, K, R, S, stride, pad, **kw):
        out, names = logged(lambda: orig['conv2d_fwd'](x, w_krsc, bias, K, R, S, stride, pad, **kw))
        key = ('fwd', tuple(x.shape), K, R, tuple(stride))
        if key not in rec and ('jfwd_kernel' in names or 'conv3x3_c64' in names):
            rec[key] = dict(x=x, w=w_krsc, K=K, R=R, stride=tuple(stride), pad=tuple(pad), out=out, names=names,
                            parked=parked.get(out.data_ptr()))
        return out

    for n, f in (('conv2d_dgrad', w_dgrad), ('conv2d_bwd1x1_lazy', w_pair), ('conv2d_dgrad_lazy', w_dlazy),
                 ('conv2d_fwd_lazyz', w_lazyz), ('conv2d_fwd_lazya', w_lazya), ('conv2d_fwd', w_fwd), ('_park_stats', park)):
        monkeypatch.setattr(ops, n, f)

    # the stem goes to the library from inside its autograd Function: hooks on the module instead
    stem = {}
    conv1 = model.conv1

    stem_fwd = conv1.forward_from_nchw          # (the model calls this method directly: module hooks do not fire)

    def stem_forward(x_nchw):
        L.cn_kernel_log(1)
        stem['w'] = conv1.weight.detach().float().cpu().clone()      # (the optimizer step at the end updates it)
        out = stem_fwd(x_nchw)
        stem['fwd_names'] = L.cn_kernel_log(0).decode()
        stem['y'] = out.detach()
        out.register_hook(lambda gr: stem.__setitem__('dy', gr.detach()))
        return out
    monkeypatch.setattr(conv1, 'forward_from_nchw', stem_forward)
    stem_bwd = ops.StemPairConvFunction.backward

    def stem_backward(ctx, dy):      # (runs on the autograd thread: the library's kernel log is per thread)
        L.cn_kernel_log(1)
        out = stem_bwd(ctx, dy)
        stem['bwd_names'] = L.cn_kernel_log(0).decode()
        return out
    monkeypatch.setattr(ops.StemPairConvFunction, 'backward', staticmethod(stem_backward))
    tr.train([(x0, t0)])
    torch.cuda.synchronize()
    monkeypatch.setattr(conv1, 'forward_from_nchw', stem_fwd)
    monkeypatch.setattr(ops.StemPairConvFunction, 'backward', staticmethod(stem_bwd))
    for n, f in orig.items():
        monkeypatch.setattr(ops, n, f)

    report, served = [], set()

    def note(kind, key, names, **errs):
        served.update(n for n in names.split(';') if n)
        report.append('%-8s %-46s %-60s %s' % (kind, str(key[1:]), names[:60], '  '.join('%s %.1e' % kv for kv in errs.items())))

    # ---------------------------------------------------------------- junction data gradients (+ the c64 halo dgrad)
    for key, r in [(k, v) for k, v in rec.items() if k[0] == 'dgrad']:
        N, H, W, C = r['x_shape']
        K, R = r['K'], r['R']
        if r['bn'] is None:      # plain 3x3 halo data gradient: images 0, 1, 254, 255 on the CPU
            imgs = [0, 1, N - 2, N - 1]
            dyc = _f(r['dy'][imgs]).permute(0, 3, 1, 2).contiguous()
            w_oihw = _f(r['w']).reshape(C, R, R, K).permute(3, 0, 1, 2).contiguous()     # CRSK -> [K][C][R][S]
            ref = torch.nn.grad.conv2d_input((len(imgs), C, H, W), w_oihw, dyc, r['stride'], r['pad'])
            e = rel_l2(_f(r['out'][imgs]).permute(0, 3, 1, 2), ref)
            note('dgrad3x3', key, r['names'], dx=e)
            assert e < 5e-3, (key, e)
            continue
        assert R == 1 and r['stride'] == (1, 1), key
        g_out, partial, rows = r['out']
        bn_y, bn_mask, bn_stats, bn_relu = r['bn']
        M = N * H * W
        idx = _rows(M, M + C)
        dy2 = r['dy'].reshape(M, K)
        dx = _f(dy2[idx.to(dev)]) @ _f(r['w']).reshape(C, K).t()
        if r['addend'] is not None:
            if r['sub'] == 2:     # the even pixels of a stride-2 projection's input gradient, zero elsewhere
                n_, rem = idx // (H * W), idx % (H * W)
                h_, w_ = rem // W, rem % W
                even = (h_ % 2 == 0) & (w_ % 2 == 0)
                aH, aW = (H + 1) // 2, (W + 1) // 2
                aidx = (n_ * aH + h_ // 2) * aW + w_ // 2
                add = _f(r['addend'].reshape(-1, C)[aidx.clamp(max=N * aH * aW - 1).to(dev)])
                dx = dx + add * even.view(-1, 1).float()
            else:
                dx = dx + _f(r['addend'].reshape(M, C)[idx.to(dev)])
        if bn_mask is not None:
            keep = _bits(bn_mask.reshape(M, C // 8)[idx.to(dev)], C)
        else:
            st = _f(bn_stats)
            keep = (_f(bn_y.reshape(M, C)[idx.to(dev)]) * st[2 * C:3 * C] + st[3 * C:]) > 0 if bn_relu else torch.ones_like(dx, dtype=torch.bool)
        g_ref = torch.where(keep, dx, torch.zeros_like(dx))
        e_g = rel_l2(_f(g_out.reshape(M, C)[idx.to(dev)]), g_ref)
        # the fused reduction: column sums of the partial rows vs the stored g over the WHOLE tensor (fp64 on the GPU)
        st = bn_stats.double()
        gd = g_out.reshape(M, C).double()
        xhat = (bn_y.reshape(M, C).double() - st[:C]) * st[C:2 * C]
        s1_ref, s2_ref = gd.sum(0).cpu(), (gd * xhat).sum(0).cpu()
        ps = partial[:rows].double().sum(0).cpu()
        e1, e2 = rel_l2(ps[:C], s1_ref), rel_l2(ps[C:], s2_ref)
        del gd, xhat
        note('junction', key, r['names'], g=e_g, sum_g=e1, sum_g_xhat=e2)
        assert e_g < 5e-3 and e1 < 1e-5 and e2 < 1e-5, (key, e_g, e1, e2)

    # ---------------------------------------------------------------- junction pair and lazy-dy data gradients
    def lazy_dy_rows(r, idx, K):
        cf = _f(r['coef'])
        gq, yq = _f(r['g'].reshape(-1, K)[idx.to(dev)]), _f(r['y'].reshape(-1, K)[idx.to(dev)])
        return (cf[:K] * gq + (cf[K:2 * K] * yq + cf[2 * K:])).to(torch.bfloat16).float()

    for key, r in [(k, v) for k, v in rec.items() if k[0] == 'pair']:
        N, H, W, C = r['x'].shape
        K, M = r['K'], N * H * W
        idx = _rows(M, M + K)
        dx_ref = lazy_dy_rows(r, idx, K) @ _f(r['w']).reshape(C, K).t()
        e_dx = rel_l2(_f(r['out'].reshape(M, C)[idx.to(dev)]), dx_ref)
        # dW over ALL pixels, fp32 on the CPU in chunks of 64k pixels
        cf = _f(r['coef'])
        dw = torch.zeros(K, C, dtype=torch.float64)
        for m0 in range(0, M, 65536):
            sl = slice(m0, min(M, m0 + 65536))
            dyq = (cf[:K] * _f(r['g'].reshape(M, K)[sl]) + (cf[K:2 * K] * _f(r['y'].reshape(M, K)[sl]) + cf[2 * K:])).to(torch.bfloat16).float()
            dw += (dyq.t() @ _f(r['x'].reshape(M, C)[sl])).double()
        got = (_f(r['dw']) - r['beta'] * _f(r['dw_before'])).reshape(K, C).double() / r['scale']
        e_dw = rel_l2(got, dw)
        note('pair', key, r['names'], dx=e_dx, dW=e_dw)
        assert e_dx < 5e-3 and e_dw < 1e-4, (key, e_dx, e_dw)

    for key, r in [(k, v) for k, v in rec.items() if k[0] == 'dlazy']:
        N, H, W, C = r['x_shape']
        K = r['K']
        if r['stride'] != (1, 1):
            continue     # the strided projection's lazy dgrad runs on the tiled kernel (tests/test_ops.py::test_lazy_dy_*)
        M = N * H * W
        idx = _rows(M, M + K + 1)
        dx_ref = lazy_dy_rows(r, idx, K) @ _f(r['w']).reshape(C, K).t()
        e = rel_l2(_f(r['out'].reshape(M, C)[idx.to(dev)]), dx_ref)
        note('dlazy', key, r['names'], dx=e)
        assert e < 5e-3, (key, e)

    # ---------------------------------------------------------------- forward: lazy z, lazy a, streaming / halo forward
    def check_stats(r, K):
        if r.get('parked') is None:
            return {}
        partial, rows, pivot = r['parked']
        assert pivot is None
        yd = r['out'].reshape(-1, K).double()
        ps = partial[:rows].double().sum(0).cpu()
        e1, e2 = rel_l2(ps[:K], yd.sum(0).cpu()), rel_l2(ps[K:], (yd * yd).sum(0).cpu())
        assert e1 < 1e-5 and e2 < 1e-5, ('statistics partials', e1, e2)
        return dict(sum_y=e1, sum_y2=e2)

    for key, r in [(k, v) for k, v in rec.items() if k[0] == 'lazyz']:
        y3, res, stats, res_stats, z, mask, relu = r['lz']
        N, H, W, C = y3.shape
        K, M = r['K'], N * H * W
        idx = _rows(M, M + 7)
        st = _f(stats)
        rr = _f(res.reshape(M, C)[idx.to(dev)])
        if res_stats is not None:    # the projection shortcut's BatchNorm applied here: round_T(res * rscale + rshift)
            rs = _f(res_stats)
            rr = (rr * rs[2 * C:3 * C] + rs[3 * C:]).to(torch.bfloat16).float()
        pre = _f(y3.reshape(M, C)[idx.to(dev)]) * st[2 * C:3 * C] + st[3 * C:] + rr
        z_ref = pre.clamp_min(0) if relu else pre
        z_got = _f(z.reshape(M, C)[idx.to(dev)])
        e_z = rel_l2(z_got, z_ref)
        if mask is not None:     # the ReLU bits: exact wherever the pre-activation is clear of zero by more than rounding
            bits = _bits(mask.reshape(M, C // 8)[idx.to(dev)], C)
            clear = pre.abs() > 1e-2 * pre.abs().mean()
            assert bool((bits == (pre > 0))[clear].all()), key
        out_ref = z_got @ _f(r['w']).reshape(K, C).t()
        e_o = rel_l2(_f(r['out'].reshape(M, K)[idx.to(dev)]), out_ref)
        extra = check_stats(r, K)
        note('lazy z', key, r['names'], z=e_z, out=e_o, **extra)
        assert e_z < 5e-3 and e_o < 5e-3, (key, e_z, e_o)

    for key, r in [(k, v) for k, v in rec.items() if k[0] == 'lazya']:
        bn_y, stats, a, relu = r['la']
        N, H, W, C = bn_y.shape
        K, M = r['K'], N * H * W
        st = _f(stats)
        if r['kernel'] == (1, 1):
            idx = _rows(M, M + 11)
            pre = _f(bn_y.reshape(M, C)[idx.to(dev)]) * st[2 * C:3 * C] + st[3 * C:]
            a_got = _f(a.reshape(M, C)[idx.to(dev)])
            e_a = rel_l2(a_got, pre.clamp_min(0) if relu else pre)
            e_o = rel_l2(_f(r['out'].reshape(M, K)[idx.to(dev)]), a_got @ _f(r['w']).reshape(K, C).t())
        else:
            imgs = [0, 1, N - 2, N - 1]
            pre = _f(bn_y[imgs]) * st[2 * C:3 * C] + st[3 * C:]
            a_got = _f(a[imgs])
            e_a = rel_l2(a_got, pre.clamp_min(0) if relu else pre)
            w_oihw = _f(r['w']).reshape(K, 3, 3, C).permute(0, 3, 1, 2).contiguous()
            ref = F.conv2d(a_got.permute(0, 3, 1, 2).contiguous(), w_oihw, padding=1)
            e_o = rel_l2(_f(r['out'][imgs]).permute(0, 3, 1, 2), ref)
        extra = check_stats(r, K)
        note('lazy a', key, r['names'], a=e_a, out=e_o, **extra)
        assert e_a < 5e-3 and e_o < 5e-3, (key, e_a, e_o)

    for key, r in [(k, v) for k, v in rec.items() if k[0] == 'fwd']:
        N, H, W, C = r['x'].shape
        K, R, M = r['K'], r['R'], N * H * W
        if R == 1:
            idx = _rows(M, M + 13)
            e_o = rel_l2(_f(r['out'].reshape(M, K)[idx.to(dev)]), _f(r['x'].reshape(M, C)[idx.to(dev)]) @ _f(r['w']).reshape(K, C).t())
        else:
            imgs = [0, 1, N - 2, N - 1]
            w_oihw = _f(r['w']).reshape(K, R, R, C).permute(0, 3, 1, 2).contiguous()
            ref = F.conv2d(_f(r['x'][imgs]).permute(0, 3, 1, 2).contiguous(), w_oihw, stride=r['stride'], padding=r['pad'])
            e_o = rel_l2(_f(r['out'][imgs]).permute(0, 3, 1, 2), ref)
        extra = check_stats(r, K)
        note('fwd', key, r['names'], out=e_o, **extra)
        assert e_o < 5e-3, (key, e_o)

    # ---------------------------------------------------------------- stem halo kernels
    xq = x0.to(torch.bfloat16).float()
    wq = stem['w'].to(torch.bfloat16).float()
    imgs = [0, B - 1]
    y_ref = F.conv2d(xq[imgs], wq, stride=2, padding=3)
    e_y = rel_l2(_f(stem['y'][imgs]).permute(0, 3, 1, 2), y_ref)
    dyc = _f(stem['dy']).permute(0, 3, 1, 2).contiguous()
    dw_ref = torch.nn.grad.conv2d_weight(xq, (64, 3, 7, 7), dyc, 2, 3)
    e_w = rel_l2(conv1.weight.grad.detach().float().cpu(), dw_ref)
    report.append('%-8s y %.1e  dW %.1e  %s | %s' % ('stem', e_y, e_w, stem.get('fwd_names'), stem.get('bwd_names')))
    assert e_y < 5e-3 and e_w < 1e-4, (e_y, e_w)
    served.update(n for n in (stem.get('fwd_names', '') + ';' + stem.get('bwd_names', '')).split(';') if n)

    print('\n'.join(report))
    names = ' '.join(sorted(served))
    print('kernels that served a checked call:', names)
    for k in WANT:
        assert k in names, (k, names)
    assert len([k for k in rec if k[0] == 'dgrad' and rec[k]['bn'] is not None]) >= 7      # the 7 junction shapes of ResNet-50
    assert len([k for k in rec if k[0] == 'pair']) >= 1 and len([k for k in rec if k[0] == 'lazyz']) >= 2
    assert len([k for k in rec if k[0] == 'lazya']) >= 4
