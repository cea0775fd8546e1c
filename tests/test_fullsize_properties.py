"""Size-independent properties checked at BASELINE.json's full sizes (ResNet-50, B=256, 3x224x224,
bf16) where the CPU oracle is too slow to be the checker:
  * run-to-run determinism of a whole training step (fixed-order reductions, no atomics),
  * exact homogeneity of the conv kernels (scaling an operand by 2 is exact in bf16/fp32),
  * BatchNorm training output has zero mean / unit variance per channel, running stats follow,
  * softmax-CE gradient rows sum to zero, loss at initialisation ~= ln(1000),
  * one bf16 step vs one fp32 step of the same model agree within the bf16 tolerance.
Plus ragged batches (B=3, odd spatial sizes) against the oracle on both the emulator and the GPU."""
import math

import pytest
import torch

from conftest import HAS_GPU
from helpers import rel_l2

MODES = [pytest.param('emul'), pytest.param('gpu', marks=pytest.mark.gpu)]


def _full_step(dtype, steps=2, seed=123, B=256):
    import convnet_amd as ca
    dev = torch.device('cuda', 0)
    torch.manual_seed(seed)
    model = ca.models.resnet(dataset='imagenet', depth=50)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device='cuda:0', dtype=dtype,
                    grad_clip=1e9, print_freq=10 ** 9)
    g = torch.Generator().manual_seed(7)
    data = [(torch.randn(B, 3, 224, 224, generator=g).to(dev), torch.randint(0, 1000, (B,), generator=g).to(dev))
            for _ in range(steps)]
    recs = [tr.train([b]) for b in data]
    torch.cuda.synchronize()
    return recs, model, tr


@pytest.mark.gpu
def test_full_size_step_is_deterministic_and_sane():
    a, ma, _ = _full_step(torch.bfloat16)
    b, mb, _ = _full_step(torch.bfloat16)
    for ra, rb in zip(a, b):
        assert ra['loss'] == rb['loss'] and ra['grad'] == rb['grad'], 'training step is not run-to-run deterministic'
    assert torch.equal(ma._cn_arena.params, mb._cn_arena.params)
    assert a[0]['loss'] == pytest.approx(math.log(1000.0), abs=0.15)   # zero-gamma residual init: near-uniform logits
    assert all(math.isfinite(r['loss']) and math.isfinite(r['grad']) and r['grad'] > 0 for r in a)
    f, _, _ = _full_step(torch.float32, steps=1)
    assert a[0]['loss'] == pytest.approx(f[0]['loss'], abs=2e-2)
    assert a[0]['grad'] == pytest.approx(f[0]['grad'], rel=5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', [(256, 56, 64, 256, 1, 1, 0), (256, 14, 256, 256, 3, 1, 1), (256, 28, 512, 1024, 1, 2, 0)])
def test_full_size_conv_is_exactly_homogeneous(cfg):
    import convnet_amd as ca
    N, H, C, K, R, st, pad = cfg
    dev = torch.device('cuda', 0)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, H, C, generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn(K, R, R, C, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    wc = w.permute(3, 1, 2, 0).contiguous()
    y1 = ca.ops.conv2d_fwd(x, w, None, K, R, R, (st, st), (pad, pad))
    y2 = ca.ops.conv2d_fwd(x * 2, w, None, K, R, R, (st, st), (pad, pad))
    assert torch.equal(y2, y1 * 2)
    dy = torch.randn(y1.shape, generator=g).to(torch.bfloat16).to(dev)
    d1 = ca.ops.conv2d_dgrad(dy, wc, x.shape, K, R, R, (st, st), (pad, pad))
    d2 = ca.ops.conv2d_dgrad(dy * 2, wc, x.shape, K, R, R, (st, st), (pad, pad))
    assert torch.equal(d2, d1 * 2)
    g1 = torch.zeros(K, R, R, C, device=dev)
    g2 = torch.zeros(K, R, R, C, device=dev)
    ca.ops.conv2d_wgrad(x, dy, g1, C, K, R, R, (st, st), (pad, pad), beta=0.0)
    ca.ops.conv2d_wgrad(x * 2, dy, g2, C, K, R, R, (st, st), (pad, pad), beta=0.0)
    assert torch.equal(g2, g1 * 2)
    # and wgrad agrees with an independent fp64 evaluation on a random sample of filter taps
    xs, dys = x.double().cpu(), dy.double().cpu()
    for (k, r, s, c) in [(0, 0, 0, 0), (K - 1, R - 1, R - 1, C - 1), (K // 2, R // 2, 0, C // 3)]:
        P = y1.shape[1]
        ph = torch.arange(P) * st + r - pad
        pw = torch.arange(P) * st + s - pad
        vh, vw = (ph >= 0) & (ph < H), (pw >= 0) & (pw < H)
        xsub = xs[:, ph[vh]][:, :, pw[vw], c]
        ref = (xsub * dys[:, vh][:, :, vw, k]).sum()
        assert float(g1[k, r, s, c]) == pytest.approx(float(ref), rel=2e-3, abs=1e-2)


@pytest.mark.gpu
def test_full_size_batchnorm_normalises():
    import convnet_amd as ca
    dev = torch.device('cuda', 0)
    C = 256
    bn = ca.nn.BatchNorm2d(C)
    m = torch.nn.Sequential(bn)
    ca.engine.prepare(m, dev, torch.bfloat16)
    y = (torch.randn(256, 56, 56, C, device=dev) * 3 + 1.5).to(torch.bfloat16)
    with torch.no_grad():
        bn.train()
        z = bn(y)
    zf = z.float().view(-1, C)
    assert zf.mean(0).abs().max() < 2e-2 and (zf.var(0, unbiased=False) - 1).abs().max() < 2e-2
    yf = y.float().view(-1, C)
    assert rel_l2(bn.running_mean, 0.1 * yf.mean(0)) < 1e-3
    assert rel_l2(bn.running_var, 0.9 + 0.1 * yf.var(0, unbiased=True)) < 1e-3


@pytest.mark.gpu
def test_full_size_softmax_gradient_rows_sum_to_zero():
    import convnet_amd as ca
    dev = torch.device('cuda', 0)
    logits = (torch.randn(256, 1000, device=dev) * 4).requires_grad_(True)
    target = torch.randint(0, 1000, (256,), device=dev)
    loss = ca.CrossEntropyLoss()(logits, target)
    loss.backward()
    assert logits.grad.sum(1).abs().max() < 1e-6
    assert float(loss) == pytest.approx(float(torch.nn.functional.cross_entropy(logits.detach().cpu(), target.cpu())), rel=1e-5)


@pytest.mark.parametrize('mode', MODES)
def test_ragged_batch_and_odd_image_size_match_oracle(mode):
    """B=3 (smaller than every tile), 36x28 inputs (odd feature-map sizes after the strided layers)."""
    if mode == 'emul' and HAS_GPU:
        pytest.skip('emulator mode is for GPU-less hosts')
    if mode == 'gpu' and not HAS_GPU:
        pytest.skip('no GPU')
    import convnet_amd as ca
    from oracle import convnet_oracle as O
    dev = 'cuda:0' if mode == 'gpu' else 'cpu'
    kw = dict(depth=18, num_classes=16, inplanes=8, width=(8, 16, 32, 64))
    g = torch.Generator().manual_seed(3)
    data = [(torch.randn(3, 3, 36, 28, generator=g), torch.randint(0, 16, (3,), generator=g)) for _ in range(2)]
    torch.manual_seed(123)
    ref = O.OracleResNet(**kw)
    rr = O.oracle_train(ref, data)
    torch.manual_seed(123)
    model = ca.models.resnet(**kw)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=dev, dtype=torch.float32,
                    grad_clip=1e9, print_freq=10 ** 9)
    for b, r in zip(data, rr):
        out = tr.train([b])
        assert out['loss'] == pytest.approx(r['loss'], abs=1e-4)
        assert out['grad'] == pytest.approx(r['grad'], rel=1e-3)
    val = tr.validate([(data[0][0][:1], data[0][1][:1])])   # batch of ONE in eval mode
    rv = O.oracle_validate(ref, [(data[0][0][:1], data[0][1][:1])])
    assert val['loss'] == pytest.approx(rv['loss'], rel=1e-3)
