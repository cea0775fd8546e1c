"""BASELINE config 5 (`resnet(quantize=True)`, simulated 8-bit training) through the HIP kernels of
csrc/quant.hip, against (a) per-operator vectors recorded from the reference's own
models/modules/quantize.py (tests/golden/quant_ops.pt) and (b) training trajectories of the reference
Trainer (tests/golden/traj_r{18,50}s_quant.json), with the CPU oracle (oracle/quant_oracle.py, pinned on the
same fixtures in fp32 and float64) run beside it on the same noise stream.

The stochastic-rounding noise of the gradient quantisers is the reference's own stream: the tests install a
noise source that draws `torch.empty(shape).uniform_(-0.5, 0.5)` from torch's global CPU generator in the
order the backward pass runs, which is the order the reference draws in.

Tolerances.  The primitives are bit-exact in fp32 (same operation order, no FMA contraction).  Whole-network
quantities pass through quantisers that turn a one-ulp difference into a full quantisation step
(range / 255), so fp32 trajectories are compared at loss abs 2e-2 / grad-norm rel 3e-2 -- the same band in
which the fp32 oracle follows the fp32 reference (tests/test_quant_oracle.py; in float64 it follows to 1e-7).
"""
import json
import os

import pytest
import torch

from conftest import HAS_GPU
from helpers import GOLDEN, rel_l2

MODES = [pytest.param('emul'), pytest.param('gpu', marks=pytest.mark.gpu)]
SMALL = dict(num_classes=16, inplanes=8, width=[8, 16, 32, 64])


def _dev(mode):
    if mode == 'emul' and HAS_GPU:
        pytest.skip('emulator mode is for GPU-less hosts')
    if mode == 'gpu' and not HAS_GPU:
        pytest.skip('no GPU')
    return torch.device('cuda', 0) if mode == 'gpu' else torch.device('cpu')


def _nhwc(x, dev):
    return x.permute(0, 2, 3, 1).contiguous().to(dev)


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous().cpu()


def _cpu_noise(shape):
    return torch.empty(shape).uniform_(-0.5, 0.5)


@pytest.fixture()
def reference_noise():
    import convnet_amd as ca
    ca.quant.set_noise_source(_cpu_noise)
    yield
    ca.quant.set_noise_source(None)


@pytest.mark.parametrize('mode', MODES)
def test_quantisers_are_bit_exact_against_the_reference_vectors(mode):
    dev = _dev(mode)
    import convnet_amd as ca
    Q = ca.quant
    g = torch.load(os.path.join(GOLDEN, 'quant_ops.pt'))
    qm = Q.QuantMeasure(8, shape_measure=(1, 1, 1, 1), flatten_dims=(1, -1)).to(dev)
    qm.train()
    x = g['qm_x'].to(dev)                       # per-sample rows: any memory order with the batch leading
    assert torch.equal(qm(x).cpu(), g['qm_train_y'])
    assert torch.equal(qm.running_range.cpu(), g['qm_running_range'])
    assert torch.equal(qm.running_zero_point.cpu(), g['qm_running_zero_point'])
    qm.eval()
    assert torch.equal(qm((g['qm_x'] * 1.5).to(dev)).cpu(), g['qm_eval_y'])
    # per-output-channel weight quantiser
    w = g['w'].to(dev)
    wq = torch.empty_like(w)
    ca._lib.check(ca._lib.load().cn_quantize_rows(w.data_ptr(), wq.data_ptr(), 8, 36, 8, ca._lib.stream_of(w)))
    assert torch.equal(wq.cpu(), g['w_q'])
    # 16-bit bias quantiser over the global range
    b = g['b'].to(dev)
    qp = Q.qparams(Q.minmax_rows(b, 1), 1, 1)
    assert torch.equal(Q.quantize(b, qp[0:1], qp[1:2], 16).cpu(), g['b_q16'])
    # bf16 storage: the same grid, rounded to bf16 on store
    xb = g['qm_x'].to(dev).to(torch.bfloat16)
    qp = Q.qparams(Q.minmax_rows(xb, 4), 4, 0)
    zp, rng = qp.cpu().tolist()
    ref = (((xb.float().cpu() - zp) / (rng / 255.)).clamp(0, 255).round() * (rng / 255.) + zp).to(torch.bfloat16)
    assert torch.equal(Q.quantize(xb, qp[0:1], qp[1:2]).cpu(), ref)
    # the built-in generator: unbiased stochastic rounding onto the same grid
    big = torch.linspace(-1, 1, 4096, device=dev).view(4, 1024).contiguous()
    qp = Q.qparams(Q.minmax_rows(big, 4), 4, 1)
    acc = torch.zeros_like(big)
    for _ in range(8):
        y = Q.quantize(big, qp[0:1], qp[1:2], 4, stochastic=True)
        lv = (y - (-1)) / (2 / 15.)
        assert (lv - lv.round()).abs().max() < 1e-4
        acc += y
    assert (acc / 8 - big).abs().mean() < 0.02


def _prepare(mod, dev, dtype=torch.float32):
    import convnet_amd as ca
    ca.engine.prepare(mod, dev, dtype)
    return mod


@pytest.mark.parametrize('mode', MODES)
def test_rangebn_forward_backward_against_the_reference_vectors(mode, reference_noise):
    dev = _dev(mode)
    import convnet_amd as ca
    g = torch.load(os.path.join(GOLDEN, 'quant_ops.pt'))
    bn = ca.quant.RangeBN(8)
    with torch.no_grad():
        bn.weight.copy_(g['rbn_w'])
        bn.bias.copy_(g['rbn_b'])
    _prepare(bn, dev)
    bn.train()
    x = _nhwc(g['rbn_x'], dev).requires_grad_(True)
    torch.manual_seed(77)
    y = bn(x)
    y.backward(_nhwc(g['rbn_gy'], dev))
    assert rel_l2(_nchw(y.detach()), g['rbn_y']) < 1e-6
    assert rel_l2(_nchw(x.grad), g['rbn_dx']) < 1e-5
    assert rel_l2(bn.weight.grad.cpu(), g['rbn_dw']) < 1e-5
    assert rel_l2(bn.bias.grad.cpu(), g['rbn_db']) < 1e-5
    assert rel_l2(bn.running_mean.cpu(), g['rbn_running_mean']) < 1e-5   # (the kernel sums in double)
    assert rel_l2(bn.running_var.cpu(), g['rbn_running_var']) < 1e-5
    assert rel_l2(bn.quantize_input.running_range.cpu(), g['rbn_qi_range']) < 1e-6   # batch mean summed in double
    # eval mode: running statistics, no autograd
    bn.eval()
    with torch.no_grad():
        ye = _nchw(bn(_nhwc(g['rbn_x'], dev)))
    qi = bn.quantize_input
    zp, rng = float(qi.running_zero_point), float(qi.running_range)
    xq = ((g['rbn_x'] - zp) / (rng / 255.)).clamp(0, 255).round() * (rng / 255.) + zp
    ref = (xq - bn.running_mean.cpu().view(1, -1, 1, 1)) / (bn.running_var.cpu().view(1, -1, 1, 1) + bn.eps) \
        * g['rbn_w'].view(1, -1, 1, 1) + g['rbn_b'].view(1, -1, 1, 1)
    assert rel_l2(ye, ref) < 1e-6


@pytest.mark.parametrize('mode', MODES)
def test_qconv_and_qlinear_against_the_reference_vectors(mode, reference_noise):
    dev = _dev(mode)
    import convnet_amd as ca
    g = torch.load(os.path.join(GOLDEN, 'quant_ops.pt'))
    conv = ca.quant.QConv2d(8, 16, 3, stride=1, padding=1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(g['qc_w'])
    _prepare(conv, dev)
    conv.train()
    x = _nhwc(g['qc_x'], dev).requires_grad_(True)
    torch.manual_seed(78)
    y = conv(x)
    y.backward(_nhwc(g['qc_gy'], dev))
    assert rel_l2(_nchw(y.detach()), g['qc_y']) < 1e-6
    assert rel_l2(_nchw(x.grad), g['qc_dx']) < 1e-5
    assert rel_l2(conv.weight.grad.cpu(), g['qc_dw']) < 1e-5
    lin = ca.quant.QLinear(32, 16)
    with torch.no_grad():
        lin.weight.copy_(g['ql_w'])
        lin.bias.copy_(g['ql_b'])
    _prepare(lin, dev)
    lin.train()
    x = g['ql_x'].to(dev).requires_grad_(True)
    torch.manual_seed(79)
    y = lin(x)
    y.backward(g['ql_gy'].to(dev))
    assert rel_l2(y.detach().cpu(), g['ql_y']) < 1e-6
    assert rel_l2(x.grad.cpu(), g['ql_dx']) < 1e-5
    assert rel_l2(lin.weight.grad.cpu(), g['ql_dw']) < 1e-5
    assert rel_l2(lin.bias.grad.cpu(), g['ql_db']) < 1e-5


def _engine_trajectory(meta, depth, dev, dtype, steps):
    import convnet_amd as ca
    from helpers import golden_batches   # noqa: F401  (same generator as the fixture)
    torch.manual_seed(123)
    model = ca.models.resnet(dataset='imagenet', quantize=True, depth=depth, **SMALL)
    sd = model.state_dict()
    assert set(sd.keys()) == set(meta['keys'].keys())
    for k, v in sd.items():
        assert list(v.shape) == meta['keys'][k], k
        if v.dtype.is_floating_point:    # seeded construction reproduces the reference's initial weights
            assert abs(float(v.double().sum()) - meta['init_sums'][k][0]) <= 1e-6 * max(1.0, meta['init_sums'][k][1]), k
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=str(dev), dtype=dtype,
                    grad_clip=1e9, print_freq=10 ** 9)
    g = torch.Generator().manual_seed(meta['seed'])
    data = [(torch.randn(meta['B'], 3, meta['size'], meta['size'], generator=g),
             torch.randint(0, meta['classes'], (meta['B'],), generator=g)) for _ in range(meta['steps'])][:steps]
    recs = []
    for x, t in data:
        r = tr.train([(x, t)])
        recs.append({k: float(r[k]) for k in ('loss', 'prec1', 'prec5', 'grad')})
    return recs, tr, model, data


# rel-L2 bounds on the tensors the config-5 fixtures keep after the last step (one 0.15 band before).
# Measured on the MI355X, fp32 engine on the reference's noise stream, maxima over ResNet-18s / ResNet-50s / full-size
# ResNet-50: conv1.weight 0.061, fc.weight 0.037, bn1.running_mean 0.043, fc range 0.0099, downsample 0.0028, fc.bias 0.0016,
# bn1.running_var 0.0014, bn1 range 0.00073, layer1.0.conv1.weight 8.6e-6, conv1 range 6.2e-8.  The stem filter and the
# classifier see every rounding flip of the three steps (the oracle itself is 7 % from the reference there).
FINAL_TOL = {'conv1.weight': 0.10, 'fc.weight': 0.07, 'bn1.running_mean': 0.08, 'fc.quantize_input.running_range': 0.03,
             'layer2.0.downsample.0.weight': 0.01, 'fc.bias': 5e-3, 'bn1.running_var': 5e-3,
             'bn1.quantize_input.running_range': 3e-3, 'layer1.0.conv1.weight': 1e-4,
             'conv1.quantize_input.running_range': 1e-5}


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('tag,depth', [('r18s_quant', 18), ('r50s_quant', 50)])
def test_quantised_fp32_trajectory_follows_the_reference(mode, tag, depth, reference_noise):
    dev = _dev(mode)
    meta = json.load(open(os.path.join(GOLDEN, 'traj_%s.json' % tag)))
    if mode == 'emul' and depth != 18:
        pytest.skip('emulated suite: one ResNet-18 step only (the GPU run does all steps of both depths)')
    steps = meta['steps'] if mode == 'gpu' else 1
    recs, tr, model, data = _engine_trajectory(meta, depth, dev, torch.float32, steps)
    for i, (r, gr) in enumerate(zip(recs, meta['records'])):
        assert r['loss'] == pytest.approx(gr['loss'], abs=1e-4 if i == 0 else 2e-2), (i, r, gr)
        assert r['grad'] == pytest.approx(gr['grad'], rel=3e-2), (i, r, gr)
        if i == 0:
            assert r['prec1'] == gr['prec1'] and r['prec5'] == gr['prec5']
    if steps == meta['steps']:
        final = torch.load(os.path.join(GOLDEN, 'traj_%s_final.pt' % tag))
        sd = model.state_dict()
        # after three steps at lr 0.1 the fp32 ORACLE already sits 0.001-7 % away from the fp32 reference on
        # these tensors (35 % on a last-BN gamma that starts at zero): rounding flips compound.  Per-tensor bounds =
        # ~2x the deviation measured on the MI355X (round 4; FINAL_TOL above), not one loose band.
        for k, tol in FINAL_TOL.items():
            assert rel_l2(sd[k].float().cpu(), final[k]) < tol, (k, rel_l2(sd[k].float().cpu(), final[k]), tol)
        val = tr.validate(data[:2])     # eval mode: running ranges / statistics
        assert val['loss'] == pytest.approx(meta['validate']['loss'], abs=5e-2)


@pytest.mark.gpu
def test_quantised_bf16_step_and_builtin_noise():
    """bf16 storage with the kernels' own rounding-noise generator (the production configuration): the first
    step's loss matches the fp32 reference (forward only depends on the deterministic quantisers), the
    gradient norm is within the quantisation noise band, training stays finite."""
    dev = _dev('gpu')
    import convnet_amd as ca
    meta = json.load(open(os.path.join(GOLDEN, 'traj_r50s_quant.json')))
    ca.quant.manual_seed(1)
    recs, tr, model, data = _engine_trajectory(meta, 50, dev, torch.bfloat16, 3)
    assert recs[0]['loss'] == pytest.approx(meta['records'][0]['loss'], abs=5e-2)
    assert recs[0]['grad'] == pytest.approx(meta['records'][0]['grad'], rel=1e-1)
    assert all(torch.isfinite(torch.tensor([r['loss'], r['grad']])).all() for r in recs)


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('cfg', [(16, 8, 32, 1, 1, 0), (32, 9, 64, 3, 1, 1), (16, 8, 128, 3, 2, 1), (64, 6, 64, 1, 2, 0),
                                 (48, 7, 136, 3, 1, 1)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_int8_mfma_forward_equals_the_simulated_convolution(mode, cfg, dtype):
    """csrc/qconv_i8.hip: the QConv2d forward product on v_mfma_i32_32x32x32_i8 (levels - 128, exact int32
    accumulation, zero-point / border-class corrections in the epilogue) against (a) the float-kernel path of the
    same module and (b) the CPU oracle's F.conv2d on the dequantised operands.  fp32 output: rel-L2 1e-5 (the
    int8 path is the more exact of the two); bf16 output: one bf16 rounding."""
    dev = _dev(mode)
    if mode == 'emul' and (cfg[2] > 64 or (dtype == torch.bfloat16 and cfg[3] != 3)):
        pytest.skip('emulated suite: the small shapes only (all of them run on the GPU)')
    import convnet_amd as ca
    from oracle import quant_oracle as QO
    C, H, K, R, stride, pad = cfg
    g = torch.Generator().manual_seed(C * 131 + K)
    conv = ca.quant.QConv2d(C, K, R, stride=stride, padding=pad, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(K, C, R, R, generator=g) * 0.1)
    _prepare(conv, dev, dtype)
    conv.train()
    x_nchw = torch.randn(4, C, H, H + 1, generator=g)
    x = _nhwc(x_nchw, dev).to(dtype)
    with torch.no_grad():
        conv.int8_forward = False
        y_sim = conv(x).float()
        conv.int8_forward = True
        y_i8 = conv(x).float()
    xq = x.float().cpu().permute(0, 3, 1, 2)
    zp, rng = QO.qparams_mean(xq)
    wzp, wrng = QO.qparams_rows(conv.weight.detach().float().cpu())
    ref = torch.nn.functional.conv2d(QO.quantize(xq, zp, rng), QO.quantize(conv.weight.detach().float().cpu(), wzp, wrng),
                                     None, stride, pad)
    # the oracle averages the per-sample min / max in fp32, the kernel in double: a last-bit difference in the
    # zero point moves a handful of elements to the neighbouring level (1/255 of the range each)
    assert rel_l2(_nchw(y_i8), ref) < (2e-3 if dtype == torch.float32 else 8e-3)
    assert rel_l2(_nchw(y_i8), _nchw(y_sim)) < (1e-5 if dtype == torch.float32 else 8e-3)
    # and through autograd: same gradients as the float-kernel path (the backward pass IS that path)
    ca.quant.set_noise_source(lambda shape: torch.zeros(shape))
    try:
        grads = []
        for flag in (False, True):
            conv.int8_forward = flag
            conv._arena.zero_grad()     # gradients accumulate into the flat arena
            xi = x.clone().requires_grad_(True)
            y = conv(xi)
            y.backward(torch.ones_like(y))
            grads.append((xi.grad.float().cpu().clone(), conv.weight.grad.float().cpu().clone()))
        assert rel_l2(grads[1][0], grads[0][0]) < 1e-6 and rel_l2(grads[1][1], grads[0][1]) < 1e-6
    finally:
        ca.quant.set_noise_source(None)


@pytest.mark.gpu
def test_full_size_quantised_resnet50_follows_the_reference(reference_noise):
    """ResNet-50 at full width (2048-channel layers, 7x7 maps: 49-value RangeBN chunks), B=16, 224x224, fp32,
    against the reference Trainer's records (tests/golden/traj_r50_quant_full.json), on the reference's noise
    stream.  Same tolerance band as the small models (quantisers amplify one-ulp differences)."""
    dev = _dev('gpu')
    import convnet_amd as ca
    meta = json.load(open(os.path.join(GOLDEN, 'traj_r50_quant_full.json')))
    torch.manual_seed(123)
    model = ca.models.resnet(dataset='imagenet', quantize=True, depth=50)
    sd = model.state_dict()
    assert set(sd.keys()) == set(meta['keys'].keys())
    for k, v in sd.items():
        if v.dtype.is_floating_point:
            assert abs(float(v.double().sum()) - meta['init_sums'][k][0]) <= 1e-6 * max(1.0, meta['init_sums'][k][1]), k
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=str(dev),
                    dtype=torch.float32, grad_clip=1e9, print_freq=10 ** 9)
    g = torch.Generator().manual_seed(meta['seed'])
    data = [(torch.randn(meta['B'], 3, meta['size'], meta['size'], generator=g),
             torch.randint(0, meta['classes'], (meta['B'],), generator=g)) for _ in range(meta['steps'])]
    for i, ((x, t), gr) in enumerate(zip(data, meta['records'])):
        r = tr.train([(x, t)])
        assert float(r['loss']) == pytest.approx(gr['loss'], abs=1e-3 if i == 0 else 3e-2), (i, r, gr)
        assert float(r['grad']) == pytest.approx(gr['grad'], rel=3e-2), (i, r, gr)
    final = torch.load(os.path.join(GOLDEN, 'traj_r50_quant_full_final.pt'))
    sd = model.state_dict()
    for k in ('conv1.weight', 'layer1.0.conv1.weight', 'bn1.running_mean', 'bn1.running_var',
              'conv1.quantize_input.running_range', 'fc.quantize_input.running_range'):
        assert rel_l2(sd[k].float().cpu(), final[k]) < FINAL_TOL[k], (k, rel_l2(sd[k].float().cpu(), final[k]))


@pytest.mark.gpu
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_bench_scale_quantised_resnet50_follows_the_reference(dtype, reference_noise):
    """Config 5 at a bench-scale batch: ResNet-50 quantize=True, B=128, 224x224 (the largest batch whose reference
    autograd graph fits the build container; the bench runs B=256), fp32 AND bf16 storage, against the reference
    Trainer's fp32 records (tests/golden/traj_r50_quant_b128.json) on the reference's rounding-noise stream, with a
    bound on loss and gradient norm at EVERY step (measured: fp32 5e-5 / 4e-4 and 4e-3 / 1e-4; bf16 4e-4 / 3e-4 and
    3e-3 / 3e-3):
        fp32:  loss abs 1e-3 (step 0) / 5e-3, gradient norm rel 2e-2
        bf16:  loss abs 5e-3 at both steps,   gradient norm rel 2e-2
    and rel-L2 0.06 on the tensors the fixture keeps after two steps at lr 0.1 on 8-bit grids (measured <= 0.03), except
    the stem BatchNorm's running mean in bf16 (a near-zero mean of rounded activations: 0.14 measured, bound 0.25)."""
    path = os.path.join(GOLDEN, 'traj_r50_quant_b128.json')
    if not os.path.exists(path):
        pytest.skip('fixture not generated (oracle/make_golden_quant.py big 128)')
    dev = _dev('gpu')
    import convnet_amd as ca
    meta = json.load(open(path))
    torch.manual_seed(123)
    model = ca.models.resnet(dataset='imagenet', quantize=True, depth=50)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=str(dev), dtype=dtype,
                    grad_clip=1e9, print_freq=10 ** 9)
    g = torch.Generator().manual_seed(meta['seed'])
    data = [(torch.randn(meta['B'], 3, meta['size'], meta['size'], generator=g),
             torch.randint(0, meta['classes'], (meta['B'],), generator=g)) for _ in range(meta['steps'])]
    f32 = dtype == torch.float32
    for i, ((x, t), gr) in enumerate(zip(data, meta['records'])):
        r = tr.train([(x, t)])
        print('step', i, dtype, float(r['loss']), gr['loss'], float(r['grad']), gr['grad'])
        assert float(r['loss']) == pytest.approx(gr['loss'], abs=(1e-3 if i == 0 else 5e-3) if f32 else 5e-3), (i, r, gr)
        assert float(r['grad']) == pytest.approx(gr['grad'], rel=2e-2), (i, r, gr)
    final = torch.load(os.path.join(GOLDEN, 'traj_r50_quant_b128_final.pt'))
    sd = model.state_dict()
    for k in ('conv1.weight', 'layer1.0.conv1.weight', 'bn1.running_mean', 'bn1.running_var',
              'conv1.quantize_input.running_range', 'fc.quantize_input.running_range'):
        print(k, rel_l2(sd[k].float().cpu(), final[k]))
        assert rel_l2(sd[k].float().cpu(), final[k]) < (0.25 if (k == 'bn1.running_mean' and not f32) else 0.06), k


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_producer_side_fusions_ops_change_no_bit(mode, dtype):
    """Round 4, op level (quantize.py:158-182, 101-112, 288-326): cn_rangebn_fwd_q (input quantiser inside the statistics
    pass, snapped tensor stored, per-sample min / max of z) == cn_quantize + cn_rangebn_fwd + cn_minmax_rows;
    cn_rangebn_bwd_mm (routing folded into the apply pass, per-sample min / max of dx) == cn_rangebn_bwd + cn_minmax_rows;
    cn_eltwise_mm == cn_eltwise + cn_minmax_rows: every tensor bit for bit."""
    dev = _dev(mode)
    import convnet_amd as ca
    Q, L, lib = ca.quant, ca._lib.load(), ca._lib
    ptr, code = lib.ptr, lib.dtype_code(dtype)
    shapes = [(4, 6, 8, 16), (2, 4, 4, 32)] if mode == 'emul' else [(32, 28, 28, 128), (16, 14, 14, 256), (64, 56, 56, 64)]
    # few pixels per sample, few channels, a large batch: the per-sample min / max partials outgrow the reduction's partial
    # rows and live behind the coefficients (this shape used to fail with CN_EWORKSPACE)
    shapes.append((128, 1, 1, 64))
    for (N, H, W, C) in shapes:
        g_ = torch.Generator().manual_seed(N + H + C)
        M, chunks = N * H * W, 16
        y = (torch.randn(N, H, W, C, generator=g_) * 1.3 + 0.2).to(dtype).to(dev)
        # ties inside a chunk (first-index rule) and a routed element that is also the tensor's extreme
        y.view(-1, C)[0, :] = 9.0
        y.view(-1, C)[1, :] = 9.0
        w = (torch.rand(C, generator=g_) + 0.5).to(dev)
        b = (torch.randn(C, generator=g_) * 0.2).to(dev)
        fix = Q._scale_fix(M // chunks)
        for relu in (True, False):
            outs = {}
            for fused in (False, True):
                rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
                stats = torch.empty(2 * C, dtype=torch.float32, device=dev)
                arg = torch.empty(C * 2 * chunks, dtype=torch.int32, device=dev)
                ws = ca.ops.workspace(L.cn_rangebn_workspace(M, C, chunks), dev, 'quant')
                z = torch.empty_like(y)
                qp = Q.qparams(Q.minmax_rows(y, N), N, 0)
                if fused:
                    qy = torch.full_like(y, float('nan'))
                    zmm = torch.empty(N * 2, dtype=torch.float32, device=dev)
                    lib.check(L.cn_rangebn_fwd_q(ptr(y), ptr(qp), 8, ptr(qy), None, ptr(z), ptr(w), ptr(b), ptr(rm), ptr(rv), 0.1,
                                                 1e-5, chunks, fix, ptr(stats), ptr(arg), M, C, int(relu), code, N, ptr(zmm),
                                                 ptr(ws), ws.numel() * 4, lib.stream_of(y)), 'cn_rangebn_fwd_q')
                else:
                    qy = Q.quantize(y, qp[0:1], qp[1:2], 8)
                    lib.check(L.cn_rangebn_fwd(ptr(qy), None, ptr(z), ptr(w), ptr(b), ptr(rm), ptr(rv), 0.1, 1e-5, chunks, fix,
                                               ptr(stats), ptr(arg), M, C, int(relu), 1, code, ptr(ws), ws.numel() * 4,
                                               lib.stream_of(y)), 'cn_rangebn_fwd')
                    zmm = Q.minmax_rows(z, N)
                # backward on a fixed gradient
                gq = (torch.randn(N, H, W, C, generator=torch.Generator().manual_seed(7)) * 0.1).to(dtype).to(dev)
                dx = torch.empty_like(y)
                dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
                if fused:
                    dxmm = torch.empty(N * 2, dtype=torch.float32, device=dev)
                    lib.check(L.cn_rangebn_bwd_mm(ptr(gq), ptr(qy), ptr(w), ptr(stats), ptr(arg), ptr(dx), ptr(dw), ptr(db), M, C,
                                                  chunks, fix, code, N, ptr(dxmm), ptr(ws), ws.numel() * 4, lib.stream_of(y)),
                              'cn_rangebn_bwd_mm')
                else:
                    lib.check(L.cn_rangebn_bwd(ptr(gq), ptr(qy), ptr(w), ptr(stats), ptr(arg), ptr(dx), ptr(dw), ptr(db), M, C,
                                               chunks, fix, code, ptr(ws), ws.numel() * 4, lib.stream_of(y)), 'cn_rangebn_bwd')
                    dxmm = Q.minmax_rows(dx, N)
                outs[fused] = [t.float().cpu().clone() for t in (qy, z, stats, arg, zmm, dx, dw, db, dxmm, rm, rv)]
            for i, (a0, a1) in enumerate(zip(outs[False], outs[True])):
                assert torch.equal(a0, a1), ((N, H, W, C), relu, i)
        # elementwise + min / max
        bb = (torch.randn(N, H, W, C, generator=g_)).to(dtype).to(dev)
        cc = (torch.randn(N, H, W, C, generator=g_)).to(dtype).to(dev)
        for op in (2, 4):
            a0 = torch.empty_like(bb)
            lib.check(L.cn_eltwise(op, ptr(a0), ptr(bb), ptr(cc), bb.numel(), code, lib.stream_of(bb)), 'cn_eltwise')
            mm0 = Q.minmax_rows(a0, N)
            a1, mm1 = Q.eltwise_mm(op, bb, cc, N)
            assert torch.equal(a0.cpu(), a1.cpu()) and torch.equal(mm0.cpu(), mm1.cpu()), ((N, H, W, C), op)


@pytest.mark.parametrize('mode', MODES)
def test_producer_side_fusions_change_no_bit_of_a_trajectory(mode, reference_noise):
    """Round 4, model level: with quant.FUSE_QUANT on (the default) the quantised ResNet trains to the same numbers, bit for
    bit, as with every quantiser running its own min / max and quantise passes - and the stash really is used (no
    cn_minmax_rows launch is left for the tensors the fused producers measured)."""
    if mode == 'emul':
        pytest.skip('GPU suite (two trajectories: 85 s on the emulator; the op-level test above covers the emulator)')
    dev = _dev(mode)
    import convnet_amd as ca
    meta = json.load(open(os.path.join(GOLDEN, 'traj_r18s_quant.json')))
    res = {}
    saved = ca.quant.FUSE_QUANT
    calls = {}
    real = ca.quant._take_minmax
    try:
        for fused in (False, True):
            ca.quant.FUSE_QUANT = fused
            hits = [0, 0]

            def counting(x, rows, hits=hits):
                r = real(x, rows)
                hits[0 if r is None else 1] += 1
                return r
            ca.quant._take_minmax = counting
            recs, tr, model, data = _engine_trajectory(meta, 18, dev, torch.float32, 2 if mode == 'gpu' else 1)
            res[fused] = (recs, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()})
            calls[fused] = tuple(hits)
    finally:
        ca.quant.FUSE_QUANT = saved
        ca.quant._take_minmax = real
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    for k, v in res[False][1].items():
        assert torch.equal(v, res[True][1][k]), k
    assert calls[False][1] == 0 and calls[True][1] > 0, calls
    # what is still measured by its own pass: the convolution outputs (RangeBN's input quantiser), the network input, the
    # classifier's input and bias - fewer than half of the quantisers
    assert calls[True][1] >= calls[True][0] * 0.8, calls


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('depth', [18, 50])
def test_block_input_gradient_sum_in_the_dgrad_epilogue_changes_no_bit_in_fp32(mode, depth, reference_noise):
    """Round 6 (quant.JUNCTION_ADD): the two gradients meeting at a quantised block's input are summed in the data
    gradient's epilogue of whichever branch arrives second (identity shortcut: conv1's; projection: conv1's or the
    projection's) instead of a separate add pass.  fp32: the same numbers bit for bit, and no separate add is launched."""
    dev = _dev(mode)
    if mode == 'emul':
        pytest.skip('GPU suite (two trajectories: 100 s on the emulator; test_quantised_block_switches_change_no_bit covers the emulator)')
    import convnet_amd as ca
    meta = json.load(open(os.path.join(GOLDEN, 'traj_r%ds_quant.json' % depth)))
    res, adds = {}, {}
    saved, real_add = ca.quant.JUNCTION_ADD, ca.ops.add_
    try:
        for fused in (False, True):
            ca.quant.JUNCTION_ADD = fused
            n = [0]

            def counting(a, b, n=n):
                n[0] += 1
                return real_add(a, b)
            ca.ops.add_ = counting
            recs, tr, model, data = _engine_trajectory(meta, depth, dev, torch.float32, 2 if mode == 'gpu' else 1)
            res[fused] = (recs, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()})
            adds[fused] = n[0]
    finally:
        ca.quant.JUNCTION_ADD = saved
        ca.ops.add_ = real_add
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    for k, v in res[False][1].items():
        assert torch.equal(v, res[True][1][k]), k
    assert adds[False] > 0 and adds[True] == 0, adds


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_8bit_level_storage_changes_no_bit_of_a_trajectory(mode, dtype, reference_noise):
    """Round 6 (quant.STORE8): RangeBN's snapped input (saved for backward) and the quantised gradient its backward kernels
    read are kept as one byte per element - the LEVELS - and de-quantised on load (cn_rangebn_fwd_q8, cn_quantize_levels,
    cn_rangebn_bwd_q8).  Same trajectory and same final state bit for bit as with the snapped values stored in the compute
    dtype, fp32 and bf16; and the level tensors really are what is saved (uint8)."""
    dev = _dev(mode)
    if mode == 'emul':
        pytest.skip('GPU suite (two trajectories: 100 s on the emulator; test_quantised_block_switches_change_no_bit covers the emulator)')
    import convnet_amd as ca
    meta = json.load(open(os.path.join(GOLDEN, 'traj_r18s_quant.json')))
    res, kinds = {}, {}
    saved = ca.quant.STORE8
    real = ca.quant.RangeBNFunction.forward
    try:
        for on in (False, True):
            ca.quant.STORE8 = on
            seen = set()

            def spy(ctx, *a, _seen=seen, **k):
                out = real(ctx, *a, **k)
                _seen.add(ctx.to_save[0].dtype if hasattr(ctx, 'to_save') else None)
                return out
            ca.quant.RangeBNFunction.forward = staticmethod(spy)
            recs, tr, model, data = _engine_trajectory(meta, 18, dev, dtype, 2 if mode == 'gpu' else 1)
            res[on] = (recs, {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()})
            kinds[on] = seen
    finally:
        ca.quant.STORE8 = saved
        ca.quant.RangeBNFunction.forward = staticmethod(real)
    assert res[False][0] == res[True][0], (res[False][0], res[True][0])
    for k, v in res[False][1].items():
        assert torch.equal(v, res[True][1][k]), k
    assert torch.uint8 in kinds[True] and torch.uint8 not in kinds[False], kinds


@pytest.mark.parametrize('mode', MODES)
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_quantised_block_switches_change_no_bit(mode, dtype, reference_noise):
    """One quantised bottleneck block (identity shortcut) + one with a projection, forward and backward, with each round-6
    switch of quant.py off and on: STORE8 (8-bit level storage of RangeBN's saved input and its output gradient),
    QP_FROM_PRODUCER (gradient producers emit the gradient quantiser's parameters) - the same output, input gradient and
    parameter gradients bit for bit in fp32 and bf16; JUNCTION_ADD (block-input gradient sum in the later data gradient's
    epilogue) - bit for bit in fp32, one rounding apart in bf16.  The fast, emulator-sized form of the trajectory tests above."""
    dev = _dev(mode)
    import convnet_amd as ca
    from convnet_amd.models import resnet as R
    torch.manual_seed(4)
    model = ca.models.resnet(dataset='imagenet', quantize=True, depth=50, num_classes=8, inplanes=8, width=[8, 16, 16, 16])
    ca.engine.prepare(model, dev, dtype)
    model.train()
    blocks = [model.layer1[0], model.layer1[1]]      # projection shortcut, identity shortcut
    g = torch.Generator().manual_seed(6)
    x0 = torch.randn(4, 8, 8, 8, generator=g).to(dev).to(dtype)      # NHWC, 8 channels = layer1's input
    gy = None

    def run():
        nonlocal gy
        model._cn_arena.zero_grad()
        x = x0.clone().requires_grad_(True)
        torch.manual_seed(11)
        y = blocks[1](blocks[0](x))
        if gy is None:
            gy = torch.randn(y.shape, generator=g).to(dev).to(dtype)
        y.backward(gy)
        return (y.detach().float().cpu().clone(), x.grad.float().cpu().clone(), model._cn_arena.grads.detach().cpu().clone())

    Q = ca.quant
    for name, exact_bf16 in (('STORE8', True), ('QP_FROM_PRODUCER', True), ('JUNCTION_ADD', False)):
        saved = getattr(Q, name)
        try:
            setattr(Q, name, False)
            a = run()
            setattr(Q, name, True)
            b = run()
        finally:
            setattr(Q, name, saved)
        for u, v in zip(a, b):
            if dtype == torch.float32 or exact_bf16:
                assert torch.equal(u, v), name
            else:
                assert rel_l2(u, v) < 1e-2, name


@pytest.mark.parametrize('mode', MODES)
def test_division_free_quantiser_levels_are_exact_at_the_rounding_boundaries(mode):
    """Round 6 (q_levels_chunk): the quantisers multiply by 1/scale where that provably gives the level of the reference's
    division and redo a chunk with the division otherwise.  Brute force against IEEE fp32 arithmetic on the CPU
    (rint(clamp((x + (-zp)) / scale + noise))): every level identical - random inputs, inputs packed within 1e-7 ... 1e-3
    (relative) of every half-integer boundary on both sides, values far outside the range, with and without rounding noise."""
    dev = _dev(mode)
    import convnet_amd as ca
    L = ca._lib.load()
    g = torch.Generator().manual_seed(17)
    zp, rng = -1.7320508, 5.4365637
    scale = torch.tensor(rng, dtype=torch.float32) / 255.0
    ks = torch.arange(0, 256, dtype=torch.float32)
    eps = torch.tensor([0.0, 1e-7, 3e-7, 1e-6, 1e-5, 1e-4, 1e-3], dtype=torch.float32)
    eps = torch.cat([eps, -eps])
    near = (ks.view(-1, 1) + 0.5) * (1.0 + eps.view(1, -1))                        # targets for (x - zp) / scale
    xs = [torch.tensor(zp, dtype=torch.float32) + near.flatten() * scale,
          torch.tensor(zp, dtype=torch.float32) + (torch.rand(1 << 16, generator=g) * 300.0 - 20.0) * scale,
          torch.tensor([1e9, -1e9, 1e30, 0.0, zp, zp + rng], dtype=torch.float32)]
    x = torch.cat(xs)
    x = torch.cat([x, torch.zeros((-x.numel()) % 4)])                              # whole fp32 chunks
    n = x.numel()
    for with_noise in (False, True):
        noise = (torch.rand(n, generator=g) - 0.5) if with_noise else None
        t = (x + (-torch.tensor(zp, dtype=torch.float32))) / scale
        if noise is not None:
            t = t + noise
        want = torch.round(t.clamp(0.0, 255.0)).to(torch.uint8)                   # torch.round: half to even, like rintf
        xd = x.to(dev)
        y8 = torch.empty(n, dtype=torch.uint8, device=dev)
        zpt = torch.tensor([zp], dtype=torch.float32, device=dev)
        rgt = torch.tensor([rng], dtype=torch.float32, device=dev)
        nd = noise.to(dev) if noise is not None else None
        ca._lib.check(L.cn_quantize_levels(xd.data_ptr(), y8.data_ptr(), n, 0, zpt.data_ptr(), rgt.data_ptr(), 8,
                                           nd.data_ptr() if nd is not None else None, int(with_noise), 0, None,
                                           ca._lib.stream_of(xd)), 'cn_quantize_levels')
        got = y8.cpu()
        bad = (got != want).nonzero().flatten()
        assert bad.numel() == 0, (with_noise, bad[:8].tolist(), x[bad[:8]].tolist(), got[bad[:8]].tolist(), want[bad[:8]].tolist())
