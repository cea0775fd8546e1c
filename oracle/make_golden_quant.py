"""Golden fixtures for BASELINE config 5 (ResNet `{'quantize': True}`: simulated 8-bit training) produced by
executing the REAL reference: /root/reference models/resnet.py:385-391 rebinds torch.nn.{Conv2d, Linear,
BatchNorm2d} to models/modules/quantize.py's QConv2d / QLinear / RangeBN, and the reference Trainer trains the
result on CPU fp32.  Runs only in the build container; separate from make_golden.py because that rebinding is
process-global and irreversible.

The reference code for this configuration does not run as written on the installed torch 2.10 (SURVEY.md
section 8c, hazard ii).  Two run-time repairs are applied here, by wrapping the reference's own functions
(nothing is copied), and are the documented deviations of this path:

  1. ``UniformQuantizeGrad.forward`` returns its input object itself (quantize.py:98); the in-place ReLU / `+=`
     that follow then trip autograd's version check.  Repair: return ``input.clone()`` (numerically identical).
  2. A tensor whose quantisation range is zero (``fc.bias`` at its zero init, quantize.py:239-242, and the
     all-zero gradient that reaches every conv behind a last-BN with gamma = 0 at step 0) gives scale = 0 and
     0/0 = NaN in ``UniformQuantize.forward`` (quantize.py:64-66).  Repair: ``calculate_qparams`` reports a
     range of 1 wherever it measured 0, which makes the quantiser the identity on such a tensor (every element
     equals the zero point, so it maps to level 0 and back to itself).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_quant.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'

_T_START = __import__('time').time()


def assert_no_new_bytecode():
    """/root/reference is read-only: nothing this process imports from it may leave bytecode behind.  (Round 3's
    oracle/time_reference_cpu.py ran without sys.dont_write_bytecode and left *.pyc there, stamped 2026-09-26 18:25;
    this tree does not own the reference and cannot delete them, so only files written since this process started count.)"""
    for d, _, files in os.walk(REF):
        if os.path.basename(d) == '__pycache__':
            new = [f for f in files if os.path.getmtime(os.path.join(d, f)) >= _T_START - 1.0]
            assert not new, 'bytecode leaked into the reference tree: %s/%s' % (d, new[:3])

sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, 'refshim'))

import torch  # noqa: E402

torch.set_num_threads(8)

import models as ref_models                     # noqa: E402
import models.modules.quantize as refq          # noqa: E402
from trainer import Trainer as RefTrainer       # noqa: E402
from utils.optim import OptimRegime             # noqa: E402
from utils.cross_entropy import CrossEntropyLoss  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

# ---- repair 1: clone in UniformQuantizeGrad.forward
_orig_qg_forward = refq.UniformQuantizeGrad.forward


def _qg_forward(ctx, input, *args, **kw):
    return _orig_qg_forward(ctx, input, *args, **kw).clone()


refq.UniformQuantizeGrad.forward = staticmethod(_qg_forward)

# ---- repair 2: zero range -> identity quantiser
_orig_calc = refq.calculate_qparams


def _calc(*args, **kw):
    qp = _orig_calc(*args, **kw)
    rng = torch.where(qp.range == 0, torch.ones_like(qp.range), qp.range)
    return refq.QParams(range=rng, zero_point=qp.zero_point, num_bits=qp.num_bits)


refq.calculate_qparams = _calc

SMALL = dict(width=[8, 16, 32, 64], inplanes=8, num_classes=16)


def tensor_sums(sd):
    return {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()}


def batches(n, B, size, classes, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(B, 3, size, size, generator=g), torch.randint(0, classes, (B,), generator=g))
            for _ in range(n)]


def trajectory(tag, model_kw, B, size, classes, steps, seed, dtype=torch.float):
    """k training steps of the reference Trainer on the quantised model.  The stochastic rounding of the
    gradient quantiser (quantize.py:67-69,111) draws from torch's global CPU generator, seeded by
    manual_seed(123) before model construction exactly as main.py:137 does; a replica has to consume that
    stream in the same order to reproduce the trajectory (tests/test_quant.py does)."""
    torch.manual_seed(123)
    model = ref_models.resnet(dataset='imagenet', quantize=True, **model_kw)
    sd0 = model.state_dict()
    init_sums = tensor_sums({k: v for k, v in sd0.items() if v.dtype.is_floating_point})
    keys = {k: list(v.shape) for k, v in sd0.items()}
    model.to(dtype)
    opt = OptimRegime(model, model.regime)
    tr = RefTrainer(model, CrossEntropyLoss(), opt, device_ids=None, device='cpu', dtype=dtype,
                    distributed=False, grad_clip=1e9, print_freq=10 ** 9)
    data = batches(steps, B, size, classes, seed)
    recs = []
    for x, t in data:
        r = tr.train([(x, t)])
        recs.append({k: float(r[k]) for k in ('loss', 'prec1', 'prec5', 'grad')})
    val = tr.validate(data[:2])
    sd = model.state_dict()
    out = {'tag': tag, 'model_kw': model_kw, 'B': B, 'size': size, 'classes': classes, 'steps': steps,
           'seed': seed, 'records': recs, 'keys': keys,
           'validate': {k: float(val[k]) for k in ('loss', 'prec1', 'prec5')},
           'init_sums': init_sums,
           'final_sums': tensor_sums({k: v for k, v in sd.items() if v.dtype.is_floating_point})}
    with open(os.path.join(OUT, 'traj_%s.json' % tag), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    keep = ['conv1.weight', 'conv1.quantize_input.running_range', 'conv1.quantize_input.running_zero_point',
            'bn1.running_mean', 'bn1.running_var', 'bn1.quantize_input.running_range', 'layer1.0.conv1.weight',
            'layer1.0.bn3.weight', 'layer2.0.downsample.0.weight', 'layer4.1.bn2.weight',
            'layer4.1.bn2.running_var', 'fc.weight', 'fc.bias', 'fc.quantize_input.running_range']
    if size > 64:   # full-size model: keep the fixture small
        keep = ['conv1.weight', 'conv1.quantize_input.running_range', 'bn1.running_mean', 'bn1.running_var',
                'layer1.0.conv1.weight', 'fc.bias', 'fc.quantize_input.running_range', 'layer4.1.bn2.running_var']
    torch.save({k: sd[k].clone() for k in keep if k in sd}, os.path.join(OUT, 'traj_%s_final.pt' % tag))
    print(tag, recs, 'val', out['validate'])


def op_vectors():
    """Per-op known-answer vectors of the reference's quantisation primitives (with the two repairs), small
    enough to commit: quantize() per tensor / per output channel, QuantMeasure train+eval, RangeBN forward and
    its autograd gradients (max / min routing), conv2d_biprec gradients with a fixed noise stream."""
    g = torch.Generator().manual_seed(5)
    out = {}
    x = torch.randn(4, 8, 6, 6, generator=g)
    qm = refq.QuantMeasure(8, shape_measure=(1, 1, 1, 1), flatten_dims=(1, -1))
    qm.train()
    out['qm_x'] = x
    out['qm_train_y'] = qm(x).clone()
    out['qm_running_range'] = qm.running_range.clone()
    out['qm_running_zero_point'] = qm.running_zero_point.clone()
    qm.eval()
    out['qm_eval_y'] = qm(x * 1.5).clone()
    w = torch.randn(8, 4, 3, 3, generator=g) * 0.2
    out['w'] = w
    out['w_q'] = refq.quantize(w, qparams=refq.calculate_qparams(w, num_bits=8, flatten_dims=(1, -1),
                                                                 reduce_dim=None)).clone()
    b = torch.randn(8, generator=g)
    out['b'] = b
    out['b_q16'] = refq.quantize(b, num_bits=16, flatten_dims=(0, -1)).clone()
    # RangeBN forward + backward (16 chunks of B*H*W/16 = 18 values)
    xb = torch.randn(8, 8, 6, 6, generator=g).requires_grad_(True)
    bn = refq.RangeBN(8)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(8, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(8, generator=g) * 0.1)
    bn.train()
    torch.manual_seed(77)     # noise of the output-gradient quantiser
    yb = bn(xb * 1.0)     # RangeBN quantises its input in place: feed a non-leaf
    gy = torch.randn(yb.shape, generator=g)
    yb.backward(gy)
    out.update(rbn_x=xb.detach().clone(), rbn_w=bn.weight.detach().clone(), rbn_b=bn.bias.detach().clone(),
               rbn_y=yb.detach().clone(), rbn_gy=gy, rbn_dx=xb.grad.clone(), rbn_dw=bn.weight.grad.clone(),
               rbn_db=bn.bias.grad.clone(), rbn_running_mean=bn.running_mean.clone(),
               rbn_running_var=bn.running_var.clone(),
               rbn_qi_range=bn.quantize_input.running_range.clone())
    # QConv2d forward + backward
    conv = refq.QConv2d(8, 16, 3, stride=1, padding=1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(16, 8, 3, 3, generator=g) * 0.1)
    conv.train()
    xc = torch.randn(4, 8, 6, 6, generator=g).requires_grad_(True)
    torch.manual_seed(78)
    yc = conv(xc)
    gc = torch.randn(yc.shape, generator=g)
    yc.backward(gc)
    out.update(qc_x=xc.detach().clone(), qc_w=conv.weight.detach().clone(), qc_y=yc.detach().clone(), qc_gy=gc,
               qc_dx=xc.grad.clone(), qc_dw=conv.weight.grad.clone())
    # QLinear forward + backward (bias quantised to 16 bits)
    lin = refq.QLinear(32, 16)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(16, 32, generator=g) * 0.1)
        lin.bias.copy_(torch.randn(16, generator=g) * 0.1)
    lin.train()
    xl = torch.randn(8, 32, generator=g).requires_grad_(True)
    torch.manual_seed(79)
    yl = lin(xl)
    gl = torch.randn(yl.shape, generator=g)
    yl.backward(gl)
    out.update(ql_x=xl.detach().clone(), ql_w=lin.weight.detach().clone(), ql_b=lin.bias.detach().clone(),
               ql_y=yl.detach().clone(), ql_gy=gl, ql_dx=xl.grad.clone(), ql_dw=lin.weight.grad.clone(),
               ql_db=lin.bias.grad.clone())
    torch.save(out, os.path.join(OUT, 'quant_ops.pt'))
    print('quant_ops.pt', {k: tuple(v.shape) for k, v in out.items()})


def full():
    """Full-size ResNet-50 with quantize=True (2048-channel layers, 7x7 maps with 49-value RangeBN chunks), B=16."""
    trajectory('r50_quant_full', dict(depth=50), B=16, size=224, classes=1000, steps=2, seed=43)


def big_batch(B=128):
    """Full-size ResNet-50, quantize=True, at a bench-scale batch (B=128: what the build container's 62 GB hold for the
    reference's fp32 autograd graph; the bench runs B=256): 2 steps.  For the bf16 / fp32 parity test at a batch where
    the RangeBN chunks span 8 samples each and the activation quantisers average over 128 per-sample ranges."""
    trajectory('r50_quant_b%d' % B, dict(depth=50), B=B, size=224, classes=1000, steps=2, seed=47)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'big':
        big_batch(int(sys.argv[2]) if len(sys.argv) > 2 else 128)
        assert_no_new_bytecode()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'full':
        full()
        assert_no_new_bytecode()
        sys.exit(0)
    op_vectors()
    trajectory('r50s_quant', dict(depth=50, **SMALL), B=16, size=64, classes=16, steps=3, seed=41)
    trajectory('r18s_quant', dict(depth=18, **SMALL), B=16, size=64, classes=16, steps=3, seed=42)
    # float64 runs: the quantisers turn one-ulp differences into whole quantisation steps, so an fp32 replica
    # can only follow the fp32 trajectories to ~1e-3; in double precision such flips are vanishingly rare and a
    # restatement with the same semantics reproduces these records to ~1e-9 (tests/test_quant_oracle.py)
    trajectory('r50s_quant_f64', dict(depth=50, **SMALL), B=16, size=64, classes=16, steps=3, seed=41,
               dtype=torch.double)
    trajectory('r18s_quant_f64', dict(depth=18, **SMALL), B=16, size=64, classes=16, steps=3, seed=42,
               dtype=torch.double)
    assert_no_new_bytecode()
