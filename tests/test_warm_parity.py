"""Warm-start parity: the headline comparison that pins the INNER-block backward.

`init_model` zeroes the last BatchNorm gamma of every residual block (/root/reference/models/resnet.py:24-28),
so in the cold-start goldens every inner convolution's weight / data gradient is exactly zero at step 0 and
~1e-4 of its natural size at step 1: a 100 % wrong inner wgrad would pass them.  The warm goldens
(oracle/make_golden.py `warm`) overwrite gamma / beta / running statistics of every BatchNorm of the REFERENCE
model with seeded non-trivial values, train 3 steps with the reference Trainer on CPU and record, besides the meters,
the RAW autograd gradient of step 0 per tensor (tensor hooks - the reference's WeightDecay regulariser adds wd*p to
p.grad in place before the optimizer step; norm of the full tensor + 2048 seeded samples) for conv1 / conv2 / conv3 /
downsample weights and BN gamma / beta of two blocks per stage, and the same tensors at the end.

Conditioning decides what can be asserted (tools/conditioning.py, profiles/r03_warm_fixture_conditioning.txt):
  * with EVERY gamma in [0.5, 1.5) the step-0 gradient of ResNet-50 is chaotic in the rounding sense: the reference's
    own fp32 run is only within 2e-2 of its float64 run and PyTorch's bf16 autocast is uncorrelated with it (1.3);
    the fixture therefore draws the LAST gamma of every block from [0.03, 0.1) - 300-1000x the cold-start values,
    every inner gradient of natural relative size - where fp32 follows float64 to ~1e-3;
  * the small fixture (batch 8, 32x32) and ResNet-18 b=256 are recorded from the reference run in FLOAT64, ResNet-50
    b=256 in fp32 (float64 would need ~50 GB here);
  * element-wise agreement of ANY bf16 run with an fp32 one is bounded by ReLU decisions that flip under a 2^-9
    perturbation (PyTorch's own autocast: 0.1 - 0.4 per tensor on this fixture): the fixture records the autocast
    error per tensor and the bf16 engine is held to it - "as close to the fp32 reference as PyTorch's own bf16" -
    while tests/test_step_local_consistency.py pins every bf16 kernel tightly on the tensors it really sees.

Stated tolerances, per tensor:
  fp32 engine vs float64 reference (small, batch 8)   : gradient norm rel 5e-3, sampled gradient rel-L2 5e-3
  fp32 engine vs float64 reference (ResNet-18 b=256)   : norm rel 1e-3, sampled gradient rel-L2 3e-3
  fp32 engine vs fp32 reference (ResNet-50 b=256)      : norm rel 2e-3, sampled gradient rel-L2 5e-3 (BOTH sides carry
                                                         ~1e-3 of fp32 rounding here); final tensors after 3 steps 5e-3
  bf16 engine                                          : per-tensor gradient norm within 5e-2 (BN gamma / beta: 1e-1);
                                                         sampled gradient rel-L2 <= max(1.5 x autocast's, 5e-2)
CPU (`-m "not gpu"`): the oracle restatement (run in float64) follows the small fixture to 1e-6 - which pins the
oracle on warm-start data - and the engine runs it through the emulator.
"""
import pytest
import torch

from conftest import HAS_GPU
from helpers import golden_batches, load_warm, rel_l2, run_engine_trajectory, sample_tensor, warm_bn_state

MODES = [pytest.param('emul'), pytest.param('gpu', marks=pytest.mark.gpu)]


def _dev(mode):
    if mode == 'emul' and HAS_GPU:
        pytest.skip('emulator mode is for GPU-less hosts')
    if mode == 'gpu' and not HAS_GPU:
        pytest.skip('no GPU')
    return torch.device('cuda', 0) if mode == 'gpu' else torch.device('cpu')


def _is_bn(name):
    return '.bn' in name or name.startswith('bn') or 'downsample.1' in name


def _check_tensors(got, gold, norm_tol, l2_tol, bn_l2_tol=None, what='grad'):
    worst = {}
    for k, g in gold.items():
        norm, val = got[k]
        assert g['norm'] > 0, k     # the fixture really exercises this tensor
        tol = bn_l2_tol if (bn_l2_tol is not None and _is_bn(k)) else l2_tol
        err = rel_l2(val, g['val'])
        worst[k] = err
        assert norm == pytest.approx(g['norm'], rel=norm_tol), (what, k, norm, g['norm'])
        assert err < tol, (what, k, err)
    return worst


def test_warm_fixture_exercises_the_inner_blocks():
    """Every recorded step-0 gradient is of natural size: nothing is scaled away by a zero gamma."""
    for tag in ('r50s_warm', 'r18_b256_warm', 'r50_b256_warm'):
        meta, tens = load_warm(tag)
        norms = meta['grad0_norms']
        inner = [k for k in norms if 'layer' in k and 'conv' in k]
        assert len(inner) >= 16
        assert min(norms[k] for k in inner) > 1e-3 * max(norms.values()), tag
        assert set(tens['grad0']) == set(norms)


def test_oracle_follows_the_warm_reference_trajectory():
    from oracle import convnet_oracle as O
    meta, tens = load_warm('r50s_warm')
    kw = meta['model_kw']
    torch.manual_seed(123)
    assert meta['reference_dtype'] == 'float64'
    model = O.OracleResNet(kw['depth'], kw['num_classes'], kw['inplanes'], tuple(kw['width']))
    warm_bn_state(model, meta['warm_seed'], bn_type=torch.nn.BatchNorm2d)
    model.double()
    data = [(x.double(), t) for x, t in golden_batches(meta)]
    params = dict(model.named_parameters())
    raw = {}     # the raw autograd gradients (the oracle's optimizer, like the reference's regulariser, adds wd*p in place)
    for k in tens['grad0']:
        params[k].register_hook(lambda g, k=k: raw.__setitem__(k, g.detach().clone()))
    recs = O.oracle_train(model, data[:1])
    got = {k: sample_tensor(raw[k], k) for k in tens['grad0']}
    _check_tensors(got, tens['grad0'], 1e-6, 1e-6)      # (the fixture stores fp32 samples of the float64 gradients)
    assert recs[0]['loss'] == pytest.approx(meta['records'][0]['loss'], rel=1e-9)
    assert recs[0]['grad'] == pytest.approx(meta['records'][0]['grad'], rel=1e-6)


@pytest.mark.parametrize('mode', MODES)
def test_fp32_engine_warm_small(mode):
    dev = _dev(mode)
    meta, tens = load_warm('r50s_warm')
    grads = {k: None for k in tens['grad0']}
    steps = None if mode == 'gpu' else 1      # the emulated CPU suite stays short
    recs, tr, model, data = run_engine_trajectory(meta, torch.float32, dev, steps, graph=False, grads_after_step0=grads)
    r, g = recs[0], meta['records'][0]
    assert r['loss'] == pytest.approx(g['loss'], abs=1e-4)
    assert r['prec1'] == g['prec1'] and r['prec5'] == g['prec5']
    assert r['grad'] == pytest.approx(g['grad'], rel=5e-3)
    worst = _check_tensors(grads, tens['grad0'], 5e-3, 5e-3)
    print('worst fp32 step-0 gradient rel-L2 vs the float64 reference:', sorted(worst.items(), key=lambda kv: -kv[1])[:3])
    # later steps of this fixture (lr 0.1, gradient norm ~280: the loss climbs 2.8 -> 6.3 -> 9.4) amplify the
    # fp32-vs-float64 difference of step 0; they are followed loosely, the b=256 fixtures carry the multi-step claim
    for r, g in zip(recs[1:], meta['records'][1:]):
        assert r['loss'] == pytest.approx(g['loss'], rel=5e-2)     # (measured on the GPU: 0.5 % / 1.4 %)
        assert r['loss'] == r['loss'] and r['grad'] == r['grad']


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['r18_b256_warm', 'r50_b256_warm'])
def test_fp32_engine_warm_at_headline_batch(tag):
    meta, tens = load_warm(tag)
    f64 = meta['reference_dtype'] == 'float64'
    grads = {k: None for k in tens['grad0']}
    recs, tr, model, data = run_engine_trajectory(meta, torch.float32, torch.device('cuda', 0), grads_after_step0=grads)
    for r, g in zip(recs, meta['records']):
        assert r['loss'] == pytest.approx(g['loss'], abs=5e-4)
        assert abs(r['prec1'] - g['prec1']) <= 100.0 / meta['B'] + 1e-6      # fp32 summation order can flip one argmax tie
        assert r['grad'] == pytest.approx(g['grad'], rel=2e-3)
    worst = _check_tensors(grads, tens['grad0'], 1e-3 if f64 else 2e-3, 3e-3 if f64 else 5e-3)
    print('worst fp32 step-0 gradient rel-L2 (%s reference):' % meta['reference_dtype'],
          sorted(worst.items(), key=lambda kv: -kv[1])[:4])
    sd = model.state_dict()
    # three steps at lr 0.1 from a non-trivial state: summation-order differences of the gradients are amplified
    _check_tensors({k: sample_tensor(sd[k], k) for k in tens['final']}, tens['final'], 5e-3, 5e-3, what='final')
    val = tr.validate(data[:2])
    assert val['loss'] == pytest.approx(meta['validate']['loss'], rel=5e-3)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['r50_b256_warm', 'r18_b256_warm'])
def test_bf16_engine_warm_at_headline_batch(tag):
    """The bench configuration (ResNet-50 bf16 b=256) on non-trivial BatchNorm state: per tensor, the step-0 gradient
    norm within 5e-2 (BN parameters 1e-1) of the reference's and its element-wise distance no larger than 1.5 x the
    distance of PyTorch's own bf16 autocast run of the reference model (recorded in the fixture), floor 5e-2."""
    meta, tens = load_warm(tag)
    grads = {k: None for k in tens['grad0']}
    recs, tr, model, data = run_engine_trajectory(meta, torch.bfloat16, torch.device('cuda', 0), steps=2,
                                                  grads_after_step0=grads)
    B = meta['B']
    for i, (r, g) in enumerate(zip(recs, meta['records'])):
        assert r['loss'] == pytest.approx(g['loss'], abs=3e-2 if i == 0 else 8e-2), i
        assert abs(r['prec1'] - g['prec1']) <= 200.0 / B + 1e-6
        assert r['grad'] == pytest.approx(g['grad'], rel=5e-2), i
    ac = meta['autocast_err']
    rows = []
    for k, g in tens['grad0'].items():
        norm, val = grads[k]
        err = rel_l2(val, g['val'])
        rows.append((k, err, ac[k][0], abs(norm - g['norm']) / g['norm']))
        assert norm == pytest.approx(g['norm'], rel=1e-1 if _is_bn(k) else 5e-2), (k, norm, g['norm'])
        assert err <= max(1.5 * ac[k][0], 5e-2), (k, err, 'autocast', ac[k][0])
    rows.sort(key=lambda r: -r[1])
    print('bf16 step-0 gradients, worst (tensor, engine rel-L2, PyTorch-autocast rel-L2, norm err):', rows[:5])
    import statistics
    print('median engine / autocast error ratio: %.2f' % statistics.median(r[1] / max(r[2], 1e-9) for r in rows))
