"""Pins oracle/quant_oracle.py (the CPU restatement of BASELINE config 5, `{'quantize': True}`) against
fixtures produced by the reference itself (oracle/make_golden_quant.py): per-operator known-answer vectors
and training trajectories.  The quantisers amplify one-ulp differences into whole quantisation steps, so the
fp32 trajectories are followed to ~1e-2 while the float64 ones pin the semantics to 1e-7."""
import json
import os

import pytest
import torch

from oracle import convnet_oracle as O
from oracle import quant_oracle as Q

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SMALL = dict(num_classes=16, inplanes=8, width=(8, 16, 32, 64))


def _close(a, b, tol):
    d = (a - b).abs().max().item()
    assert d <= tol * max(b.abs().max().item(), 1e-30), (d, b.abs().max().item())


def test_quantisation_primitives_match_the_reference_vectors():
    g = torch.load(os.path.join(GOLD, 'quant_ops.pt'))
    qm = Q.QuantMeasureState((1, 1, 1, 1))
    assert torch.equal(qm(g['qm_x'], True), g['qm_train_y'])
    assert torch.equal(qm.running_range, g['qm_running_range'])
    assert torch.equal(qm.running_zero_point, g['qm_running_zero_point'])
    assert torch.equal(qm(g['qm_x'] * 1.5, False), g['qm_eval_y'])
    zp, r = Q.qparams_rows(g['w'])
    assert torch.equal(Q.quantize(g['w'], zp, r), g['w_q'])
    zp, r = Q.qparams_extreme(g['b'])
    assert torch.equal(Q.quantize(g['b'], zp, r, 16), g['b_q16'])


def test_rangebn_qconv_qlinear_forward_backward_match_the_reference_vectors():
    g = torch.load(os.path.join(GOLD, 'quant_ops.pt'))
    bn = Q.OracleRangeBN(8)
    bn.weight.data.copy_(g['rbn_w'])
    bn.bias.data.copy_(g['rbn_b'])
    bn.train()
    x = g['rbn_x'].clone().requires_grad_(True)
    torch.manual_seed(77)
    y = bn(x)
    y.backward(g['rbn_gy'])
    for k, v in (('rbn_y', y), ('rbn_dx', x.grad), ('rbn_dw', bn.weight.grad), ('rbn_db', bn.bias.grad),
                 ('rbn_running_mean', bn.running_mean), ('rbn_running_var', bn.running_var)):
        _close(v.detach(), g[k], 1e-6)
    conv = Q.OracleQConv2d(8, 16, 3, 1, 1)
    conv.weight.data.copy_(g['qc_w'])
    conv.train()
    x = g['qc_x'].clone().requires_grad_(True)
    torch.manual_seed(78)
    y = conv(x)
    y.backward(g['qc_gy'])
    for k, v in (('qc_y', y), ('qc_dx', x.grad), ('qc_dw', conv.weight.grad)):
        _close(v.detach(), g[k], 1e-6)
    lin = Q.OracleQLinear(32, 16)
    lin.weight.data.copy_(g['ql_w'])
    lin.bias.data.copy_(g['ql_b'])
    lin.train()
    x = g['ql_x'].clone().requires_grad_(True)
    torch.manual_seed(79)
    y = lin(x)
    y.backward(g['ql_gy'])
    for k, v in (('ql_y', y), ('ql_dx', x.grad), ('ql_dw', lin.weight.grad), ('ql_db', lin.bias.grad)):
        _close(v.detach(), g[k], 1e-6)


@pytest.mark.parametrize('tag,depth,dtype,ltol,gtol', [
    ('r50s_quant_f64', 50, torch.double, 1e-7, 1e-6), ('r18s_quant_f64', 18, torch.double, 1e-7, 1e-6),
    ('r50s_quant', 50, torch.float, 2e-2, 3e-2), ('r18s_quant', 18, torch.float, 2e-2, 3e-2)])
def test_quantised_training_trajectory_follows_the_reference(tag, depth, dtype, ltol, gtol):
    gj = json.load(open(os.path.join(GOLD, 'traj_%s.json' % tag)))
    torch.manual_seed(123)
    m = Q.OracleQuantResNet(depth, **SMALL)
    sd = Q.state_dict_like_reference(m)
    assert set(sd.keys()) == set(gj['keys'].keys())
    for k, v in sd.items():
        assert list(v.shape) == gj['keys'][k], k
        if v.dtype.is_floating_point:    # same seeded construction: same initial weights
            assert abs(float(v.double().sum()) - gj['init_sums'][k][0]) <= 1e-6 * max(1.0, gj['init_sums'][k][1]), k
    m.to(dtype)
    for mod in m.modules():
        if hasattr(mod, 'measure'):
            mod.measure.running_range = mod.measure.running_range.to(dtype)
            mod.measure.running_zero_point = mod.measure.running_zero_point.to(dtype)
    data = [(x.to(dtype), t) for x, t in O.synthetic_batches(gj['steps'], gj['B'], gj['size'], gj['classes'],
                                                             gj['seed'])]
    recs = O.oracle_train(m, data, wd_filter=Q.quant_weight_decay_filter)
    for r, rr in zip(recs, gj['records']):
        assert abs(r['loss'] - rr['loss']) <= ltol, (r, rr)
        assert abs(r['grad'] - rr['grad']) <= gtol * rr['grad'], (r, rr)
    if dtype == torch.double:
        assert [r['prec1'] for r in recs] == [rr['prec1'] for rr in gj['records']]
        val = O.oracle_validate(m, data[:2])
        assert abs(val['loss'] - gj['validate']['loss']) <= 1e-7
        assert val['prec5'] == gj['validate']['prec5']
        fs = gj['final_sums']
        sd = Q.state_dict_like_reference(m)
        for k, v in sd.items():
            if v.dtype.is_floating_point:
                assert abs(float(v.double().sum()) - fs[k][0]) <= 1e-7 * max(1.0, fs[k][1]), k
