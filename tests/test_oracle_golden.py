"""Pins oracle/convnet_oracle.py (the travelling CPU restatement) against golden vectors produced
by the real reference (oracle/make_golden.py -> tests/golden/)."""
import json
import os

import pytest
import torch

from helpers import GOLDEN, golden_batches, load_traj, tensor_sums
from oracle import convnet_oracle as O


def _structure():
    with open(os.path.join(GOLDEN, 'structure.json')) as f:
        return json.load(f)


@pytest.mark.parametrize('depth', [18, 34, 50, 101])
def test_structure_matches_reference(depth):
    g = _structure()['resnet%d' % depth]
    torch.manual_seed(123)
    m = O.OracleResNet(depth)
    sd = m.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == g['keys']
    assert sum(p.numel() for p in m.parameters()) == g['params']
    if 'init_sums' in g:   # seeded construction reproduces the reference's initial weights
        mine = tensor_sums(sd)
        for k, (s, a) in g['init_sums'].items():
            assert mine[k][0] == pytest.approx(s, rel=1e-9, abs=1e-9) and mine[k][1] == pytest.approx(a, rel=1e-9)


def test_mnist_structure_and_eval():
    g = _structure()['mnist']
    torch.manual_seed(123)
    m = O.OracleMnist()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == g['keys']
    assert sum(p.numel() for p in m.parameters()) == g['params'] == 131978
    fx = torch.load(os.path.join(GOLDEN, 'mnist_eval.pt'))
    m.eval()
    with torch.no_grad():
        y = m(fx['x'])
    assert torch.allclose(y, fx['logits'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('tag', ['r50s', 'r18s', 'r50s_clip'])
def test_trajectory_matches_reference_trainer(tag):
    meta, final = load_traj(tag)
    torch.manual_seed(123)
    kw = meta['model_kw']
    model = O.OracleResNet(kw['depth'], kw['num_classes'], kw['inplanes'], tuple(kw['width']))
    mine = tensor_sums(model.state_dict())
    for k, (s, a) in meta['init_sums'].items():
        assert mine[k][1] == pytest.approx(a, rel=1e-9)
    recs = O.oracle_train(model, golden_batches(meta), lr=0.1, momentum=0.9, weight_decay=1e-4,
                          loss_scale=meta['loss_scale'], grad_clip=meta['grad_clip'],
                          smooth_eps=meta['smooth_eps'], chunk_batch=meta['chunk_batch'])
    for r, g in zip(recs, meta['records']):
        assert r['loss'] == pytest.approx(g['loss'], rel=2e-5, abs=2e-5)
        assert r['prec1'] == g['prec1'] and r['prec5'] == g['prec5']
        assert r['grad'] == pytest.approx(g['grad'], rel=1e-4)
    sd = model.state_dict()
    for k, v in final.items():
        assert torch.allclose(sd[k], v, rtol=1e-4, atol=1e-6), k
    val = O.oracle_validate(model, golden_batches(meta)[:2], meta['smooth_eps'])
    assert val['loss'] == pytest.approx(meta['validate']['loss'], rel=1e-4)
    assert val['prec1'] == meta['validate']['prec1']


def test_mnist_trajectory_matches_reference_trainer():
    """BASELINE config 0: models/mnist.py + reference Trainer on CPU (Dropout active, same RNG stream)."""
    with open(os.path.join(GOLDEN, 'traj_mnist.json')) as f:
        meta = json.load(f)
    torch.manual_seed(123)
    model = O.OracleMnist()
    g = torch.Generator().manual_seed(meta['seed'])
    data = [(torch.randn(meta['B'], 1, 28, 28, generator=g), torch.randint(0, 10, (meta['B'],), generator=g))
            for _ in range(meta['steps'])]
    recs = O.oracle_train(model, data, lr=0.1, momentum=0.9, weight_decay=0, wd_filter=None)
    for r, gr in zip(recs, meta['records']):
        assert r['loss'] == pytest.approx(gr['loss'], rel=2e-5)
        assert r['prec1'] == gr['prec1'] and r['prec5'] == gr['prec5']
        assert r['grad'] == pytest.approx(gr['grad'], rel=1e-4)
