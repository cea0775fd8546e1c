"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_traj(tag):
    with open(os.path.join(GOLDEN, 'traj_%s.json' % tag)) as f:
        meta = json.load(f)
    final = torch.load(os.path.join(GOLDEN, 'traj_%s_final.pt' % tag))
    return meta, final


def golden_batches(meta):
    g = torch.Generator().manual_seed(meta['seed'])
    data = [(torch.randn(meta['B'], 3, meta['size'], meta['size'], generator=g),
             torch.randint(0, meta['classes'], (meta['B'],), generator=g)) for _ in range(meta['steps'])]
    for (x, t), (sx, st) in zip(data, meta['input_sums']):   # the seeded generator reproduces the inputs
        assert abs(float(x.double().sum()) - sx) < 1e-6 * max(1.0, abs(sx)) and float(t.sum()) == st
    return data


def tensor_sums(sd):
    return {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()
            if v.dtype.is_floating_point}


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def run_engine_trajectory(meta, dtype, device, steps=None, graph=True):
    """Train our engine exactly as oracle/make_golden.py trained the reference; returns
    (per-step records, validate dict, model)."""
    import convnet_amd as ca
    torch.manual_seed(123)
    kw = dict(meta['model_kw'])
    model = ca.models.resnet(dataset='imagenet', **kw)
    crit = ca.CrossEntropyLoss(smooth_eps=meta['smooth_eps']) if meta['smooth_eps'] else ca.CrossEntropyLoss()
    opt = ca.OptimRegime(model, model.regime)
    tr = ca.Trainer(model, crit, opt, device=str(device), dtype=dtype, loss_scale=meta['loss_scale'],
                    grad_clip=meta['grad_clip'], print_freq=10 ** 9)
    if not graph:
        tr._use_graph = False     # eager launches only (tests that count Python-side fusion decisions per step)
    data = golden_batches(meta)
    if steps is not None:
        data = data[:steps]
    recs = []
    for x, t in data:
        r = tr.train([(x, t)], chunk_batch=meta['chunk_batch'])
        recs.append({k: float(r[k]) for k in ('loss', 'prec1', 'prec5', 'grad')})
    return recs, tr, model, data
