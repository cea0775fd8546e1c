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


WARM_SAMPLE = 2048     # == oracle/make_golden.py


def warm_bn_state(model, seed, bn_type=None, last_gamma=(0.03, 0.1)):
    """The recipe of oracle/make_golden.py:warm_bn_state applied to OUR model (same module names and order):
    seeded non-trivial gamma / beta / running statistics for every BatchNorm, so that the inner blocks' weight
    and data gradients are non-zero from step 0 (init_model zeroes the last gamma of every block).  The last
    BatchNorm of every block draws gamma from `last_gamma` (conditioning: see the generator), the others from
    [0.5, 1.5)."""
    if bn_type is None:
        import convnet_amd as ca
        bn_type = ca.nn.BatchNorm2d
    g = torch.Generator().manual_seed(seed)
    names = set(n for n, _ in model.named_modules())
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, bn_type):
                C = m.num_features
                last = name.endswith('.bn3') or (name.endswith('.bn2') and (name[:-4] + '.bn3') not in names)
                lo, hi = last_gamma if last else (0.5, 1.5)
                m.weight.copy_(torch.rand(C, generator=g) * (hi - lo) + lo)
                m.bias.copy_(torch.randn(C, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(C, generator=g) + 0.5)


def sample_index(name, n):
    if n <= WARM_SAMPLE:
        return torch.arange(n)
    gi = torch.Generator().manual_seed(sum(map(ord, name)) + n)
    return torch.randperm(n, generator=gi)[:WARM_SAMPLE].sort().values


def sample_tensor(v, name):
    """(norm, sampled values) of a tensor in its logical (reference-shape, row-major) element order."""
    v = v.detach().float().cpu().contiguous().flatten()
    return float(v.double().norm()), v[sample_index(name, v.numel())]


def load_warm(tag):
    with open(os.path.join(GOLDEN, 'traj_%s.json' % tag)) as f:
        meta = json.load(f)
    return meta, torch.load(os.path.join(GOLDEN, 'traj_%s_tensors.pt' % tag))


def run_engine_trajectory(meta, dtype, device, steps=None, graph=True, grads_after_step0=None):
    """Train our engine exactly as oracle/make_golden.py trained the reference; returns
    (per-step records, validate dict, model).  `meta['warm_seed']` (warm-start goldens): the BatchNorm state is
    overwritten by the seeded recipe first.  grads_after_step0: dict filled with {name: (norm, sample)} of the
    named parameters' gradients after the first step."""
    import convnet_amd as ca
    torch.manual_seed(123)
    kw = dict(meta['model_kw'])
    model = ca.models.resnet(dataset='imagenet', **kw)
    if meta.get('warm_seed') is not None:
        warm_bn_state(model, meta['warm_seed'], last_gamma=tuple(meta['warm_last_gamma']))
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
    for i, (x, t) in enumerate(data):
        r = tr.train([(x, t)], chunk_batch=meta['chunk_batch'])
        recs.append({k: float(r[k]) for k in ('loss', 'prec1', 'prec5', 'grad')})
        if i == 0 and grads_after_step0 is not None:   # zero_grad runs at the START of a step: these are step 0's
            params = dict(model.named_parameters())
            for k in list(grads_after_step0):
                grads_after_step0[k] = sample_tensor(params[k].grad, k)
    return recs, tr, model, data


# ---- the 2-rank data-parallel workers and their oracle-side checks (tests/test_cli_and_dp.py: gloo, CPU emulator or
# one shared GPU; tests/test_rccl_gpu.py: direct RCCL, one GPU per rank)
DP_WORKER = r'''
import os, sys, json, torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import convnet_amd as ca
DEV = %(dev)r
rank = int(os.environ['RANK'])
if DEV != 'cpu':
    DEV = DEV %% rank if '%%d' in DEV else DEV        # 'cuda:%%d': one device per rank (direct RCCL); 'cuda:0': shared
    torch.cuda.set_device(torch.device(DEV))
    assert not ca._lib.is_emulated()
dist.init_process_group(%(backend)r, init_method='env://')
torch.manual_seed(123 + 7 * rank)          # different initial weights per rank: the broadcast must fix that
model = ca.models.resnet(depth=18, width=(8, 16, 32, 64), inplanes=8, num_classes=16)
tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=DEV,
                dtype=torch.float32, distributed=True, local_rank=rank, grad_clip=1e9, print_freq=10**9,
                bucket_mb=0.05)
g = torch.Generator().manual_seed(77)
data = [(torch.randn(8, 3, 32, 32, generator=g), torch.randint(0, 16, (8,), generator=g)) for _ in range(2)]
recs = []
for x, t in data:
    r = tr.train([(x[rank * 4:(rank + 1) * 4], t[rank * 4:(rank + 1) * 4])])
    recs.append({k: float(r[k]) for k in ('loss', 'grad')})
sd = {k: v.float().cpu() for k, v in model.state_dict().items() if v.dtype.is_floating_point}
torch.save({'recs': recs, 'sd': sd, 'nbuckets': len(tr.arena.buckets), 'transport': tr.reducer.describe()}, %(out)r %% rank)
ca.comm.destroy_default()
dist.destroy_process_group()
'''


def check_dp_against_oracle(outs):
    """Oracle-side DDP semantics (/root/reference/trainer.py:79-82): rank-0 initial weights everywhere, per-rank BN
    statistics, gradients averaged over the 2 ranks, one SGD step per iteration."""
    import pytest
    from oracle import convnet_oracle as O
    torch.manual_seed(123)
    replicas = [O.OracleResNet(18, 16, 8, (8, 16, 32, 64)) for _ in range(2)]
    replicas[1].load_state_dict(replicas[0].state_dict())
    opts = [O.OracleSGD(m) for m in replicas]
    g = torch.Generator().manual_seed(77)
    data = [(torch.randn(8, 3, 32, 32, generator=g), torch.randint(0, 16, (8,), generator=g)) for _ in range(2)]
    for step, (x, t) in enumerate(data):
        losses = []
        for r, m in enumerate(replicas):
            m.train()
            opts[r].zero_grad()
            loss = O.oracle_cross_entropy(m(x[r * 4:(r + 1) * 4]), t[r * 4:(r + 1) * 4])
            loss.backward()
            losses.append(float(loss))
        for p0, p1 in zip(replicas[0].parameters(), replicas[1].parameters()):
            avg = (p0.grad + p1.grad) / 2
            p0.grad.copy_(avg)
            p1.grad.copy_(avg)
        gnorm = torch.norm(torch.stack([p.grad.norm(2) for p in replicas[0].parameters()]), 2).item()
        for o in opts:
            o.step()
        for r in range(2):
            assert outs[r]['recs'][step]['loss'] == pytest.approx(losses[r], abs=1e-4)
            assert outs[r]['recs'][step]['grad'] == pytest.approx(gnorm, rel=1e-3)
    ref_sd = replicas[0].state_dict()
    for k in ('conv1.weight', 'layer2.0.downsample.0.weight', 'fc.weight', 'layer4.1.bn2.bias'):
        assert rel_l2(outs[0]['sd'][k], ref_sd[k]) < 1e-4, k


def check_syncbn_against_oracle(outs):
    """--sync-bn: 2 ranks x 4 samples with synchronised statistics == one process on the 8-sample batch."""
    import pytest
    from oracle import convnet_oracle as O
    torch.manual_seed(123)
    model = O.OracleResNet(18, 16, 8, (8, 16, 32, 64))
    opt = O.OracleSGD(model)
    g = torch.Generator().manual_seed(77)
    data = [(torch.randn(8, 3, 32, 32, generator=g), torch.randint(0, 16, (8,), generator=g)) for _ in range(2)]
    for step, (x, t) in enumerate(data):
        model.train()
        opt.zero_grad()
        loss = O.oracle_cross_entropy(model(x), t)
        loss.backward()
        gnorm = torch.norm(torch.stack([p.grad.norm(2) for p in model.parameters()]), 2).item()
        opt.step()
        rank_mean = 0.5 * (outs[0]['recs'][step]['loss'] + outs[1]['recs'][step]['loss'])
        assert rank_mean == pytest.approx(float(loss), abs=1e-4)
        assert outs[0]['recs'][step]['grad'] == pytest.approx(gnorm, rel=1e-3)
    ref_sd = model.state_dict()
    for k in ('conv1.weight', 'layer2.0.downsample.0.weight', 'fc.weight', 'layer4.1.bn2.bias',
              'bn1.running_mean', 'layer3.0.bn1.running_var', 'layer1.1.bn2.weight'):
        assert rel_l2(outs[0]['sd'][k], ref_sd[k]) < 1e-4, k
