"""engine.DeferredWgrad (round 5): the weight gradients of the late stages are parked by backward and run - with the SGD
update of exactly those filters and the refresh of their 16-bit copies - on the side stream beside the NEXT step's forward
pass.  Same kernels, same operands, same floating-point order: a training run with the deferral is bit-identical to one
without (losses at every step, every parameter, every momentum buffer, the statistics), whatever follows a deferring step
(another one, a validation pass, a step that clips, the end of the loop).  Replaces nothing in the reference: it is the
schedule of `loss.backward()` + `optimizer.step()` of /root/reference trainer.py:162,173."""
import pytest
import torch

from helpers import warm_bn_state

KW = dict(depth=50, num_classes=16, inplanes=8, width=(8, 16, 32, 64))
MODES = ['emul', pytest.param('gpu', marks=pytest.mark.gpu)]


def _dev(mode):
    import convnet_amd as ca
    if mode == 'emul':
        if not ca._lib.is_emulated():
            pytest.skip('the HIP library is bound (GPU box): the emulator case runs in the CPU suite')
        return torch.device('cpu')
    assert not ca._lib.is_emulated()
    return torch.device('cuda', 0)


def _run(ca, dev, defer, dtype, monkeypatch, plan, B=8, size=64, clip_at=None):
    """plan: list of 'train' / 'val' entries; returns (records, state tensors, deferral counters)."""
    monkeypatch.setitem(ca.flags._VALUES, 'wgrad_defer', defer)
    torch.manual_seed(123)
    model = ca.models.resnet(**KW)
    warm_bn_state(model, 7)
    regime = [dict(r) for r in model.regime]
    for r in regime:                   # (a tiny batch on a warm model: keep the run finite)
        if 'lr' in r:
            r['lr'] = r['lr'] * 0.05
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, regime), device=str(dev), dtype=dtype,
                    print_freq=10 ** 9)
    g = torch.Generator().manual_seed(5)
    recs = []
    for i, what in enumerate(plan):
        batch = [(torch.randn(B, 3, size, size, generator=g), torch.randint(0, 16, (B,), generator=g))]
        if what == 'train2':           # two steps inside ONE loop: the flush at the start of the second step
            batch = batch + [(torch.randn(B, 3, size, size, generator=g), torch.randint(0, 16, (B,), generator=g))]
        if what == 'clip':             # a step that needs the global gradient norm cannot defer: it starts with flush + gate
            tr.grad_clip = 1e9
            r = tr.train(batch)
            tr.grad_clip = -1
        elif what == 'val':
            r = tr.validate(batch)
        else:
            r = tr.train(batch)
        recs.append((what, r['loss'], r['prec1']))
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    d = tr.arena.defer
    state = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
    state['__momentum'] = tr.optimizer.momentum_buf.detach().cpu().clone()
    state['__wbuf'] = tr.arena.wbuf.detach().float().cpu().clone()
    return recs, state, (d.flushes if d is not None else None, d.state if d is not None else None,
                         len(d.slots) if d is not None else 0)


@pytest.mark.parametrize('mode,dtype', [('emul', torch.bfloat16),
                                        pytest.param('gpu', torch.bfloat16, marks=pytest.mark.gpu),
                                        pytest.param('gpu', torch.float32, marks=pytest.mark.gpu)])
def test_deferred_weight_gradients_change_no_bit(mode, dtype, monkeypatch):
    dev = _dev(mode)
    import convnet_amd as ca
    plan = ['train2', 'train', 'val', 'train2', 'clip', 'train', 'val']
    small = dict(B=2, size=32) if mode == 'emul' else {}
    base, s0, c0 = _run(ca, dev, '', dtype, monkeypatch, plan, **small)
    defr, s1, c1 = _run(ca, dev, 'layer3+layer4', dtype, monkeypatch, plan, **small)
    assert c0 == (None, None, 0)
    # 6 deferring steps (train2 = 2, train, train2 = 2, train after clip); the clipped step does not defer
    assert c1[0] == 6 and c1[1] == 'idle' and c1[2] == 9 * 3 + 2, c1     # layer3: 6 blocks, layer4: 3 (x 3 convs) + 2 projections
    assert all(r[1] == r[1] and r[1] < 50 for r in base), base           # finite, sane
    assert base == defr, (base, defr)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


@pytest.mark.parametrize('mode', MODES)
def test_deferred_ranges_and_optimizer_split(mode, monkeypatch):
    """The deferred arena ranges are exactly the named filters, the optimizer's two calls cover the arena exactly once."""
    dev = _dev(mode)
    import convnet_amd as ca
    from convnet_amd import optim
    monkeypatch.setitem(ca.flags._VALUES, 'wgrad_defer', 'layer4')
    torch.manual_seed(123)
    model = ca.models.resnet(**KW)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=str(dev),
                    dtype=torch.bfloat16, print_freq=10 ** 9)
    d = tr.arena.defer
    names = sorted(s.name for s in d.slots)
    assert all(n.startswith('layer4.') and n.endswith('.weight') and ('conv' in n or 'downsample.0' in n) for n in names)
    assert len(names) == 3 * 3 + 1
    cover = sorted(list(d.ranges) + d.complement())
    assert cover[0][0] == 0 and cover[-1][1] == tr.arena.total
    assert all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    tr.optimizer._bind()
    tr.optimizer._build_runs()
    runs = tr.optimizer._runs
    both = sorted(optim._intersect_runs(runs, d.ranges) + optim._intersect_runs(runs, optim._complement(d.ranges, tr.arena.total)))
    assert sum(b - a for a, b, _ in both) == sum(b - a for a, b, _ in runs)
    assert all(x[1] <= y[0] for x, y in zip(both, both[1:]))
