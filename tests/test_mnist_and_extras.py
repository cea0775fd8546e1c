"""BASELINE config 0 (models/mnist.py, b<=64, SGD) on the HIP path against the golden records of the
reference's CPU Trainer, plus Trainer.calibrate_bn (trainer.py:277-285) against the oracle."""
import json
import os

import pytest
import torch

from conftest import HAS_GPU
from helpers import GOLDEN, rel_l2

MODES = [pytest.param('emul'), pytest.param('gpu', marks=pytest.mark.gpu)]


def _dev(mode):
    if mode == 'emul' and HAS_GPU:
        pytest.skip('emulator mode is for GPU-less hosts')
    if mode == 'gpu' and not HAS_GPU:
        pytest.skip('no GPU')
    return torch.device('cuda', 0) if mode == 'gpu' else torch.device('cpu')


@pytest.mark.parametrize('mode', MODES)
def test_mnist_training_matches_reference(mode):
    dev = _dev(mode)
    import convnet_amd as ca
    with open(os.path.join(GOLDEN, 'traj_mnist.json')) as f:
        meta = json.load(f)
    final = torch.load(os.path.join(GOLDEN, 'traj_mnist_final.pt'))
    torch.manual_seed(123)
    model = ca.models.mnist()
    regime = [{'epoch': 0, 'optimizer': 'SGD', 'lr': 0.1, 'momentum': 0.9, 'weight_decay': 0}]   # main.py:243-247
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, regime), device=str(dev),
                    dtype=torch.float32, grad_clip=1e9, print_freq=10 ** 9)
    g = torch.Generator().manual_seed(meta['seed'])
    data = [(torch.randn(meta['B'], 1, 28, 28, generator=g), torch.randint(0, 10, (meta['B'],), generator=g))
            for _ in range(meta['steps'])]
    steps = meta['steps'] if mode == 'gpu' else 1
    for (x, t), gr in list(zip(data, meta['records']))[:steps]:
        r = tr.train([(x, t)])
        assert r['loss'] == pytest.approx(gr['loss'], abs=1e-4)
        assert r['prec1'] == gr['prec1'] and r['prec5'] == gr['prec5']
        assert r['grad'] == pytest.approx(gr['grad'], rel=1e-3)
    if steps == meta['steps']:
        sd = model.state_dict()
        for k, v in final.items():
            assert rel_l2(sd[k].float().cpu(), v) < 1e-4, k
        val = tr.validate(data[:2])
        assert val['loss'] == pytest.approx(meta['validate']['loss'], rel=1e-4)
        assert val['prec1'] == meta['validate']['prec1']


@pytest.mark.parametrize('mode', MODES)
def test_mnist_eval_logits_match_reference(mode):
    dev = _dev(mode)
    import convnet_amd as ca
    fx = torch.load(os.path.join(GOLDEN, 'mnist_eval.pt'))
    torch.manual_seed(123)
    model = ca.models.mnist()
    ca.engine.prepare(model, dev, torch.float32)
    model.eval()
    with torch.no_grad():
        y = model(fx['x'].to(dev))
    assert torch.allclose(y.cpu(), fx['logits'], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('mode', MODES)
def test_calibrate_bn_cumulative_statistics(mode):
    """calibrate_bn: momentum=None => running stats become the plain average over the batches."""
    dev = _dev(mode)
    import convnet_amd as ca
    from oracle import convnet_oracle as O
    kw = dict(depth=18, num_classes=16, inplanes=8, width=(8, 16, 32, 64))
    torch.manual_seed(123)
    model = ca.models.resnet(**kw)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=str(dev),
                    dtype=torch.float32, print_freq=10 ** 9)
    data = O.synthetic_batches(3, 4, size=32, classes=16, seed=9)
    tr.calibrate_bn(data)
    torch.manual_seed(123)
    ref = O.OracleResNet(**kw)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = None
            m.reset_running_stats()
    ref.train()
    with torch.no_grad():
        for x, _ in data:
            ref(x)
    sd, rsd = model.state_dict(), ref.state_dict()
    for k in ('bn1.running_mean', 'bn1.running_var', 'layer3.0.downsample.1.running_var', 'layer4.1.bn2.running_mean'):
        assert rel_l2(sd[k].float().cpu(), rsd[k]) < 1e-4, k
    assert int(sd['bn1.num_batches_tracked']) == 3


@pytest.mark.gpu
def test_resnet_learns_a_separable_task_bf16():
    """End-to-end learning sanity on the MI355X with every fusion on (beyond the few-step trajectory
    parity): ResNet-18 bf16 on 8 classes of noisy class templates must fit the training set."""
    import convnet_amd as ca
    dev = 'cuda:0'
    torch.manual_seed(123)
    model = ca.models.resnet(dataset='imagenet', depth=18, num_classes=8)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=dev,
                    dtype=torch.bfloat16, print_freq=10 ** 9)
    g = torch.Generator().manual_seed(5)
    templates = torch.randn(8, 3, 64, 64, generator=g)
    def batch():
        t = torch.randint(0, 8, (64,), generator=g)
        return templates[t] * 0.7 + torch.randn(64, 3, 64, 64, generator=g), t
    first = tr.train([batch() for _ in range(5)])
    for _ in range(5):
        last = tr.train([batch() for _ in range(10)])
    assert first['loss'] > 1.5
    assert last['loss'] < 0.35 and last['prec1'] > 90.0, (first, last)
    val = tr.validate([batch() for _ in range(4)])
    assert val['prec1'] > 85.0, val


def test_optimizer_resume_past_a_regime_boundary_keeps_weight_decay(device):
    """a checkpoint saved after the first LR boundary (epoch >= 30 of models/resnet.py:250-256)
    must resume with the cumulative regime setting - in particular the WeightDecay regulariser that only
    phase 0 names - exactly like a fresh OptimRegime started at that epoch."""
    import convnet_amd as ca
    kw = dict(depth=18, width=(8, 16, 32, 64), inplanes=8, num_classes=16)
    torch.manual_seed(1)
    model = ca.models.resnet(**kw)
    ca.engine.prepare(model, device, torch.float32)
    opt = ca.OptimRegime(model, model.regime)
    for ep in (0, 29, 30, 31):
        opt.update(ep, ep * 10)
    assert opt.hyper['lr'] == pytest.approx(0.01) and opt.regularizer_cfg
    opt._bind()
    opt.momentum_buf.fill_(0.25)
    state = opt.state_dict()

    torch.manual_seed(1)
    model2 = ca.models.resnet(**kw)
    ca.engine.prepare(model2, device, torch.float32)
    opt2 = ca.OptimRegime(model2, model2.regime)
    opt2.load_state_dict(state)
    opt2.update(31, 310)
    fresh = ca.OptimRegime(model2, model2.regime)
    fresh.update(31, 310)
    assert opt2.hyper['lr'] == pytest.approx(fresh.hyper['lr']) == pytest.approx(0.01)
    assert [r['name'] for r in opt2.regularizer_cfg] == ['WeightDecay']
    opt2._build_runs()
    wds = {wd for _, _, wd in opt2._runs}
    assert 1e-4 in wds and 0.0 in wds          # conv / fc weights decayed, BN + bias not
    assert float(opt2.momentum_buf[:8].mean()) == pytest.approx(0.25)
    # a foreign optimizer state (e.g. the reference's torch.optim dict) is refused, not silently ignored
    with pytest.raises(ca._lib.ConvNetHipError):
        opt2.load_state_dict({'state': {}, 'param_groups': []})


def test_nonfinite_running_mean_does_not_poison_the_batch_statistics(device):
    """with centred statistics the conv epilogue and the statistics pass use the BatchNorm's running mean
    as the pivot of their sums.  In the reference the batch statistics do not depend on the running buffers, so a
    non-finite running mean (bad checkpoint, diverged step) must not reach them: the kernels replace such a pivot
    by 0 (cn_pivot).  Training output / loss / gradients of a model whose running means hold inf and NaN equal the
    clean model's."""
    import convnet_amd as ca
    kw = dict(depth=18, width=(8, 16, 32, 64), inplanes=8, num_classes=16)
    g = torch.Generator().manual_seed(3)
    x, t = torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 16, (4,), generator=g)
    res = []
    for poison in (False, True):
        torch.manual_seed(123)
        model = ca.models.resnet(**kw)
        tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=str(device),
                        dtype=torch.float32, grad_clip=1e9, print_freq=10 ** 9)
        tr._use_graph = False
        if poison:
            with torch.no_grad():
                model.bn1.running_mean[0] = float('inf')
                model.bn1.running_mean[1] = float('nan')
                model.layer2[0].bn1.running_mean.fill_(float('-inf'))
                model.layer3[1].bn2.running_mean[3] = float('nan')
        r = tr.train([(x, t)])
        res.append((float(r['loss']), float(r['grad']), model.fc.weight.detach().float().cpu().clone()))
    (l0, g0, w0), (l1, g1, w1) = res
    assert l1 == l1 and g1 == g1 and torch.isfinite(w1).all()
    assert l1 == pytest.approx(l0, rel=1e-5) and g1 == pytest.approx(g0, rel=1e-4)
    assert torch.allclose(w0, w1, rtol=1e-4, atol=1e-6)


def test_reference_optimizer_state_resumes_with_its_momentum(device):
    """a checkpoint written by the REFERENCE carries torch.optim.SGD's state_dict as `optim_state_dict`
    (bare, or under 'optimizer_state').  OptimRegime.load_state_dict imports its momentum buffers (index i of
    param_groups[*]['params'] = i-th model parameter) instead of refusing the file or restarting from zero; a state
    that is neither format is still refused loudly."""
    import convnet_amd as ca
    from oracle import convnet_oracle as O
    kw = dict(depth=18, width=(8, 16, 32, 64), inplanes=8, num_classes=16)
    torch.manual_seed(123)
    ref = O.OracleResNet(18, 16, 8, (8, 16, 32, 64))
    sgd = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9)
    g = torch.Generator().manual_seed(9)
    x, t = torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 16, (4,), generator=g)
    O.oracle_cross_entropy(ref(x), t).backward()
    sgd.step()
    ref_state = sgd.state_dict()
    names = [n for n, _ in ref.named_parameters()]
    for wrap in (lambda s: s, lambda s: {'optimizer_state': s, 'regime': []}):
        torch.manual_seed(123)
        model = ca.models.resnet(**kw)
        opt = ca.OptimRegime(model, model.regime)
        ca.Trainer(model, ca.CrossEntropyLoss(), opt, device=str(device), dtype=torch.float32, print_freq=10 ** 9)
        opt.load_state_dict(wrap(ref_state))
        mine = opt.state_dict()['momentum_buffer']
        for i, n in enumerate(names):
            assert torch.equal(mine[n].float().cpu(), ref_state['state'][i]['momentum_buffer']), n
    with pytest.raises(ca._lib.ConvNetHipError):
        opt.load_state_dict({'something': 'else'})


def test_batch_of_one_on_a_1x1_map_raises_like_the_reference(device):
    """Error behaviour at the operator seam (seam 2): torch's BatchNorm2d - the class the reference model
    instantiates, models/resnet.py:128 - refuses to train on one value per channel; so does ours, with the same
    exception type and message (it used to run on and train on zeros)."""
    import torch.nn.functional as F
    import convnet_amd as ca
    with pytest.raises(ValueError) as ref:
        F.batch_norm(torch.randn(1, 64, 1, 1), torch.zeros(64), torch.ones(64), training=True)
    kw = dict(depth=18, num_classes=10, inplanes=8, width=(8, 16, 32, 64))
    torch.manual_seed(123)
    model = ca.models.resnet(**kw)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=str(device),
                    dtype=torch.float32, print_freq=10 ** 9)
    with pytest.raises(ValueError) as ours:
        tr.train([(torch.randn(1, 3, 32, 32), torch.tensor([3]))])
    assert str(ours.value) == str(ref.value)
    # ... and evaluates such a batch fine, like the reference (running statistics)
    r = tr.validate([(torch.randn(1, 3, 32, 32), torch.tensor([3]))])
    assert r['loss'] == r['loss']
