"""Generate the golden fixtures under tests/golden/ by executing the REAL reference.

Runs only in the build container (needs /root/reference, read-only).  The reference's own
``models/`` and ``trainer.Trainer`` are imported unmodified through ``oracle/refshim`` (re-stated
``utils.*`` + a torchvision stub, placed ahead of the reference on sys.path) and driven on CPU fp32
with fixed seeds.  Output: small JSON / .pt files that the CPU tests use to pin
``oracle/convnet_oracle.py`` and that the GPU tests compare the HIP path against.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py
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

import models as ref_models            # noqa: E402  (the reference's registry)
from trainer import Trainer as RefTrainer  # noqa: E402  (the reference's step engine)
from utils.optim import OptimRegime        # noqa: E402  (refshim restatement)
from utils.cross_entropy import CrossEntropyLoss  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)

SMALL = dict(width=[8, 16, 32, 64], inplanes=8, num_classes=16)


def tensor_sums(sd):
    return {k: [float(v.double().sum()), float(v.double().abs().sum())] for k, v in sd.items()}


def batches(n, B, size, classes, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(B, 3, size, size, generator=g), torch.randint(0, classes, (B,), generator=g))
            for _ in range(n)]


def structure():
    out = {}
    for name, kw in [('resnet18', dict(depth=18)), ('resnet50', dict(depth=50)), ('resnet34', dict(depth=34)),
                     ('resnet101', dict(depth=101))]:
        torch.manual_seed(123)
        m = ref_models.resnet(dataset='imagenet', **kw)
        sd = m.state_dict()
        out[name] = {'params': sum(p.numel() for p in m.parameters()),
                     'keys': {k: list(v.shape) for k, v in sd.items()},
                     'regime': [{k: v for k, v in r.items() if k != 'regularizer'} for r in m.regime]}
        if name in ('resnet18', 'resnet50'):
            out[name]['init_sums'] = tensor_sums({k: v for k, v in sd.items() if v.dtype.is_floating_point})
    torch.manual_seed(123)
    m = ref_models.mnist()
    sd = m.state_dict()
    out['mnist'] = {'params': sum(p.numel() for p in m.parameters()),
                    'keys': {k: list(v.shape) for k, v in sd.items()},
                    'init_sums': tensor_sums({k: v for k, v in sd.items() if v.dtype.is_floating_point})}
    out['registry'] = sorted(n for n in ref_models.__dict__ if n.islower() and not n.startswith('__')
                             and callable(ref_models.__dict__[n]))
    with open(os.path.join(OUT, 'structure.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('structure.json', {k: v.get('params') for k, v in out.items() if isinstance(v, dict)})


def trajectory(tag, model_kw, B, size, classes, steps, seed, loss_scale=1.0, grad_clip=1e9, chunk_batch=1,
               smooth=0.0):
    """k training steps of the reference Trainer (one-batch loaders -> per-step meter values)."""
    torch.manual_seed(123)
    model = ref_models.resnet(dataset='imagenet', **model_kw)
    init_sums = tensor_sums({k: v for k, v in model.state_dict().items() if v.dtype.is_floating_point})
    crit = CrossEntropyLoss(smooth_eps=smooth) if smooth else CrossEntropyLoss()
    opt = OptimRegime(model, model.regime)
    tr = RefTrainer(model, crit, opt, device_ids=None, device='cpu', dtype=torch.float, distributed=False,
                    loss_scale=loss_scale, grad_clip=grad_clip, print_freq=10 ** 9)
    data = batches(steps, B, size, classes, seed)
    recs = []
    for x, t in data:
        r = tr.train([(x, t)], chunk_batch=chunk_batch)
        recs.append({k: float(r[k]) for k in ('loss', 'prec1', 'prec5', 'grad')})
    val = tr.validate(data[:2])
    sd = model.state_dict()
    out = {'tag': tag, 'model_kw': model_kw, 'B': B, 'size': size, 'classes': classes, 'steps': steps,
           'seed': seed, 'loss_scale': loss_scale, 'grad_clip': grad_clip, 'chunk_batch': chunk_batch,
           'smooth_eps': smooth, 'records': recs,
           'validate': {k: float(val[k]) for k in ('loss', 'prec1', 'prec5')},
           'input_sums': [[float(x.double().sum()), float(t.sum())] for x, t in data],
           'init_sums': init_sums,
           'final_sums': tensor_sums({k: v for k, v in sd.items() if v.dtype.is_floating_point}),
           'num_batches_tracked': int(sd['bn1.num_batches_tracked'])}
    with open(os.path.join(OUT, 'traj_%s.json' % tag), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    # a few full tensors for element-wise comparison
    keep = ['conv1.weight', 'bn1.running_mean', 'bn1.running_var', 'layer1.0.conv1.weight',
            'layer2.0.downsample.0.weight', 'layer4.1.bn2.weight', 'fc.weight', 'fc.bias']
    if size > 64:   # full-size models: keep the fixture small
        keep = ['conv1.weight', 'bn1.running_mean', 'bn1.running_var', 'layer1.0.conv1.weight', 'fc.bias']
    torch.save({k: sd[k].clone() for k in keep if k in sd}, os.path.join(OUT, 'traj_%s_final.pt' % tag))
    print(tag, recs[0], recs[-1], 'val', out['validate'])


def mnist_trajectory(steps=3, B=16, seed=31):
    """BASELINE config 0 shape: models/mnist.py trained by the reference Trainer on CPU with the CLI
    default regime (main.py:243-247: SGD lr 0.1, momentum 0.9, wd 0).  Dropout(0.5) is active and
    draws from torch's global CPU generator after manual_seed(123) + model construction."""
    torch.manual_seed(123)
    model = ref_models.mnist()
    opt = OptimRegime(model, [{'epoch': 0, 'optimizer': 'SGD', 'lr': 0.1, 'momentum': 0.9, 'weight_decay': 0}])
    tr = RefTrainer(model, CrossEntropyLoss(), opt, device_ids=None, device='cpu', dtype=torch.float,
                    distributed=False, grad_clip=1e9, print_freq=10 ** 9)
    g = torch.Generator().manual_seed(seed)
    data = [(torch.randn(B, 1, 28, 28, generator=g), torch.randint(0, 10, (B,), generator=g)) for _ in range(steps)]
    recs = []
    for x, t in data:
        r = tr.train([(x, t)])
        recs.append({k: float(r[k]) for k in ('loss', 'prec1', 'prec5', 'grad')})
    val = tr.validate(data[:2])
    sd = model.state_dict()
    out = {'B': B, 'steps': steps, 'seed': seed, 'records': recs,
           'validate': {k: float(val[k]) for k in ('loss', 'prec1', 'prec5')},
           'final_sums': tensor_sums({k: v for k, v in sd.items() if v.dtype.is_floating_point})}
    with open(os.path.join(OUT, 'traj_mnist.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    torch.save({k: sd[k].clone() for k in ('feats.0.weight', 'feats.0.bias', 'feats.3.running_var',
                                           'classifier.weight', 'classifier.bias')},
               os.path.join(OUT, 'traj_mnist_final.pt'))
    print('mnist', recs[0], recs[-1], out['validate'])


def mnist_eval():
    torch.manual_seed(123)
    m = ref_models.mnist()
    m.eval()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(4, 1, 28, 28, generator=g)
    with torch.no_grad():
        y = m(x)
    torch.save({'x': x, 'logits': y}, os.path.join(OUT, 'mnist_eval.pt'))
    print('mnist_eval', y[0, :4])


def big():
    """Headline-size goldens: the reference Trainer at the batch sizes BASELINE.json
    quotes (configs[1] ResNet-18 fp32 b=256, configs[2] ResNet-50 b=256) plus ResNet-50 b=64.
    ~22 GB of activations and ~15 s per ResNet-50 b=256 step on 8 cores."""
    trajectory('r50_b64', dict(depth=50), B=64, size=224, classes=1000, steps=2, seed=23)
    trajectory('r18_b256', dict(depth=18), B=256, size=224, classes=1000, steps=2, seed=25)
    trajectory('r50_b256', dict(depth=50), B=256, size=224, classes=1000, steps=2, seed=24)


# ---------------------------------------------------------------------------------------------------------
# Warm-start goldens.  `init_model` zeroes the last BatchNorm gamma of every block
# (/root/reference/models/resnet.py:24-28), so in a cold-start trajectory every inner convolution's weight /
# data gradient is exactly zero at step 0 and ~1e-4 of its natural size at step 1: the cold goldens cannot
# see a wrong inner wgrad / dgrad.  Here every BatchNorm of the REFERENCE model gets seeded non-trivial
# gamma / beta / running statistics before training, and the per-tensor gradients of step 0 are recorded.
#
# Conditioning (tools/conditioning.py, profiles/r03_warm_fixture_conditioning.txt): with EVERY gamma drawn from
# [0.5, 1.5) the 16 residual branches run at full strength and the step-0 gradient of ResNet-50 becomes chaotic in the
# rounding sense - the reference's own fp32 run is only within 2e-2 of its float64 run per tensor, and PyTorch's own
# bf16 autocast run is 1.3 (uncorrelated, equal norm) - ReLU decisions that flip under a 2^-9 perturbation re-route the
# backward signal.  The LAST gamma of every block is therefore drawn from [0.03, 0.1): 300-1000x the cold-start values
# (every inner gradient is of natural relative size and any wrong kernel shows), fp32 within ~1e-3 of float64.
# Element-wise agreement of a bf16 run with an fp32 one is bounded by the same mechanism whatever the implementation
# (autocast: ~0.2 per tensor): the fixture records PyTorch's own bf16-autocast error per tensor as the yardstick.
WARM_SEED = 977
WARM_SAMPLE = 2048
WARM_LAST_GAMMA = (0.03, 0.1)


def warm_bn_state(model, seed=WARM_SEED, last_gamma=WARM_LAST_GAMMA):
    """Overwrite gamma / beta / running_mean / running_var of every BatchNorm (module order) from one seeded
    generator; the last BatchNorm of every residual block (bn3 of a Bottleneck, bn2 of a BasicBlock) draws gamma
    from `last_gamma`, every other one from [0.5, 1.5).  tests/helpers.py applies the same recipe to our model (same
    module names and order)."""
    g = torch.Generator().manual_seed(seed)
    names = set(n for n, _ in model.named_modules())
    with torch.no_grad():
        for name, m in model.named_modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                C = m.num_features
                last = name.endswith('.bn3') or (name.endswith('.bn2') and (name[:-4] + '.bn3') not in names)
                lo, hi = last_gamma if last else (0.5, 1.5)
                m.weight.copy_(torch.rand(C, generator=g) * (hi - lo) + lo)
                m.bias.copy_(torch.randn(C, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(C, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(C, generator=g) + 0.5)


def warm_tensor_names(depth):
    names = ['conv1.weight', 'bn1.weight', 'bn1.bias', 'fc.weight']
    third = depth >= 50
    for s in (1, 2, 3, 4):
        for blk in (0, 1):
            b = 'layer%d.%d.' % (s, blk)
            names += [b + 'conv1.weight', b + 'conv2.weight', b + 'bn1.weight', b + 'bn2.weight', b + 'bn2.bias']
            if third:
                names += [b + 'conv3.weight', b + 'bn3.weight', b + 'bn3.bias']
        if s > 1 or third:
            names += ['layer%d.0.downsample.0.weight' % s, 'layer%d.0.downsample.1.weight' % s]
    return names


def sample_index(name, n):
    """Up to WARM_SAMPLE seeded flat positions of a logical (reference-shape, row-major) tensor; the position set
    depends on the tensor's name and size only (tests/helpers.py draws the same set)."""
    if n <= WARM_SAMPLE:
        return torch.arange(n)
    gi = torch.Generator().manual_seed(sum(map(ord, name)) + n)
    return torch.randperm(n, generator=gi)[:WARM_SAMPLE].sort().values


def sample_tensor(v, name):
    """norm / sum of the full tensor + its elements at sample_index()."""
    v = v.detach().float().contiguous().flatten()
    n = v.numel()
    idx = sample_index(name, n)
    return {'norm': float(v.double().norm()), 'sum': float(v.double().sum()), 'numel': n, 'val': v[idx].clone()}


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def warm_trajectory(tag, model_kw, B, size, classes, steps, seed, dtype=torch.float):
    """dtype=torch.double: the reference itself run in float64 (the small and the ResNet-18 fixtures): the engine is
    measured against the truth instead of against another fp32 rounding pattern.
    Recorded per named tensor: the RAW autograd gradient of step 0 (tensor hooks: the reference's WeightDecay
    regulariser adds wd*p to p.grad in place before optimizer.step(), so p.grad after the step is not the gradient)
    - norm of the full tensor + WARM_SAMPLE seeded samples - and the same for the tensors after `steps` steps; plus
    `autocast_err`: the per-tensor rel-L2 of the SAME model's step-0 gradient computed under torch.autocast(bfloat16)
    (PyTorch's own bf16 arithmetic on the reference model code) against the recorded one."""
    import copy
    torch.manual_seed(123)
    model = ref_models.resnet(dataset='imagenet', **model_kw)
    warm_bn_state(model)
    data = batches(steps, B, size, classes, seed)
    names = warm_tensor_names(model_kw['depth'])
    # PyTorch's bf16 (autocast) on a float32 copy of the same start state, batch 0
    mb = copy.deepcopy(model).float()
    mb.train()
    with torch.autocast('cpu', dtype=torch.bfloat16):
        out_b = mb(data[0][0])
    torch.nn.functional.cross_entropy(out_b.float(), data[0][1]).backward()
    ac_grads = {k: p.grad.detach().clone() for k, p in mb.named_parameters() if k in names}
    del mb, out_b
    model.to(dtype)
    start_sums = tensor_sums({k: v for k, v in model.state_dict().items() if v.dtype.is_floating_point})
    opt = OptimRegime(model, model.regime)
    tr = RefTrainer(model, CrossEntropyLoss(), opt, device_ids=None, device='cpu', dtype=dtype,
                    distributed=False, grad_clip=1e9, print_freq=10 ** 9)
    params = dict(model.named_parameters())
    raw, handles = {}, []
    for k in names:
        handles.append(params[k].register_hook(lambda g, k=k: raw.__setitem__(k, g.detach().clone())))
    recs, grads0, autocast_err = [], None, None
    for i, (x, t) in enumerate(data):
        r = tr.train([(x, t)])
        recs.append({k: float(r[k]) for k in ('loss', 'prec1', 'prec5', 'grad')})
        if i == 0:
            grads0 = {k: sample_tensor(raw[k], k) for k in names}
            autocast_err = {k: [rel_l2(ac_grads[k], raw[k]), float(ac_grads[k].double().norm())] for k in names}
            for h in handles:
                h.remove()
            raw.clear()
    val = tr.validate(data[:2])
    sd = model.state_dict()
    out = {'tag': tag, 'model_kw': model_kw, 'B': B, 'size': size, 'classes': classes, 'steps': steps,
           'seed': seed, 'loss_scale': 1.0, 'grad_clip': 1e9, 'chunk_batch': 1, 'smooth_eps': 0.0,
           'warm_seed': WARM_SEED, 'warm_last_gamma': list(WARM_LAST_GAMMA),
           'reference_dtype': str(dtype).replace('torch.', ''), 'records': recs,
           'validate': {k: float(val[k]) for k in ('loss', 'prec1', 'prec5')},
           'input_sums': [[float(x.double().sum()), float(t.sum())] for x, t in data],
           'start_sums': start_sums,
           'final_sums': tensor_sums({k: v for k, v in sd.items() if v.dtype.is_floating_point}),
           'grad0_norms': {k: v['norm'] for k, v in grads0.items()},
           'autocast_err': autocast_err,
           'num_batches_tracked': int(sd['bn1.num_batches_tracked'])}
    with open(os.path.join(OUT, 'traj_%s.json' % tag), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    final = {k: sample_tensor(sd[k], k) for k in names}
    for k in ('bn1.running_mean', 'bn1.running_var', 'layer3.1.bn2.running_mean', 'layer3.1.bn2.running_var'):
        final[k] = sample_tensor(sd[k], k)
    torch.save({'grad0': grads0, 'final': final}, os.path.join(OUT, 'traj_%s_tensors.pt' % tag))
    print(tag, recs, 'val', out['validate'])
    ae = sorted(autocast_err.items(), key=lambda kv: -kv[1][0])
    print('  PyTorch bf16-autocast vs recorded step-0 gradient, rel-L2: worst', [(k, round(v[0], 3)) for k, v in ae[:3]],
          'best', [(k, round(v[0], 4)) for k, v in ae[-2:]])
    print('  smallest / largest recorded step-0 gradient norms:',
          sorted(out['grad0_norms'].items(), key=lambda kv: kv[1])[:3],
          sorted(out['grad0_norms'].items(), key=lambda kv: kv[1])[-3:])


def warm(which=('small', 'r18', 'r50')):
    if 'small' in which:
        warm_trajectory('r50s_warm', dict(depth=50, **SMALL), B=8, size=32, classes=16, steps=3, seed=41,
                        dtype=torch.double)
    if 'r18' in which:
        warm_trajectory('r18_b256_warm', dict(depth=18), B=256, size=224, classes=1000, steps=3, seed=43,
                        dtype=torch.double)
    if 'r50' in which:   # float64 at this size would need ~50 GB; at 12544..802816 values per channel fp32 is well conditioned
        warm_trajectory('r50_b256_warm', dict(depth=50), B=256, size=224, classes=1000, steps=3, seed=42)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] in ('big', 'warm'):
        big() if sys.argv[1] == 'big' else warm(tuple(sys.argv[2:]) or ('small', 'r18', 'r50'))
        assert_no_new_bytecode()
        sys.exit(0)
    structure()
    trajectory('r50s', dict(depth=50, **SMALL), B=8, size=32, classes=16, steps=4, seed=11)
    trajectory('r18s', dict(depth=18, **SMALL), B=8, size=32, classes=16, steps=4, seed=12)
    trajectory('r50s_clip', dict(depth=50, **SMALL), B=8, size=32, classes=16, steps=3, seed=13, loss_scale=8.0,
               grad_clip=0.5, chunk_batch=2, smooth=0.1)
    trajectory('r18_full', dict(depth=18), B=4, size=224, classes=1000, steps=2, seed=21)
    trajectory('r50_full', dict(depth=50), B=4, size=224, classes=1000, steps=2, seed=22)
    mnist_eval()
    mnist_trajectory()
    big()
    warm()
    assert_no_new_bytecode()
