"""CPU oracle for the hot path -- TEST INFRASTRUCTURE, NOT A PRODUCT PATH.

A plain-PyTorch (CPU, fp32, NCHW) restatement of what the reference executes for
``trainer.Trainer.train/forward/_step`` over ``models/resnet.py`` / ``models/mnist.py``.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
only as the checker / reported baseline; the product package never does.

Every function cites the reference lines it restates (paths are into /root/reference, which does
not exist on the GPU box - hence this travelling restatement).  Parity pinning: the reference has
NO tests or golden vectors of its own, so this oracle is pinned against
outputs of the reference itself, executed in the build container through ``oracle/refshim`` by
``oracle/make_golden.py`` and committed under ``tests/golden/`` (state_dict structure, seeded
initial weights, per-step loss / prec@1 / prec@5 / grad-norm trajectories of the reference Trainer,
final weights).  ``tests/test_oracle_golden.py`` checks the restatement against those fixtures.

Third-party arithmetic: torch (unpinned in requirements.txt:1-4; installed 2.10.0+rocm7.0 CPU
kernels = oneDNN / ATen native) and the un-vendored eladhoffer/utils.pytorch (branch master, no
recoverable pin) whose SGD / WeightDecay / CrossEntropy / accuracy semantics are re-stated from the
reference's call sites.
"""
import math
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------------
# model structure (models/resnet.py:81-165 blocks, :168-213 ResNet, :216-244 ImageNet variant,
# :403-419 depth table)

DEPTHS = {18: ('basic', (2, 2, 2, 2)), 34: ('basic', (3, 4, 6, 3)), 50: ('bottleneck', (3, 4, 6, 3)),
          101: ('bottleneck', (3, 4, 23, 3)), 152: ('bottleneck', (3, 8, 36, 3)),
          200: ('bottleneck', (3, 24, 36, 3))}


class OracleBasic(nn.Module):
    def __init__(self, cin, planes, stride, expansion, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes * expansion, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes * expansion)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(out + res)


class OracleBottleneck(nn.Module):
    def __init__(self, cin, planes, stride, expansion, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)   # stride on the 3x3 (:129)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        res = x if self.downsample is None else self.downsample(x)
        return self.relu(out + res)


class OracleResNet(nn.Module):
    def __init__(self, depth=50, num_classes=1000, inplanes=64, width=(64, 128, 256, 512)):
        super().__init__()
        kind, layers = DEPTHS[depth]
        expansion = 1 if kind == 'basic' else 4
        block = OracleBasic if kind == 'basic' else OracleBottleneck
        self.conv1 = nn.Conv2d(3, inplanes, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = inplanes
        for i, n in enumerate(layers):
            stride = 1 if i == 0 else 2
            cout = width[i] * expansion
            ds = None
            if stride != 1 or cin != cout:   # models/resnet.py:176-181
                ds = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
            blocks = [block(cin, width[i], stride, expansion, ds)]
            cin = cout
            blocks += [block(cin, width[i], 1, expansion, None) for _ in range(1, n)]
            setattr(self, 'layer%d' % (i + 1), nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(cin, num_classes)
        oracle_init_resnet(self)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.avgpool(x).flatten(1))


def oracle_init_resnet(model):
    """models/resnet.py:16-31 (fan-out normal, BN (1,0), zero last-BN gamma, fc N(0,.01))."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2. / n))
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
    for m in model.modules():
        if isinstance(m, OracleBottleneck):
            nn.init.constant_(m.bn3.weight, 0)
        elif isinstance(m, OracleBasic):
            nn.init.constant_(m.bn2.weight, 0)
    model.fc.weight.data.normal_(0, 0.01)
    model.fc.bias.data.zero_()


class OracleMnist(nn.Module):
    """models/mnist.py:10-40."""

    def __init__(self):
        super().__init__()
        self.feats = nn.Sequential(
            nn.Conv2d(1, 32, 5, 1, 1), nn.MaxPool2d(2, 2), nn.ReLU(True), nn.BatchNorm2d(32),
            nn.Conv2d(32, 64, 3, 1, 1), nn.ReLU(True), nn.BatchNorm2d(64),
            nn.Conv2d(64, 64, 3, 1, 1), nn.MaxPool2d(2, 2), nn.ReLU(True), nn.BatchNorm2d(64),
            nn.Conv2d(64, 128, 3, 1, 1), nn.ReLU(True), nn.BatchNorm2d(128))
        self.classifier = nn.Conv2d(128, 10, 1)
        self.avgpool = nn.AvgPool2d(6, 6)
        self.dropout = nn.Dropout(0.5)

    def forward(self, x):
        out = self.dropout(self.feats(x))
        return self.avgpool(self.classifier(out)).view(-1, 10)


# ------------------------------------------------------------------------------------------------
# optimizer / loss / meters (utils submodule semantics, re-stated from call sites)

def resnet_weight_decay_filter(name, module):
    """models/resnet.py:34-40: decay iff the name does not end in 'bias' and the owning module is not
    a BatchNorm2d."""
    return (not name.endswith('bias')) and (not isinstance(module, nn.BatchNorm2d))


class OracleSGD(object):
    """torch.optim.SGD(momentum) + WeightDecay regulariser applied as g += wd*p before the step
    (trainer.py:173 -> optimizer.step(); regime at models/resnet.py:250-252)."""

    def __init__(self, model, lr=0.1, momentum=0.9, weight_decay=1e-4, wd_filter=resnet_weight_decay_filter):
        self.model = model
        self.opt = torch.optim.SGD(model.parameters(), lr=lr, momentum=momentum)
        self.decayed = []
        for mname, mod in model.named_modules():
            for pname, p in mod.named_parameters(recurse=False):
                full = (mname + '.' if mname else '') + pname
                if weight_decay and (wd_filter is None or wd_filter(full, mod)):
                    self.decayed.append(p)
        self.weight_decay = weight_decay

    def zero_grad(self):
        self.opt.zero_grad()

    def set_lr(self, lr):
        for g in self.opt.param_groups:
            g['lr'] = lr

    def step(self):
        with torch.no_grad():
            for p in self.decayed:
                if p.grad is not None:
                    p.grad.add_(p, alpha=self.weight_decay)
        self.opt.step()


def oracle_cross_entropy(logits, target, smooth_eps=0.0):
    """main.py:231-235 criterion; == F.cross_entropy for smooth_eps = 0."""
    if not smooth_eps:
        return F.cross_entropy(logits, target)
    lsm = F.log_softmax(logits, dim=-1)
    nll = -lsm.gather(-1, target.unsqueeze(-1)).squeeze(-1)
    return ((1 - smooth_eps) * nll + smooth_eps * (-lsm.mean(dim=-1))).mean()


def oracle_accuracy(output, target, topk=(1, 5)):
    """utils.meters.accuracy as called at trainer.py:224: prec@k in percent."""
    maxk = max(topk)
    _, pred = output.float().topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1))
    return [correct[:k].reshape(-1).float().sum().item() * 100.0 / target.size(0) for k in topk]


def oracle_train(model, batches, lr=0.1, momentum=0.9, weight_decay=1e-4, loss_scale=1.0, grad_clip=-1.0,
                 smooth_eps=0.0, chunk_batch=1, wd_filter=resnet_weight_decay_filter, optimizer=None):
    """Trainer._step / Trainer.forward(training=True) (trainer.py:106-177,198-233) over a list of
    (inputs NCHW fp32, target int64) batches.  Returns per-step records
    {loss, prec1, prec5, grad} where grad is the total L2 norm of the (unscaled) gradient as
    clip_grad_norm_ reports it (computed always; clipping applied only when grad_clip > 0)."""
    opt = optimizer or OracleSGD(model, lr, momentum, weight_decay, wd_filter)
    model.train()
    records = []
    for inputs, target in batches:
        opt.zero_grad()
        outs, total = [], 0.0
        for xi, ti in zip(inputs.chunk(chunk_batch, 0), target.chunk(chunk_batch, 0)):
            out = model(xi)
            loss = oracle_cross_entropy(out, ti, smooth_eps)
            if chunk_batch > 1:
                loss = loss / chunk_batch
            outs.append(out.detach())
            total += float(loss)
            (loss * loss_scale).backward()
        params = [p for p in model.parameters() if p.grad is not None]
        for p in params:
            p.grad.div_(loss_scale)
        gnorm = torch.norm(torch.stack([p.grad.norm(2) for p in params]), 2).item()
        if grad_clip > 0:
            torch.nn.utils.clip_grad_norm_(params, grad_clip)
        opt.step()
        out = torch.cat(outs, 0)
        p1, p5 = oracle_accuracy(out, target, (1, 5))
        records.append({'loss': total, 'prec1': p1, 'prec5': p5, 'grad': gnorm})
    return records


@torch.no_grad()
def oracle_validate(model, batches, smooth_eps=0.0):
    """Trainer.validate (trainer.py:271-275): eval mode, no_grad, batch-size weighted meters."""
    model.eval()
    tot, l, a1, a5 = 0, 0.0, 0.0, 0.0
    for inputs, target in batches:
        out = model(inputs)
        n = inputs.size(0)
        p1, p5 = oracle_accuracy(out, target, (1, 5))
        l += float(oracle_cross_entropy(out, target, smooth_eps)) * n
        a1 += p1 * n
        a5 += p5 * n
        tot += n
    return {'loss': l / tot, 'prec1': a1 / tot, 'prec5': a5 / tot}


def synthetic_batches(n_batches, batch, size=224, classes=1000, seed=123, channels=3):
    """SURVEY.md section 8d synthetic inputs: unit-normal images, uniform integer targets."""
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(batch, channels, size, size, generator=g),
             torch.randint(0, classes, (batch,), generator=g)) for _ in range(n_batches)]


def usable_cpus():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def time_cpu_baseline(depth=50, batch=32, steps=2, warmup=1, size=224, threads=None):
    """Reported CPU baseline (kind='port'): this oracle's training step timed on the host cores."""
    threads = threads or min(usable_cpus(), 64)   # oneDNN stops scaling (and starts thrashing) far below 256
    torch.set_num_threads(threads)
    torch.manual_seed(123)
    model = OracleResNet(depth)
    batches = synthetic_batches(min(warmup + steps, 4), batch, size)     # (cycled: the timing does not depend on the data)
    opt = OracleSGD(model)
    oracle_train(model, batches[:warmup], optimizer=opt)
    per_step = []
    for i in range(steps):          # every step timed on its own: the line reports the MEDIAN and the spread
        t0 = time.perf_counter()
        oracle_train(model, [batches[(warmup + i) % len(batches)]], optimizer=opt)
        per_step.append(time.perf_counter() - t0)
    srt = sorted(per_step)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    dt = sum(per_step)
    return {'img_per_s': batch / med, 'img_per_s_mean': batch * steps / dt, 'img_per_s_min': batch / srt[-1],
            'img_per_s_max': batch / srt[0], 's_per_step': med, 's_total': dt, 'cores': torch.get_num_threads(),
            'batch': batch, 'steps': steps}
