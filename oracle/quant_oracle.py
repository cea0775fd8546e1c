"""CPU oracle for BASELINE config 5 (ResNet `{'quantize': True}`) -- TEST INFRASTRUCTURE, NOT A PRODUCT PATH.

Plain-PyTorch CPU fp32 restatement of /root/reference models/modules/quantize.py (the simulated 8-bit
training operators that models/resnet.py:387-391 swaps in for Conv2d / Linear / BatchNorm2d), written as
explicit forward / backward formulas instead of the reference's autograd tricks, so that every quantity the
HIP path has to reproduce is visible.  Only tests/ may import it.

Pinned (tests/test_quant_oracle.py) against tests/golden/quant_ops.pt and traj_r{50,18}s_quant.json, which
oracle/make_golden_quant.py produced by running the reference itself with two documented repairs:
  1. UniformQuantizeGrad.forward clones (quantize.py:98 returns its input, which breaks in-place ReLU);
  2. a zero quantisation range is treated as range 1, i.e. the quantiser is the identity on a constant tensor
     (quantize.py:64-66 divides by scale = 0 otherwise: fc.bias at init, gradients behind a gamma = 0 BN).

Stochastic rounding (quantize.py:67-69): noise ~ U(-0.5, 0.5) drawn with `tensor.new(shape).uniform_()`
from torch's global CPU generator, one draw of the gradient's NCHW shape per quantize_grad backward, in the
order autograd executes them.  `noise_fn(grad)` below is that draw; tests hand the same stream to the HIP
path.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import convnet_oracle as O


def default_noise(like):
    return like.new(like.shape).uniform_(-0.5, 0.5)


def qparams_mean(x):
    """calculate_qparams(x, flatten_dims=(1,-1), reduce_dim=0, 'mean') (quantize.py:19-38): per-sample
    min / max, averaged over the batch -> (zero_point, range) scalars."""
    flat = x.flatten(1)
    mn, mx = flat.min(-1)[0].mean(), flat.max(-1)[0].mean()
    return mn, _fix_range(mx - mn)


def qparams_extreme(x):
    """... reduce_type='extreme' (quantize.py:31-33): global min / max."""
    mn, mx = x.min(), x.max()
    return mn, _fix_range(mx - mn)


def qparams_rows(w):
    """flatten_dims=(1,-1), reduce_dim=None: per output channel (quantize.py:201-202)."""
    flat = w.flatten(1)
    mn, mx = flat.min(-1)[0], flat.max(-1)[0]
    shape = [-1] + [1] * (w.dim() - 1)
    return mn.view(shape), _fix_range(mx - mn).view(shape)


def _fix_range(r):
    return torch.where(r == 0, torch.ones_like(r), r)   # repair 2


def quantize(x, zero_point, rng, num_bits=8, noise=None):
    """UniformQuantize.forward (quantize.py:41-76), unsigned, dequantised; same operation order."""
    qmax = 2. ** num_bits - 1.
    scale = rng / qmax
    out = (x + (-zero_point)) / scale
    if noise is not None:
        out = out + noise
    out = out.clamp(0., qmax).round()
    return out * scale + zero_point


class QuantMeasureState(object):
    """QuantMeasure (quantize.py:140-182): running range / zero point with momentum 0.1 applied as
    running = running * momentum + new * (1 - momentum)."""

    def __init__(self, shape, momentum=0.1):
        self.running_zero_point = torch.zeros(*shape)
        self.running_range = torch.zeros(*shape)
        self.momentum = momentum

    def __call__(self, x, training):
        if training:
            zp, rng = qparams_mean(x)
            self.running_zero_point.mul_(self.momentum).add_(zp * (1 - self.momentum))
            self.running_range.mul_(self.momentum).add_(rng * (1 - self.momentum))
        else:
            zp, rng = self.running_zero_point.reshape(()), self.running_range.reshape(())
        return quantize(x, zp, rng)


class _QConvFn(torch.autograd.Function):
    """QConv2d.forward + conv2d_biprec (quantize.py:115-121,195-219): y = conv(q(x), q(w)); the weight
    gradient sees the full-precision dy, the input gradient sees dy quantised to 8 bits with stochastic
    rounding over its global min / max; both quantisers are straight-through."""

    @staticmethod
    def forward(ctx, x, w, mod):
        qx = mod.measure(x, mod.training)
        zp, rng = qparams_rows(w)
        qw = quantize(w, zp, rng)
        ctx.save_for_backward(qx, qw)
        ctx.mod = mod
        return F.conv2d(qx, qw, None, mod.stride, mod.padding)

    @staticmethod
    def backward(ctx, dy):
        qx, qw = ctx.saved_tensors
        mod = ctx.mod
        dx = None
        if ctx.needs_input_grad[0]:   # the stem's input needs no gradient: that quantiser (and its noise draw) never runs
            zp, rng = qparams_extreme(dy)
            gq = quantize(dy, zp, rng, noise=mod.noise_fn(dy))
            dx = torch.nn.grad.conv2d_input(qx.shape, qw, gq, mod.stride, mod.padding)
        dw = torch.nn.grad.conv2d_weight(qx, qw.shape, dy, mod.stride, mod.padding)
        return dx, dw, None


class OracleQConv2d(nn.Conv2d):
    def __init__(self, cin, cout, k, stride=1, padding=0, bias=False):
        super().__init__(cin, cout, k, stride, padding, bias=bias)
        assert not bias
        self.measure = QuantMeasureState((1, 1, 1, 1))
        self.noise_fn = default_noise

    def forward(self, x):
        return _QConvFn.apply(x, self.weight, self)


class _QLinearFn(torch.autograd.Function):
    """QLinear.forward + linear_biprec (quantize.py:123-129,233-253); bias quantised to 16 bits over its
    global range."""

    @staticmethod
    def forward(ctx, x, w, b, mod):
        qx = mod.measure(x, mod.training)
        zp, rng = qparams_rows(w)
        qw = quantize(w, zp, rng)
        bz, br = qparams_extreme(b)
        qb = quantize(b, bz, br, num_bits=16)
        ctx.save_for_backward(qx, qw)
        ctx.mod = mod
        return F.linear(qx, qw, qb)

    @staticmethod
    def backward(ctx, dy):
        qx, qw = ctx.saved_tensors
        mod = ctx.mod
        zp, rng = qparams_extreme(dy)
        gq = quantize(dy, zp, rng, noise=mod.noise_fn(dy))
        return gq @ qw, dy.t() @ qx, dy.sum(0), None


class OracleQLinear(nn.Linear):
    def __init__(self, fin, fout):
        super().__init__(fin, fout)
        self.measure = QuantMeasureState((1,))
        self.noise_fn = default_noise

    def forward(self, x):
        return _QLinearFn.apply(x, self.weight, self.bias, self)


class _RangeBNFn(torch.autograd.Function):
    """RangeBN.forward (quantize.py:283-330) and the gradient autograd derives for it.

    x is first quantised (QuantMeasure, straight-through).  Training statistics per channel over the
    M = B*H*W values taken in (b, h, w) order and cut into 16 consecutive chunks: mean, and
    scale = (mean of chunk maxima - mean of chunk minima) * scale_fix.  y = (x - mean) / (scale + eps) * w + b.
    The output gradient is quantised (8 bits, stochastic, global range).  Backward: with r = 1/(scale+eps),
    S1 = sum g, S2 = sum g*(x-mean):  db = S1, dw = r*S2, dL/dscale = -w*r^2*S2,
    dx = g*w*r - w*r*S1/M  + dL/dscale * scale_fix/16 * ([first argmax of its chunk] - [first argmin])."""

    @staticmethod
    def forward(ctx, x, w, b, mod):
        x = mod.measure(x, mod.training)
        B, C, H, W = x.shape
        if mod.training:
            y = x.transpose(0, 1).reshape(C, mod.num_chunks, (B * H * W) // mod.num_chunks)
            mx, imx = y.max(-1)
            mn, imn = y.min(-1)
            mean = y.reshape(C, -1).mean(-1)
            fix = (0.5 * 0.35) * (1 + (math.pi * math.log(4)) ** 0.5) / ((2 * math.log(y.size(-1))) ** 0.5)
            scale = (mx.mean(-1) - mn.mean(-1)) * fix
            mod.running_mean.mul_(mod.momentum).add_(mean * (1 - mod.momentum))
            mod.running_var.mul_(mod.momentum).add_(scale * (1 - mod.momentum))
            ctx.idx = (imx, imn, fix)
        else:
            mean, scale = mod.running_mean, mod.running_var
        r = 1.0 / (scale + mod.eps)
        ctx.save_for_backward(x, w, mean, r)
        ctx.mod = mod
        out = (x - mean.view(1, -1, 1, 1)) / (scale.view(1, -1, 1, 1) + mod.eps)
        return out * w.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, r = ctx.saved_tensors
        mod = ctx.mod
        B, C, H, W = x.shape
        zp, rng = qparams_extreme(dy)
        g = quantize(dy, zp, rng, noise=mod.noise_fn(dy))
        xc = x - mean.view(1, -1, 1, 1)
        S1 = g.sum((0, 2, 3))
        S2 = (g * xc).sum((0, 2, 3))
        M = B * H * W
        dx = g * (w * r).view(1, -1, 1, 1) - (w * r * S1 / M).view(1, -1, 1, 1)
        imx, imn, fix = ctx.idx
        dscale = -w * r * r * S2
        L = M // mod.num_chunks
        flat = dx.transpose(0, 1).reshape(C, -1).clone()
        base = torch.arange(mod.num_chunks).view(1, -1) * L
        coef = (dscale * fix / mod.num_chunks).view(-1, 1).expand(C, mod.num_chunks)
        flat.scatter_add_(1, base + imx, coef)
        flat.scatter_add_(1, base + imn, -coef)
        dx = flat.reshape(C, B, H, W).transpose(0, 1)
        return dx, r * S2, S1, None


class OracleRangeBN(nn.Module):
    def __init__(self, c, momentum=0.1, eps=1e-5, num_chunks=16):
        super().__init__()
        self.register_buffer('running_mean', torch.zeros(c))
        self.register_buffer('running_var', torch.zeros(c))
        self.bias = nn.Parameter(torch.empty(c))
        self.weight = nn.Parameter(torch.empty(c))
        self.momentum, self.eps, self.num_chunks = momentum, eps, num_chunks
        self.measure = QuantMeasureState((1, 1, 1, 1))
        self.noise_fn = default_noise
        self.weight.data.uniform_()     # RangeBN.reset_params (quantize.py:277-281): consumes the RNG stream
        self.bias.data.zero_()

    def forward(self, x):
        return _RangeBNFn.apply(x, self.weight, self.bias, self)


class _Block(nn.Module):
    def __init__(self, kind, cin, planes, stride, expansion, downsample):
        super().__init__()
        self.kind = kind
        if kind == 'basic':
            self.conv1 = OracleQConv2d(cin, planes, 3, stride, 1)
            self.bn1 = OracleRangeBN(planes)
            self.conv2 = OracleQConv2d(planes, planes * expansion, 3, 1, 1)
            self.bn2 = OracleRangeBN(planes * expansion)
        else:
            self.conv1 = OracleQConv2d(cin, planes, 1)
            self.bn1 = OracleRangeBN(planes)
            self.conv2 = OracleQConv2d(planes, planes, 3, stride, 1)
            self.bn2 = OracleRangeBN(planes)
            self.conv3 = OracleQConv2d(planes, planes * expansion, 1)
            self.bn3 = OracleRangeBN(planes * expansion)
        self.downsample = downsample

    def last_bn(self):
        return self.bn2 if self.kind == 'basic' else self.bn3

    def forward(self, x):
        out = F.relu(self.bn1(self.conv1(x)))
        if self.kind == 'basic':
            out = self.bn2(self.conv2(out))
        else:
            out = F.relu(self.bn2(self.conv2(out)))
            out = self.bn3(self.conv3(out))
        res = x if self.downsample is None else self.downsample(x)   # after the branch, like the reference
        return F.relu(out + res)


class OracleQuantResNet(nn.Module):
    """models/resnet.py:216-244 built from the quantised operators; construction order (hence RNG
    consumption and state_dict order) follows the reference."""

    def __init__(self, depth=50, num_classes=1000, inplanes=64, width=(64, 128, 256, 512)):
        super().__init__()
        kind, layers = O.DEPTHS[depth]
        expansion = 1 if kind == 'basic' else 4
        self.conv1 = OracleQConv2d(3, inplanes, 7, 2, 3)
        self.bn1 = OracleRangeBN(inplanes)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = inplanes
        for i, n in enumerate(layers):
            stride = 1 if i == 0 else 2
            cout = width[i] * expansion
            ds = None
            if stride != 1 or cin != cout:
                ds = nn.Sequential(OracleQConv2d(cin, cout, 1, stride), OracleRangeBN(cout))
            blocks = [_Block(kind, cin, width[i], stride, expansion, ds)]
            cin = cout
            blocks += [_Block(kind, cin, width[i], 1, expansion, None) for _ in range(1, n)]
            setattr(self, 'layer%d' % (i + 1), nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = OracleQLinear(cin, num_classes)
        for m in self.modules():        # init_model (models/resnet.py:16-31) as it acts on the rebound classes
            if isinstance(m, OracleQConv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, OracleRangeBN):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        for m in self.modules():
            if isinstance(m, _Block):
                nn.init.constant_(m.last_bn().weight, 0)
        self.fc.weight.data.normal_(0, 0.01)
        self.fc.bias.data.zero_()

    def set_noise(self, fn):
        for m in self.modules():
            if hasattr(m, 'noise_fn'):
                m.noise_fn = fn

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.avgpool(x).flatten(1))


def quant_weight_decay_filter(name, module):
    """models/resnet.py:34-40 with nn.BatchNorm2d rebound to RangeBN."""
    return (not name.endswith('bias')) and (not isinstance(module, OracleRangeBN))


def state_dict_like_reference(model):
    """The oracle keeps QuantMeasure's two buffers as plain attributes; export them under the reference's
    state_dict names (`<module>.quantize_input.running_zero_point` / `.running_range`)."""
    sd = dict(model.state_dict())
    for name, m in model.named_modules():
        if hasattr(m, 'measure'):
            sd[name + '.quantize_input.running_zero_point'] = m.measure.running_zero_point
            sd[name + '.quantize_input.running_range'] = m.measure.running_range
    return sd
