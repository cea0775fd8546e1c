"""Operator modules with the constructor signatures and ``state_dict`` layout of the torch.nn
classes the reference instantiates (/root/reference models/resnet.py:75-78,126-135,226-242;
models/mnist.py:10-33), backed by the HIP kernels instead of ATen.

The reference's own operator plug-in mechanism is rebinding ``torch.nn.Conv2d`` etc. before model
construction (models/resnet.py:387-399); here the model files construct these classes explicitly.

Internal activation layout is NHWC in the compute dtype; `to_nhwc` / `Linear` are the only places
where the reference's NCHW / [B, features] shapes appear.
"""
import math

import torch
import torch.nn as tnn
from torch.nn import init

from . import _lib, ops


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class _ArenaModule(tnn.Module):
    """Mixin: parameters live in a ParamArena once engine.prepare() ran."""

    def __init__(self):
        super().__init__()
        self._arena = None
        self._slots = {}
        self.compute_dtype = torch.float32

    def _bind_arena(self, arena, slot):
        self._arena = arena
        self._slots[slot.name.rsplit('.', 1)[-1]] = slot

    def _set_compute_dtype(self, dtype):
        self.compute_dtype = dtype

    def _require_prepared(self):
        if self._arena is None:
            raise _lib.ConvNetHipError(
                '%s used before engine.prepare(model, device, dtype): parameters are not on the device arena'
                % type(self).__name__)

    def grad_view(self, pname):
        """Flat fp32 gradient segment of parameter `pname` in kernel (KRSC) memory order."""
        s = self._slots[pname]
        return self._arena.grads[s.offset:s.offset + s.numel]

    def master_view(self, pname):
        s = self._slots[pname]
        return self._arena.params[s.offset:s.offset + s.numel]

    def _notify_grad_ready(self):
        self._arena.module_ready(self)


class Conv2d(_ArenaModule):
    """nn.Conv2d(in, out, kernel_size, stride, padding, dilation, groups, bias) - groups=1,
    dilation=1 (all that models/resnet.py and models/mnist.py use)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, padding_mode='zeros'):
        super().__init__()
        if groups != 1 or _pair(dilation) != (1, 1) or padding_mode != 'zeros':
            raise NotImplementedError('HIP Conv2d supports groups=1, dilation=1, zero padding')
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _pair(kernel_size), _pair(stride), _pair(padding)
        self.dilation, self.groups = (1, 1), 1
        self.out_f32 = False
        # same RNG consumption as torch.nn.Conv2d.reset_parameters so that seeded model
        # construction reproduces the reference's initial weights bit for bit
        self.weight = tnn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            self.bias = tnn.Parameter(torch.empty(out_channels))
            fan_in = in_channels * self.kernel_size[0] * self.kernel_size[1]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            init.uniform_(self.bias, -bound, bound)
        else:
            self.register_parameter('bias', None)
        self.w_krsc = None      # compute-dtype filter copies: views into ParamArena.wbuf
        self.w_crsk = None
        self.needs_dgrad = True

    def padded_in_channels(self):
        ch = _lib.chunk_elems(self.compute_dtype)
        return (self.in_channels + ch - 1) // ch * ch

    def ensure_prepared(self):
        """Compute-dtype filter copies (KRSC + CRSK) are refreshed for the whole model in one launch
        whenever the fp32 masters changed (engine.ParamArena.prepare_weights)."""
        self._require_prepared()
        self._arena.prepare_weights()

    def forward(self, x):
        self._require_prepared()
        if x.shape[-1] != self.padded_in_channels():
            raise _lib.ConvNetHipError('Conv2d expected %d (padded) NHWC channels, got %s'
                                       % (self.padded_in_channels(), tuple(x.shape)))
        if self.out_channels % _lib.chunk_elems(self.compute_dtype) != 0:
            # ragged width (MNIST's 10-way 1x1 classifier): only as a dense layer on a 1x1 map
            if self.kernel_size != (1, 1) or x.shape[1] != 1 or x.shape[2] != 1:
                raise NotImplementedError('Conv2d with out_channels %% chunk != 0 is only supported as a 1x1 conv '
                                          'on a 1x1 map (dense head)')
            y = ops.SmallLinearFunction.apply(x.reshape(x.shape[0], self.in_channels), self.weight, self.bias, self)
            return y.view(x.shape[0], 1, 1, self.out_channels)
        return ops.Conv2dFunction.apply(x, self.weight, self.bias, self)

    def pair_eligible(self, x_nchw):
        """The pixel-pair stem form (ops.StemPairConvFunction) applies: bf16, <= 4 input channels, stride 2,
        no bias, no input gradient needed, even padded width."""
        return (ops.STEM_PAIRS and self.compute_dtype == torch.bfloat16 and self.in_channels <= 4
                and self.bias is None and self.stride == (2, 2) and not getattr(self, 'needs_dgrad', True)
                and x_nchw.dim() == 4 and (x_nchw.shape[3] + 2 * self.padding[1]) % 2 == 0
                and x_nchw.shape[2] + 2 * self.padding[0] >= self.kernel_size[0])

    def forward_from_nchw(self, x_nchw):
        """The host->device boundary of trainer.py:116-117 fused with the first convolution: fp32 NCHW
        batch -> layout conversion in the form this conv consumes -> conv."""
        self._require_prepared()
        if self.pair_eligible(x_nchw):
            xp = ops.nchw_to_pairs(x_nchw, self.padding)
            return ops.StemPairConvFunction.apply(xp, self.weight, self)
        return self.forward(ops.nchw_to_nhwc(x_nchw, self.compute_dtype, self.padded_in_channels()))

    def extra_repr(self):
        return '{}, {}, kernel_size={}, stride={}, padding={}, bias={}'.format(
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding,
            self.bias is not None)


class Linear(_ArenaModule):
    """nn.Linear(in_features, out_features, bias): a 1x1 convolution over a 1x1 image; returns fp32
    [B, out_features] (the criterion consumes fp32 logits)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.in_channels, self.out_channels = in_features, out_features
        self.kernel_size, self.stride, self.padding = (1, 1), (1, 1), (0, 0)
        self.out_f32 = True
        self.weight = tnn.Parameter(torch.empty(out_features, in_features))
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            self.bias = tnn.Parameter(torch.empty(out_features))
            bound = 1 / math.sqrt(in_features) if in_features > 0 else 0
            init.uniform_(self.bias, -bound, bound)
        else:
            self.register_parameter('bias', None)
        self.w_krsc = None
        self.w_crsk = None

    def padded_in_channels(self):
        return self.in_features

    def ensure_prepared(self):
        self._require_prepared()
        self._arena.prepare_weights()

    def forward(self, x):
        self._require_prepared()
        B = x.shape[0]
        if self.out_features % _lib.chunk_elems(self.compute_dtype) != 0:
            # ragged width (e.g. a 3- or 10-way classifier): plain dot products from the fp32 master
            return ops.SmallLinearFunction.apply(x.reshape(B, self.in_features), self.weight, self.bias, self)
        y = ops.Conv2dFunction.apply(x.reshape(B, 1, 1, self.in_features), self.weight, self.bias, self)
        return y.view(B, self.out_features)

    def extra_repr(self):
        return 'in_features={}, out_features={}, bias={}'.format(self.in_features, self.out_features,
                                                                 self.bias is not None)


class BatchNorm2d(_ArenaModule):
    """nn.BatchNorm2d(num_features, eps, momentum, affine, track_running_stats) with optional fused
    residual add + ReLU: ``bn(y, residual=None, relu=False)``."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
        super().__init__()
        if not affine:
            raise NotImplementedError('HIP BatchNorm2d is affine')
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.affine, self.track_running_stats = affine, track_running_stats
        self.weight = tnn.Parameter(torch.ones(num_features))
        self.bias = tnn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        self._host_batches = 0  # host mirror of num_batches_tracked (momentum=None averaging)
        self.sync_group = None  # set by convert_sync_batchnorm: batch statistics over all ranks

    def reset_running_stats(self):
        self.running_mean.zero_()
        self.running_var.fill_(1)
        self.num_batches_tracked.zero_()
        self._host_batches = 0

    def effective_momentum(self):
        if self.momentum is None:  # cumulative moving average (trainer.calibrate_bn)
            self._host_batches += 1
            return 1.0 / float(self._host_batches)
        return float(self.momentum)

    def _stats_span_ranks(self):
        """True when this BatchNorm's batch statistics are all-reduced over >= 2 ranks (--sync-bn)."""
        return ops._sync_group(self) is not None      # ((group, world) only for an initialised group of > 1 rank)

    def forward(self, y, residual=None, relu=False, defer_apply=False, residual_bn=None):
        """defer_apply / residual_bn (ops.DUAL_BN): a projection shortcut's BatchNorm called with defer_apply=True only
        finalises its batch statistics and returns its input; the junction BatchNorm, given that tensor as `residual`
        and the shortcut BatchNorm as `residual_bn`, applies both in its one apply pass.  Ignored in eval mode."""
        self._require_prepared()
        if self.training or not self.track_running_stats:
            if y.dim() == 4 and y.shape[0] * y.shape[1] * y.shape[2] <= 1 and not self._stats_span_ranks():
                # torch.nn.functional.batch_norm's check, keyed like torch's bn_training (training OR no running
                # statistics to fall back on); the reference stops here too: a batch of one image on a 1 x 1 map has no
                # variance.  SyncBatchNorm over >= 2 ranks has one value per channel PER RANK: torch accepts that, so does
                # this.  The size is printed the way the reference sees the tensor, NCHW
                raise ValueError('Expected more than 1 value per channel when training, got input size {}'.format(
                    torch.Size([y.shape[0], y.shape[3], y.shape[1], y.shape[2]])))
            if torch.is_grad_enabled():
                return ops.BatchNormActFunction.apply(y, self.weight, self.bias, residual, self, relu, defer_apply,
                                                      residual_bn)
            with torch.no_grad():
                return ops.BatchNormActFunction.apply(y, self.weight, self.bias, residual, self, relu, defer_apply,
                                                      residual_bn)
        return ops.batch_norm_infer(y, residual, self, relu)

    def extra_repr(self):
        return '{}, eps={}, momentum={}'.format(self.num_features, self.eps, self.momentum)


class ReLU(tnn.Module):
    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace  # accepted for signature parity; the HIP op is out-of-place

    def forward(self, x):
        return ops.ReLUFunction.apply(x)


class MaxPool2d(tnn.Module):
    def __init__(self, kernel_size, stride=None, padding=0):
        super().__init__()
        self.kernel_size = kernel_size
        self.stride = stride if stride is not None else kernel_size
        self.padding = padding

    def forward(self, x):
        return ops.MaxPool2dFunction.apply(x, self.kernel_size, self.stride, self.padding)


class AdaptiveAvgPool2d(tnn.Module):
    def __init__(self, output_size):
        super().__init__()
        if output_size not in (1, (1, 1)):
            raise NotImplementedError('HIP AdaptiveAvgPool2d supports output_size=1')
        self.output_size = output_size

    def forward(self, x):
        return ops.GlobalAvgPoolFunction.apply(x)


class Dropout(tnn.Module):
    """nn.Dropout(p).  The reference's ResNet blocks only use p = 0 (identity); models/mnist.py:32
    uses p = 0.5.  The keep mask is drawn on the host from torch's global CPU generator in the
    reference's NCHW element order (`empty_like(x).bernoulli_(1 - p)`, what ATen's CPU dropout
    does), so a seeded run consumes the same random stream as the reference; the multiply (forward
    and backward) is the HIP eltwise kernel."""

    def __init__(self, p=0.5, inplace=False):
        super().__init__()
        self.p = p

    def forward(self, x):
        if self.p == 0 or not self.training:
            return x
        N, H, W, C = x.shape
        keep = torch.empty(N, C, H, W).bernoulli_(1 - self.p).div_(1 - self.p)
        mask = ops.nchw_to_nhwc(keep.to(x.device), x.dtype, C)
        return ops.DropoutFunction.apply(x, mask)


def bn_relu_maxpool(bn, pool, y):
    """pool(relu(bn(y))): one fused op in training mode (ops.BnReluMaxPoolFunction), the separate modules
    otherwise (eval mode, synchronised BatchNorm, non-square pooling arguments)."""
    simple = all(isinstance(v, int) for v in (pool.kernel_size, pool.stride, pool.padding)) \
        and pool.kernel_size <= 2 * pool.stride
    if (ops.FUSE_STEM_POOL and simple and type(bn) is BatchNorm2d and bn._arena is not None and (bn.training or not bn.track_running_stats)
            and ops._sync_group(bn) is None):
        fn = ops.BnReluMaxPoolFunction.apply
        if torch.is_grad_enabled():
            return fn(y, bn.weight, bn.bias, bn, pool.kernel_size, pool.stride, pool.padding)
        with torch.no_grad():
            return fn(y, bn.weight, bn.bias, bn, pool.kernel_size, pool.stride, pool.padding)
    return pool(bn(y, relu=True))


def convert_sync_batchnorm(model, process_group=None):
    """nn.SyncBatchNorm.convert_sync_batchnorm(model) of the reference's --sync-bn (main.py:190-191):
    every BatchNorm2d of `model` takes its training-mode batch statistics (and the matching backward
    sums) over the batches of all ranks of `process_group` (default group when None)."""
    for m in model.modules():
        if isinstance(m, BatchNorm2d):
            m.sync_group = process_group if process_group is not None else True
    return model


def fork(x, holder=None):
    """Duplicate an activation for two consumers; gradients are summed by our kernel (or, with a
    ResGradHolder, inside the conv-branch dgrad epilogue)."""
    if torch.is_grad_enabled() and x.requires_grad:
        return ops.ForkFunction.apply(x, holder)
    return x, x


def to_nhwc(x_nchw, dtype, c_pad=None):
    """The H2D boundary of trainer.py:116-117: fp32 NCHW -> NHWC compute dtype (zero-padded C)."""
    return ops.nchw_to_nhwc(x_nchw, dtype, c_pad)
