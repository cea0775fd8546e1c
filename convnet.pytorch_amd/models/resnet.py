"""ResNet family (ImageNet variant) on the HIP operator modules.

Same factory surface, module tree / ``state_dict`` keys, initialisation and optimisation regimes
as /root/reference models/resnet.py (factory :385-431, ResNet_imagenet :216-317, blocks :81-165,
init_model :16-31, weight-decay filter :34-40), re-expressed as a declarative stage table instead
of hand-written block classes.  ``quantize=True`` (BASELINE config 5; the reference rebinds torch.nn's
Conv2d / Linear / BatchNorm2d to its simulated-8-bit classes, :387-391) builds the same tree from
``convnet.pytorch_amd.quant``'s QConv2d / QLinear / RangeBN.  Out of scope here (not reachable from the
BASELINE configs): ResNet_cifar, resnet_se, mixed-size "sampled" regimes, checkpoint_segments, bn_norm.

Module construction order and registration order deliberately match the reference, so seeding
torch's RNG and building ``resnet(depth=50)`` yields bit-identical initial weights.
"""
import math

import torch
import torch.nn as tnn

from .. import nn as cnn

__all__ = ['resnet']

# depth -> (block kind, blocks per stage)      (reference models/resnet.py:403-419)
_IMAGENET_DEPTHS = {
    18: ('basic', (2, 2, 2, 2)),
    34: ('basic', (3, 4, 6, 3)),
    50: ('bottleneck', (3, 4, 6, 3)),
    101: ('bottleneck', (3, 4, 23, 3)),
    152: ('bottleneck', (3, 8, 36, 3)),
    200: ('bottleneck', (3, 24, 36, 3)),
}

# conv plan of a residual branch: (kernel, width multiplier key, takes the block stride?)
_BRANCH = {
    'basic': ((3, 'planes', True), (3, 'out', False)),
    'bottleneck': ((1, 'planes', False), (3, 'planes', True), (1, 'out', False)),
}


def weight_decay_config(value=1e-4, log=False):
    """Regulariser spec of the reference regime (models/resnet.py:34-40): decay every parameter
    whose name does not end in 'bias' and whose module is not a BatchNorm2d."""
    return {'name': 'WeightDecay', 'value': value, 'log': log,
            'filter': {'parameter_name': lambda n: not n.endswith('bias'),
                       'module': lambda m: not isinstance(m, cnn.BatchNorm2d)}}


def linear_scale(lr0, lrT, T, t0=0):
    rate = (lrT - lr0) / T
    return "lambda t: {'lr': max(%s + (t - %s) * %s, 0)}" % (lr0, t0, rate)


class ResidualBlock(tnn.Module):
    """BasicBlock / Bottleneck of the reference (models/resnet.py:81-165) built from a branch plan.
    The last BN of the branch fuses `+ residual` and the final ReLU; inner BNs fuse their ReLU."""

    def __init__(self, kind, inplanes, planes, stride, expansion, downsample, op_classes=None):
        super().__init__()
        self.kind = kind
        self.quantized = op_classes is not None
        Conv, Norm = op_classes[:2] if self.quantized else (cnn.Conv2d, cnn.BatchNorm2d)
        widths = {'planes': planes, 'out': planes * expansion}
        cin = inplanes
        self.n_convs = len(_BRANCH[kind])
        for i, (k, wkey, strided) in enumerate(_BRANCH[kind], start=1):
            cout = widths[wkey]
            setattr(self, 'conv%d' % i, Conv(cin, cout, kernel_size=k, stride=stride if strided else 1,
                                             padding=k // 2, bias=False))
            setattr(self, 'bn%d' % i, Norm(cout))
            if not self.quantized:
                getattr(self, 'conv%d' % i).feeds_batchnorm = True   # BN statistics come out of the conv epilogue
                getattr(self, 'conv%d' % i).__dict__['stats_bn'] = getattr(self, 'bn%d' % i)   # ... centred on its running mean
                getattr(self, 'bn%d' % i).__dict__['producer_conv'] = getattr(self, 'conv%d' % i)   # (lazy dy: ops.LAZY_DY)
            if i > 1 and not self.quantized:   # this conv reads relu(bn_{i-1}(.)): its dgrad epilogue does that BN's backward reduction
                # (instance dict, not setattr: the BN must not become a registered sub-module of the conv)
                getattr(self, 'conv%d' % i).__dict__['input_bn'] = getattr(self, 'bn%d' % (i - 1))
            if i == 1 and kind == 'basic':
                self.relu = cnn.ReLU(inplace=True)  # registration order of the reference BasicBlock
            cin = cout
        if kind == 'bottleneck':
            self.relu = cnn.ReLU(inplace=True)
            self.dropout = cnn.Dropout(0)
        self.downsample = downsample
        if kind == 'basic':
            self.dropout = cnn.Dropout(0)
        self.stride = stride
        self.expansion = expansion
        if self.quantized:   # the reference's operator chain; what is shared / fused is listed in quant.py
            # the two gradients meeting at the block input are summed in the later data gradient's epilogue
            # (quant.JUNCTION_ADD): conv1 + the identity shortcut's gradient (parked by quant.add_relu), or conv1 + the
            # projection convolution
            from ..ops import ResGradHolder
            self._holder = ResGradHolder()
            self.conv1._res_holder = self._holder
            if downsample is not None:
                downsample[0]._res_holder = self._holder
            if downsample is not None and hasattr(downsample[0], 'quantize_input'):
                # conv1 and the projection read the same block input: one min / max + quantise pass for both
                self.conv1.__dict__['share_q_out'] = True
                downsample[0].__dict__['share_q_from'] = self.conv1
            return
        # the two gradients meeting at the block input are summed inside a dgrad epilogue
        from ..ops import ResGradHolder
        self._holder = ResGradHolder()
        self.conv1._res_holder = self._holder
        self.conv1.__dict__['junction_conv1'] = True   # (never a lazy-dy consumer: ops._lazy_dy_ok)
        if downsample is None:
            self.last_bn()._res_holder = self._holder     # identity: last BN's dres + conv1 dgrad
        else:
            downsample[0]._res_holder = self._holder      # downsample conv dgrad + conv1 dgrad
            downsample[0].feeds_batchnorm = True
            downsample[0].__dict__['stats_bn'] = downsample[1]
            downsample[1].__dict__['producer_conv'] = downsample[0]

    def last_bn(self):
        return getattr(self, 'bn%d' % self.n_convs)

    def link_inner_consumers(self):
        """bn_i -> relu -> conv_{i+1} run back to back in forward(): a 1x1 convolution on the streaming kernel / a 3x3
        convolution on the 64-channel halo kernel applies that BatchNorm on its operand path (ops.LAZY_A)."""
        if not self.quantized:
            for i in range(1, self.n_convs):
                getattr(self, 'bn%d' % i).__dict__['inner_consumer_conv'] = getattr(self, 'conv%d' % (i + 1))

    def set_input_bn(self, bn):
        """The block input is the output of `bn` (the previous block's last BN, ReLU and residual fused):
        whichever of conv1 / downsample conv completes the input gradient reduces it for that BN."""
        if self.quantized:
            return
        self.conv1.__dict__['input_bn'] = bn
        bn.__dict__['consumer_conv'] = self.conv1   # (lazy z: conv1 can apply that junction on its operand load, ops.LAZY_Z)
        if self.downsample is not None:
            self.downsample[0].__dict__['input_bn'] = bn

    def forward(self, x):
        xa, xb = cnn.fork(x, self._holder)
        out = xa
        for i in range(1, self.n_convs):
            out = getattr(self, 'conv%d' % i)(out)
            out = getattr(self, 'bn%d' % i)(out, relu=True)
        out = getattr(self, 'conv%d' % self.n_convs)(out)
        residual = xb
        if self.quantized:   # bn -> (downsample) -> add -> relu as separate operators, in the reference's order
            out = self.last_bn()(out)
            if self.downsample is not None:
                residual = self.downsample[1](self.downsample[0](xb))
            from ..quant import add_relu
            return add_relu(out, residual, self._holder if self.downsample is None else None)
        if self.downsample is not None:
            ds_conv, ds_bn = self.downsample[0], self.downsample[1]
            from .. import ops
            if ops.DUAL_BN and ds_bn.training and self.last_bn().training and isinstance(ds_bn, cnn.BatchNorm2d) \
                    and ops._sync_group(ds_bn) is None and ops._sync_group(self.last_bn()) is None:
                # the shortcut BatchNorm finalises its statistics only; the junction's apply pass applies both
                residual = ds_bn(ds_conv(xb), defer_apply=True)
                return self.last_bn()(out, residual=residual, relu=True, residual_bn=ds_bn)
            residual = ds_bn(ds_conv(xb))
        return self.last_bn()(out, residual=residual, relu=True)


def init_model(model):
    """models/resnet.py:16-31: fan-out normal for convs, BN (1, 0), zero gamma on the last BN of every
    residual block, N(0, 0.01) classifier with zero bias - same draw order as the reference."""
    for m in model.modules():
        if isinstance(m, cnn.Conv2d):
            n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
            m.weight.data.normal_(0, math.sqrt(2. / n))
        elif isinstance(m, cnn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()
    for m in model.modules():
        if isinstance(m, ResidualBlock):
            tnn.init.constant_(m.last_bn().weight, 0)
    model.fc.weight.data.normal_(0, 0.01)
    model.fc.bias.data.zero_()


class ResNetImagenet(tnn.Module):
    num_train_images = 1281167

    def __init__(self, num_classes=1000, inplanes=64, block='bottleneck', layers=(3, 4, 23, 3),
                 width=(64, 128, 256, 512), expansion=4, regime='normal', scale_lr=1, ramp_up_lr=True,
                 ramp_up_epochs=5, epochs=90, base_devices=4, base_device_batch=64, quantize=False):
        super().__init__()
        self.inplanes = inplanes
        self.op_classes = None
        if quantize:
            from .. import quant
            self.op_classes = (quant.QConv2d, quant.RangeBN, quant.QLinear)
        Conv, Norm, Dense = self.op_classes or (cnn.Conv2d, cnn.BatchNorm2d, cnn.Linear)
        self.conv1 = Conv(3, inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.conv1.needs_dgrad = False  # network input needs no gradient
        self.conv1.feeds_batchnorm = not quantize
        self.bn1 = Norm(inplanes)
        if not quantize:
            self.conv1.__dict__['stats_bn'] = self.bn1   # the stem's statistics partials are centred on bn1.running_mean
        self.relu = cnn.ReLU(inplace=True)
        self.maxpool = cnn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i, nblocks in enumerate(layers):
            setattr(self, 'layer%d' % (i + 1),
                    self._make_layer(block, width[i], nblocks, expansion, stride=1 if i == 0 else 2))
        prev = None
        for m in self.modules():   # registration order = execution order of the residual blocks
            if isinstance(m, ResidualBlock):
                if prev is not None:
                    m.set_input_bn(prev.last_bn())
                m.link_inner_consumers()
                prev = m
        self.avgpool = cnn.AdaptiveAvgPool2d(1)
        self.fc = Dense(width[-1] * expansion, num_classes)
        init_model(self)

        batch_size = base_devices * base_device_batch
        num_steps_epoch = math.floor(self.num_train_images / batch_size)
        ramp_up_steps = num_steps_epoch * ramp_up_epochs
        self.regime = [
            {'epoch': 0, 'optimizer': 'SGD', 'lr': scale_lr * 1e-1, 'momentum': 0.9,
             'regularizer': weight_decay_config(1e-4)},
            {'epoch': 30, 'lr': scale_lr * 1e-2},
            {'epoch': 60, 'lr': scale_lr * 1e-3},
            {'epoch': 80, 'lr': scale_lr * 1e-4},
        ]
        if 'cutmix' in regime:
            self.regime = [
                {'epoch': 0, 'optimizer': 'SGD', 'lr': scale_lr * 1e-1, 'momentum': 0.9,
                 'regularizer': weight_decay_config(1e-4)},
                {'epoch': 75, 'lr': scale_lr * 1e-2},
                {'epoch': 150, 'lr': scale_lr * 1e-3},
                {'epoch': 225, 'lr': scale_lr * 1e-4},
            ]
        if 'linear' in regime:
            self.regime = [
                {'epoch': 0, 'optimizer': 'SGD', 'lr': scale_lr * 1e-1, 'momentum': 0.9,
                 'regularizer': weight_decay_config(1e-4),
                 'step_lambda': linear_scale(scale_lr * 1e-1, 0, num_steps_epoch * epochs)},
            ]
            if ramp_up_lr:
                # the reference (models/resnet.py:269-276) would start this regime at lr 0 and append a ramp-up
                # phase, but crashes on `self.regime['step_lambda']` (a list indexed by a string); here the
                # linear decay runs without warm-up, and says so
                import logging
                logging.warning("resnet(regime='linear'): ramp_up_lr is ignored (linear decay from %g without "
                                "warm-up; the reference raises a TypeError for this combination)", scale_lr * 1e-1)
            ramp_up_lr = False
        if ramp_up_lr and scale_lr > 1:  # learning-rate ramp-up (models/resnet.py:313-317)
            self.regime[0]['step_lambda'] = linear_scale(0.1, 0.1 * scale_lr, ramp_up_steps)
            self.regime.insert(1, {'epoch': ramp_up_epochs, 'lr': scale_lr * 1e-1})

    def _make_layer(self, kind, planes, blocks, expansion, stride):
        out_planes = planes * expansion
        downsample = None
        if stride != 1 or self.inplanes != out_planes:  # models/resnet.py:176-181
            Conv, Norm = (self.op_classes or (cnn.Conv2d, cnn.BatchNorm2d))[:2]
            downsample = tnn.Sequential(
                Conv(self.inplanes, out_planes, kernel_size=1, stride=stride, bias=False),
                Norm(out_planes))
        stage = [ResidualBlock(kind, self.inplanes, planes, stride, expansion, downsample, self.op_classes)]
        self.inplanes = out_planes
        for _ in range(1, blocks):
            stage.append(ResidualBlock(kind, self.inplanes, planes, 1, expansion, None, self.op_classes))
        return tnn.Sequential(*stage)

    def features(self, x):
        """x: fp32 NCHW (the loader layout) or an already-converted NHWC compute tensor."""
        if self.op_classes is not None and self.training:
            from .. import quant
            quant.tick(x.device)    # fresh stochastic-rounding noise for this step's gradient quantisers
        if x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3 and x.shape[-1] != self.conv1.padded_in_channels():
            x = self.conv1.forward_from_nchw(x)
        else:
            x = self.conv1(x)
        x = cnn.bn_relu_maxpool(self.bn1, self.maxpool, x)
        from .. import ops
        ops.LAZY_Z_SCOPE[0] += 1    # block b's junction may be left to block b+1's conv1: it runs next, by construction
        try:
            x = self.layer1(x)
            x = self.layer2(x)
            x = self.layer3(x)
            x = self.layer4(x)
        finally:
            ops.LAZY_Z_SCOPE[0] -= 1
        x = self.avgpool(x)
        return x.view(x.size(0), -1)

    def forward(self, x):
        return self.fc(self.features(x))


def resnet(**config):
    """Factory with the reference's call shape: resnet(dataset=..., depth=..., **kw)."""
    dataset = config.pop('dataset', 'imagenet')
    if config.pop('bn_norm', None):
        raise NotImplementedError("resnet(bn_norm=...) is not part of the MI355X hot path")
    config['quantize'] = bool(config.pop('quantize', False))
    if 'imagenet' not in dataset:
        raise NotImplementedError("only the ImageNet ResNet variant is built natively (dataset=%r)" % dataset)
    config.setdefault('num_classes', 1000)
    depth = config.pop('depth', 50)
    if depth not in _IMAGENET_DEPTHS:
        raise ValueError('unsupported ResNet depth %r' % depth)
    kind, layers = _IMAGENET_DEPTHS[depth]
    config.update(block=kind, layers=layers)
    if kind == 'basic':
        config['expansion'] = 1
    return ResNetImagenet(**config)
