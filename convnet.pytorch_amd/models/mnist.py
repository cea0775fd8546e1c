"""MNIST 5-conv model of /root/reference models/mnist.py:10-40 on the HIP operator modules
(BASELINE config 0 is the reference's own CPU run; this is the same network, registry name and
module tree - state_dict keys ``feats.N.*`` / ``classifier.*`` - for the GPU).  Training and
inference both run on the HIP path; Dropout(0.5) draws its mask from torch's CPU generator in the
reference's element order, so seeded trajectories are comparable with the reference's."""
import torch
import torch.nn as tnn

from .. import nn as cnn
from .. import ops

__all__ = ['mnist']


class MnistModel(tnn.Module):
    def __init__(self):
        super().__init__()
        self.feats = tnn.Sequential(
            cnn.Conv2d(1, 32, 5, 1, 1), cnn.MaxPool2d(2, 2), cnn.ReLU(True), cnn.BatchNorm2d(32),
            cnn.Conv2d(32, 64, 3, 1, 1), cnn.ReLU(True), cnn.BatchNorm2d(64),
            cnn.Conv2d(64, 64, 3, 1, 1), cnn.MaxPool2d(2, 2), cnn.ReLU(True), cnn.BatchNorm2d(64),
            cnn.Conv2d(64, 128, 3, 1, 1), cnn.ReLU(True), cnn.BatchNorm2d(128))
        self.feats[0].needs_dgrad = False
        self.classifier = cnn.Conv2d(128, 10, 1)
        self.avgpool = cnn.AdaptiveAvgPool2d(1)  # AvgPool2d(6, 6) on the 6x6 map == global average
        self.dropout = cnn.Dropout(0.5)

    def forward(self, inputs):
        first = self.feats[0]
        x = inputs
        if x.dim() == 4 and x.shape[1] == 1 and x.dtype == torch.float32:
            x = cnn.to_nhwc(x, first.compute_dtype, first.padded_in_channels())
        out = self.feats(x)
        out = self.dropout(out)
        # the reference applies the 1x1 classifier, then the 6x6 average; both are linear, so the
        # average is taken first (on 128 channels, a multiple of the 16-byte chunk) and the
        # classifier runs as a [B,128]x[128,10] product with fp32 output.
        out = self.avgpool(out)
        out = self.classifier(out)      # 10-way dense head on the pooled [B,1,1,128] map, fp32 logits
        return out.view(-1, 10)


def mnist(**kwargs):
    return MnistModel()
