"""utils.cross_entropy.CrossEntropyLoss(**{'smooth_eps': e}) as main.py:231-235 builds it:
F.cross_entropy with optional uniform label smoothing, mean reduction."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def cross_entropy(inputs, target, smooth_eps=None):
    smooth_eps = smooth_eps or 0
    if smooth_eps == 0:
        return F.cross_entropy(inputs, target)
    lsm = F.log_softmax(inputs, dim=-1)
    nll = -lsm.gather(-1, target.unsqueeze(-1)).squeeze(-1)
    smooth = -lsm.mean(dim=-1)
    return ((1 - smooth_eps) * nll + smooth_eps * smooth).mean()


class CrossEntropyLoss(nn.CrossEntropyLoss):
    def __init__(self, smooth_eps=None, **kw):
        super().__init__(**kw)
        self.smooth_eps = smooth_eps

    def forward(self, input, target):
        return cross_entropy(input, target, self.smooth_eps)
