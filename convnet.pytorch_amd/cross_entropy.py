"""CrossEntropyLoss(smooth_eps=...) as main.py:231-235 constructs it (class from the un-vendored
utils.cross_entropy; equals F.cross_entropy for smooth_eps = 0, mean reduction).  Forward and
gradient are the HIP softmax-CE kernel; when a Trainer attaches `meters` the same launch also
accumulates the loss / prec@1 / prec@5 meters on the device."""
import torch.nn as tnn

from . import ops


class CrossEntropyLoss(tnn.Module):
    def __init__(self, weight=None, ignore_index=-100, reduction='mean', smooth_eps=None, smooth_dist=None,
                 from_logits=True):
        super().__init__()
        if weight is not None or reduction != 'mean' or smooth_dist is not None or not from_logits:
            raise NotImplementedError('HIP CrossEntropyLoss: mean reduction, uniform smoothing only')
        self.smooth_eps = float(smooth_eps or 0.0)
        self.meters = None     # optional device buffer [loss*B, prec1*B, prec5*B, B] (set by Trainer)
        self.last_step = None  # [loss, prec1, prec5] of the latest batch (device)

    def to(self, *args, **kwargs):  # no parameters; accept the reference's criterion.to(device, dtype)
        return self

    def forward(self, output, target):
        return ops.SoftmaxCrossEntropyFunction.apply(output, target, self)
