"""utils.mixup: only the class names are needed (trainer.py:8, models/resnet.py:10); mixup /
cutmix are outside the hot path."""
import torch.nn as nn


class MixUp(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def reset(self):
        pass

    def forward(self, x):
        return x


class CutMix(MixUp):
    pass
