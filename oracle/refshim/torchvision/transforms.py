"""Stub of torchvision.transforms (import-time only)."""


class _Unavailable(object):
    def __init__(self, *a, **k):
        raise RuntimeError('torchvision is not installed; transforms are outside the oracle path')


Compose = Normalize = ToTensor = Resize = CenterCrop = RandomCrop = RandomHorizontalFlip = _Unavailable
RandomResizedCrop = ColorJitter = Lambda = Pad = _Unavailable
