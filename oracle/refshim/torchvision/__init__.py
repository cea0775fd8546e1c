"""Stub: the reference imports torchvision.transforms at module import time
(models/resnet.py:3) but the ResNet / MNIST paths never call it."""
from . import transforms  # noqa: F401
