import torch.nn as nn


def is_bn(module):
    return isinstance(module, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d))


class FilterModules(nn.Module):
    def __init__(self, source, module=None):
        super().__init__()
        self._mods = nn.ModuleList([m for m in source.modules() if module is None or module(m)])
