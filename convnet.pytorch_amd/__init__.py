"""convnet.pytorch_amd -- MI355X-native (gfx950 / CDNA4) engine for the data-parallel training hot
path of eladhoffer/convNet.pytorch: trainer.Trainer.train/forward/_step over models/resnet.py.

The directory name contains a dot, so it is imported through the root-level ``convnet_amd`` loader
(``import convnet_amd``), which registers this package under that name.

    host side (Python, mirrors the reference's interface)      device side (hand-written HIP)
    ---------------------------------------------------------  --------------------------------
    models/ registry, nn.py operator modules, trainer.Trainer,  csrc/*.hip behind the C ABI of
    optim.OptimRegime, cross_entropy, meters, engine (arenas)   include/convnet_hip.h
"""
from . import _lib            # noqa: F401
from . import ops             # noqa: F401
from . import nn              # noqa: F401
from . import engine          # noqa: F401
from . import comm            # noqa: F401
from . import quant           # noqa: F401
from . import models          # noqa: F401
from .trainer import Trainer  # noqa: F401
from .optim import OptimRegime, Regime  # noqa: F401
from .cross_entropy import CrossEntropyLoss  # noqa: F401
from .meters import AverageMeter, accuracy   # noqa: F401

torch_dtypes = {  # utils.misc.torch_dtypes of the reference (main.py:18,136) + bfloat16
    'float': __import__('torch').float, 'float32': __import__('torch').float32,
    'bfloat16': __import__('torch').bfloat16, 'bf16': __import__('torch').bfloat16,
    'half': __import__('torch').float16, 'float16': __import__('torch').float16,
}
