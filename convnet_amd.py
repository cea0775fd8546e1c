"""Import shim: ``import convnet_amd`` loads the package that lives in ``convnet.pytorch_amd/``
(a directory name Python cannot import directly) and registers it as ``convnet_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'convnet.pytorch_amd')
_spec = importlib.util.spec_from_file_location('convnet_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['convnet_amd'] = _mod
_spec.loader.exec_module(_mod)
