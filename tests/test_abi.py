"""The C-ABI library loads and exports every symbol include/convnet_hip.h declares (no compute
calls, so this runs without a GPU); the product loader refuses to fall back."""
import ctypes
import os
import re

import pytest

from helpers import ROOT

HIP_LIB = os.path.join(ROOT, 'convnet.pytorch_amd', 'libconvnet_hip.so')


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'convnet_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(cn_[a-z0-9_]+)\s*\(', txt)))


def test_header_and_binding_agree():
    import convnet_amd as ca
    assert _header_symbols() == list(ca._lib.EXPORTED_SYMBOLS)


def test_hip_library_exports_every_declared_symbol():
    if not os.path.exists(HIP_LIB):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(HIP_LIB)
    for name in _header_symbols():
        assert hasattr(lib, name), name
    lib.cn_is_emulator.restype = ctypes.c_int
    assert lib.cn_is_emulator() == 0
    lib.cn_build_info.restype = ctypes.c_char_p
    assert b'gfx950' in lib.cn_build_info()


def test_hip_library_was_built_from_this_tree():
    """cn_build_info() carries the content hash of the sources + headers the binary was compiled from (csrc/build.sh);
    _lib.source_hash() is the same recipe over the tree: a stale libconvnet_hip.so - which gpurun would ship to the GPU box
    and bench.py would measure - fails here (bench.py prints both hashes in its line)."""
    import convnet_amd as ca
    lib = ctypes.CDLL(HIP_LIB)
    lib.cn_build_info.restype = ctypes.c_char_p
    info = lib.cn_build_info().decode()
    if ca._lib.source_hash() not in info:
        import __graft_entry__ as g
        g.build()          # (rebuilt: the next process loads the fresh binary; this one checks the file again)
        import subprocess
        import sys
        out = subprocess.check_output([sys.executable, '-c', 'import ctypes; l = ctypes.CDLL(%r); '
                                       'l.cn_build_info.restype = ctypes.c_char_p; print(l.cn_build_info().decode())' % HIP_LIB])
        info = out.decode()
    assert ca._lib.source_hash() in info, (ca._lib.source_hash(), info)


def test_loader_fails_loudly_without_library(monkeypatch, tmp_path):
    import convnet_amd as ca
    monkeypatch.setattr(ca._lib, '_lib', None)
    monkeypatch.setattr(ca._lib, 'HIP_LIB', str(tmp_path / 'missing.so'))
    monkeypatch.setenv('CONVNET_AMD_EMULATE', '0')
    with pytest.raises(ca._lib.ConvNetHipError):
        ca._lib.load()


def test_ops_refuse_host_tensors_on_the_product_library():
    """With the real HIP library bound, CPU tensors are rejected instead of silently computed."""
    import torch
    import convnet_amd as ca
    if ca._lib.is_emulated():
        pytest.skip('emulator bound in this process')
    with pytest.raises(ca._lib.ConvNetHipError):
        ca.ops.nchw_to_nhwc(torch.zeros(1, 3, 4, 4), torch.float32)
