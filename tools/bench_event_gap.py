"""What does a cross-stream hand-off cost the stream that records the event?  Chain of dependent small kernels on
the main stream; every iteration forks a kernel onto a side stream (as Conv2dFunction.backward does for the weight
gradient).  Variants: no fork, torch Stream.wait_stream, cn_stream_fork with / without the system-scope fence."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import convnet_amd as ca
from convnet_amd import _lib
L = _lib.load()
dev = torch.device('cuda', 0)
main = torch.cuda.Stream(dev, priority=-1)
side = torch.cuda.Stream(dev)
BIG = os.environ.get('GAP_BIG', '0') == '1'     # long kernels: the host runs far ahead of the device, as in a real step
a = torch.zeros(1 << (26 if BIG else 20), device=dev)
b = torch.zeros(1 << (25 if BIG else 19), device=dev)   # a different grid size: tells the streams apart in a kernel trace
N = 100 if BIG else 400


def fill(t, s):
    _lib.check(L.cn_fill_f32(_lib.ptr(t), t.numel(), 1.0, s.cuda_stream), 'fill')


def run(mode):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(main):
        e0.record()
        for _ in range(N):
            if mode == 'cn_alloc':       # what an autograd node does: fresh tensors, recorded on the side stream, freed
                t1 = torch.empty(1 << 22, device=dev)
                fill(t1, main)
                _lib.check(L.cn_stream_fork(main.cuda_stream, side.cuda_stream), 'fork')
                with torch.cuda.stream(side):
                    t2 = torch.empty(1 << 22, device=dev)
                    fill(b, side)
                t1.record_stream(side)
                del t1, t2
            elif mode == 'cn_mark':
                h = L.cn_stream_arm()
                fill(a, main)
                assert L.cn_stream_disarm() == 1
                _lib.check(L.cn_stream_wait_mark(h, side.cuda_stream), 'wait_mark')
                fill(b, side)
            else:
                fill(a, main)
            if mode == 'torch':
                side.wait_stream(main)
                fill(b, side)
            elif mode in ('cn', 'cn_sysfence'):
                _lib.check(L.cn_stream_fork(main.cuda_stream, side.cuda_stream), 'fork')
                fill(b, side)
            elif mode == 'cn_mark':
                pass
            elif mode == 'side_nowait':
                fill(b, side)
            fill(a, main)
        main.wait_stream(side)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N


MODES = sys.argv[1:] or None
for rep in range(2):
    for mode in MODES or ('none', 'side_nowait', 'torch', 'cn', 'cn_sysfence', 'cn_mark'):
        L.cn_set_option(b'fork_sysfence', 1 if mode == 'cn_sysfence' else 0)
        run(mode)
        print('%-12s %6.2f us per iteration (2 main kernels%s)' % (mode, run(mode), '' if mode == 'none' else ' + 1 side kernel'))
