#!/usr/bin/env python
"""graph = auto under load (GPU): 300 ResNet-50 b=256 steps in one loop (the eager verdict must hold: img/s, the watch's
reference and recent step periods), then 40 steps behind a loader that sleeps 40 ms per batch (the loader wait is
subtracted from the step period - no capture may happen).  Prints the trainer's DEBUG log."""
import sys, logging, time, torch
sys.path.insert(0,'.')
import convnet_amd as ca
logging.basicConfig(level=logging.DEBUG, stream=sys.stdout, format='%(message)s')
torch.manual_seed(123)
model = ca.models.resnet(dataset='imagenet', depth=50)
tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device='cuda:0', dtype=torch.bfloat16, print_freq=10**9)
g = torch.Generator().manual_seed(1)
pool = [(torch.randn(256,3,224,224,generator=g).cuda(), torch.randint(0,1000,(256,),generator=g).cuda()) for _ in range(4)]
tr.train([pool[i%4] for i in range(10)])
torch.cuda.synchronize(); t0=time.perf_counter()
r = tr.train([pool[i%4] for i in range(300)])
torch.cuda.synchronize(); dt=time.perf_counter()-t0
print('300 steps', dt, 'img/s', 300*256/dt, 'loss', r['loss'])
print('eager_for', len(tr._graph_eager_for), 'watch', {k[0]: (w.ref_ms, w.recent_ms(), len(w.periods)) for k,w in tr._watch.items()}, 'graphs', [g['graph'] is not None for g in tr._gstates.values()])
# a loader-bound loop: a slow iterable (sleep 40 ms before every batch) must not trigger a capture
class Slow:
    def __init__(s,n): s.n=n
    def __len__(s): return s.n
    def __iter__(s):
        for i in range(s.n):
            time.sleep(0.04); yield pool[i%4]
t0=time.perf_counter(); r = tr.train(Slow(40)); torch.cuda.synchronize(); dt=time.perf_counter()-t0
print('slow loader 40 steps', dt, 'watch', {k[0]: (round(w.ref_ms,2), round(w.recent_ms(),2), len(w.periods)) for k,w in tr._watch.items()}, 'graphs', [g['graph'] is not None for g in tr._gstates.values()], 'eager_for', len(tr._graph_eager_for))
