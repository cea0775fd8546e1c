#!/usr/bin/env python
"""cProfile of the HOST side of the eager training step (ResNet-50 bf16 b=256): where the ~11-12 ms of host time per step go.
    python tools/host_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import sys

os.environ.setdefault('CONVNET_AMD_FLAGS', 'graph=0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch          # noqa: E402
import convnet_amd as ca   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device('cuda', 0)
    torch.manual_seed(123)
    model = ca.models.resnet(dataset='imagenet', depth=50)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device=str(dev),
                    dtype=torch.bfloat16, print_freq=10 ** 9)
    g = torch.Generator().manual_seed(123)
    pool = [(torch.randn(256, 3, 224, 224, generator=g).to(dev), torch.randint(0, 1000, (256,), generator=g).to(dev))
            for _ in range(2)]
    tr.train([pool[i % 2] for i in range(6)])
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    tr.train([pool[i % 2] for i in range(n)])
    pr.disable()
    torch.cuda.synchronize()
    for key in ('tottime', 'cumulative'):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(28)
        print('==== by %s (%d steps)' % (key, n))
        print('\n'.join(l[:150] for l in s.getvalue().split('\n')[4:44]))


if __name__ == '__main__':
    main()
