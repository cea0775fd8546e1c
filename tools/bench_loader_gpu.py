#!/usr/bin/env python
"""The image-folder pipeline end to end on the GPU box: worker processes (decode + crop [+ resize]) -> pinned batches ->
copy stream -> device-side resize / ToTensor / Normalize (trainer.DevicePrefetcher) [-> ResNet-50 bf16 training steps].
    python tools/bench_loader_gpu.py [--workers 16] [--images 4096] [--batch 256] [--train]
Modes: host (everything in the workers, the reference's pipeline), device-normalize, device-resize (round 6)."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=4096)
    ap.add_argument('--workers', type=int, default=16)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--train', action='store_true', help='feed ResNet-50 bf16 training steps (else just drain the prefetcher)')
    ap.add_argument('--modes', default='host,device-normalize,device-resize')
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    from PIL import Image
    import torch
    import convnet_amd as ca
    from convnet_amd import data as D
    root = tempfile.mkdtemp()
    rng = np.random.RandomState(0)
    for c in range(8):
        os.makedirs(os.path.join(root, 'imagenet', 'train', 'c%d' % c))
    base = [(rng.rand(375, 500, 3) * 60 + np.linspace(0, 180, 500)[None, :, None]).astype(np.uint8) for _ in range(16)]
    for i in range(args.images):     # ~500x375 photo-like noise + gradient, ~90 KB JPEGs (ImageNet's average is ~110 KB)
        Image.fromarray(np.roll(base[i % 16], i, axis=1)).save(
            os.path.join(root, 'imagenet', 'train', 'c%d' % (i % 8), '%05d.jpg' % i), quality=90)
    dev = torch.device('cuda', 0)
    res = {'workers': args.workers, 'images': args.images, 'batch': args.batch, 'train': args.train,
           'cores_allowed': len(os.sched_getaffinity(0))}
    tr = None
    if args.train:
        torch.manual_seed(1)
        model = ca.models.resnet(dataset='imagenet', depth=50, num_classes=8)
        tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device='cuda:0', dtype=torch.bfloat16,
                        print_freq=10 ** 9)
    for mode in args.modes.split(','):
        dr = D.DataRegime([{'epoch': 0}], defaults={'datasets_path': root, 'name': 'imagenet', 'split': 'train',
                                                     'augment': True, 'input_size': 224, 'batch_size': args.batch,
                                                     'shuffle': True, 'num_workers': args.workers, 'drop_last': True,
                                                     'pin_memory': True, 'device_normalize': mode != 'host',
                                                     'device_resize': mode == 'device-resize'})
        loader = dr.get_loader()
        rates = []
        for ep in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            n = 0
            if tr is not None:
                r = tr.train(loader)
                n = len(loader) * args.batch
            else:
                for x, t in ca.trainer.DevicePrefetcher(loader, dev):
                    n += t.shape[0]
            torch.cuda.synchronize()
            rates.append(n / (time.time() - t0))
        res[mode] = {'img_s_epochs': [round(v, 1) for v in rates], 'img_s': round(max(rates[1:]), 1)}
        print('%-17s %s img/s (epochs: %s)' % (mode, res[mode]['img_s'], res[mode]['img_s_epochs']))
        del loader, dr
    print(json.dumps(res))
    if args.out:
        json.dump(res, open(args.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
