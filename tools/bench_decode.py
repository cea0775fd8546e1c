#!/usr/bin/env python
"""Decode + augment throughput of the image-folder input pipeline (convnet.pytorch_amd/data.py: PIL decode,
RandomResizedCrop(224), flip, ToTensor, Normalize) on synthetic ImageNet-sized JPEGs, per worker count.
CPU only (measurement aid; SURVEY.md section 8f-3: the step needs ~12k img/s per GPU)."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('CONVNET_AMD_EMULATE', '1')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--images', type=int, default=512)
    ap.add_argument('--workers', default='1,4,8')
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--device-normalize', action='store_true', help='workers stop at the uint8 crop (ToTensor + Normalize on the device)')
    ap.add_argument('--device-resize', action='store_true', help='... and at the unresized crop (PIL resize on the device too)')
    args = ap.parse_args()
    from PIL import Image
    import torch
    from convnet_amd import data as D
    root = tempfile.mkdtemp()
    rng = np.random.RandomState(0)
    for c in range(4):
        os.makedirs(os.path.join(root, 'imagenet', 'train', 'c%d' % c))
    for i in range(args.images):     # ~500x375 photos-like noise + gradients (average ImageNet size ~110 KB)
        a = (rng.rand(375, 500, 3) * 60 + np.linspace(0, 180, 500)[None, :, None]).astype(np.uint8)
        Image.fromarray(a).save(os.path.join(root, 'imagenet', 'train', 'c%d' % (i % 4), '%05d.jpg' % i), quality=90)
    for nw in [int(w) for w in args.workers.split(',')]:
        dr = D.DataRegime([{'epoch': 0}], defaults={'datasets_path': root, 'name': 'imagenet', 'split': 'train',
                                                     'augment': True, 'input_size': 224, 'batch_size': args.batch,
                                                     'shuffle': True, 'num_workers': nw, 'drop_last': True,
                                                     'device_normalize': args.device_normalize or args.device_resize,
                                                     'device_resize': args.device_resize})
        loader = dr.get_loader()
        n = 0
        for x, t in loader:     # warm-up epoch (worker start-up, page cache)
            n += t.shape[0]
        t0 = time.time()
        n = 0
        for x, t in loader:
            n += t.shape[0]
        dt = time.time() - t0
        print('workers %2d: %7.1f img/s (%d images, %.2f s; %.1f img/s per worker)' % (nw, n / dt, n, dt, n / dt / max(nw, 1)))
        del loader, dr
    print('host: %d usable cores (torch threads %d)' % (len(os.sched_getaffinity(0)), torch.get_num_threads()))


if __name__ == '__main__':
    main()
