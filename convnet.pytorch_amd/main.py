"""Command-line driver with the flag surface and control flow of /root/reference main.py:28-364
(`python main.py --model resnet --model-config "{'depth': 50}" -b 256 --dtype bfloat16 ...`),
re-authored for the MI355X engine:

* `models.__dict__[args.model](**{'dataset': ..., **literal_eval(--model-config)})` registry call,
  model-attached `regime` picked up exactly like main.py:243-253;
* `--dtype float|bfloat16|half` (the 16-bit types get the reference's `half` policy, main.py:239-250: BN
  statistics / parameters and master weights in fp32; fp16 wants `--loss-scale`, as in the reference); `--device cuda`; distributed via the launcher env (`--local_rank`,
  `--dist-init env://`, `--dist-backend nccl` = RCCL);
* results dir with `config.json`, `log.txt`, `results.csv`, `checkpoint.pth.tar` /
  `model_best.pth.tar` holding the reference's keys (epoch, model, config, state_dict,
  optim_state_dict, best_prec1); `state_dict` tensors have the reference's OIHW shapes, so
  checkpoints load both ways;
* data: a synthetic ImageNet-shaped dataset (`--dataset imagenet-synthetic`, the default here);
  `--dataset imagenet --datasets-dir ...` drives the image-folder pipeline of data.py.
"""
import argparse
import csv
import json
import logging
import os
import shutil
import time
from ast import literal_eval
from datetime import datetime

import torch
import torch.distributed as dist

from . import engine, models, torch_dtypes
from .cross_entropy import CrossEntropyLoss
from .optim import OptimRegime
from .trainer import Trainer

model_names = sorted(name for name in models.__dict__
                     if name.islower() and not name.startswith('__') and callable(models.__dict__[name]))


def build_parser():
    p = argparse.ArgumentParser(description='ConvNet training on MI355X (convNet.pytorch CLI surface)')
    p.add_argument('--config-file', default=None, help='json configuration file')
    p.add_argument('--results-dir', metavar='RESULTS_DIR', default='./results', help='results dir')
    p.add_argument('--save', metavar='SAVE', default='', help='saved folder')
    p.add_argument('--datasets-dir', metavar='DATASETS_DIR', default='~/Datasets', help='datasets dir')
    p.add_argument('--dataset', metavar='DATASET', default='imagenet-synthetic', help='dataset name')
    p.add_argument('--model', '-a', metavar='MODEL', default='resnet', choices=model_names,
                   help='model architecture: ' + ' | '.join(model_names))
    p.add_argument('--input-size', type=int, default=None, help='image input size')
    p.add_argument('--model-config', default='', help='additional architecture configuration')
    p.add_argument('--dtype', default='float', help='type of tensor: ' + ' | '.join(torch_dtypes.keys()))
    p.add_argument('--device', default='cuda', help='device assignment ("cuda")')
    p.add_argument('--device-ids', default=[0], type=int, nargs='+', help='device ids assignment')
    p.add_argument('--world-size', default=-1, type=int, help='number of distributed processes')
    p.add_argument('--local_rank', '--local-rank', default=-1, type=int, help='rank of distributed processes')
    p.add_argument('--dist-init', default='env://', type=str, help='init used to set up distributed training')
    p.add_argument('--dist-backend', default='nccl', type=str, help='distributed backend (nccl = RCCL)')
    p.add_argument('-j', '--workers', default=8, type=int, metavar='N')
    p.add_argument('--host-normalize', action='store_true',
                   help='(not in the reference) keep ToTensor + Normalize in the loader workers; default: uint8 crops leave '
                        'the workers and the two transforms run on the device behind the copy - same batch, bit for bit')
    p.add_argument('--device-resize', action='store_true',
                   help='(not in the reference) also move the Resize step to the device: the uint8 crops leave the workers '
                        'unresized and PIL\'s fixed-point BILINEAR resampler runs in csrc/resize.hip - same batch, bit for bit.  '
                        'Opt-in: it pays where the host cores are slow (+15 %% per worker in the build container) and is a wash on '
                        'the MI355X box (8.2k vs 8.7k img/s on its 16-core quota; profiles/r06_loader_gpu_box.txt)')
    p.add_argument('--epochs', default=90, type=int, metavar='N')
    p.add_argument('--start-epoch', default=-1, type=int, metavar='N')
    p.add_argument('-b', '--batch-size', default=256, type=int, metavar='N')
    p.add_argument('--eval-batch-size', default=-1, type=int)
    p.add_argument('--optimizer', default='SGD', type=str, metavar='OPT')
    p.add_argument('--drop-optim-state', action='store_true', default=False)
    p.add_argument('--save-all', action='store_true', default=False)
    p.add_argument('--label-smoothing', default=0, type=float)
    p.add_argument('--sync-bn', action='store_true', default=False)
    p.add_argument('--mixup', default=None, type=float)
    p.add_argument('--cutmix', default=None, type=float)
    p.add_argument('--duplicates', default=1, type=int)
    p.add_argument('--chunk-batch', default=1, type=int)
    p.add_argument('--cutout', action='store_true', default=False)
    p.add_argument('--autoaugment', action='store_true', default=False)
    p.add_argument('--grad-clip', default=-1, type=float)
    p.add_argument('--loss-scale', default=1, type=float)
    p.add_argument('--lr', '--learning-rate', default=0.1, type=float, metavar='LR')
    p.add_argument('--momentum', default=0.9, type=float, metavar='M')
    p.add_argument('--weight-decay', '--wd', default=0, type=float, metavar='W')
    p.add_argument('--print-freq', '-p', default=10, type=int, metavar='N')
    p.add_argument('--adapt-grad-norm', default=None, type=int)
    p.add_argument('--resume', default='', type=str, metavar='PATH')
    p.add_argument('-e', '--evaluate', type=str, metavar='FILE')
    p.add_argument('--seed', default=123, type=int)
    p.add_argument('--tensorwatch', action='store_true', default=False)
    p.add_argument('--tensorwatch-port', default=0, type=int)
    # synthetic-data knobs (not in the reference: it has no synthetic dataset)
    p.add_argument('--steps-per-epoch', default=100, type=int, help='synthetic: training batches per epoch')
    p.add_argument('--val-steps', default=10, type=int, help='synthetic: validation batches per epoch')
    return p


class SyntheticLoader(object):
    """Iterable with __len__ yielding (inputs NCHW fp32, target int64) - the only interface Trainer
    needs from a loader (trainer.py:198,235).  A small pool of seeded batches is cycled."""

    def __init__(self, n_batches, batch, size, classes, channels, seed, device=None, pool=8):
        g = torch.Generator().manual_seed(seed)
        self.pool = [(torch.randn(batch, channels, size, size, generator=g),
                      torch.randint(0, classes, (batch,), generator=g)) for _ in range(min(pool, n_batches))]
        if device is not None:
            self.pool = [(x.to(device), t.to(device)) for x, t in self.pool]
        self.n = n_batches

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.pool[i % len(self.pool)]


def save_checkpoint(state, is_best, path='.', filename='checkpoint.pth.tar', save_all=False):
    """utils.log.save_checkpoint as called at main.py:324-331."""
    filename = os.path.join(path, filename)
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, os.path.join(path, 'model_best.pth.tar'))
    if save_all:
        shutil.copyfile(filename, os.path.join(path, 'checkpoint_epoch_%s.pth.tar' % state['epoch']))


class ResultsLog(object):
    """results.csv with the reference's columns (epoch, steps, training *, validation *)."""

    def __init__(self, path):
        self.path = path + '.csv'
        self.rows = []

    def load(self, path):
        if os.path.isfile(path):
            with open(path) as f:
                self.rows = list(csv.DictReader(f))

    def add(self, **kw):
        self.rows.append(kw)

    def save(self):
        if not self.rows:
            return
        keys = list(self.rows[-1].keys())
        with open(self.path, 'w', newline='') as f:
            w = csv.DictWriter(f, fieldnames=keys, extrasaction='ignore')
            w.writeheader()
            for r in self.rows:
                w.writerow(r)


def main(argv=None):
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.config_file is not None:
        with open(args.config_file) as f:
            parser.set_defaults(**json.loads(f.read()))
        args = parser.parse_args(argv)
    return main_worker(args)


def main_worker(args):
    best_prec1 = 0
    if args.dtype not in torch_dtypes:
        raise SystemExit("--dtype %r unsupported on the MI355X path (choose from %s)" % (args.dtype,
                                                                                         list(torch_dtypes)))
    dtype = torch_dtypes[args.dtype]
    torch.manual_seed(args.seed)
    time_stamp = datetime.now().strftime('%Y-%m-%d_%H-%M-%S')
    if args.evaluate:
        args.results_dir = '/tmp'
    if args.save == '':
        args.save = time_stamp
    save_path = os.path.join(args.results_dir, args.save)

    if args.local_rank < 0 and 'LOCAL_RANK' in os.environ and int(os.environ.get('WORLD_SIZE', 1)) > 1:
        args.local_rank = int(os.environ['LOCAL_RANK'])   # torchrun exports it instead of --local_rank
    args.distributed = args.local_rank >= 0 or args.world_size > 1
    if args.distributed:
        dist.init_process_group(backend=args.dist_backend, init_method=args.dist_init,
                                world_size=args.world_size, rank=args.local_rank if args.dist_init != 'env://' else -1)
        args.local_rank = dist.get_rank()
        args.world_size = dist.get_world_size()
        args.device_ids = [args.local_rank % max(torch.cuda.device_count(), 1)]
    main_rank = not (args.distributed and args.local_rank > 0)

    if main_rank:
        os.makedirs(save_path, exist_ok=True)
        with open(os.path.join(save_path, 'config.json'), 'w') as f:
            json.dump({k: v for k, v in vars(args).items()}, f, indent=2, sort_keys=True, default=str)
        logging.basicConfig(level=logging.INFO, format='%(asctime)s - %(levelname)s - %(message)s',
                            handlers=[logging.FileHandler(os.path.join(save_path, 'log.txt')),
                                      logging.StreamHandler()], force=True)
    else:
        logging.basicConfig(level=logging.ERROR, force=True)
    results = ResultsLog(os.path.join(save_path, 'results'))
    logging.info('saving to %s', save_path)
    logging.info('creating model %s', args.model)

    from . import _lib
    if 'cuda' in args.device and torch.cuda.is_available():
        torch.cuda.set_device(args.device_ids[0])
        device = torch.device('cuda', args.device_ids[0])
    elif _lib.emulation_requested():
        device = torch.device('cpu')   # TEST-ONLY emulator run of the CLI (CONVNET_AMD_EMULATE=1)
    else:
        raise SystemExit('this engine runs on an MI355X (--device cuda): there is no CPU path')

    model_config = {'dataset': args.dataset}
    if args.model_config != '':
        model_config = dict(model_config, **literal_eval(args.model_config))
    model = models.__dict__[args.model](**model_config)
    if args.sync_bn:   # main.py:190-191
        from . import nn as cnn
        model = cnn.convert_sync_batchnorm(model)
    logging.info('created model with configuration: %s', model_config)
    logging.info('number of parameters: %d', sum(p.nelement() for p in model.parameters()))

    optim_state_dict = None
    if args.evaluate:
        if not os.path.isfile(args.evaluate):
            parser = build_parser()
            parser.error('invalid checkpoint: {}'.format(args.evaluate))
        checkpoint = torch.load(args.evaluate, map_location='cpu')
        model.load_state_dict(checkpoint['state_dict'])
        logging.info("loaded checkpoint '%s' (epoch %s)", args.evaluate, checkpoint['epoch'])
    if args.resume:
        checkpoint_file = args.resume
        if os.path.isdir(checkpoint_file):
            results.load(os.path.join(checkpoint_file, 'results.csv'))
            checkpoint_file = os.path.join(checkpoint_file, 'model_best.pth.tar')
        if os.path.isfile(checkpoint_file):
            checkpoint = torch.load(checkpoint_file, map_location='cpu')
            if args.start_epoch < 0:
                args.start_epoch = checkpoint['epoch']
            best_prec1 = checkpoint['best_prec1']
            model.load_state_dict(checkpoint['state_dict'])
            optim_state_dict = checkpoint.get('optim_state_dict', None)
            logging.info("loaded checkpoint '%s' (epoch %s)", checkpoint_file, checkpoint['epoch'])
        else:
            logging.error("no checkpoint found at '%s'", args.resume)

    loss_params = {}
    if args.label_smoothing > 0:
        loss_params['smooth_eps'] = args.label_smoothing
    criterion = getattr(model, 'criterion', CrossEntropyLoss)(**loss_params)
    # model.to(device, dtype) of the reference (main.py:236): fp32 master arena + compute dtype
    engine.prepare(model, device, dtype)

    optim_regime = getattr(model, 'regime', [{'epoch': 0, 'optimizer': args.optimizer, 'lr': args.lr,
                                              'momentum': args.momentum, 'weight_decay': args.weight_decay}])
    optimizer = optim_regime if isinstance(optim_regime, OptimRegime) \
        else OptimRegime(model, optim_regime, use_float_copy=True)
    if optim_state_dict is not None:
        optimizer.load_state_dict(optim_state_dict)

    trainer = Trainer(model, criterion, optimizer, device_ids=args.device_ids, device=str(device), dtype=dtype,
                      print_freq=args.print_freq, distributed=args.distributed, local_rank=args.local_rank,
                      mixup=args.mixup, cutmix=args.cutmix, loss_scale=args.loss_scale, grad_clip=args.grad_clip,
                      adapt_grad_norm=args.adapt_grad_norm)

    args.eval_batch_size = args.eval_batch_size if args.eval_batch_size > 0 else args.batch_size
    if 'synthetic' in args.dataset:
        is_mnist = args.model == 'mnist'
        size = args.input_size or (28 if is_mnist else 224)
        classes, channels = (10, 1) if is_mnist else (model_config.get('num_classes', 1000), 3)
        rank = max(args.local_rank, 0)
        val_data = SyntheticLoader(args.val_steps, args.eval_batch_size, size, classes, channels, args.seed + 10000)
        val_loader = lambda: val_data                                   # noqa: E731
        if not args.evaluate:
            train_data = SyntheticLoader(args.steps_per_epoch, args.batch_size, size, classes, channels,
                                         args.seed + 1 + rank)
            train_loader = lambda: train_data                           # noqa: E731
    else:
        # real image folders through DataRegime, exactly the settings of main.py:264-293
        from .data import DataRegime
        if hasattr(model, 'sampled_data_regime'):
            raise NotImplementedError('sampled (mixed-size) data regimes are not part of the hot path')
        val_data = DataRegime(getattr(model, 'data_eval_regime', None),
                              defaults={'datasets_path': args.datasets_dir, 'name': args.dataset, 'split': 'val',
                                        'augment': False, 'input_size': args.input_size,
                                        'batch_size': args.eval_batch_size, 'shuffle': False,
                                        'num_workers': args.workers, 'pin_memory': True, 'drop_last': False,
                                        'device_normalize': not args.host_normalize,
                                        'device_resize': args.device_resize and not args.host_normalize})
        val_loader = val_data.get_loader
        if not args.evaluate:
            train_data = DataRegime(getattr(model, 'data_regime', None),
                                    defaults={'datasets_path': args.datasets_dir, 'name': args.dataset,
                                              'split': 'train', 'augment': True, 'input_size': args.input_size,
                                              'batch_size': args.batch_size, 'shuffle': True,
                                              'num_workers': args.workers, 'pin_memory': True, 'drop_last': True,
                                              'distributed': args.distributed, 'duplicates': args.duplicates,
                                              'autoaugment': args.autoaugment,
                                              'cutout': {'holes': 1, 'length': 16} if args.cutout else None,
                                              'device_normalize': not args.host_normalize,
                                        'device_resize': args.device_resize and not args.host_normalize})
            train_loader = train_data.get_loader
            logging.info('data regime: %s', train_data)
    if args.evaluate:
        res = trainer.validate(val_loader())
        logging.info(res)
        return res

    logging.info('optimization regime: %s', optim_regime)
    args.start_epoch = max(args.start_epoch, 0)
    trainer.training_steps = args.start_epoch * len(train_data)
    train_results = val_results = None
    for epoch in range(args.start_epoch, args.epochs):
        trainer.epoch = epoch
        logging.info('\nStarting Epoch: {0}\n'.format(epoch + 1))
        t0 = time.time()
        if hasattr(train_data, 'set_epoch'):   # main.py:301-302
            train_data.set_epoch(epoch)
            val_data.set_epoch(epoch)
        train_results = trainer.train(train_loader(), chunk_batch=args.chunk_batch)
        val_results = trainer.validate(val_loader())
        if not main_rank:
            continue
        is_best = val_results['prec1'] > best_prec1
        best_prec1 = max(val_results['prec1'], best_prec1)
        save_checkpoint({'epoch': epoch + 1, 'model': args.model, 'config': args.model_config,
                         'state_dict': model.state_dict(),
                         'optim_state_dict': None if args.drop_optim_state else optimizer.state_dict(),
                         'best_prec1': best_prec1}, is_best, path=save_path, save_all=args.save_all)
        logging.info('\nResults - Epoch: {0}\nTraining Loss {train[loss]:.4f} \tTraining Prec@1 {train[prec1]:.3f} \t'
                     'Training Prec@5 {train[prec5]:.3f} \tValidation Loss {val[loss]:.4f} \t'
                     'Validation Prec@1 {val[prec1]:.3f} \tValidation Prec@5 {val[prec5]:.3f} \t'
                     '[{ips:.0f} img/s]\n'.format(epoch + 1, train=train_results, val=val_results,
                                                  ips=len(train_loader()) * args.batch_size / (time.time() - t0)))
        values = dict(epoch=epoch + 1, steps=trainer.training_steps)
        values.update({'training ' + k: v for k, v in train_results.items()})
        values.update({'validation ' + k: v for k, v in val_results.items()})
        results.add(**values)
        results.save()
    return {'train': train_results, 'val': val_results}


if __name__ == '__main__':
    main()
