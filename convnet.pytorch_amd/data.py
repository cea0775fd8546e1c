"""Input pipeline for real image folders: the host side of the reference's `data.py` + `preprocess.py`
(/root/reference data.py:17-125, preprocess.py:21-161) for the datasets the hot path is benchmarked on.

The reference builds its pipeline from torchvision (`datasets.ImageFolder`, `transforms.*`), which is
not part of this image; the same steps are re-stated here on PIL + numpy:

    ImageFolder (class sub-directories, sorted)                       data.py:45-53
    train : RandomResizedCrop(input_size) -> RandomHorizontalFlip
            -> ToTensor -> Normalize(mean, std)                       preprocess.py:71-77 (inception_preprocess)
    eval  : Resize(scale_size = input_size*8/7) -> CenterCrop(input_size)
            -> ToTensor -> Normalize                                  preprocess.py:21-41 (scale_crop)
    DataRegime: epoch-keyed loader settings, DistributedSampler       data.py:74-125

Batches leave this module exactly as the reference's loader hands them to `Trainer`: fp32 NCHW
`inputs`, int64 `target`, pinned when asked.  The device side (copy-stream prefetch, fused cast to the
bf16 NHWC / pixel-pair layout) is trainer.DevicePrefetcher + csrc/pool.hip.

Randomness: the random transforms draw from torch's global generator with exactly the calls, arguments and
order of torchvision's implementations (>= 0.8: `RandomResizedCrop.get_params`: `torch.empty(1).uniform_(scale)`,
`torch.empty(1).uniform_(log ratio)`, `torch.randint` for the top / left corner, ten attempts, centre-crop
fallback; `RandomHorizontalFlip`: `torch.rand(1) < p`), so a DataLoader worker seeded like the reference's
(base_seed + worker_id -> torch.manual_seed) consumes the same stream and takes the same crops / flips as the
reference's torchvision pipeline.  torchvision is absent from this image, so this is pinned by construction
(the algorithm is restated from its published source) plus a recorded draw sequence in tests/test_data.py, not
by running torchvision; the deterministic eval transform is pinned against PIL there as well.
"""
import math
import os
import random
from copy import deepcopy

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset, Subset
from torch.utils.data.distributed import DistributedSampler

from .optim import Regime

_IMAGENET_STATS = {'mean': [0.485, 0.456, 0.406], 'std': [0.229, 0.224, 0.225]}   # preprocess.py:7-8
IMG_EXTENSIONS = ('.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif', '.tiff', '.webp')


def _pil():
    from PIL import Image
    return Image


# ---------------------------------------------------------------------------------------------
# transforms (callables on PIL images / CHW float tensors)

class Compose(object):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img

    def __repr__(self):
        return 'Compose(%s)' % ', '.join(repr(t) for t in self.transforms)


class Resize(object):
    """Shorter side -> `size`, aspect ratio kept, bilinear (torchvision.transforms.Resize(int))."""

    def __init__(self, size):
        self.size = int(size)

    def __call__(self, img):
        w, h = img.size
        if (w <= h and w == self.size) or (h <= w and h == self.size):
            return img
        if w < h:
            ow, oh = self.size, int(self.size * h / w)
        else:
            oh, ow = self.size, int(self.size * w / h)
        return img.resize((ow, oh), _pil().BILINEAR)

    def __repr__(self):
        return 'Resize(%d)' % self.size


class CenterCrop(object):
    def __init__(self, size):
        self.size = int(size)

    def __call__(self, img):
        w, h = img.size
        th = tw = self.size
        if w < tw or h < th:    # pad symmetrically with zeros, like torchvision
            Image = _pil()
            canvas = Image.new(img.mode, (max(w, tw), max(h, th)))
            canvas.paste(img, ((canvas.size[0] - w) // 2, (canvas.size[1] - h) // 2))
            img, (w, h) = canvas, canvas.size
        left, top = int(round((w - tw) / 2.0)), int(round((h - th) / 2.0))
        return img.crop((left, top, left + tw, top + th))

    def __repr__(self):
        return 'CenterCrop(%d)' % self.size


class RandomResizedCrop(object):
    """Inception-style crop: area fraction in `scale`, aspect ratio log-uniform in `ratio`, ten
    attempts then a centre crop clamped to the ratio range; resized to size x size (bilinear)."""

    def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), device_resize=False):
        self.size, self.scale, self.ratio = int(size), scale, ratio
        self.device_resize = device_resize      # True: stop at the crop; the resize runs on the device (CropForDevice)
        lr = torch.log(torch.tensor(self.ratio))
        self._log_ratio = (lr[0].item(), lr[1].item())

    def get_params(self, w, h):
        """torchvision.transforms.RandomResizedCrop.get_params, draw for draw (returns left, top, cw, ch)."""
        area = h * w
        log_ratio = self._log_ratio      # torch.log(torch.tensor(self.ratio)): the same fp32 values, computed once
        for _ in range(10):
            target_area = area * torch.empty(1).uniform_(self.scale[0], self.scale[1]).item()
            aspect = torch.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1])).item()
            cw = int(round(math.sqrt(target_area * aspect)))
            ch = int(round(math.sqrt(target_area / aspect)))
            if 0 < cw <= w and 0 < ch <= h:
                top = torch.randint(0, h - ch + 1, size=(1,)).item()
                left = torch.randint(0, w - cw + 1, size=(1,)).item()
                return left, top, cw, ch
        in_ratio = float(w) / float(h)
        if in_ratio < min(self.ratio):
            cw, ch = w, int(round(w / min(self.ratio)))
        elif in_ratio > max(self.ratio):
            ch, cw = h, int(round(h * max(self.ratio)))
        else:
            cw, ch = w, h
        return (w - cw) // 2, (h - ch) // 2, cw, ch

    def __call__(self, img):
        # torchvision's F.resized_crop: crop FIRST, then resize the crop (PIL: img.crop(...).resize(...)).  NOT PIL's
        # resize(box=...): that form lets the antialiasing window read source pixels outside the box, which changes the
        # outermost ring of the 224 x 224 result (up to 18 grey levels on noise; round 4 fix)
        left, top, cw, ch = self.get_params(*img.size)
        crop = img.crop((left, top, left + cw, top + ch))
        if self.device_resize:
            return CropForDevice(np.array(crop, dtype=np.uint8), resample_table_cached(cw, self.size),
                                 resample_table_cached(ch, self.size), self.size)
        return crop.resize((self.size, self.size), _pil().BILINEAR)

    def __repr__(self):
        return 'RandomResizedCrop(%d)' % self.size


# ---- the Resize step on the device (csrc/resize.hip) -------------------------------------------------------------
_PRECISION_BITS = 32 - 8 - 2      # PIL libImaging/Resample.c


_TABLES = {}


def resample_table_cached(in_size, out_size):
    """resample_table for the whole output range, memoised per (in, out) size: a worker meets a few hundred distinct crop
    widths / heights, each table is ~10 KB (callers must not modify the returned array)."""
    t = _TABLES.get((in_size, out_size))
    if t is None:
        if len(_TABLES) > 4096:
            _TABLES.clear()
        t = _TABLES[(in_size, out_size)] = resample_table(in_size, out_size)
    return t


def resample_table(in_size, out_size, lo=0, n=None):
    """PIL's BILINEAR resampling coefficients for output indices [lo, lo + n) of a resize of `in_size` pixels to
    `out_size` (libImaging/Resample.c: precompute_coeffs + normalize_coeffs_8bpc, operation for operation in float64):
    int32 [n][2 + ksize] = {first input index, count, round(k * 2^22) ...}.  With these, the 8-bit resize is pure integer
    arithmetic: out = clip8((2^21 + sum_j in[xmin + j] * kk[j]) >> 22), which cn_resize_u8_crops reproduces exactly."""
    n = out_size - lo if n is None else n
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale                      # bilinear: support 1
    ksize = int(math.ceil(support)) * 2 + 1
    xx = np.arange(lo, lo + n, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = (center - support + 0.5).astype(np.int64)          # (int) truncates toward zero; negatives clamp to 0 either way
    xmin = np.maximum(xmin, 0)
    xmax = (center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    x = np.arange(ksize, dtype=np.int64)[None, :]
    arg = ((x + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss
    arg = np.abs(arg)
    w = np.where(arg < 1.0, 1.0 - arg, 0.0)
    valid = x < xmax[:, None]
    w = np.where(valid, w, 0.0)
    ww = np.zeros(n, dtype=np.float64)
    for j in range(ksize):                                   # sequential sum, like the C loop (not numpy's pairwise sum)
        ww = np.where(valid[:, j], ww + w[:, j], ww)
    k = np.where((ww != 0.0)[:, None] & valid, w / np.where(ww != 0.0, ww, 1.0)[:, None], w)
    kk = np.where(k < 0, -0.5 + k * (1 << _PRECISION_BITS), 0.5 + k * (1 << _PRECISION_BITS)).astype(np.int32)   # (int): toward zero
    kk = np.where(valid, kk, 0).astype(np.int32)
    out = np.empty((n, 2 + ksize), dtype=np.int32)
    out[:, 0] = xmin
    out[:, 1] = xmax
    out[:, 2:] = kk
    return out


class CropForDevice(object):
    """What a loader worker hands over when the Resize step runs on the device: the uint8 source region (HWC), the two
    coefficient tables that turn it into the size x size result, and whether the result is mirrored."""
    __slots__ = ('pix', 'th', 'tv', 'flip', 'size')

    def __init__(self, pix, th, tv, size):
        self.pix, self.th, self.tv, self.size, self.flip = pix, th, tv, size, False


def collate_crops(batch):
    """collate_fn of a device_resize loader: B CropForDevice samples -> the flat buffers cn_resize_u8_crops takes."""
    crops = [b[0] for b in batch]
    target = torch.tensor([b[1] for b in batch], dtype=torch.int64)
    B = len(crops)
    meta = np.zeros((B, 8), dtype=np.int64)
    hs = np.array([c.pix.shape[0] for c in crops], dtype=np.int64)
    pixels = torch.empty(sum(c.pix.size for c in crops), dtype=torch.uint8)      # (one copy of every crop, straight into the batch)
    tables = torch.empty(sum(c.th.size + c.tv.size for c in crops), dtype=torch.int32)
    pn, tn = pixels.numpy(), tables.numpy()
    poff = toff = 0
    for i, c in enumerate(crops):
        h, w = c.pix.shape[0], c.pix.shape[1]
        meta[i] = (poff, h, w, int(c.flip), toff, c.th.shape[1] - 2, toff + c.th.size, c.tv.shape[1] - 2)
        pn[poff:poff + c.pix.size] = c.pix.reshape(-1)
        tn[toff:toff + c.th.size] = c.th.reshape(-1)
        tn[toff + c.th.size:toff + c.th.size + c.tv.size] = c.tv.reshape(-1)
        poff += c.pix.size
        toff += c.th.size + c.tv.size
    row_off = np.concatenate([[0], np.cumsum(hs)[:-1]]).astype(np.int32)
    row_owner = np.repeat(np.arange(B, dtype=np.int32), hs)
    inputs = {'crops': pixels, 'meta': torch.from_numpy(meta), 'tables': tables, 'row_owner': torch.from_numpy(row_owner),
              'row_off': torch.from_numpy(row_off), 'size': torch.tensor([crops[0].size, crops[0].pix.shape[2]], dtype=torch.int32)}
    return inputs, target


class ResizeCenterCropForDevice(object):
    """Resize(scale_size) -> CenterCrop(size) (scale_crop) with the resize left to the device: the worker ships the source
    region the size x size centre window of the resized image depends on, and that window's coefficient tables."""

    def __init__(self, scale_size, size):
        self.scale_size, self.size = int(scale_size), int(size)

    def __call__(self, img):
        w, h = img.size
        s = self.scale_size
        if (w <= h and w == s) or (h <= w and h == s):
            ow, oh = w, h
        elif w < h:
            ow, oh = s, int(s * h / w)
        else:
            oh, ow = s, int(s * w / h)
        if ow < self.size or oh < self.size:
            raise NotImplementedError('device_resize: the resized image is smaller than the crop (zero padding) - use the host pipeline')
        left, top = int(round((ow - self.size) / 2.0)), int(round((oh - self.size) / 2.0))
        th = resample_table(w, ow, left, self.size)
        tv = resample_table(h, oh, top, self.size)
        x0, x1 = int(th[:, 0].min()), int((th[:, 0] + th[:, 1]).max())
        y0, y1 = int(tv[:, 0].min()), int((tv[:, 0] + tv[:, 1]).max())
        th[:, 0] -= x0
        tv[:, 0] -= y0
        return CropForDevice(np.array(img.crop((x0, y0, x1, y1)), dtype=np.uint8), th, tv, self.size)

    def __repr__(self):
        return 'ResizeCenterCropForDevice(%d, %d)' % (self.scale_size, self.size)


class RandomHorizontalFlip(object):
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        if torch.rand(1) < self.p:       # torchvision.transforms.RandomHorizontalFlip.forward
            if isinstance(img, CropForDevice):
                img.flip = not img.flip      # (the device mirrors the output columns of the resize)
                return img
            return img.transpose(_pil().FLIP_LEFT_RIGHT)
        return img

    def __repr__(self):
        return 'RandomHorizontalFlip(%g)' % self.p


class ToTensor(object):
    """PIL image -> float32 CHW in [0, 1]."""

    def __call__(self, img):
        a = np.asarray(img, dtype=np.uint8)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div_(255.0)

    def __repr__(self):
        return 'ToTensor()'


class ToUint8HWC(object):
    """PIL image -> uint8 HWC tensor: what leaves the workers when ToTensor + Normalize run on the device
    (`device_normalize`; ops.u8_nhwc_to_nchw, bit-identical to the two host transforms)."""

    def __call__(self, img):
        a = np.array(img, dtype=np.uint8)      # (a writable copy: torch.from_numpy refuses to share PIL's read-only buffer)
        if a.ndim == 2:
            a = a[:, :, None]
        return torch.from_numpy(a)

    def __repr__(self):
        return 'ToUint8HWC()'


class Normalize(object):
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, t):
        return (t - self.mean) / self.std

    def __repr__(self):
        return 'Normalize(mean=%s, std=%s)' % (self.mean.flatten().tolist(), self.std.flatten().tolist())


def _to_tensor(normalize, device_normalize):
    return [ToUint8HWC()] if device_normalize else [ToTensor(), Normalize(**normalize)]


def scale_crop(input_size, scale_size=None, normalize=None, device_normalize=False, device_resize=False):
    """Evaluation transform (preprocess.py:21-41, num_crops = 1)."""
    normalize = normalize or _IMAGENET_STATS
    if device_resize:
        return Compose([ResizeCenterCropForDevice(scale_size, input_size)])
    t = [CenterCrop(input_size)] + _to_tensor(normalize, device_normalize)
    if scale_size != input_size:
        t = [Resize(scale_size)] + t
    return Compose(t)


def inception_preprocess(input_size, normalize=None, device_normalize=False, device_resize=False):
    """Training transform (preprocess.py:71-77)."""
    normalize = normalize or _IMAGENET_STATS
    if device_resize:
        return Compose([RandomResizedCrop(input_size, device_resize=True), RandomHorizontalFlip()])
    return Compose([RandomResizedCrop(input_size), RandomHorizontalFlip()] + _to_tensor(normalize, device_normalize))


def get_transform(transform_name='imagenet', input_size=None, scale_size=None, normalize=None, augment=True,
                  cutout=None, autoaugment=False, padding=None, duplicates=1, num_crops=1, device_normalize=False,
                  device_resize=False):
    """preprocess.get_transform (preprocess.py:115-161) for the ImageNet family; the research
    augmentations (autoaugment, cutout, duplicates, multi-crop) are outside the hot path."""
    if 'imagenet' not in transform_name:
        raise NotImplementedError('transform %r: only the ImageNet pipeline is built' % transform_name)
    if autoaugment or cutout is not None or duplicates != 1 or num_crops != 1:
        raise NotImplementedError('autoaugment / cutout / duplicates / multi-crop are not part of the hot path')
    input_size = input_size or 224
    scale_size = scale_size or int(input_size * 8 / 7)
    # device_normalize (not in the reference; main.py turns it on unless --host-normalize is given): the workers stop at the uint8 crop and ToTensor + Normalize
    # run on the device behind the host->device copy (trainer.DevicePrefetcher) - the same fp32 NCHW batch, bit for bit
    # device_resize (round 6; implies device_normalize): the workers stop at the uint8 CROP (any size) and PIL's fixed-point
    # BILINEAR resize runs on the device too (csrc/resize.hip; coefficient tables computed here, by PIL's recipe): the same
    # batch bit for bit, 0.64 of 2.8 ms per image less work in the workers
    if device_resize and not device_normalize:
        raise ValueError('device_resize needs device_normalize (the crops leave the workers as uint8)')
    if augment:
        return inception_preprocess(input_size, normalize=normalize, device_normalize=device_normalize,
                                    device_resize=device_resize)
    return scale_crop(input_size=input_size, scale_size=scale_size, normalize=normalize, device_normalize=device_normalize,
                      device_resize=device_resize)


# ---------------------------------------------------------------------------------------------
# datasets

class ImageFolder(Dataset):
    """root/<class>/<image>: classes = sorted sub-directory names, samples sorted by path
    (torchvision.datasets.ImageFolder semantics, data.py:45-53)."""

    def __init__(self, root, transform=None, target_transform=None):
        root = os.path.expanduser(root)
        if not os.path.isdir(root):
            raise FileNotFoundError('image folder %r does not exist' % root)
        self.root = root
        self.classes = sorted(d.name for d in os.scandir(root) if d.is_dir())
        if not self.classes:
            raise FileNotFoundError('no class folders under %r' % root)
        self.class_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.samples = []
        for c in self.classes:
            for dirpath, _, files in sorted(os.walk(os.path.join(root, c), followlinks=True)):
                for f in sorted(files):
                    if f.lower().endswith(IMG_EXTENSIONS):
                        self.samples.append((os.path.join(dirpath, f), self.class_to_idx[c]))
        if not self.samples:
            raise FileNotFoundError('no images under %r' % root)
        self.targets = [t for _, t in self.samples]
        self.transform, self.target_transform = transform, target_transform

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, i):
        path, target = self.samples[i]
        with open(path, 'rb') as f:
            img = _pil().open(f)
            img.load()                      # (decode while the file is open)
            if img.mode != 'RGB':           # the reference's pil_loader converts unconditionally; for an RGB image that is a
                img = img.convert('RGB')    # plain copy of the decoded pixels (0.1 ms of the 2 ms per image) - skipped
        if self.transform is not None:
            img = self.transform(img)
        if self.target_transform is not None:
            target = self.target_transform(target)
        return img, target


def get_dataset(name, split='train', transform=None, target_transform=None, download=True,
                datasets_path='~/Datasets'):
    """data.get_dataset (data.py:17-70) for `imagenet`: <datasets_path>/imagenet/{train,val}/<class>/*."""
    if name != 'imagenet':
        raise NotImplementedError('dataset %r: the folder pipeline is built for "imagenet"' % name)
    root = os.path.join(os.path.expanduser(datasets_path), name, 'train' if split == 'train' else 'val')
    return ImageFolder(root, transform=transform, target_transform=target_transform)


_DATA_ARGS = {'name', 'split', 'transform', 'target_transform', 'download', 'datasets_path'}
_DATALOADER_ARGS = {'batch_size', 'shuffle', 'sampler', 'batch_sampler', 'num_workers', 'collate_fn', 'pin_memory',
                    'drop_last', 'timeout', 'worker_init_fn'}
_TRANSFORM_ARGS = {'transform_name', 'input_size', 'scale_size', 'normalize', 'augment', 'cutout', 'duplicates',
                   'num_crops', 'autoaugment', 'device_normalize', 'device_resize'}
_OTHER_ARGS = {'distributed'}


def _seed_worker(worker_id):
    seed = torch.initial_seed() % 2 ** 32
    random.seed(seed)
    np.random.seed(seed)


class DataRegime(object):
    """Epoch-keyed data settings -> DataLoader (data.py:74-125): the same setting groups, the same
    `get_loader / set_epoch / get / __len__` surface main.py drives (main.py:264-310)."""

    def __init__(self, regime, defaults={}):
        self.regime = Regime(regime, deepcopy(defaults))
        self.epoch = 0
        self.steps = None
        self._sampler = None
        self.get_loader(True)

    def get_setting(self):
        setting = self.regime.setting
        out = {'data': {k: v for k, v in setting.items() if k in _DATA_ARGS},
               'loader': {k: v for k, v in setting.items() if k in _DATALOADER_ARGS},
               'transform': {k: v for k, v in setting.items() if k in _TRANSFORM_ARGS},
               'other': {k: v for k, v in setting.items() if k in _OTHER_ARGS}}
        out['transform'].setdefault('transform_name', out['data']['name'])
        return out

    def get(self, key, default=None):
        return self.regime.setting.get(key, default)

    def get_loader(self, force_update=False, override_settings=None, subset_indices=None):
        if force_update or self.regime.update(self.epoch, self.steps):
            setting = self.get_setting()
            if override_settings is not None:
                setting.update(override_settings)
            self._transform = get_transform(**setting['transform'])
            setting['data'].setdefault('transform', self._transform)
            self._data = get_dataset(**setting['data'])
            if subset_indices is not None:
                self._data = Subset(self._data, subset_indices)
            loader = dict(setting['loader'])
            if setting['other'].get('distributed', False):
                loader['sampler'] = DistributedSampler(self._data)
                loader['shuffle'] = None
            self._sampler = loader.get('sampler', None)
            if setting['transform'].get('device_resize'):
                loader.setdefault('collate_fn', collate_crops)
            if loader.get('num_workers', 0) > 0:
                loader.setdefault('worker_init_fn', _seed_worker)
                loader.setdefault('persistent_workers', True)
            self._loader = DataLoader(self._data, **loader)
            # uint8 HWC batches carry what the device-side ToTensor + Normalize needs (trainer.DevicePrefetcher reads it)
            self._loader.device_normalize = dict(setting['transform'].get('normalize') or _IMAGENET_STATS) \
                if setting['transform'].get('device_normalize') else None
        return self._loader

    def set_epoch(self, epoch):
        self.epoch = epoch
        if self._sampler is not None and hasattr(self._sampler, 'set_epoch'):
            self._sampler.set_epoch(epoch)

    def __len__(self):
        return len(self._data)

    def __repr__(self):
        return str(self.regime.setting)
