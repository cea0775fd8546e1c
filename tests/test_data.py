"""Input pipeline (convnet.pytorch_amd/data.py; the reference's data.py / preprocess.py re-stated on
PIL + numpy because torchvision is not installed here).

Pinned: the deterministic evaluation transform against an independent PIL + numpy computation, the
ImageFolder class / sample ordering, the DataRegime -> DataLoader plumbing (shapes, dtypes, labels,
epoch-keyed settings) and a CLI run of `main.py --dataset imagenet` on a tiny folder through the
TEST-ONLY emulator.  Unpinned (stated in data.py): the random draws of the training augmentation -
only their invariants are tested (crop inside the image, area / ratio ranges, output size, flips)."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from helpers import ROOT  # noqa: F401  (puts the repo root on sys.path)

PIL = pytest.importorskip('PIL')
from PIL import Image  # noqa: E402


def _make_folder(root, classes=('cat', 'apple', 'bird'), per_class=3, size=(40, 52), seed=0):
    rng = np.random.RandomState(seed)
    for split in ('train', 'val'):
        for ci, c in enumerate(classes):
            d = os.path.join(root, 'imagenet', split, c)
            os.makedirs(d, exist_ok=True)
            for i in range(per_class):
                w, h = size[0] + 7 * i, size[1] - 5 * ci
                arr = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
                Image.fromarray(arr).save(os.path.join(d, 'img_%d.png' % i))
    return os.path.join(root, 'imagenet')


def test_imagefolder_ordering_and_eval_transform(tmp_path):
    import convnet_amd as ca
    from convnet_amd import data as D
    root = _make_folder(str(tmp_path))
    tf = D.get_transform('imagenet', input_size=28, augment=False)          # Resize(32) -> CenterCrop(28)
    ds = D.get_dataset('imagenet', split='val', transform=tf, datasets_path=str(tmp_path))
    assert ds.classes == ['apple', 'bird', 'cat'] and ds.class_to_idx['cat'] == 2
    assert len(ds) == 9 and [t for _, t in ds.samples] == [0, 0, 0, 1, 1, 1, 2, 2, 2]
    assert [os.path.basename(p) for p, _ in ds.samples[:3]] == ['img_0.png', 'img_1.png', 'img_2.png']
    x, t = ds[4]
    assert x.dtype == torch.float32 and tuple(x.shape) == (3, 28, 28) and t == 1
    # independent computation of the same deterministic pipeline
    img = Image.open(ds.samples[4][0]).convert('RGB')
    w, h = img.size
    scale = int(28 * 8 / 7)
    if w < h:
        ow, oh = scale, int(scale * h / w)
    else:
        oh, ow = scale, int(scale * w / h)
    r = img.resize((ow, oh), Image.BILINEAR)
    left, top = int(round((ow - 28) / 2.0)), int(round((oh - 28) / 2.0))
    a = np.asarray(r.crop((left, top, left + 28, top + 28)), dtype=np.float32) / 255.0
    mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
    ref = torch.from_numpy(((a - mean) / std).transpose(2, 0, 1).copy())
    assert torch.allclose(x, ref, atol=1e-6)
    assert ca is not None


def test_random_resized_crop_invariants():
    from convnet_amd import data as D
    torch.manual_seed(3)
    rrc = D.RandomResizedCrop(24)
    w, h = 90, 60
    for _ in range(200):
        left, top, cw, ch = rrc.get_params(w, h)
        assert 0 <= left and 0 <= top and left + cw <= w and top + ch <= h and cw > 0 and ch > 0
        assert 0.08 * w * h * 0.9 <= cw * ch <= w * h
        assert 3.0 / 4.0 * 0.9 <= cw / float(ch) <= 4.0 / 3.0 * 1.1
    img = Image.fromarray(np.arange(60 * 90 * 3, dtype=np.uint32).reshape(60, 90, 3).astype(np.uint8))
    out = rrc(img)
    assert out.size == (24, 24)
    flip = D.RandomHorizontalFlip(p=1.0)
    assert np.array_equal(np.asarray(flip(img)), np.asarray(img)[:, ::-1])
    # seeded runs reproduce the same crops
    torch.manual_seed(11)
    a = [rrc.get_params(w, h) for _ in range(5)]
    torch.manual_seed(11)
    assert a == [rrc.get_params(w, h) for _ in range(5)]


def test_random_resized_crop_is_crop_then_resize():
    """torchvision.transforms.functional.resized_crop on a PIL image = img.crop(box) THEN .resize(size, BILINEAR)
    (/root/reference preprocess.py:73 uses transforms.RandomResizedCrop): the resampling window is clamped at the CROP's
    edges.  PIL's resize(box=...) reads beyond the box and differs on the outermost ring of the result - the form this
    module used until round 4."""
    from convnet_amd import data as D
    rng = np.random.RandomState(0)
    img = Image.fromarray((rng.rand(375, 500, 3) * 255).astype(np.uint8))
    rrc = D.RandomResizedCrop(224)
    for seed in (0, 1, 2):
        torch.manual_seed(seed)
        left, top, cw, ch = rrc.get_params(*img.size)
        torch.manual_seed(seed)
        got = np.asarray(rrc(img))
        want = np.asarray(img.crop((left, top, left + cw, top + ch)).resize((224, 224), Image.BILINEAR))
        assert np.array_equal(got, want)
        boxed = np.asarray(img.resize((224, 224), Image.BILINEAR, box=(left, top, left + cw, top + ch)))
        if (left, top, cw, ch) != (0, 0, 500, 375):
            assert not np.array_equal(boxed, want)       # the two forms really are different operations
            assert np.array_equal(boxed[8:-8, 8:-8], want[8:-8, 8:-8])   # ... on the border ring only


def test_augmentation_draws_follow_torchvisions_sequence():
    """The crop / flip draws are torchvision's, call for call (RandomResizedCrop.get_params: uniform_(scale),
    uniform_(log ratio), randint(top), randint(left); RandomHorizontalFlip: rand(1) < p), on torch's global
    generator.  torchvision is not installed here: the sequence below is what that published algorithm yields
    for torch.manual_seed(0) on a 500x375 image -- recomputed independently in this test from raw torch draws
    and pinned as recorded values."""
    import math
    from convnet_amd import data as D
    torch.manual_seed(0)
    rrc = D.RandomResizedCrop(224)
    got = [rrc.get_params(500, 375) for _ in range(4)]
    flips = [bool(torch.rand(1) < 0.5) for _ in range(4)]
    assert got == [(80, 17, 343, 294), (183, 52, 271, 251), (182, 26, 318, 295), (56, 28, 125, 151)]   # (left, top, w, h)
    assert flips == [False, False, True, True]
    # the same numbers from the raw draw order, written out once more without the class
    torch.manual_seed(0)
    lr = torch.log(torch.tensor((3.0 / 4.0, 4.0 / 3.0)))
    exp = []
    while len(exp) < 4:
        area = 500 * 375 * torch.empty(1).uniform_(0.08, 1.0).item()
        ar = torch.exp(torch.empty(1).uniform_(lr[0], lr[1])).item()
        cw, ch = int(round(math.sqrt(area * ar))), int(round(math.sqrt(area / ar)))
        if 0 < cw <= 500 and 0 < ch <= 375:
            top = torch.randint(0, 375 - ch + 1, size=(1,)).item()
            left = torch.randint(0, 500 - cw + 1, size=(1,)).item()
            exp.append((left, top, cw, ch))
    assert exp == got


def test_data_regime_loader_and_epoch_keyed_settings(tmp_path):
    from convnet_amd import data as D
    _make_folder(str(tmp_path))
    regime = [{'epoch': 0, 'input_size': 16}, {'epoch': 2, 'input_size': 24, 'batch_size': 2}]
    dr = D.DataRegime(regime, defaults={'datasets_path': str(tmp_path), 'name': 'imagenet', 'split': 'train',
                                        'augment': True, 'input_size': None, 'batch_size': 4, 'shuffle': True,
                                        'num_workers': 0, 'pin_memory': False, 'drop_last': True})
    assert len(dr) == 9
    torch.manual_seed(0)
    batches = list(dr.get_loader())
    assert len(batches) == 2
    x, t = batches[0]
    assert tuple(x.shape) == (4, 3, 16, 16) and x.dtype == torch.float32 and t.dtype == torch.int64
    assert int(t.min()) >= 0 and int(t.max()) <= 2
    dr.set_epoch(2)
    x, t = next(iter(dr.get_loader()))
    assert tuple(x.shape) == (2, 3, 24, 24)            # the epoch-2 phase of the regime took over
    assert dr.get('input_size') == 24


def test_cli_trains_on_an_image_folder(tmp_path):
    """`main.py --dataset imagenet --datasets-dir ...` end to end on a tiny folder (emulator)."""
    from conftest import HAS_GPU
    if HAS_GPU:
        pytest.skip('emulator CLI run is for GPU-less hosts')
    _make_folder(str(tmp_path), per_class=4)
    from convnet_amd.main import main
    out = main(['--dataset', 'imagenet', '--datasets-dir', str(tmp_path), '--model', 'resnet',
                '--model-config', "{'depth': 18, 'width': [8, 16, 32, 64], 'inplanes': 8, 'num_classes': 3}",
                '--input-size', '32', '-b', '4', '--epochs', '1', '--workers', '0', '--device', 'cpu',
                '--dtype', 'float', '--results-dir', str(tmp_path / 'results'), '--save', 'run', '--print-freq', '1'])
    assert out is not None
    ck = torch.load(tmp_path / 'results' / 'run' / 'checkpoint.pth.tar', map_location='cpu')
    assert ck['epoch'] == 1 and 'conv1.weight' in ck['state_dict']


@pytest.mark.gpu
def test_folder_pipeline_feeds_the_gpu_trainer(tmp_path):
    """Image folder -> worker processes -> pinned fp32 NCHW batches -> copy-stream prefetch -> fused
    layout cast -> two bf16 training steps on the MI355X (finite loss, parameters move)."""
    import convnet_amd as ca
    from convnet_amd import data as D
    assert not ca._lib.is_emulated()
    _make_folder(str(tmp_path), per_class=6, size=(70, 64))
    dr = D.DataRegime(None, defaults={'datasets_path': str(tmp_path), 'name': 'imagenet', 'split': 'train',
                                      'augment': True, 'input_size': 64, 'batch_size': 8, 'shuffle': True,
                                      'num_workers': 2, 'pin_memory': True, 'drop_last': True})
    torch.manual_seed(123)
    model = ca.models.resnet(dataset='imagenet', depth=18, num_classes=8)
    tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device='cuda:0',
                    dtype=torch.bfloat16, print_freq=10 ** 9)
    w0 = model.fc.weight.detach().float().cpu().clone()
    res = tr.train(dr.get_loader())
    assert res['loss'] == res['loss'] and 0 < res['loss'] < 20       # finite
    assert not torch.equal(model.fc.weight.detach().float().cpu(), w0)
    val = tr.validate(dr.get_loader())
    assert val['loss'] == val['loss']


@pytest.mark.gpu
def test_device_side_totensor_normalize_on_the_gpu(tmp_path):
    """The same on the MI355X (cn_u8_nhwc_to_nchw_lut on the copy stream, pinned uint8 batches, two workers), plus a
    training pass fed that way: the loss equals the host-normalised run's, bit for bit."""
    import convnet_amd as ca
    from convnet_amd import data as D
    assert not ca._lib.is_emulated()
    test_device_side_totensor_normalize_is_bit_identical(tmp_path)
    _make_folder(str(tmp_path / 'f'), per_class=6, size=(70, 64))
    losses = []
    for dn in (False, True):
        torch.manual_seed(7)
        dr = D.DataRegime(None, defaults={'datasets_path': str(tmp_path / 'f'), 'name': 'imagenet', 'split': 'train',
                                          'augment': True, 'input_size': 64, 'batch_size': 8, 'shuffle': False,
                                          'num_workers': 0, 'pin_memory': True, 'drop_last': True, 'device_normalize': dn})
        torch.manual_seed(123)
        model = ca.models.resnet(dataset='imagenet', depth=18, num_classes=8)
        tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, model.regime), device='cuda:0',
                        dtype=torch.bfloat16, print_freq=10 ** 9)
        torch.manual_seed(11)
        losses.append(tr.train(dr.get_loader())['loss'])
    assert losses[0] == losses[1] and 0 < losses[0] < 20, losses


def test_device_side_totensor_normalize_is_bit_identical(tmp_path):
    """`device_normalize` (DataRegime setting; main.py default, `--host-normalize` turns it off): the workers stop at the
    uint8 HWC crop, trainer.DevicePrefetcher runs ToTensor + Normalize behind the copy (ops.u8_nhwc_to_nchw: a per-channel
    table of the reference's own fp32 arithmetic, preprocess.py:23-25).  The batches the step receives are the host
    pipeline's, bit for bit - train (same crops / flips under the same seed) and eval."""
    from convnet_amd import data as D
    import convnet_amd as ca
    root = tmp_path / 'ds'
    rng = np.random.RandomState(1)
    for split in ('train', 'val'):
        for c in range(3):
            d = root / 'imagenet' / split / ('c%d' % c)
            d.mkdir(parents=True)
            for i in range(4):
                a = (rng.rand(40 + 7 * i, 50 + 5 * c, 3) * 255).astype(np.uint8)
                Image.fromarray(a).save(str(d / ('%d.png' % i)))
    dev = torch.device('cpu')        # the emulator build serves host tensors; on a GPU box the same test runs on cuda:0
    if torch.cuda.is_available() and not ca._lib.is_emulated():
        dev = torch.device('cuda', 0)
    for split, augment in (('train', True), ('val', False)):
        got = {}
        for dn in (False, True):
            torch.manual_seed(5)
            dr = D.DataRegime([{'epoch': 0}], defaults={'datasets_path': str(root), 'name': 'imagenet', 'split': split,
                                                         'augment': augment, 'input_size': 32, 'batch_size': 4,
                                                         'shuffle': False, 'num_workers': 0, 'drop_last': False,
                                                         'device_normalize': dn})
            loader = dr.get_loader()
            assert (loader.device_normalize is not None) == dn
            first = next(iter(loader))[0]
            assert first.dtype == (torch.uint8 if dn else torch.float32)
            assert tuple(first.shape) == ((4, 32, 32, 3) if dn else (4, 3, 32, 32))
            torch.manual_seed(5)
            got[dn] = [(x.cpu().clone(), t.cpu().clone()) for x, t in ca.trainer.DevicePrefetcher(loader, dev)]
        assert len(got[False]) == len(got[True]) == 3
        for (x0, t0), (x1, t1) in zip(got[False], got[True]):
            assert x1.dtype == torch.float32 and torch.equal(x0, x1) and torch.equal(t0, t1)
    # uint8 batches without the loader's mean / std are refused, not guessed
    with pytest.raises(ValueError):
        list(ca.trainer.DevicePrefetcher([(torch.zeros(1, 4, 4, 3, dtype=torch.uint8), torch.zeros(1, dtype=torch.long))], dev))


def _device_for_data():
    import convnet_amd as ca
    if torch.cuda.is_available() and not ca._lib.is_emulated():
        return torch.device('cuda', 0)
    return torch.device('cpu')


def test_device_resize_matches_pil_bit_for_bit():
    """Round 6 (csrc/resize.hip + data.resample_table): PIL's 8-bit BILINEAR resize is fixed-point arithmetic on coefficients
    computed in double; the tables are computed by PIL's recipe on the host, the two integer passes run on the device.
    Every byte equals Image.resize's, for down-scaling with antialiasing support, up-scaling, identity and degenerate
    sizes, with and without the horizontal flip that follows the resize in the training transform."""
    import convnet_amd as ca
    from convnet_amd import data as D
    dev = _device_for_data()
    rng = np.random.RandomState(3)
    S = 64
    sizes = [(70, 90), (64, 64), (20, 33), (200, 64), (64, 150), (3, 5), (131, 257), (500, 375)]
    batch, ref = [], []
    for i, (h, w) in enumerate(sizes):
        a = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        c = D.CropForDevice(a, D.resample_table(w, S), D.resample_table(h, S), S)
        c.flip = bool(i % 2)
        batch.append((c, i))
        r = Image.fromarray(a).resize((S, S), Image.BILINEAR)
        if c.flip:
            r = r.transpose(Image.FLIP_LEFT_RIGHT)
        ref.append(np.array(r))
    inputs, target = D.collate_crops(batch)
    assert target.tolist() == list(range(len(sizes)))
    out = ca.ops.resize_crops({k: (v if k == 'size' else v.to(dev)) for k, v in inputs.items()}).cpu().numpy()
    for i, r in enumerate(ref):
        assert np.array_equal(out[i], r), (sizes[i], int(np.abs(out[i].astype(int) - r.astype(int)).max()))


@pytest.mark.gpu
def test_device_resize_matches_pil_bit_for_bit_gpu():
    test_device_resize_matches_pil_bit_for_bit()


def test_device_resize_loader_batches_are_the_host_pipelines(tmp_path):
    """`device_resize` (DataRegime setting; main.py --device-resize): the workers
    stop at the uint8 CROP of RandomResizedCrop (train) / the source window of Resize + CenterCrop (eval); trainer.
    DevicePrefetcher resizes, flips and normalises behind the copy.  The fp32 NCHW batches the step receives are the host
    pipeline's bit for bit - same crops and flips under the same seed - for both transforms."""
    from convnet_amd import data as D
    import convnet_amd as ca
    root = tmp_path / 'ds'
    rng = np.random.RandomState(2)
    for split in ('train', 'val'):
        for c in range(3):
            d = root / 'imagenet' / split / ('c%d' % c)
            d.mkdir(parents=True)
            for i in range(4):
                a = (rng.rand(60 + 17 * i, 90 - 11 * c, 3) * 255).astype(np.uint8)
                Image.fromarray(a).save(str(d / ('%d.png' % i)))
    dev = _device_for_data()
    for split, augment in (('train', True), ('val', False)):
        got = {}
        for dr_on in (False, True):
            torch.manual_seed(9)
            reg = D.DataRegime([{'epoch': 0}], defaults={'datasets_path': str(root), 'name': 'imagenet', 'split': split,
                                                          'augment': augment, 'input_size': 32, 'batch_size': 4,
                                                          'shuffle': False, 'num_workers': 0, 'drop_last': False,
                                                          'device_normalize': True, 'device_resize': dr_on})
            loader = reg.get_loader()
            first = next(iter(loader))[0]
            assert isinstance(first, dict) == dr_on
            torch.manual_seed(9)
            got[dr_on] = [(x.cpu().clone(), t.cpu().clone()) for x, t in ca.trainer.DevicePrefetcher(loader, dev)]
        assert len(got[False]) == len(got[True]) == 3
        for (x0, t0), (x1, t1) in zip(got[False], got[True]):
            assert x1.dtype == torch.float32 and tuple(x1.shape[1:]) == (3, 32, 32)
            assert torch.equal(x0, x1) and torch.equal(t0, t1)
    with pytest.raises(ValueError):
        D.get_transform('imagenet', input_size=32, device_resize=True, device_normalize=False)


@pytest.mark.gpu
def test_device_resize_loader_on_the_gpu(tmp_path):
    """The same through two worker processes and pinned buffers onto the MI355X."""
    test_device_resize_loader_batches_are_the_host_pipelines(tmp_path)
