"""Round 5, "lazy dy1" at the model level: with ops.LAZY_DY1 the inner BatchNorm behind every block's conv1 (bn1) runs its
backward reduction + finalize only and conv1's streaming junction kernel forms dy1 on its operand path
(cn_conv2d_dgrad_junction_lazy) - a full-width ResNet-50 trains to the same numbers, bit for bit, as with the apply pass, and
the lazy path really served every block it can serve (all of layer1 / layer2 but their first blocks' ... see the count).
Replaces the bn1 / conv1 backward pair of /root/reference models/resnet.py:141-147 (run in reverse by trainer.py:162)."""
import pytest
import torch

from helpers import warm_bn_state

MODES = ['emul', pytest.param('gpu', marks=pytest.mark.gpu)]


def _run(ca, dev, lazy1, dtype, N, size, steps, width=(64, 128, 256, 512)):
    saved = (ca.ops.LAZY_DY1, ca.ops.LAZY_DY1_MIN_MB)
    ca.ops.LAZY_DY1, ca.ops.LAZY_DY1_MIN_MB = lazy1, 0.0
    for k in list(ca.ops.COUNTERS):
        ca.ops.COUNTERS[k] = 0
    try:
        torch.manual_seed(123)
        model = ca.models.resnet(depth=50, num_classes=16, width=width)
        warm_bn_state(model, 7)
        regime = [dict(r) for r in model.regime]
        for r in regime:
            if 'lr' in r:
                r['lr'] = r['lr'] * 0.05
        tr = ca.Trainer(model, ca.CrossEntropyLoss(), ca.OptimRegime(model, regime), device=str(dev), dtype=dtype,
                        print_freq=10 ** 9)
        tr._use_graph = False
        g = torch.Generator().manual_seed(5)
        recs = []
        for _ in range(steps):
            r = tr.train([(torch.randn(N, 3, size, size, generator=g), torch.randint(0, 16, (N,), generator=g))])
            recs.append((r['loss'], r['prec1']))
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        state = {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items()}
        state['__momentum'] = tr.optimizer.momentum_buf.detach().cpu().clone()
        return recs, state, dict(ca.ops.COUNTERS)
    finally:
        ca.ops.LAZY_DY1, ca.ops.LAZY_DY1_MIN_MB = saved


@pytest.mark.parametrize('mode', MODES)
def test_lazy_dy1_training_is_bit_identical(mode):
    import convnet_amd as ca
    if mode == 'emul':
        if not ca._lib.is_emulated():
            pytest.skip('the HIP library is bound (GPU box): the emulator case runs in the CPU suite')
        # (full width where the streaming junction kernel lives - the first two stages -, thin behind: a short emulated run)
        dev, N, size, steps, width = torch.device('cpu'), 2, 32, 1, (64, 128, 8, 8)
    else:
        assert not ca._lib.is_emulated()
        dev, N, size, steps, width = torch.device('cuda', 0), 16, 64, 3, (64, 128, 256, 512)
    r0, s0, c0 = _run(ca, dev, False, torch.bfloat16, N, size, steps, width)
    r1, s1, c1 = _run(ca, dev, True, torch.bfloat16, N, size, steps, width)
    assert c0.get('bn_bwd_lazy1', 0) == 0 and c0.get('jdgrad_lazy1', 0) == 0
    # blocks whose conv1 data gradient is the K <= 128 streaming junction kernel: layer1.1, layer1.2, layer2.0 .. layer2.3
    assert c1.get('bn_bwd_lazy1', 0) == 6 * steps, c1
    assert c1.get('jdgrad_lazy1', 0) == 6 * steps and c1.get('lazy_dy1_fallback', 0) == 0, c1
    assert all(l == l for l, _ in r0)
    assert r0 == r1, (r0, r1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
