"""Pins the gfx950 lane maps the kernels assume (cn_common.h): MFMA 32x32x16 bf16 / 32x32x2 f32
operand + accumulator layout and the ds_read_b64_tr_b16 shuffle.  On the emulator this only checks
the harness; on the GPU it checks the hardware."""
import pytest
import torch

from conftest import HAS_GPU

MODES = [pytest.param('emul'), pytest.param('gpu', marks=pytest.mark.gpu)]


def _setup(mode):
    if mode == 'emul' and HAS_GPU:
        pytest.skip('emulator mode is for GPU-less hosts')
    if mode == 'gpu' and not HAS_GPU:
        pytest.skip('no GPU')
    import convnet_amd as ca
    dev = torch.device('cuda', 0) if mode == 'gpu' else torch.device('cpu')
    stream = torch.cuda.current_stream().cuda_stream if mode == 'gpu' else None
    return ca, dev, stream


@pytest.mark.parametrize('mode', MODES)
def test_mfma_bf16_lane_map(mode):
    ca, dev, stream = _setup(mode)
    g = torch.Generator().manual_seed(0)
    A = torch.randint(-4, 5, (32, 16), generator=g).float()     # asymmetric, exactly representable
    B = torch.randint(-4, 5, (16, 32), generator=g).float()
    Ab = A.to(torch.bfloat16).view(torch.int16).to(dev)
    Bb = B.to(torch.bfloat16).view(torch.int16).to(dev)
    D = torch.zeros(32, 32, device=dev)
    ca._lib.check(ca._lib.load().cn_probe_mfma_bf16(Ab.data_ptr(), Bb.data_ptr(), D.data_ptr(), stream))
    assert torch.equal(D.cpu(), A @ B)


@pytest.mark.parametrize('mode', MODES)
def test_mfma_f16_lane_map(mode):
    ca, dev, stream = _setup(mode)
    g = torch.Generator().manual_seed(3)
    A = torch.randint(-4, 5, (32, 16), generator=g).float()
    B = torch.randint(-4, 5, (16, 32), generator=g).float()
    Ah = A.to(torch.float16).view(torch.int16).to(dev)
    Bh = B.to(torch.float16).view(torch.int16).to(dev)
    D = torch.zeros(32, 32, device=dev)
    ca._lib.check(ca._lib.load().cn_probe_mfma_f16(Ah.data_ptr(), Bh.data_ptr(), D.data_ptr(), stream))
    assert torch.equal(D.cpu(), A @ B)


@pytest.mark.parametrize('mode', MODES)
def test_mfma_f32_lane_map(mode):
    ca, dev, stream = _setup(mode)
    g = torch.Generator().manual_seed(1)
    A = torch.randint(-4, 5, (32, 2), generator=g).float()
    B = torch.randint(-4, 5, (2, 32), generator=g).float()
    D = torch.zeros(32, 32, device=dev)
    Ad, Bd = A.to(dev), B.to(dev)   # keep the device copies alive across the launch
    ca._lib.check(ca._lib.load().cn_probe_mfma_f32(Ad.data_ptr(), Bd.data_ptr(), D.data_ptr(), stream))
    assert torch.equal(D.cpu(), A @ B)


@pytest.mark.parametrize('mode', MODES)
def test_lds_transpose_read_map(mode):
    ca, dev, stream = _setup(mode)
    src = torch.arange(256, dtype=torch.int16).to(dev)
    out = torch.zeros(64, 4, dtype=torch.int16, device=dev)
    ca._lib.check(ca._lib.load().cn_probe_tr16(src.data_ptr(), out.data_ptr(), stream))
    exp = torch.zeros(64, 4, dtype=torch.int16)
    for l in range(64):
        g0, i = l & ~15, l & 15
        for j in range(4):
            src_lane = g0 + 4 * j + (i >> 2)        # in[L][e] = 4*L + e for the linear 8-byte/lane image
            exp[l, j] = 4 * src_lane + (i & 3)
    got = out.cpu()
    assert torch.equal(got, exp), 'ds_read_b64_tr_b16 map differs:\n%s' % got[:16].tolist()


@pytest.mark.parametrize('mode', MODES)
def test_mfma_i8_lane_map(mode):
    """v_mfma_i32_32x32x32_i8 as csrc/qconv_i8.hip assumes it: lane l holds A[l & 31][16*(l >> 5) + e], e = 0..15."""
    ca, dev, stream = _setup(mode)
    g = torch.Generator().manual_seed(2)
    A = torch.randint(-128, 128, (32, 32), generator=g, dtype=torch.int32)     # asymmetric, full int8 range
    B = torch.randint(-128, 128, (32, 32), generator=g, dtype=torch.int32)
    Ad, Bd = A.to(torch.int8).to(dev), B.to(torch.int8).to(dev)
    D = torch.zeros(32, 32, dtype=torch.int32, device=dev)
    ca._lib.check(ca._lib.load().cn_probe_mfma_i8(Ad.data_ptr(), Bd.data_ptr(), D.data_ptr(), stream))
    assert torch.equal(D.cpu(), A @ B)


@pytest.mark.gpu
@pytest.mark.parametrize('how', ['fork', 'mark'])
def test_cross_stream_hand_off_orders_the_consumer(how):
    """cn_stream_fork (event ring) and cn_stream_arm / cn_stream_wait_mark (the producer kernel's own completion
    event): a kernel on the second stream that is handed a tensor must see what the long producer kernel on the
    first stream wrote, every time (stale reads would show the previous round's value)."""
    ca, dev, _ = _setup('gpu')
    L, lib = ca._lib.load(), ca._lib
    s1, s2 = torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev)
    n = 1 << 27                                   # 512 MB: the producer runs for ~100 us
    a = torch.zeros(n, dtype=torch.float32, device=dev)
    out = torch.zeros(n, dtype=torch.float32, device=dev)
    zero = torch.zeros(n, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    for it in range(1, 13):
        if how == 'mark':
            h = L.cn_stream_arm()
            lib.check(L.cn_fill_f32(a.data_ptr(), n, float(it), s1.cuda_stream), 'fill')
            assert L.cn_stream_disarm() == 1 and h >= 0
            lib.check(L.cn_stream_wait_mark(h, s2.cuda_stream), 'wait_mark')
        else:
            lib.check(L.cn_fill_f32(a.data_ptr(), n, float(it), s1.cuda_stream), 'fill')
            lib.check(L.cn_stream_fork(s1.cuda_stream, s2.cuda_stream), 'fork')
        # out = relu(a + 0) on the second stream (cn_eltwise op 4)
        lib.check(L.cn_eltwise(4, out.data_ptr(), a.data_ptr(), zero.data_ptr(), n, lib.F32, s2.cuda_stream), 'eltwise')
        lib.check(L.cn_stream_fork(s2.cuda_stream, s1.cuda_stream), 'join')   # the next fill must not overtake the read
        s2.synchronize()
        assert float(out.min()) == float(it) and float(out.max()) == float(it), (how, it)
    torch.cuda.synchronize()
