"""Functional wrappers + autograd Functions over the HIP kernels (C ABI in include/convnet_hip.h).

Tensor conventions of this layer (not of the user-facing modules, which keep the reference's
NCHW / OIHW shapes at their boundary): activations are contiguous NHWC tensors in the compute dtype
(bf16 or fp32), filters are KRSC, parameters / gradients / statistics are fp32.

PyTorch is used here for allocation, stream handles and the autograd tape only; every arithmetic
pass over tensor data is one of the `cn_*` kernels.
"""
import collections
import contextlib
import os
import weakref

import torch
import torch.distributed as dist
from torch.autograd import Function

from . import _lib, flags
from ._lib import check, dtype_code, ptr, stream_of

# ---------------------------------------------------------------------------------------------
# workspaces: one grow-only scratch buffer per (device, tag, stream).  Kernels on one stream are ordered,
# so reuse across ops on that stream is safe; keying by stream keeps a buffer from being shared (or, when
# it is replaced by a bigger one, recycled by the caching allocator) across streams without an event.
_WS = {}


def workspace(nbytes, device, tag='main'):
    dev = torch.device(device)
    sid = torch.cuda.current_stream(dev).cuda_stream if dev.type == 'cuda' else 0
    key = (str(dev), tag, sid)
    buf = _WS.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        n = max(int(nbytes + 3) // 4, 1024)
        old = buf
        buf = torch.empty(n, dtype=torch.float32, device=dev)
        if old is not None and dev.type == 'cuda':
            old.record_stream(torch.cuda.current_stream(dev))   # kernels still reading the old block finish first
        _WS[key] = buf
    return buf


def _L():
    return _lib.load()


class KernelProfiler(object):
    """Optional per-call HIP-event timing of the kernel wrappers (used by bench.py for the live
    roofline numbers).  Events are recorded on torch's current stream, which is the stream every
    kernel of this package is launched on.  Disabled by default: zero overhead in the hot loop."""

    def __init__(self):
        self.enabled = False
        self.records = []
        # True: the weight-gradient side stream is folded into the main stream while recording, so every kernel
        # runs ALONE (per-kernel / per-layer tables).  False: the step keeps its two-stream schedule and the events
        # sit on whichever stream the call is launched on: durations of the OVERLAPPED step, i.e. what rocprofv3
        # reports for the timed region (used to choose and price the dominant kernel).
        self.fold_streams = True

    def run(self, name, launches, flops, nbytes, fn, device, detail=None):
        """name: label, or a callable evaluated AFTER fn() (the library reports which kernel instantiation(s) its
        dispatcher picked: cn_kernel_log); detail: optional per-shape label (conv layers).  A call that launched
        several DIFFERENT GEMM instantiations (a strided dgrad: one launch per parity class) is recorded under the
        joined label 'a + b' with its true launch count - never under the last launch's name."""
        if not self.enabled or device.type != 'cuda':
            return fn()
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        if callable(name):
            _L().cn_kernel_log(1)
        s.record()
        out = fn()
        e.record()
        if callable(name):
            name = name()
        # (which stream the call was launched on: the weight-gradient side stream's launches are told apart)
        self.records.append((name, launches, flops, nbytes, s, e, detail, torch.cuda.current_stream(device).cuda_stream))
        return out

    def summary(self, by_detail=False):
        torch.cuda.synchronize()
        agg = {}
        # the stream the step runs along = the one most calls were launched on (the trainer's own high-priority stream);
        # everything else is the weight-gradient side stream
        cnt = {}
        for r in self.records:
            cnt[r[7]] = cnt.get(r[7], 0) + 1
        main = max(cnt, key=cnt.get) if cnt else None
        for name, launches, flops, nbytes, s, e, detail, stream in self.records:
            if by_detail and detail is None:
                continue
            a = agg.setdefault(detail if by_detail else name,
                               {'calls': 0, 'launches': 0, 'ms': 0.0, 'flops': 0.0, 'bytes': 0.0, 'records': [], 'side_ms': 0.0})
            ms = s.elapsed_time(e)
            if main is not None and stream != main:
                a['side_ms'] += ms
            a['calls'] += 1
            a['launches'] += launches
            a['ms'] += ms
            a['flops'] += flops
            a['bytes'] += nbytes
            a['records'].append((ms, flops, nbytes))
        return agg


PROFILER = KernelProfiler()
# The switches below mirror flags.py (ONE table: defaults, meaning, how to override for an A/B); tests flip these module
# attributes to compare a fused path with the one it replaces.
# 0: the stride-2 projection shortcut's input gradient as a dense tensor
SUBSAMPLED_SHORTCUT_GRAD = flags.on('subsampled_shortcut_grad')
# 0: the fused stem's BatchNorm-backward sums over the input map with the pool gather
STEM_XMAX = flags.on('stem_xmax')
# 0: plain sum / sum-of-squares statistics partials; 1: centred on the running mean for fp32 models; all: every dtype
CENTERED_STATS = 'all' if flags.text('centered_stats') == 'all' else flags.on('centered_stats')


class SideStream(object):
    """Optional second HIP stream for the weight-gradient kernels: wgrad(x, dy) and dgrad(dy, w) of a
    layer are independent, and the late layers' launches are too small to fill 256 CUs on their
    own, so running wgrad beside the main backward stream packs the machine better.
    Measured +5.2 % on ResNet-50 b=256 (10.04k -> 10.56k img/s).  On by default (flag wgrad_stream).  Trainer /
    BucketReducer join the stream before the gradient all-reduce and the optimizer step; while the KernelProfiler is
    recording in "alone" mode everything stays on one stream.  ONE side stream at the runtime's default priority, all
    CUs: two or three streams, a high-priority side stream and CU masks were measured slower or neutral
    (profiles/r03_ab_whole_step_knobs.txt, profiles/r04_ab_wgrad_reduce_lanes_and_side_streams_rejected.txt)."""

    def __init__(self):
        self.enabled = flags.on('wgrad_stream')
        self._streams = {}
        self.used = False
        self.marks = flags.on('marks')     # 0: always hand off with an event record
        self.capturing = False      # set by Trainer around HIP-graph capture
        self.hold = flags.on('side_hold')
        self._held = collections.deque()
        # a caller that runs backward passes without ever joining (anything but Trainer / BucketReducer) must not pin
        # every operand for ever: beyond this many pending hand-offs the oldest go back to the allocator the
        # record_stream way (correct without a join, just slower).  ResNet-200 with 8 accumulation chunks stays below.
        self.hold_max = 2048
        # ... nor more than this many bytes of them (a count alone lets thousands of activation-sized operands pile up:
        # one ResNet-50 b=256 step holds ~20 GB)
        self.hold_max_bytes = 64 * 2 ** 30
        self._held_bytes = 0
        self._mark = None

    def get(self, device):
        """The side stream the weight-gradient launches go to."""
        st = self._streams.get(device)
        if st is None:
            st = self._streams[device] = torch.cuda.Stream(device)
        return st

    @contextlib.contextmanager
    def mark(self, dy):
        """Around the library call that produces the gradient tensor `dy`: its last kernel signals a mark event on
        completion (cn_stream_arm), so the weight-gradient launch that consumes `dy` can wait for exactly that
        kernel (hipExtLaunchKernel's stop event) instead of an event recorded behind it in the chain's queue."""
        self._mark = None
        if not (self.marks and self.active(dy)):
            yield
            return
        h = _L().cn_stream_arm()
        try:
            yield
        finally:
            if _L().cn_stream_disarm() == 1 and h >= 0:
                self._mark = (dy.data_ptr(), h, dy)   # dy held: its memory cannot be recycled under the mark

    def submit(self, device, launch, notify, dy=None):
        """Run `launch()` (returns the tensors it reads / scratch it uses) on the side stream once `dy` - and
        everything else queued on the current stream so far - is ready, then `notify()`."""
        cur = torch.cuda.current_stream(device)
        side = self.get(device)
        mark, self._mark = self._mark, None
        if mark is not None and dy is not None and dy.data_ptr() == mark[0]:
            check(_L().cn_stream_wait_mark(mark[1], side.cuda_stream), 'cn_stream_wait_mark')   # dy's producer is done
        else:
            self.fork(cur, side)
        with torch.cuda.stream(side):
            held = launch()
        if self.hold:
            # the operands stay referenced until the chain has waited for the side stream (join, once per step): the
            # caching allocator then needs no cross-stream bookkeeping for them.  record_stream would make it record an
            # event (with torch's default flags: a system-scope fence) on the side stream at every free and poll it
            nbytes = sum(t.numel() * t.element_size() for t in held)
            self._held.append((held, nbytes))
            self._held_bytes += nbytes
            while len(self._held) > self.hold_max or (self._held_bytes > self.hold_max_bytes and len(self._held) > 1):
                old, ob = self._held.popleft()
                self._held_bytes -= ob
                for t in old:
                    t.record_stream(side)
        else:
            for t in held:
                t.record_stream(side)
        self.used = True
        notify()

    def gather(self, device):
        """The stream a collective (or a join) has to order itself behind."""
        return self.get(device)

    def active(self, t):
        # not while a HIP graph is being captured: a replayed graph runs faster as one chain (b=8 +4 %, b=32 +3 %,
        # b=64 +2 %, b=128 +1 %; the runtime's graph queues put the chain behind a weight gradient every few layers)
        return self.enabled and t.is_cuda and not (PROFILER.enabled and PROFILER.fold_streams) and not self.capturing

    def fork(self, cur, side):
        """`side` waits for everything queued on `cur` so far (one device-scope event from the library's ring)."""
        check(_L().cn_stream_fork(cur.cuda_stream, side.cuda_stream), 'cn_stream_fork')

    def join(self, device):
        """Make the current stream wait for everything queued on the side stream(s).  MANDATORY once per step for whoever
        drives backward passes of this package by hand (Trainer._body and BucketReducer.finish do it): the optimizer
        step must not read weight gradients the side stream is still writing, and the operands held for it are released
        here."""
        if self.used and device.type == 'cuda':
            # the library's device-scope ring event, as for the hand-offs in the other direction (torch's wait_stream
            # records an event with the default flags: a system-scope fence on the side stream)
            self.fork(self.gather(device), torch.cuda.current_stream(device))
            self.used = False
        self._held.clear()     # (what the current stream does from here on is ordered behind the side stream's reads)
        self._held_bytes = 0


SIDE = SideStream()


def _esize(t):
    return t.element_size()


def _last_kernel(suffix=''):
    """Label of the GEMM-class kernel(s) the library's dispatcher launched in the call just made (cn_kernel_log):
    one name, or the distinct names joined with ' + ' when the call's launches used different instantiations."""
    def label():
        names = [n for n in _L().cn_kernel_log(0).decode().split(';') if n]
        if not names:
            names = [_L().cn_last_kernel_name().decode()]
        uniq = []
        for n in names:
            if n not in uniq:
                uniq.append(n)
        return ' + '.join(uniq) + suffix
    return label


def _conv_detail(kind, C, H, K, R, stride):
    return '%s %d,%d->%d %dx%d/%d' % (kind, C, H, K, R, R, stride[0])


def conv_out_hw(H, W, R, S, stride, pad):
    return (H + 2 * pad[0] - R) // stride[0] + 1, (W + 2 * pad[1] - S) // stride[1] + 1


# ---------------------------------------------------------------------------------------------
# raw (non-autograd) kernel calls

# 0 = every BatchNorm re-reads its input for the statistics (bn_stats_kernel)
FUSE_BN_STATS = flags.on('fuse_bn_stats')
# 0 = the stem runs as a 49-tap conv on a 3->8 channel padded image instead of the pixel-pair form
STEM_PAIRS = flags.on('stem_pairs')
# 0 = the pixel-pair stem runs through the tiled implicit-GEMM kernel instead of the halo kernel (csrc/stem.hip)
STEM_HALO = flags.on('stem_halo')
# 0 = the 64 -> 64 channel 3x3 convolutions run through the tiled implicit-GEMM kernel instead of the halo kernel
# (csrc/conv3x3.hip)
CONV3X3_HALO = flags.on('conv3x3_halo')
# 0 = conv3 / the stride-1 projection forward through the tiled kernel instead of the streaming kernel
CONV1X1_STREAM = flags.on('conv1x1_stream')
# 0 = the stem's bn1 -> relu -> maxpool runs as separate BatchNorm and max-pool passes
FUSE_STEM_POOL = flags.on('fuse_stem_pool')
# 0 = BatchNorm backward always runs its own reduction pass over (dz, y).  The fusion is used at the residual JUNCTIONS
# only (the block-input dgrad's epilogue adds the other branch's gradient anyway and no dres tensor is written); at the
# BatchNorms inside a block the epilogue's extra operand stream costs the producing kernel more than the separate pass
# costs (profiles/r01_epilogue_fusions_per_layer.txt; whole step: junctions only 21.34 ms vs everywhere 21.71 ms).
FUSE_BN_BWD = flags.on('fuse_bn_bwd')
# "Lazy dy" (round 3): at a residual junction whose BatchNorm follows a 1x1 convolution (bn3(conv3(.)), the projection
# shortcut's BatchNorm) the backward apply pass  dy = c1*g + c2*y + c3  is not run: the convolution's dgrad and wgrad
# form dy on their operand loads (cn_conv2d_dgrad_lazy / cn_conv2d_wgrad_lazy) - one write and one read of the
# largest tensors of the step less on the critical chain.  For BN inputs that every pass streams from HBM
# (flags.LAZY_MIN_BYTES: 0.6 of the Infinity Cache; the smaller maps keep the LDS-DMA weight-gradient kernels).
LAZY_DY = flags.on('lazy_dy')
LAZY_DY_MIN_MB = float(flags.get('lazy_min_mb'))
# Junction behind a projection shortcut (round 3): the shortcut BatchNorm only finalises its statistics; its apply runs
# inside the junction BatchNorm's apply pass (cn_bn_apply_dual), so the normalised shortcut tensor is neither written nor
# re-read.  Bit-identical.
DUAL_BN = flags.on('dual_bn')
# "Lazy z" (round 3): a residual junction whose output feeds a 1x1 / stride-1 convolution of at most 128 output channels
# (the next bottleneck's conv1) is finalised but not applied; that convolution forms z = relu(bn(y) + residual) on its
# operand load and stores it (cn_conv2d_fwd_lazyz): the junction's apply pass (read y, read residual, write z) and the
# convolution's re-read of z become one read of y and the residual and one write of z.  Bit-identical.  Only inside a
# model forward that guarantees the convolution runs next (LAZY_Z_SCOPE, set by ResNetImagenet.features) and for
# junction tensors of at least LAZY_Z_MIN_MB (same size rule as lazy dy).
LAZY_Z = flags.on('lazy_z')
LAZY_Z_MIN_MB = float(flags.get('lazy_min_mb'))
LAZY_Z_SCOPE = [0]
# Junction pair (round 3): where lazy dy applies and the convolution has an instantiated shape (64 -> 256 channels: conv3
# and the projection of ResNet-50's first stage), its data gradient and weight gradient run as ONE kernel on the backward
# chain (cn_conv2d_bwd1x1_lazy): g and y are read once instead of twice.  dx bit-identical, dW differs by fp32
# summation order.
JPAIR = flags.on('jpair')
# Streaming junction kernel (round 3): the fused junction data gradient (conv1's dgrad + shortcut gradient + ReLU mask +
# BatchNorm-backward sums) of the instantiated large shapes runs as one persistent streaming kernel
# (cn_conv2d_dgrad_junction, csrc/junction.hip) instead of the tiled GEMM kernel's epilogue.  g bit-identical, the
# partial sums in another fp32 association.
JDGRAD = flags.on('jdgrad')
# 1 = a convolution's backward launches its data gradient (main stream) before it hands the weight gradient to the side stream
DGRAD_FIRST = flags.on('dgrad_first')


# how often each BatchNorm path ran (tests assert that the fused paths really are the ones in use)
COUNTERS = {'bn_fwd_fused': 0, 'bn_fwd_plain': 0, 'bn_bwd_fused': 0, 'bn_bwd_plain': 0, 'bn_bwd_lazy': 0}


class _PendingStats(object):
    """BatchNorm statistics partials a convolution emitted for its output tensor.  They travel ON that tensor
    object (attribute `_cn_stats`, set by the producing convolution): the BatchNorm that consumes the very same
    object takes them, a copy / view / re-materialised tensor has none and falls back to the statistics pass, and
    the partial buffer dies with the tensor.  No module-level table, nothing keyed by id()."""
    __slots__ = ('partial', 'rows', 'pivot')   # pivot: data_ptr of the running mean the sums are centred on (or None)


def _park_stats(y, partial, rows, pivot=None):
    ps = _PendingStats()
    ps.partial, ps.rows = partial, rows
    ps.pivot = pivot.data_ptr() if pivot is not None else None
    y._cn_stats = ps


def take_pending_stats(y):
    """The partials emitted for exactly this tensor object (taken: a second call returns None), else None."""
    ps = getattr(y, '_cn_stats', None)
    if ps is not None:
        del y._cn_stats
    return ps


def stats_pivot(conv_mod):
    """The running mean of the BatchNorm that consumes this convolution's output (wired by the model as
    conv.stats_bn): the pivot of the centred statistics the epilogue emits.  None: plain sums."""
    bn = getattr(conv_mod, 'stats_bn', None)
    if bn is None or not CENTERED_STATS or not getattr(bn, 'track_running_stats', False) \
            or getattr(bn, 'running_mean', None) is None or _sync_group(bn) is not None:
        return None
    # fp32 storage only: with 16-bit storage y itself carries 2^-8 (2^-11) of relative rounding, so once |mean| >> sigma
    # the information is gone before any sum is taken, and the epilogue's pivot loads cost the bf16 step 0.3 %
    if getattr(conv_mod, 'compute_dtype', None) != torch.float32 and CENTERED_STATS != 'all':
        return None
    return bn.running_mean


def _halo3x3_ok(x, C, K, R, S, stride, pad):
    return (CONV3X3_HALO and (R, S) == (3, 3) and tuple(stride) == (1, 1) and tuple(pad) == (1, 1)
            and _L().cn_conv3x3_c64_ok(x.shape[1], x.shape[2], C, K, dtype_code(x.dtype)))


def conv2d_fwd(x, w_krsc, bias, K, R, S, stride, pad, out_f32=False, relu=False, bn_stats=False, pivot=None):
    N, H, W, C = x.shape
    P, Q = conv_out_hw(H, W, R, S, stride, pad)
    y = torch.empty((N, P, Q, K), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    if CONV1X1_STREAM and bias is None and not out_f32 and not relu and pivot is None and (R, S) == (1, 1) \
            and tuple(stride) == (1, 1) and tuple(pad) == (0, 0) \
            and _L().cn_conv1x1_stream_fwd_ok(C, K, dtype_code(x.dtype)):
        L = _L()
        want = bn_stats
        rows = L.cn_conv1x1_stream_fwd_rows(N, H, W, K) if want else 0
        partial = torch.empty((rows, 2 * K), dtype=torch.float32, device=x.device) if want else None
        PROFILER.run(_last_kernel(), 1, 2.0 * N * P * Q * K * C,
                     x.numel() * _esize(x) + y.numel() * _esize(y) + K * C * _esize(x),
                     lambda: check(L.cn_conv1x1_stream_fwd(ptr(x), ptr(w_krsc), ptr(y), N, H, W, C, K, dtype_code(x.dtype),
                                                           ptr(partial), rows, stream_of(x)), 'cn_conv1x1_stream_fwd'),
                     x.device, detail=_conv_detail('fwd', C, H, K, R, stride))
        if want:
            _park_stats(y, partial, rows, None)
        return y
    if bias is None and not out_f32 and not relu and pivot is None and _halo3x3_ok(x, C, K, R, S, stride, pad):
        L = _L()
        want = bn_stats
        rows = L.cn_conv3x3_c64_rows(N, H) if want else 0
        partial = torch.empty((rows, 2 * K), dtype=torch.float32, device=x.device) if want else None
        PROFILER.run(_last_kernel(), 1, 2.0 * N * P * Q * K * C * R * S,
                     x.numel() * _esize(x) + y.numel() * _esize(y) + K * R * S * C * _esize(x),
                     lambda: check(L.cn_conv3x3_c64(ptr(x), ptr(w_krsc), ptr(y), N, H, W, dtype_code(x.dtype), 0,
                                                    ptr(partial), rows, stream_of(x)), 'cn_conv3x3_c64'),
                     x.device, detail=_conv_detail('fwd', C, H, K, R, stride))
        if want:
            _park_stats(y, partial, rows, None)
        return y
    if bn_stats and not out_f32:
        L = _L()
        rows = L.cn_conv2d_bnstats_rows(N * P * Q)
        partial = torch.empty((rows, 2 * K), dtype=torch.float32, device=x.device)
        PROFILER.run(_last_kernel(),
                     1, 2.0 * N * P * Q * K * C * R * S,
                     x.numel() * _esize(x) + y.numel() * _esize(y) + K * R * S * C * _esize(x) + partial.numel() * 4,
                     (lambda: check(L.cn_conv2d_fwd_bnstats(ptr(x), ptr(w_krsc), ptr(y), ptr(bias), N, H, W, C, K, R, S,
                                                            stride[0], stride[1], pad[0], pad[1], dtype_code(x.dtype),
                                                            int(relu), ptr(partial), rows, stream_of(x)),
                                    'cn_conv2d_fwd_bnstats')) if pivot is None else
                     (lambda: check(L.cn_conv2d_fwd_bnstats_centered(ptr(x), ptr(w_krsc), ptr(y), ptr(bias), N, H, W, C,
                                                                     K, R, S, stride[0], stride[1], pad[0], pad[1],
                                                                     dtype_code(x.dtype), int(relu), ptr(partial), rows,
                                                                     ptr(pivot), stream_of(x)),
                                    'cn_conv2d_fwd_bnstats_centered')),
                     x.device, detail=_conv_detail('fwd', C, H, K, R, stride))
        _park_stats(y, partial, rows, pivot)
        return y
    PROFILER.run(_last_kernel(),
                 1, 2.0 * N * P * Q * K * C * R * S,
                 x.numel() * _esize(x) + y.numel() * _esize(y) + K * R * S * C * _esize(x),
                 lambda: check(_L().cn_conv2d_fwd(ptr(x), ptr(w_krsc), ptr(y), ptr(bias), N, H, W, C, K, R, S,
                                                  stride[0], stride[1], pad[0], pad[1], dtype_code(x.dtype),
                                                  int(out_f32), int(relu), stream_of(x)), 'cn_conv2d_fwd'),
                 x.device, detail=_conv_detail('fwd', C, H, K, R, stride))
    return y


def conv2d_fwd_lazyz(lz, w_krsc, K, bn_stats=False, pivot=None):
    """conv1x1(z) with z = relu(bn(y) + residual) formed on the operand load and stored by the kernel
    (cn_conv2d_fwd_lazyz).  lz: the junction's parked state (BatchNormActFunction.forward)."""
    y3, res, stats, res_stats, z, mask, relu = lz
    N, H, W, C = y3.shape
    out = torch.empty((N, H, W, K), dtype=y3.dtype, device=y3.device)
    L = _L()
    partial, rows = None, 0
    if bn_stats:
        rows = L.cn_conv2d_bnstats_rows(N * H * W)
        partial = torch.empty((rows, 2 * K), dtype=torch.float32, device=y3.device)
    nb = y3.numel() * _esize(y3)
    PROFILER.run(_last_kernel(' [lazy z]'), 1, 2.0 * N * H * W * K * C,
                 3 * nb + (mask.numel() if mask is not None else 0) + out.numel() * _esize(out) + K * C * _esize(y3)
                 + (partial.numel() * 4 if partial is not None else 0),
                 lambda: check(L.cn_conv2d_fwd_lazyz(ptr(y3), ptr(res), ptr(stats), ptr(res_stats), int(relu), ptr(z),
                                                     ptr(mask), ptr(w_krsc), ptr(out), N, H, W, C, K, dtype_code(y3.dtype),
                                                     ptr(partial), rows, ptr(pivot), stream_of(y3)), 'cn_conv2d_fwd_lazyz'),
                 y3.device, detail=_conv_detail('fwd', C, H, K, 1, (1, 1)))
    if bn_stats:
        _park_stats(out, partial, rows, pivot)
    return out


# "Lazy a" (round 3, third session): an INNER BatchNorm (no residual, ReLU) whose consumer is a 1x1 convolution on the
# streaming forward kernel only finalises its statistics; the convolution forms a = relu(y * scale + shift) on its
# operand path and writes it as a side output (cn_conv1x1_stream_fwd_lazya): the apply pass - one read of y, one write
# of a - and the convolution's read of a become one read of y and one write of a.  Same scope rule as lazy z (the
# consumer runs next by construction of the block's forward).
LAZY_A = flags.on('lazy_a')
LAZY_A_3X3 = flags.text('lazy_a') != '1x1'      # ('1x1': the streaming 1x1 consumers only)


def lazy_a_consumer_ok(conv, y):
    """conv (the module behind an inner BatchNorm whose input is y, NHWC) will run on the streaming 1x1 forward kernel
    or on the 64-channel 3x3 halo kernel."""
    C, dtype = y.shape[-1], y.dtype
    if conv is None or getattr(conv, 'bias', None) is not None or not conv.training or getattr(conv, 'out_f32', False) \
            or conv.in_channels != C or stats_pivot(conv) is not None or dtype not in (torch.bfloat16, torch.float16) \
            or getattr(conv, 'stride', None) != (1, 1):
        return False
    if getattr(conv, 'kernel_size', None) == (1, 1) and conv.padding == (0, 0):
        return (CONV1X1_STREAM
                and bool(_L().cn_conv1x1_stream_fwd_ok(C, conv.out_channels, dtype_code(dtype))))
    if getattr(conv, 'kernel_size', None) == (3, 3) and conv.padding == (1, 1):
        return (CONV3X3_HALO and LAZY_A_3X3 and bool(_L().cn_conv3x3_c64_ok(y.shape[1], y.shape[2], C, conv.out_channels, dtype_code(dtype))))
    return False


def conv2d_fwd_lazya(la, w_krsc, K, bn_stats=False, kernel=(1, 1)):
    """la = (bn_y, stats, a, relu) parked by BatchNormActFunction: y = conv(relu?(bn_y * scale + shift)), a written
    (1x1: the streaming kernel; 3x3: the 64-channel halo kernel)."""
    bn_y, stats, a, relu = la
    N, H, W, C = bn_y.shape
    L = _L()
    y = torch.empty((N, H, W, K), dtype=bn_y.dtype, device=bn_y.device)
    want = bn_stats
    if tuple(kernel) == (3, 3):
        rows = L.cn_conv3x3_c64_rows(N, H) if want else 0
        partial = torch.empty((rows, 2 * K), dtype=torch.float32, device=bn_y.device) if want else None
        PROFILER.run(_last_kernel(' [lazy a]'), 1, 2.0 * N * H * W * K * C * 9,
                     2 * bn_y.numel() * _esize(bn_y) + y.numel() * _esize(y) + K * 9 * C * _esize(bn_y),
                     lambda: check(L.cn_conv3x3_c64_lazya(ptr(bn_y), ptr(stats), int(relu), ptr(a), ptr(w_krsc), ptr(y),
                                                          N, H, W, dtype_code(bn_y.dtype), ptr(partial), rows,
                                                          stream_of(bn_y)), 'cn_conv3x3_c64_lazya'),
                     bn_y.device, detail=_conv_detail('fwd', C, H, K, 3, (1, 1)))
        if want:
            _park_stats(y, partial, rows, None)
        return y
    rows = L.cn_conv1x1_stream_fwd_rows(N, H, W, K) if want else 0
    partial = torch.empty((rows, 2 * K), dtype=torch.float32, device=bn_y.device) if want else None
    PROFILER.run(_last_kernel(' [lazy a]'), 1, 2.0 * N * H * W * K * C,
                 2 * bn_y.numel() * _esize(bn_y) + y.numel() * _esize(y) + K * C * _esize(bn_y),
                 lambda: check(L.cn_conv1x1_stream_fwd_lazya(ptr(bn_y), ptr(stats), int(relu), ptr(a), ptr(w_krsc), ptr(y),
                                                             N, H, W, C, K, dtype_code(bn_y.dtype), ptr(partial), rows,
                                                             stream_of(bn_y)), 'cn_conv1x1_stream_fwd_lazya'),
                 bn_y.device, detail=_conv_detail('fwd', C, H, K, 1, (1, 1)))
    if want:
        _park_stats(y, partial, rows, None)
    return y


def lazy_z_consumer_ok(conv):
    """conv can take an unapplied junction as its input (cn_conv2d_fwd_lazyz's shape limits)."""
    return (conv is not None and getattr(conv, 'kernel_size', None) == (1, 1) and getattr(conv, 'stride', None) == (1, 1)
            and getattr(conv, 'padding', None) == (0, 0) and getattr(conv, 'bias', None) is None
            and conv.out_channels <= 128 and conv.in_channels <= 512 and conv.training
            and not getattr(conv, 'out_f32', False))


def conv2d_dgrad(dy, w_crsk, x_shape, K, R, S, stride, pad, addend=None, bn=None, addend_sub=1):
    """dx (NHWC).  With bn = (bn_y, bn_mask_or_None, bn_stats[4C], relu) the epilogue also does the
    reduction half of that BatchNorm's backward: returns (g = dx*relu_mask, partial, rows).
    addend_sub = 2: `addend` holds only the even (h, w) pixels of a gradient that is zero elsewhere."""
    N, H, W, C = x_shape
    dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
    name = _last_kernel()
    detail = _conv_detail('dgrad', C, H, K, R, stride)
    flops = 2.0 * dy.numel() * C * R * S
    nbytes = dy.numel() * _esize(dy) + dx.numel() * _esize(dx) + K * R * S * C * _esize(dy) \
        + (addend.numel() * _esize(addend) if addend is not None else 0)
    if addend is not None and addend_sub == 2:
        assert tuple(addend.shape) == (N, (H + 1) // 2, (W + 1) // 2, C), 'subsampled addend shape'
    if bn is None and addend is None and _halo3x3_ok(dy, K, C, R, S, stride, pad) and tuple(dy.shape[1:3]) == (H, W):
        PROFILER.run(name, 1, flops, nbytes,
                     lambda: check(_L().cn_conv3x3_c64(ptr(dy), ptr(w_crsk), ptr(dx), N, H, W, dtype_code(dy.dtype), 1, None,
                                                       0, stream_of(dy)), 'cn_conv3x3_c64'),
                     dy.device, detail=detail)
        return dx
    if bn is None:
        PROFILER.run(name, stride[0] * stride[1], flops, nbytes,
                     lambda: check(_L().cn_conv2d_dgrad_sa(ptr(dy), ptr(w_crsk), ptr(dx), ptr(addend), int(addend_sub), N, H,
                                                           W, C, K, R, S, stride[0], stride[1], pad[0], pad[1],
                                                           dtype_code(dy.dtype), 0, stream_of(dy)), 'cn_conv2d_dgrad'),
                     dy.device, detail=detail)
        return dx
    bn_y, bn_mask, bn_stats, bn_relu = bn
    L = _L()
    if JDGRAD and (R, S) == (1, 1) and tuple(stride) == (1, 1) and tuple(pad) == (0, 0) and bn_mask is not None \
            and addend is not None and L.cn_conv2d_dgrad_junction_ok(C, K, dtype_code(dy.dtype)):
        rows = L.cn_conv2d_dgrad_junction_rows_k(N, H, W, C, K)
        partial = torch.empty((rows, 2 * C), dtype=torch.float32, device=dy.device)
        PROFILER.run(_last_kernel(), 1, flops, nbytes + dx.numel() * _esize(dx) + partial.numel() * 4,
                     lambda: check(L.cn_conv2d_dgrad_junction(ptr(dy), ptr(w_crsk), ptr(dx), ptr(addend), int(addend_sub), N,
                                                              H, W, C, K, dtype_code(dy.dtype), ptr(bn_y), ptr(bn_mask),
                                                              ptr(bn_stats), ptr(partial), rows, stream_of(dy)),
                                   'cn_conv2d_dgrad_junction'),
                     dy.device, detail=detail)
        COUNTERS['jdgrad'] = COUNTERS.get('jdgrad', 0) + 1
        return dx, partial, rows
    rows = L.cn_conv2d_dgrad_bnbwd_rows(N, H, W, C, stride[0], stride[1])
    partial = torch.empty((rows, 2 * C), dtype=torch.float32, device=dy.device)
    PROFILER.run(name, stride[0] * stride[1], flops, nbytes + dx.numel() * _esize(dx) + partial.numel() * 4,
                 lambda: check(L.cn_conv2d_dgrad_bnbwd_sa(ptr(dy), ptr(w_crsk), ptr(dx), ptr(addend), int(addend_sub), N, H,
                                                          W, C, K, R, S, stride[0], stride[1], pad[0], pad[1],
                                                          dtype_code(dy.dtype), ptr(bn_y), ptr(bn_mask), ptr(bn_stats),
                                                          int(bn_relu), ptr(partial), rows, stream_of(dy)),
                               'cn_conv2d_dgrad_bnbwd'),
                 dy.device, detail=detail)
    return dx, partial, rows


def conv2d_wgrad(x, dy, dw_krsc, c_real, K, R, S, stride, pad, beta=1.0, scale=1.0, tag='main'):
    """dw_krsc (fp32, [K][R][S][c_real] memory order) = beta*dw + scale*wgrad."""
    N, H, W, C = x.shape
    code = dtype_code(x.dtype)
    L = _L()
    need = L.cn_conv2d_wgrad_workspace(N, H, W, C, K, R, S, stride[0], stride[1], pad[0], pad[1], code)
    ws = workspace(need, x.device, tag)
    def call():
        check(L.cn_conv2d_wgrad(ptr(x), ptr(dy), ptr(dw_krsc), c_real, N, H, W, C, K, R, S, stride[0], stride[1],
                                pad[0], pad[1], code, beta, scale, ptr(ws), ws.numel() * 4, stream_of(x)),
              'cn_conv2d_wgrad')
    if not (PROFILER.enabled and x.is_cuda):
        return call()
    # profiled: the call's two launches (partial products, then the fixed-order reduction of the splits) are timed
    # separately - measurement-only knob "wgrad_phase" - so each is reported under its own kernel name
    flops = 2.0 * dy.numel() * C * R * S
    part_bytes = float(need)
    detail = _conv_detail('wgrad', C, H, K, R, stride)
    L.cn_set_option(b'wgrad_phase', 1)
    try:
        PROFILER.run(_last_kernel(), 1, flops, x.numel() * _esize(x) + dy.numel() * _esize(dy) + part_bytes, call,
                     x.device, detail=detail)
        L.cn_set_option(b'wgrad_phase', 2)
        PROFILER.run('wgrad_reduce_kernel', 1, 0.0, part_bytes + K * R * S * c_real * 4, call, x.device,
                     detail=detail + ' [reduce]')
    finally:
        L.cn_set_option(b'wgrad_phase', 0)


def conv2d_dgrad_lazy(g, bn_y, coef, w_crsk, x_shape, K, R, S, stride, pad):
    """dgrad whose upstream gradient dy = c1*g + c2*bn_y + c3 is formed on the operand load (cn_conv2d_dgrad_lazy)."""
    N, H, W, C = x_shape
    dx = torch.empty((N, H, W, C), dtype=g.dtype, device=g.device)
    if (R, S) == (1, 1) and tuple(stride) == (1, 1) and tuple(pad) == (0, 0) and tuple(g.shape[1:3]) == (H, W) \
            and _L().cn_conv2d_dgrad_lazy_stream_ok(C, K, dtype_code(g.dtype)):
        PROFILER.run(_last_kernel(' [lazy dy]'), 1, 2.0 * g.numel() * C,
                     2 * g.numel() * _esize(g) + dx.numel() * _esize(dx) + K * C * _esize(g),
                     lambda: check(_L().cn_conv2d_dgrad_lazy_stream(ptr(g), ptr(bn_y), ptr(coef), ptr(w_crsk), ptr(dx), N, H, W,
                                                                    C, K, dtype_code(g.dtype), stream_of(g)),
                                   'cn_conv2d_dgrad_lazy_stream'),
                     g.device, detail=_conv_detail('dgrad', C, H, K, R, stride))
        return dx
    PROFILER.run(_last_kernel(' [lazy dy]'), stride[0] * stride[1], 2.0 * g.numel() * C * R * S,
                 2 * g.numel() * _esize(g) + dx.numel() * _esize(dx) + K * R * S * C * _esize(g),
                 lambda: check(_L().cn_conv2d_dgrad_lazy(ptr(g), ptr(bn_y), ptr(coef), ptr(w_crsk), ptr(dx), N, H, W, C, K,
                                                         R, S, stride[0], stride[1], pad[0], pad[1], dtype_code(g.dtype),
                                                         stream_of(g)), 'cn_conv2d_dgrad_lazy'),
                 g.device, detail=_conv_detail('dgrad', C, H, K, R, stride))
    return dx


def conv2d_wgrad_lazy(x, g, bn_y, coef, dw_krsc, c_real, K, R, S, stride, pad, beta=1.0, scale=1.0, tag='main'):
    """wgrad whose dy operand is formed on load from (g, bn_y, coef) (cn_conv2d_wgrad_lazy)."""
    N, H, W, C = x.shape
    code = dtype_code(x.dtype)
    L = _L()
    need = L.cn_conv2d_wgrad_workspace(N, H, W, C, K, R, S, stride[0], stride[1], pad[0], pad[1], code)
    ws = workspace(need, x.device, tag)
    PROFILER.run(_last_kernel(' [lazy dy] (+wgrad_reduce)'), 2, 2.0 * g.numel() * C * R * S,
                 x.numel() * _esize(x) + 2 * g.numel() * _esize(g) + float(need),
                 lambda: check(L.cn_conv2d_wgrad_lazy(ptr(x), ptr(g), ptr(bn_y), ptr(coef), ptr(dw_krsc), c_real, N, H, W, C,
                                                      K, R, S, stride[0], stride[1], pad[0], pad[1], code, beta, scale,
                                                      ptr(ws), ws.numel() * 4, stream_of(x)), 'cn_conv2d_wgrad_lazy'),
                 x.device, detail=_conv_detail('wgrad', C, H, K, R, stride))


def conv2d_bwd1x1_lazy(x, g, bn_y, coef, w_crsk, dw_krsc, K, beta=1.0, scale=1.0):
    """dx and the weight gradient of a 1x1 / stride-1 convolution for the lazy upstream gradient (g, bn_y, coef), one
    pass over g and bn_y (cn_conv2d_bwd1x1_lazy)."""
    N, H, W, C = x.shape
    code = dtype_code(x.dtype)
    L = _L()
    need = L.cn_conv2d_bwd1x1_lazy_workspace(N, H, W, C, K)
    ws = workspace(need, x.device, 'main')
    dx = torch.empty_like(x)

    def call():
        check(L.cn_conv2d_bwd1x1_lazy(ptr(x), ptr(g), ptr(bn_y), ptr(coef), ptr(w_crsk), ptr(dx), ptr(dw_krsc), N, H, W, C,
                                      K, code, beta, scale, ptr(ws), ws.numel() * 4, stream_of(x)), 'cn_conv2d_bwd1x1_lazy')
    if not (PROFILER.enabled and x.is_cuda):
        call()
        return dx
    flops = 4.0 * g.numel() * C
    detail = _conv_detail('dgrad+wgrad', C, H, K, 1, (1, 1))
    L.cn_set_option(b'wgrad_phase', 1)
    try:
        PROFILER.run(_last_kernel(' [lazy dy]'), 1, flops, 2 * g.numel() * _esize(g) + 2 * x.numel() * _esize(x)
                     + float(need), call, x.device, detail=detail)
        L.cn_set_option(b'wgrad_phase', 2)
        PROFILER.run('wgrad_reduce_kernel', 1, 0.0, float(need) + K * C * 4, call, x.device, detail=detail + ' [reduce]')
    finally:
        L.cn_set_option(b'wgrad_phase', 0)
    return dx


def weight_prep(w_master_krsc, w_krsc, w_crsk, Co, taps, c_real, c_pad):
    PROFILER.run('weight_prep', 1, 0.0, Co * taps * c_real * 4 + Co * taps * c_pad * _esize(w_krsc) * (2 if w_crsk is not None else 1),
                 lambda: check(_L().cn_weight_prep(ptr(w_master_krsc), ptr(w_krsc), ptr(w_crsk), Co, taps, c_real,
                                                   c_pad, dtype_code(w_krsc.dtype), stream_of(w_krsc)),
                               'cn_weight_prep'), w_krsc.device)


def colsum(x2d, out, beta=1.0, scale=1.0):
    M, C = x2d.shape
    L = _L()
    ws = workspace(L.cn_colsum_workspace(C), x2d.device, 'colsum')
    check(L.cn_colsum(ptr(x2d), ptr(out), M, C, dtype_code(x2d.dtype), beta, scale, ptr(ws), stream_of(x2d)),
          'cn_colsum')


def nchw_to_nhwc(x_nchw, dtype, c_pad=None):
    """fp32 NCHW (the loader / reference layout) -> NHWC compute dtype, channels zero-padded."""
    N, C, H, W = x_nchw.shape
    ch = _lib.chunk_elems(dtype)
    if c_pad is None:
        c_pad = (C + ch - 1) // ch * ch
    x_nchw = x_nchw.contiguous()
    if x_nchw.dtype != torch.float32:
        raise _lib.ConvNetHipError('nchw_to_nhwc expects float32 input, got %s' % x_nchw.dtype)
    y = torch.empty((N, H, W, c_pad), dtype=dtype, device=x_nchw.device)
    PROFILER.run('nchw_to_nhwc', 1, 0.0, x_nchw.numel() * 4 + y.numel() * _esize(y),
                 lambda: check(_L().cn_nchw_to_nhwc(ptr(x_nchw), ptr(y), N, C, H, W, c_pad, dtype_code(dtype),
                                                    stream_of(x_nchw)), 'cn_nchw_to_nhwc'), x_nchw.device)
    return y


def normalize_lut(mean, std):
    """lut[c][u] = ToTensor + Normalize of byte u in channel c, with the reference pipeline's own fp32 operations
    (`.float().div_(255)`, then `(t - mean) / std`): what cn_u8_nhwc_to_nchw_lut looks up."""
    u = torch.arange(256, dtype=torch.uint8).float().div_(255.0).view(1, 256)
    m = torch.tensor(mean, dtype=torch.float32).view(-1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(-1, 1)
    return ((u - m) / s).contiguous()


def resize_crops(batch):
    """A device_resize loader's batch (data.collate_crops, already on the device) -> uint8 [B, S, S, C]: PIL's BILINEAR
    resize of every crop, bit for bit (cn_resize_u8_crops), mirrored where the loader drew a horizontal flip."""
    px, meta, tables = batch['crops'], batch['meta'], batch['tables']
    row_owner, row_off = batch['row_owner'], batch['row_off']
    S, C = (int(v) for v in batch['size'].tolist())
    B, rows = meta.shape[0], row_owner.shape[0]
    tmp = torch.empty(rows * S * C, dtype=torch.uint8, device=px.device)
    out = torch.empty((B, S, S, C), dtype=torch.uint8, device=px.device)
    PROFILER.run('resize_u8_crops', 2, 0.0, px.numel() + 2 * tmp.numel() + out.numel(),
                 lambda: check(_L().cn_resize_u8_crops(ptr(px), ptr(meta), ptr(tables), ptr(row_owner), ptr(row_off), ptr(tmp),
                                                       ptr(out), B, rows, S, C, stream_of(px)), 'cn_resize_u8_crops'), px.device)
    return out


def u8_nhwc_to_nchw(x_u8, lut):
    """uint8 [N, H, W, C] crops -> the normalised fp32 [N, C, H, W] batch (device-side ToTensor + Normalize)."""
    if x_u8.dtype != torch.uint8 or x_u8.dim() != 4:
        raise _lib.ConvNetHipError('u8_nhwc_to_nchw expects uint8 [N, H, W, C], got %s %s' % (x_u8.dtype, tuple(x_u8.shape)))
    N, H, W, C = x_u8.shape
    x_u8 = x_u8.contiguous()
    lut = lut.to(device=x_u8.device, dtype=torch.float32).contiguous()
    if tuple(lut.shape) != (C, 256):
        raise _lib.ConvNetHipError('u8_nhwc_to_nchw: lut %s for %d channels' % (tuple(lut.shape), C))
    y = torch.empty((N, C, H, W), dtype=torch.float32, device=x_u8.device)
    PROFILER.run('u8_nhwc_to_nchw', 1, 0.0, x_u8.numel() + y.numel() * 4,
                 lambda: check(_L().cn_u8_nhwc_to_nchw_lut(ptr(x_u8), ptr(y), N, H, W, C, ptr(lut), stream_of(x_u8)),
                               'cn_u8_nhwc_to_nchw_lut'), x_u8.device)
    return y


def nchw_to_pairs(x_nchw, pad):
    """fp32 NCHW (C <= 4) -> zero-padded bf16 pixel-pair image [N, H+2ph, (W+2pw)/2, 8] (cn_nchw_to_pairs)."""
    N, C, H, W = x_nchw.shape
    x_nchw = x_nchw.contiguous()
    if x_nchw.dtype != torch.float32:
        raise _lib.ConvNetHipError('nchw_to_pairs expects float32 input, got %s' % x_nchw.dtype)
    y = torch.empty((N, H + 2 * pad[0], (W + 2 * pad[1]) // 2, 8), dtype=torch.bfloat16, device=x_nchw.device)
    PROFILER.run('nchw_to_pairs', 1, 0.0, x_nchw.numel() * 4 + y.numel() * 2,
                 lambda: check(_L().cn_nchw_to_pairs(ptr(x_nchw), ptr(y), N, C, H, W, pad[0], pad[1],
                                                     stream_of(x_nchw)), 'cn_nchw_to_pairs'), x_nchw.device)
    return y


def nhwc_to_nchw(x_nhwc, C=None):
    N, H, W, Cp = x_nhwc.shape
    C = C or Cp
    y = torch.empty((N, C, H, W), dtype=torch.float32, device=x_nhwc.device)
    check(_L().cn_nhwc_to_nchw(ptr(x_nhwc), ptr(y), N, C, H, W, Cp, dtype_code(x_nhwc.dtype), stream_of(x_nhwc)),
          'cn_nhwc_to_nchw')
    return y


def add_(a, b):
    PROFILER.run('eltwise_add', 1, 0.0, 3 * a.numel() * _esize(a),
                 lambda: check(_L().cn_eltwise(0, ptr(a), ptr(b), None, a.numel(), dtype_code(a.dtype),
                                               stream_of(a)), 'cn_eltwise'), a.device)
    return a


def cast_from_f32(x, dtype):
    y = torch.empty(x.shape, dtype=dtype, device=x.device)
    check(_L().cn_cast_from_f32(ptr(x), ptr(y), x.numel(), dtype_code(dtype), stream_of(x)), 'cn_cast_from_f32')
    return y


def fill_f32_(x, v=0.0):
    check(_L().cn_fill_f32(ptr(x), x.numel(), float(v), stream_of(x)), 'cn_fill_f32')
    return x


# ---------------------------------------------------------------------------------------------
# autograd Functions.  Parameter gradients are accumulated *by the kernels* straight into the flat
# gradient arena (`mod.grad_view(name)`), so the Functions return None for parameters and autograd
# never runs an accumulation kernel of its own; `mod._notify_grad_ready()` lets the data-parallel
# bucket manager start the all-reduce of a finished bucket while backward continues.

def _input_bn_state(conv_mod, x):
    """(bn module, y, mask, stats, relu) of the BatchNorm whose output *is* the convolution input `x`
    (wired by the model as conv.input_bn), when that BatchNorm's forward state is still alive and
    belongs to this very tensor; else None."""
    bn_mod = getattr(conv_mod, 'input_bn', None)
    if bn_mod is None:
        return None
    ref = getattr(bn_mod, '_fwd_ctx', None)
    bctx = ref() if ref is not None else None
    if bctx is None or getattr(bctx, 'out_ptr', None) != x.data_ptr():
        return None
    try:
        saved = bctx.saved_tensors
    except RuntimeError:      # already released
        return None
    y, stats = saved[0], saved[1]
    mask = saved[2] if len(saved) > 2 else None
    if tuple(y.shape) != tuple(x.shape) or y.dtype != x.dtype:
        return None
    if bctx.has_res and bctx.relu and mask is None:
        return None
    if y.shape[-1] % _lib.chunk_elems(y.dtype) != 0:
        return None
    return bn_mod, y, mask, stats, bctx.relu


_ZEROS = {}


def _zero_like_placeholder(x):
    """A storage-free all-zero tensor of x's shape (one cached zero element, expanded): what autograd is handed when the
    real gradient travels through a ResGradHolder."""
    key = (x.device, x.dtype)
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(1, dtype=x.dtype, device=x.device)
    return z.expand(x.shape)


def _is_zero_placeholder(t):
    z = _ZEROS.get((t.device, t.dtype))
    return z is not None and t.data_ptr() == z.data_ptr() and t.dim() > 0 and all(st == 0 for st in t.stride())


class Conv2dFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mod):
        mod.ensure_prepared()
        lz = mod.__dict__.pop('_lazy_z', None)
        if lz is not None:    # x is a junction output that exists only as (y, residual, statistics): this kernel writes it
            if lz[0] != x.data_ptr() or bias is not None:
                raise _lib.ConvNetHipError('lazy z: the parked junction is not this convolution\'s input')
            y = conv2d_fwd_lazyz(lz[1:], mod.w_krsc, mod.out_channels,
                                 bn_stats=FUSE_BN_STATS and mod.training and getattr(mod, 'feeds_batchnorm', False),
                                 pivot=stats_pivot(mod))
            COUNTERS['lazy_z'] = COUNTERS.get('lazy_z', 0) + 1
        elif mod.__dict__.get('_lazy_a') is not None:   # x is an inner BatchNorm's output that exists only as (y, statistics)
            la = mod.__dict__.pop('_lazy_a')
            if la[0] != x.data_ptr() or bias is not None:
                raise _lib.ConvNetHipError('lazy a: the parked BatchNorm output is not this convolution\'s input')
            y = conv2d_fwd_lazya(la[1:], mod.w_krsc, mod.out_channels,
                                 bn_stats=FUSE_BN_STATS and mod.training and getattr(mod, 'feeds_batchnorm', False),
                                 kernel=mod.kernel_size)
            COUNTERS['lazy_a'] = COUNTERS.get('lazy_a', 0) + 1
        else:
            y = conv2d_fwd(x, mod.w_krsc, bias, mod.out_channels, mod.kernel_size[0], mod.kernel_size[1],
                           mod.stride, mod.padding, out_f32=mod.out_f32,
                           bn_stats=FUSE_BN_STATS and mod.training and getattr(mod, 'feeds_batchnorm', False),
                           pivot=stats_pivot(mod))
        ctx.mod = mod
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x)
        # the lazy-dy mailbox is matched against THIS output in backward; an entry left by an aborted backward dies here
        ctx.out_ptr, ctx.out_shape = y.data_ptr(), tuple(y.shape)
        mod._lazy_dy = None
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        mod = ctx.mod
        lazy = getattr(mod, '_lazy_dy', None)     # (g, bn_y, coef): the BatchNorm behind this conv left dy unformed
        mod._lazy_dy = None
        R, S = mod.kernel_size
        if lazy is None and _is_zero_placeholder(dy):
            raise _lib.ConvNetHipError('lazy dy: the gradient placeholder reached a convolution with an empty mailbox '
                                       '(the BatchNorm that parked the gradient is not this convolution\'s consumer)')
        if lazy is not None:
            g, bn_y, coef = lazy
            if bn_y.data_ptr() != ctx.out_ptr or tuple(bn_y.shape) != ctx.out_shape or tuple(g.shape) != tuple(dy.shape):
                raise _lib.ConvNetHipError('lazy dy: the parked gradient does not belong to this convolution\'s output '
                                           '(stale mailbox entry or a second consumer)')
            if JPAIR and (R, S) == (1, 1) and mod.stride == (1, 1) and mod.padding == (0, 0) and ctx.needs_input_grad[0] \
                    and mod.in_channels == x.shape[-1] \
                    and _L().cn_conv2d_bwd1x1_lazy_ok(x.shape[-1], mod.out_channels, dtype_code(x.dtype)):
                # junction pair: both gradients of this convolution in one pass over (g, bn_y), on the backward chain
                holder = getattr(mod, '_res_holder', None)
                if holder is not None and holder.dres is not None:
                    raise _lib.ConvNetHipError('lazy dy met a fused-addend dgrad: the junction layout changed')
                dx = conv2d_bwd1x1_lazy(x, g, bn_y, coef, mod.w_crsk, mod.grad_view('weight'), mod.out_channels)
                mod._notify_grad_ready()
                COUNTERS['jpair'] = COUNTERS.get('jpair', 0) + 1
                if holder is not None:
                    holder.dres, holder.sub, holder.fused = dx, 1, False
                return dx, None, None, None
            if SIDE.active(x):
                def launch():
                    conv2d_wgrad_lazy(x, g, bn_y, coef, mod.grad_view('weight'), mod.in_channels, mod.out_channels, R, S,
                                      mod.stride, mod.padding, tag='side')
                    return (x, g, bn_y, coef)
                SIDE.submit(x.device, launch, mod._notify_grad_ready, coef)
            else:
                conv2d_wgrad_lazy(x, g, bn_y, coef, mod.grad_view('weight'), mod.in_channels, mod.out_channels, R, S,
                                  mod.stride, mod.padding)
                mod._notify_grad_ready()
            if not ctx.needs_input_grad[0]:
                return None, None, None, None
            holder = getattr(mod, '_res_holder', None)
            if holder is not None and holder.dres is not None:
                raise _lib.ConvNetHipError('lazy dy met a fused-addend dgrad: the junction layout changed')
            if holder is not None and SUBSAMPLED_SHORTCUT_GRAD and (R, S) == (1, 1) and mod.stride == (2, 2) \
                    and mod.padding == (0, 0):
                N_, H_, W_, C_ = x.shape
                compact = conv2d_dgrad_lazy(g, bn_y, coef, mod.w_crsk, (N_, (H_ + 1) // 2, (W_ + 1) // 2, C_),
                                            mod.out_channels, 1, 1, (1, 1), (0, 0))
                holder.dres, holder.sub, holder.fused = compact, 2, False
                return _zero_like_placeholder(x), None, None, None
            dx = conv2d_dgrad_lazy(g, bn_y, coef, mod.w_crsk, x.shape, mod.out_channels, R, S, mod.stride, mod.padding)
            if holder is not None:
                holder.dres, holder.sub, holder.fused = dx, 1, False
            return dx, None, None, None
        dy = dy.contiguous()
        if dy.dtype != x.dtype:  # fp32 logits gradient -> compute dtype
            dy = cast_from_f32(dy, x.dtype)
        if ctx.has_bias:
            colsum(dy.view(-1, mod.out_channels), mod.grad_view('bias'))
        def submit_wgrad():
            if SIDE.active(x):
                def launch():
                    conv2d_wgrad(x, dy, mod.grad_view('weight'), mod.in_channels, mod.out_channels, R, S, mod.stride,
                                 mod.padding, tag='side')
                    return (x, dy)
                SIDE.submit(x.device, launch, mod._notify_grad_ready, dy)
            else:
                conv2d_wgrad(x, dy, mod.grad_view('weight'), mod.in_channels, mod.out_channels, R, S, mod.stride,
                             mod.padding)
                mod._notify_grad_ready()

        def dgrad_part():
            dx = None
            if ctx.needs_input_grad[0]:
                addend, addend_sub = None, 1
                holder = getattr(mod, '_res_holder', None)
                if holder is not None and holder.dres is not None and holder.dres.dtype == dy.dtype \
                        and (holder.dres.shape == x.shape if holder.sub == 1 else
                             tuple(holder.dres.shape) == (x.shape[0], (x.shape[1] + 1) // 2, (x.shape[2] + 1) // 2, x.shape[3])):
                    addend, addend_sub = holder.dres, holder.sub   # the other branch's gradient, folded into this dgrad epilogue
                    holder.fused = True
                elif holder is not None and holder.dres is None and SUBSAMPLED_SHORTCUT_GRAD \
                        and (R, S) == (1, 1) and mod.stride == (2, 2) and mod.padding == (0, 0):
                    # stride-2 1x1 projection shortcut, first of the two gradients that meet at the block input: its input
                    # gradient is zero off the even pixels - compute those on the coarse grid (a stride-1 dgrad), park
                    # them, and hand autograd a storage-free placeholder (ForkFunction returns the other branch's sum)
                    N_, H_, W_, C_ = x.shape
                    compact = conv2d_dgrad(dy, mod.w_crsk, (N_, (H_ + 1) // 2, (W_ + 1) // 2, C_), mod.out_channels, 1, 1,
                                           (1, 1), (0, 0))
                    holder.dres, holder.sub, holder.fused = compact, 2, False
                    return _zero_like_placeholder(x), None, None, None
                # this dgrad is the last contribution to the gradient of x when x has no other consumer
                # (inner convs) or when the other branch's gradient is being added right here
                final = holder is None or addend is not None
                bn_args = _input_bn_state(mod, x) if (final and FUSE_BN_BWD) else None
                if bn_args is not None and holder is None:
                    bn_args = None      # junctions only (see FUSE_BN_BWD)
                if bn_args is not None:
                    bn_mod, bn_y, bn_mask, bn_stats, bn_relu = bn_args
                    dx, partial, rows = conv2d_dgrad(dy, mod.w_crsk, x.shape, mod.out_channels, R, S, mod.stride,
                                                     mod.padding, addend=addend, bn=(bn_y, bn_mask, bn_stats, bn_relu),
                                                     addend_sub=addend_sub)
                    bn_mod._bwd_partials = (dx.data_ptr(), tuple(dx.shape), partial, rows)
                else:
                    dx = conv2d_dgrad(dy, mod.w_crsk, x.shape, mod.out_channels, R, S, mod.stride, mod.padding,
                                      addend=addend, addend_sub=addend_sub)
                if holder is not None and addend is None:
                    holder.dres, holder.sub = dx, 1   # first producer of the fork gradient: park it for the other
                    holder.fused = False
            return dx, None, None, None

        # The data gradient is the backward chain's next kernel; the weight gradient only has to start some time.  Launching
        # the chain's kernel FIRST keeps the host-side cost of the side-stream hand-off (fork / mark wait + a second ctypes
        # call, ~40 us) off the chain where the host runs only just ahead of the device (the late, small layers): the
        # rocprofv3 trace showed 10-14 us of idle main queue in front of 26 of these data gradients.  Untraced the effect is at
        # the noise level (14871 vs 14819 img/s, three interleaved rounds); kept because it is the natural order (flag dgrad_first).
        if not DGRAD_FIRST:
            submit_wgrad()
        out = dgrad_part()
        if DGRAD_FIRST:
            submit_wgrad()
        return out


def _sync_group(mod):
    """(process group, world size) when this BatchNorm synchronises its batch statistics across ranks
    (nn.convert_sync_batchnorm + an initialised process group of more than one rank), else None."""
    sg = getattr(mod, 'sync_group', None)
    if sg is None or not dist.is_available() or not dist.is_initialized():
        return None
    group = None if sg is True else sg
    world = dist.get_world_size(group)
    return (group, world) if world > 1 else None


def _sync_all_reduce(t, group, world):
    """SyncBatchNorm statistics exchange: one in-stream RCCL all-reduce on the compute stream when the
    direct communicator spans this group (no stream hop, no host-side work object), else torch.distributed."""
    from . import comm as _comm
    c = _comm.default()
    if c is not None and t.is_cuda and c.world == world and (group is None or group is c.pg):
        c.allreduce_(t)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class StemPairConvFunction(Function):
    """The stride-2 stem convolution on the pixel-pair image (ops.nchw_to_pairs): same result as
    Conv2dFunction on the channel-padded NHWC image, with ceil(S/2) instead of S reduction chunks per
    filter row and no bounds tests.  The network input needs no gradient, so backward is wgrad only."""

    @staticmethod
    def forward(ctx, x_pairs, weight, mod):
        K, (R, S) = mod.out_channels, mod.kernel_size
        S2 = (S + 1) // 2
        L = _L()
        wp = torch.empty(K * R * S2 * 8, dtype=torch.bfloat16, device=x_pairs.device)
        check(L.cn_weight_prep_pairs(ptr(mod.master_view('weight')), ptr(wp), K, R, S, mod.in_channels,
                                     stream_of(x_pairs)), 'cn_weight_prep_pairs')
        want_stats = FUSE_BN_STATS and mod.training and getattr(mod, 'feeds_batchnorm', False)
        N_, Hp_, Jp_, _ = x_pairs.shape
        if STEM_HALO and want_stats and stats_pivot(mod) is None and mod.stride[0] == 2 and x_pairs.dtype == torch.bfloat16 \
                and L.cn_stem_fwd_ok(K, R, S2, Jp_, dtype_code(x_pairs.dtype)):
            P_, Q_ = (Hp_ - R) // 2 + 1, Jp_ - S2 + 1
            y = torch.empty((N_, P_, Q_, K), dtype=x_pairs.dtype, device=x_pairs.device)
            rows = L.cn_stem_fwd_rows(N_, P_)
            partial = torch.empty((rows, 2 * K), dtype=torch.float32, device=x_pairs.device)
            PROFILER.run(_last_kernel(), 1, 2.0 * N_ * P_ * Q_ * K * 8 * R * S2,
                         x_pairs.numel() * 2 + y.numel() * 2 + wp.numel() * 2 + partial.numel() * 4,
                         lambda: check(L.cn_stem_fwd(ptr(x_pairs), ptr(wp), ptr(y), N_, Hp_, Jp_, dtype_code(x_pairs.dtype),
                                                     ptr(partial), rows, stream_of(x_pairs)), 'cn_stem_fwd'),
                         x_pairs.device, detail=_conv_detail('fwd', 8, Hp_, K, R, (2, 1)))
            _park_stats(y, partial, rows, None)
        else:
            y = conv2d_fwd(x_pairs, wp, None, K, R, S2, (mod.stride[0], 1), (0, 0), bn_stats=want_stats,
                           pivot=stats_pivot(mod))
        ctx.mod = mod
        ctx.save_for_backward(x_pairs)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        mod = ctx.mod
        dy = dy.contiguous()
        K, (R, S) = mod.out_channels, mod.kernel_size
        S2 = (S + 1) // 2
        L = _L()

        def run(tag):
            tmp = torch.empty(K * R * S2 * 8, dtype=torch.float32, device=x.device)
            N_, Hp_, Jp_, _ = x.shape
            code = dtype_code(x.dtype)
            if STEM_HALO and mod.stride[0] == 2 and x.dtype == torch.bfloat16 and L.cn_stem_wgrad_ok(K, R, S2, Jp_, code):
                need = L.cn_stem_wgrad_workspace(N_, Hp_)
                ws = workspace(need, x.device, tag)

                def call():
                    check(L.cn_stem_wgrad(ptr(x), ptr(dy), ptr(tmp), N_, Hp_, Jp_, code, 0.0, 1.0, ptr(ws), ws.numel() * 4,
                                          stream_of(x)), 'cn_stem_wgrad')
                if PROFILER.enabled and x.is_cuda:
                    detail = _conv_detail('wgrad', 8, Hp_, K, R, (2, 1))
                    L.cn_set_option(b'wgrad_phase', 1)
                    try:
                        PROFILER.run(_last_kernel(), 1, 2.0 * dy.numel() * 8 * R * S2,
                                     x.numel() * 2 + dy.numel() * 2 + float(need), call, x.device, detail=detail)
                        L.cn_set_option(b'wgrad_phase', 2)
                        PROFILER.run('wgrad_reduce_kernel', 1, 0.0, float(need) + tmp.numel() * 4, call, x.device,
                                     detail=detail + ' [reduce]')
                    finally:
                        L.cn_set_option(b'wgrad_phase', 0)
                else:
                    call()
            else:
                conv2d_wgrad(x, dy, tmp, 8, K, R, S2, (mod.stride[0], 1), (0, 0), beta=0.0, tag=tag)
            check(L.cn_wgrad_unpack_pairs(ptr(tmp), ptr(mod.grad_view('weight')), K, R, S, mod.in_channels, 1.0,
                                          stream_of(x)), 'cn_wgrad_unpack_pairs')
            return tmp

        if SIDE.active(x):
            SIDE.submit(x.device, lambda: (x, dy, run('side')), mod._notify_grad_ready, dy)
        else:
            run('main')
            mod._notify_grad_ready()
        return None, None, None


def _lazy_dy_ok(bn_mod, y):
    """This junction BatchNorm's backward apply can be left to the convolution that produced its input."""
    conv = getattr(bn_mod, 'producer_conv', None)
    if not LAZY_DY or conv is None:
        return False
    if getattr(conv, 'junction_conv1', False):
        return False    # conv1 of a block: its data gradient is the junction kernel (addend + reduction), not a lazy consumer
    if getattr(conv, 'kernel_size', None) != (1, 1) or getattr(conv, 'bias', None) is not None:
        return False
    if y.shape[-1] > 512 or y.dtype not in (torch.bfloat16, torch.float16, torch.float32):
        return False
    return y.numel() * _esize(y) >= LAZY_DY_MIN_MB * 2 ** 20


class BatchNormActFunction(Function):
    """z = act(BN(y) + residual), training mode (batch statistics)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, residual, mod, relu, defer_apply=False, res_bn=None):
        N, H, W, C = y.shape
        M = N * H * W
        L = _L()
        code = dtype_code(y.dtype)
        ws = workspace(L.cn_bn_workspace(M, C, code), y.device)
        # defer_apply (a projection shortcut's BatchNorm, DUAL_BN): statistics only, the junction applies them; the
        # "output" handed to autograd is the input itself (the junction reads it through this BatchNorm's stats)
        defer = bool(defer_apply) and residual is None and not relu and _sync_group(mod) is None
        # res_bn: `residual` is that BatchNorm's deferred input
        dual = None
        if res_bn is not None:
            d = getattr(res_bn, '_deferred', None)
            res_bn._deferred = None
            if d is not None and residual is not None and d[0] == residual.data_ptr() and _sync_group(mod) is None \
                    and tuple(residual.shape) == tuple(y.shape) and residual.dtype == y.dtype:
                dual = d[1]
            elif d is not None:
                raise _lib.ConvNetHipError('deferred shortcut BatchNorm met a junction that cannot apply it')
        z = None if defer else torch.empty_like(y)
        # lazy z: the junction is finalised here and applied by the 1x1 convolution that consumes it (LAZY_Z)
        cons = getattr(mod, 'consumer_conv', None)
        lazyz = (LAZY_Z and LAZY_Z_SCOPE[0] > 0 and not defer and relu and residual is not None and C <= 512
                 and _sync_group(mod) is None and y.numel() * _esize(y) >= LAZY_Z_MIN_MB * 2 ** 20
                 and tuple(residual.shape) == tuple(y.shape) and residual.dtype == y.dtype and lazy_z_consumer_ok(cons)
                 and cons.in_channels == C)
        icons = getattr(mod, 'inner_consumer_conv', None)
        lazya = (LAZY_A and LAZY_Z_SCOPE[0] > 0 and not defer and relu and residual is None and _sync_group(mod) is None
                 and mod.training and lazy_a_consumer_ok(icons, y))
        zk = None if (defer or dual is not None or lazyz or lazya) else z     # what the statistics call applies itself
        stats = torch.empty(4 * C, dtype=torch.float32, device=y.device)
        mask = None
        if relu and residual is not None:   # 1 bit per output instead of re-reading z in backward
            mask = torch.empty(M * (C // _lib.chunk_elems(y.dtype)), dtype=torch.uint8, device=y.device)
        momentum = mod.effective_momentum()
        track = mod.track_running_stats
        nb = y.numel() * _esize(y)
        ps = take_pending_stats(y)
        if ps is not None and ps.pivot is not None and not (track and ps.pivot == mod.running_mean.data_ptr()):
            ps = None     # sums centred on something that is not this BatchNorm's running mean: take the statistics pass
        COUNTERS['bn_fwd_fused' if ps is not None else 'bn_fwd_plain'] += 1
        sync = _sync_group(mod)
        ctx.sync = sync
        if sync is not None:
            # nn.SyncBatchNorm semantics (main.py:190-191): statistics over the batch of every rank.  Each
            # rank reduces its rows to 2*C doubles, one small all-reduce, then normalise with global stats.
            # Equal per-rank batch sizes (DistributedSampler pads to that) give the global row count.
            group, world = sync
            sums = torch.empty(2 * C, dtype=torch.float64, device=y.device)
            check(L.cn_bn_local_sums(ptr(y), M, C, code, ptr(ps.partial) if ps is not None else None,
                                     ps.rows if ps is not None else 0, ptr(sums), ptr(ws), ws.numel() * 4,
                                     stream_of(y)), 'cn_bn_local_sums')
            _sync_all_reduce(sums, group, world)
            check(L.cn_bn_fwd_train_sums(ptr(y), ptr(residual), ptr(z), ptr(mask), ptr(gamma), ptr(beta),
                                         ptr(mod.running_mean) if track else None,
                                         ptr(mod.running_var) if track else None,
                                         ptr(mod.num_batches_tracked) if track else None, momentum, mod.eps,
                                         ptr(stats), M, C, int(relu), code, ptr(sums), M * world, stream_of(y)),
                  'cn_bn_fwd_train_sums')
        elif ps is not None:   # statistics came out of the producing convolution's epilogue: no pass over y
            PROFILER.run('bn_finalize+bn_apply (stats from conv epilogue)' if zk is not None
                         else 'bn_finalize (stats from conv epilogue; applied by the junction)',
                         (2 if ps.rows <= 512 else 3) - (0 if zk is not None else 1), 0.0,
                         (nb * (3 if residual is not None else 2) + (mask.numel() if mask is not None else 0)
                          if zk is not None else 0) + ps.partial.numel() * 4,
                         lambda: check((L.cn_bn_fwd_train_partials if ps.pivot is None
                                        else L.cn_bn_fwd_train_partials_centered)(
                             ptr(y), ptr(residual) if zk is not None else None, ptr(zk), ptr(mask) if zk is not None else None, ptr(gamma), ptr(beta),
                             ptr(mod.running_mean) if track else None, ptr(mod.running_var) if track else None,
                             ptr(mod.num_batches_tracked) if track else None, momentum, mod.eps, ptr(stats), M, C,
                             int(relu), code, ptr(ps.partial), ps.rows, ptr(ws), ws.numel() * 4, stream_of(y)),
                             'cn_bn_fwd_train_partials'),
                         y.device)
        else:
            PROFILER.run('bn_stats+bn_finalize+bn_apply' if zk is not None else 'bn_stats+bn_finalize', 3 if zk is not None else 2, 0.0,
                         nb * (4 if residual is not None else 3) + (mask.numel() if mask is not None else 0)
                         if zk is not None else nb,
                         lambda: check(L.cn_bn_fwd_train(
                             ptr(y), ptr(residual) if zk is not None else None, ptr(zk), ptr(mask) if zk is not None else None, ptr(gamma), ptr(beta),
                             ptr(mod.running_mean) if track else None, ptr(mod.running_var) if track else None,
                             ptr(mod.num_batches_tracked) if track else None, momentum, mod.eps, ptr(stats), M, C,
                             int(relu), code, ptr(ws), ws.numel() * 4, stream_of(y)), 'cn_bn_fwd_train'),
                         y.device)
        if lazyz:
            cons.__dict__['_lazy_z'] = (z.data_ptr(), y, residual.contiguous(), stats, dual, z, mask, relu)
        elif lazya:
            icons.__dict__['_lazy_a'] = (z.data_ptr(), y, stats, z, relu)
        elif dual is not None:    # both BatchNorms finalised: one apply pass reads y and the shortcut's raw input
            PROFILER.run('bn_apply (junction + projection-shortcut BatchNorm)', 1, 0.0,
                         nb * 3 + (mask.numel() if mask is not None else 0),
                         lambda: check(L.cn_bn_apply_dual(ptr(y), ptr(residual), ptr(z), ptr(mask), ptr(stats), ptr(dual),
                                                          M, C, int(relu), code, stream_of(y)), 'cn_bn_apply_dual'),
                         y.device)
            COUNTERS['bn_fwd_dual'] = COUNTERS.get('bn_fwd_dual', 0) + 1
        ctx.mod = mod
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.out_ptr = z.data_ptr() if z is not None else None
        mod._fwd_ctx = weakref.ref(ctx)      # lets the consumer conv's dgrad fuse this BN's backward reduction
        if mask is not None:
            ctx.save_for_backward(y, stats, mask)
        else:
            ctx.save_for_backward(y, stats)
        if defer:
            mod._deferred = (y.data_ptr(), stats)
            return y     # (autograd hands back an alias of the input: same storage, this node as its grad_fn)
        return z

    @staticmethod
    def backward(ctx, dz):
        saved = ctx.saved_tensors
        y, stats = saved[0], saved[1]
        zmask = saved[2] if len(saved) > 2 else None
        mod = ctx.mod
        N, H, W, C = y.shape
        M = N * H * W
        L = _L()
        code = dtype_code(y.dtype)
        dz = dz.contiguous()
        ws = workspace(L.cn_bn_workspace(M, C, code), y.device)
        dy = torch.empty_like(y)
        want_res = ctx.has_res and ctx.needs_input_grad[3]
        coef = torch.empty(3 * C, dtype=torch.float32, device=y.device)
        nb = y.numel() * _esize(y)
        pp = getattr(mod, '_bwd_partials', None)
        mod._bwd_partials = None
        fused_in = pp is not None and pp[0] == dz.data_ptr() and pp[1] == tuple(dz.shape) and dz.dtype == y.dtype
        if ctx.sync is not None:
            group, world = ctx.sync
            COUNTERS['bn_bwd_fused' if fused_in else 'bn_bwd_plain'] += 1
            local = torch.empty(2 * C, dtype=torch.float64, device=y.device)
            check(L.cn_bn_bwd_local_sums(ptr(dz), ptr(y), ptr(zmask), ptr(stats), M, C, int(ctx.relu), code,
                                         ptr(pp[2]) if fused_in else None, pp[3] if fused_in else 0, ptr(local),
                                         ptr(ws), ws.numel() * 4, stream_of(y)), 'cn_bn_bwd_local_sums')
            glob = local.clone()
            _sync_all_reduce(glob, group, world)
            if fused_in:
                dres = dz if want_res else None
            else:
                dres = torch.empty_like(y) if want_res else None
            check(L.cn_bn_bwd_sums(ptr(dz), ptr(y), ptr(zmask), ptr(mod.weight), ptr(stats), ptr(dy),
                                   None if fused_in else ptr(dres), ptr(mod.grad_view('weight')),
                                   ptr(mod.grad_view('bias')), 1.0, 1.0, ptr(coef), M, C, int(ctx.relu),
                                   int(fused_in), code, ptr(local), ptr(glob), M * world, stream_of(y)),
                  'cn_bn_bwd_sums')
        elif fused_in and _lazy_dy_ok(mod, y):
            # ... and the apply pass is left to the consumers: finalize only, dy = c1*g + c2*y + c3 is formed on the
            # operand loads of the producing convolution's dgrad / wgrad (see LAZY_DY)
            _, _, partial, rows = pp
            COUNTERS['bn_bwd_fused'] += 1
            COUNTERS['bn_bwd_lazy'] = COUNTERS.get('bn_bwd_lazy', 0) + 1
            dres = dz if want_res else None
            with SIDE.mark(coef):
                PROFILER.run('bn_bwd_finalize (lazy dy)', 1 if rows <= 512 else 2, 0.0, partial.numel() * 4,
                             lambda: check(L.cn_bn_bwd_partials(ptr(dz), ptr(y), ptr(mod.weight), ptr(stats), None,
                                                                ptr(mod.grad_view('weight')), ptr(mod.grad_view('bias')),
                                                                1.0, 1.0, ptr(coef), M, C, code, ptr(partial), rows,
                                                                ptr(ws), ws.numel() * 4, stream_of(y)),
                                           'cn_bn_bwd_partials'),
                             y.device)
            mod.producer_conv._lazy_dy = (dz, y, coef)
            dy = _zero_like_placeholder(y)
        elif fused_in:
            # dz arrived masked (g) with its reduction partials from the producing dgrad's epilogue
            _, _, partial, rows = pp
            COUNTERS['bn_bwd_fused'] += 1
            dres = dz if want_res else None       # the residual branch's gradient is g itself
            with SIDE.mark(dy):
                PROFILER.run('bn_bwd_finalize+bn_bwd_apply (reduce in dgrad epilogue)', 2 if rows <= 512 else 3, 0.0,
                             nb * 3 + partial.numel() * 4,
                             lambda: check(L.cn_bn_bwd_partials(ptr(dz), ptr(y), ptr(mod.weight), ptr(stats), ptr(dy),
                                                                ptr(mod.grad_view('weight')), ptr(mod.grad_view('bias')),
                                                                1.0, 1.0, ptr(coef), M, C, code, ptr(partial), rows,
                                                                ptr(ws), ws.numel() * 4, stream_of(y)),
                                           'cn_bn_bwd_partials'),
                             y.device)
        elif not ctx.relu and not want_res and zmask is None and _lazy_dy_ok(mod, y):
            # a BatchNorm with no activation behind it (the projection shortcut's): reduce + finalize, apply left to the
            # producing convolution's dgrad / wgrad (dz needs no mask)
            COUNTERS['bn_bwd_plain'] += 1
            COUNTERS['bn_bwd_lazy'] = COUNTERS.get('bn_bwd_lazy', 0) + 1
            dres = None
            with SIDE.mark(coef):
                PROFILER.run('bn_bwd_reduce+bn_bwd_finalize (lazy dy)', 2, 0.0, nb * 2,
                             lambda: check(L.cn_bn_bwd(ptr(dz), ptr(y), None, ptr(mod.weight), ptr(stats), None,
                                                       None, ptr(mod.grad_view('weight')), ptr(mod.grad_view('bias')),
                                                       1.0, 1.0, ptr(coef), M, C, 0, code, ptr(ws),
                                                       ws.numel() * 4, stream_of(y)), 'cn_bn_bwd'),
                             y.device)
            mod.producer_conv._lazy_dy = (dz, y, coef)
            dy = _zero_like_placeholder(y)
        else:
            COUNTERS['bn_bwd_plain'] += 1
            dres = torch.empty_like(y) if want_res else None
            with SIDE.mark(dy):
                PROFILER.run('bn_bwd_reduce+bn_bwd_finalize+bn_bwd_apply', 3, 0.0,
                             nb * (5 + (1 if dres is not None else 0)) + (2 * zmask.numel() if zmask is not None else 0),
                             lambda: check(L.cn_bn_bwd(ptr(dz), ptr(y), ptr(zmask), ptr(mod.weight), ptr(stats), ptr(dy),
                                                       ptr(dres), ptr(mod.grad_view('weight')), ptr(mod.grad_view('bias')),
                                                       1.0, 1.0, ptr(coef), M, C, int(ctx.relu), code, ptr(ws),
                                                       ws.numel() * 4, stream_of(y)), 'cn_bn_bwd'),
                             y.device)
        mod._notify_grad_ready()
        holder = getattr(mod, '_res_holder', None)
        if holder is not None:
            holder.dres, holder.sub = dres, 1
            holder.fused = False
        return dy, None, None, dres, None, None, None, None


def batch_norm_infer(y, residual, mod, relu):
    take_pending_stats(y)     # eval mode: the partials (if any) are not needed
    N, H, W, C = y.shape
    z = torch.empty_like(y)
    coeffs = torch.empty(2 * C, dtype=torch.float32, device=y.device)
    check(_L().cn_bn_fwd_infer(ptr(y), ptr(residual), ptr(z), ptr(mod.weight), ptr(mod.bias),
                               ptr(mod.running_mean), ptr(mod.running_var), mod.eps, ptr(coeffs), N * H * W, C,
                               int(relu), dtype_code(y.dtype), stream_of(y)), 'cn_bn_fwd_infer')
    return z


class MaxPool2dFunction(Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        N, H, W, C = x.shape
        P, Q = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = torch.empty((N, P, Q, C), dtype=x.dtype, device=x.device)
        idx = torch.empty((N, P, Q, C), dtype=torch.uint8, device=x.device)
        PROFILER.run('maxpool_fwd', 1, 0.0, x.numel() * _esize(x) + y.numel() * (_esize(y) + 1),
                     lambda: check(_L().cn_maxpool_fwd(ptr(x), ptr(y), ptr(idx), N, H, W, C, k, stride, pad,
                                                       dtype_code(x.dtype), stream_of(x)), 'cn_maxpool_fwd'),
                     x.device)
        ctx.cfg = (N, H, W, C, k, stride, pad)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, H, W, C, k, stride, pad = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
        PROFILER.run('maxpool_bwd', 1, 0.0, dx.numel() * _esize(dx) + dy.numel() * (_esize(dy) + 1),
                     lambda: check(_L().cn_maxpool_bwd(ptr(dy), ptr(idx), ptr(dx), N, H, W, C, k, stride, pad,
                                                       dtype_code(dy.dtype), stream_of(dy)), 'cn_maxpool_bwd'),
                     dy.device)
        return dx, None, None, None


class BnReluMaxPoolFunction(Function):
    """maxpool(relu(BN(y))) of the stem (models/resnet.py:228-230) without the normalised 112x112 map:
    forward = BatchNorm statistics / coefficients (from the conv epilogue partials when present) +
    one pooling pass over the pre-BN tensor; backward = cn_bn_bwd_maxpool (the pool's gather backward
    folded into both BatchNorm-backward passes).  Bit-identical to the unfused chain."""

    @staticmethod
    def forward(ctx, y, gamma, beta, mod, k, stride, pad):
        N, H, W, C = y.shape
        M = N * H * W
        L = _L()
        code = dtype_code(y.dtype)
        ws = workspace(L.cn_bn_workspace(M, C, code), y.device)
        stats = torch.empty(4 * C, dtype=torch.float32, device=y.device)
        momentum = mod.effective_momentum()
        track = mod.track_running_stats
        rm = ptr(mod.running_mean) if track else None
        rv = ptr(mod.running_var) if track else None
        nbt = ptr(mod.num_batches_tracked) if track else None
        ps = take_pending_stats(y)
        if ps is not None and ps.pivot is not None and not (track and ps.pivot == mod.running_mean.data_ptr()):
            ps = None
        COUNTERS['bn_fwd_fused' if ps is not None else 'bn_fwd_plain'] += 1
        P, Q = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        out = torch.empty((N, P, Q, C), dtype=y.dtype, device=y.device)
        idx = torch.empty((N, P, Q, C), dtype=torch.uint8, device=y.device)
        # the pre-BatchNorm value of every winning tap: the backward sums then run over the pooled map
        xmax = torch.empty_like(out) if (STEM_XMAX and ctx.needs_input_grad[0]) else None

        def run():
            if ps is not None:
                fn = L.cn_bn_fwd_train_partials if ps.pivot is None else L.cn_bn_fwd_train_partials_centered
                check(fn(ptr(y), None, None, None, ptr(gamma), ptr(beta), rm, rv, nbt, momentum,
                         mod.eps, ptr(stats), M, C, 1, code, ptr(ps.partial), ps.rows, ptr(ws),
                         ws.numel() * 4, stream_of(y)), 'cn_bn_fwd_train_partials')
            else:
                check(L.cn_bn_fwd_train(ptr(y), None, None, None, ptr(gamma), ptr(beta), rm, rv, nbt, momentum, mod.eps,
                                        ptr(stats), M, C, 1, code, ptr(ws), ws.numel() * 4, stream_of(y)),
                      'cn_bn_fwd_train')
            if xmax is not None:
                check(L.cn_maxpool_fwd_bnrelu_xmax(ptr(y), ptr(stats[2 * C:3 * C]), ptr(stats[3 * C:]), ptr(out), ptr(idx),
                                                   ptr(xmax), N, H, W, C, k, stride, pad, code, stream_of(y)),
                      'cn_maxpool_fwd_bnrelu_xmax')
            else:
                check(L.cn_maxpool_fwd_bnrelu(ptr(y), ptr(stats[2 * C:3 * C]), ptr(stats[3 * C:]), ptr(out), ptr(idx), N, H,
                                              W, C, k, stride, pad, code, stream_of(y)), 'cn_maxpool_fwd_bnrelu')
        PROFILER.run('bn_finalize+maxpool_fwd_bnrelu (stem)', 3, 0.0,
                     y.numel() * _esize(y) * (1 if ps is not None else 2)
                     + out.numel() * (_esize(out) * (2 if xmax is not None else 1) + 1), run, y.device)
        ctx.mod = mod
        ctx.cfg = (N, H, W, C, k, stride, pad)
        ctx.has_xmax = xmax is not None
        if xmax is not None:
            ctx.save_for_backward(y, stats, idx, xmax)
        else:
            ctx.save_for_backward(y, stats, idx)
        return out

    @staticmethod
    def backward(ctx, dpool):
        y, stats, idx = ctx.saved_tensors[:3]
        xmax = ctx.saved_tensors[3] if ctx.has_xmax else None
        mod = ctx.mod
        N, H, W, C, k, stride, pad = ctx.cfg
        L = _L()
        code = dtype_code(y.dtype)
        dpool = dpool.contiguous()
        ws = workspace(L.cn_bn_workspace(N * H * W, C, code), y.device)
        dy = torch.empty_like(y)
        coef = torch.empty(3 * C, dtype=torch.float32, device=y.device)
        COUNTERS['bn_bwd_plain'] += 1
        with SIDE.mark(dy):
            PROFILER.run('bn_bwd_maxpool (stem)', 3, 0.0,
                         (y.numel() * _esize(y) * 3 + dpool.numel() * (_esize(dpool) + 1) * 2) if xmax is None else
                         (y.numel() * _esize(y) * 2 + dpool.numel() * (3 * _esize(dpool) + 1)),
                         (lambda: check(L.cn_bn_bwd_maxpool(ptr(dpool), ptr(idx), ptr(y), ptr(mod.weight), ptr(stats), ptr(dy),
                                                            ptr(mod.grad_view('weight')), ptr(mod.grad_view('bias')), 1.0, 1.0,
                                                            ptr(coef), N, H, W, C, k, stride, pad, code, ptr(ws),
                                                            ws.numel() * 4, stream_of(y)), 'cn_bn_bwd_maxpool'))
                         if xmax is None else
                         (lambda: check(L.cn_bn_bwd_maxpool_xmax(ptr(dpool), ptr(idx), ptr(y), ptr(xmax), ptr(mod.weight),
                                                                 ptr(stats), ptr(dy), ptr(mod.grad_view('weight')),
                                                                 ptr(mod.grad_view('bias')), 1.0, 1.0, ptr(coef), N, H, W, C,
                                                                 k, stride, pad, code, ptr(ws), ws.numel() * 4,
                                                                 stream_of(y)), 'cn_bn_bwd_maxpool_xmax')),
                         y.device)
        mod._notify_grad_ready()
        return dy, None, None, None, None, None, None


class GlobalAvgPoolFunction(Function):
    @staticmethod
    def forward(ctx, x):
        N, H, W, C = x.shape
        y = torch.empty((N, 1, 1, C), dtype=x.dtype, device=x.device)
        check(_L().cn_avgpool_fwd(ptr(x), ptr(y), N, H * W, C, dtype_code(x.dtype), stream_of(x)),
              'cn_avgpool_fwd')
        ctx.shape = (N, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
        check(_L().cn_avgpool_bwd(ptr(dy), ptr(dx), N, H * W, C, dtype_code(dy.dtype), stream_of(dy)),
              'cn_avgpool_bwd')
        return dx


class ReLUFunction(Function):
    @staticmethod
    def forward(ctx, x):
        z = torch.empty_like(x)
        check(_L().cn_eltwise(1, ptr(z), ptr(x), None, x.numel(), dtype_code(x.dtype), stream_of(x)), 'cn_eltwise')
        ctx.save_for_backward(z)
        return z

    @staticmethod
    def backward(ctx, dz):
        (z,) = ctx.saved_tensors
        dz = dz.contiguous()
        dx = torch.empty_like(dz)
        check(_L().cn_eltwise(2, ptr(dx), ptr(dz), ptr(z), dz.numel(), dtype_code(dz.dtype), stream_of(dz)),
              'cn_eltwise')
        return dx


class DropoutFunction(Function):
    """z = x * mask; `mask` (compute dtype, NHWC) already holds keep/(1-p)."""

    @staticmethod
    def forward(ctx, x, mask):
        z = torch.empty_like(x)
        check(_L().cn_eltwise(3, ptr(z), ptr(x), ptr(mask), x.numel(), dtype_code(x.dtype), stream_of(x)), 'cn_eltwise')
        ctx.save_for_backward(mask)
        return z

    @staticmethod
    def backward(ctx, dz):
        (mask,) = ctx.saved_tensors
        dz = dz.contiguous()
        dx = torch.empty_like(dz)
        check(_L().cn_eltwise(3, ptr(dx), ptr(dz), ptr(mask), dz.numel(), dtype_code(dz.dtype), stream_of(dz)),
              'cn_eltwise')
        return dx, None


class SmallLinearFunction(Function):
    """Dense layer whose output width is not a multiple of the 16-byte chunk (MNIST's 10-way head)."""

    @staticmethod
    def forward(ctx, x2d, weight, bias, mod):
        B, C = x2d.shape
        K = mod.out_channels
        y = torch.empty((B, K), dtype=torch.float32, device=x2d.device)
        check(_L().cn_small_linear(0, ptr(x2d), ptr(mod.master_view('weight')), ptr(bias), ptr(y), None, None, B, C,
                                   K, dtype_code(x2d.dtype), stream_of(x2d)), 'cn_small_linear')
        ctx.mod = mod
        ctx.save_for_backward(x2d)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x2d,) = ctx.saved_tensors
        mod = ctx.mod
        B, C = x2d.shape
        K = mod.out_channels
        dy = dy.contiguous().float()
        L = _L()
        code = dtype_code(x2d.dtype)
        db = mod.grad_view('bias') if mod.bias is not None else None
        check(L.cn_small_linear(2, ptr(x2d), None, None, ptr(dy), ptr(mod.grad_view('weight')), ptr(db), B, C, K, code,
                                stream_of(x2d)), 'cn_small_linear')
        mod._notify_grad_ready()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2d)
            check(L.cn_small_linear(1, ptr(dy), ptr(mod.master_view('weight')), None, ptr(dx), None, None, B, C, K,
                                    code, stream_of(x2d)), 'cn_small_linear')
        return dx, None, None, None


class ForkFunction(Function):
    """Two aliases of one activation (block input -> conv branch + residual branch).  Backward sums
    the two incoming gradients with our own kernel, so autograd never launches its accumulate."""

    @staticmethod
    def forward(ctx, x, holder=None):
        ctx.holder = holder
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        holder = ctx.holder
        fused = holder is not None and holder.fused
        first = holder.dres if holder is not None else None
        sub = holder.sub if holder is not None else 1
        if holder is not None:
            holder.dres, holder.fused, holder.sub = None, False, 1
        if sub == 2 and not fused and first is not None:
            # a subsampled projection-shortcut gradient that no dgrad epilogue picked up (not the case in the ResNet
            # blocks: conv1 always does): scatter it into a dense tensor here (data movement only) and add
            dense = torch.zeros((first.shape[0], gb.shape[1], gb.shape[2], first.shape[3]), dtype=first.dtype,
                                device=first.device)
            dense[:, ::2, ::2, :] = first
            return (add_(ga.contiguous(), dense) if ga is not None else dense), None
        if ga is None:
            return gb, None
        if gb is None:
            return ga, None
        if fused:   # the later of the two branch gradients already contains the earlier one
            return (gb if (first is not None and first.data_ptr() == ga.data_ptr()) else ga), None
        ga = ga.contiguous()
        return add_(ga, gb.contiguous()), None


class ResGradHolder(object):
    """Per-block mailbox for the two gradients that meet at a residual block's input: whichever branch
    finishes first (the last BN's `dres` in identity blocks, one of the two convs in downsample
    blocks) parks its gradient here; the other branch's conv dgrad adds it in its epilogue (one pass
    instead of a separate add kernel), independent of autograd's execution order."""
    __slots__ = ('dres', 'fused', 'sub')   # sub = 2: dres holds only the even (h, w) pixels (stride-2 1x1 projection)

    def __init__(self):
        self.dres, self.fused, self.sub = None, False, 1


class SoftmaxCrossEntropyFunction(Function):
    """loss = mean_b CE(logits_b, target_b) (+ label smoothing); also accumulates the reference's
    loss / prec@1 / prec@5 meters on the device (crit.meters) when a buffer is attached."""

    @staticmethod
    def forward(ctx, logits, target, crit):
        B, K = logits.shape
        logits = logits.contiguous()
        if logits.dtype != torch.float32:
            raise _lib.ConvNetHipError('criterion expects fp32 logits')
        target = target.contiguous()
        row = torch.empty(3 * B, dtype=torch.float32, device=logits.device)
        step_out = torch.empty(3, dtype=torch.float32, device=logits.device)
        check(_L().cn_softmax_ce(ptr(logits), ptr(target), None, _lib.F32, ptr(row), ptr(step_out),
                                 ptr(crit.meters), B, K, 1.0, None, crit.smooth_eps, stream_of(logits)),
              'cn_softmax_ce')
        crit.last_step = step_out
        ctx.crit = crit
        ctx.save_for_backward(logits, target, row)
        return step_out[0]

    @staticmethod
    def backward(ctx, go):
        logits, target, row = ctx.saved_tensors
        B, K = logits.shape
        dlogits = torch.empty_like(logits)
        go = go.contiguous().to(torch.float32)
        check(_L().cn_softmax_ce(ptr(logits), ptr(target), ptr(dlogits), _lib.F32, ptr(row), None, None, B, K,
                                 1.0 / B, ptr(go), ctx.crit.smooth_eps, stream_of(logits)), 'cn_softmax_ce')
        return dlogits, None, None


def accuracy_counts(logits, target):
    """Returns a device tensor [mean CE loss, prec@1 (%), prec@5 (%)] of this batch."""
    B, K = logits.shape
    logits = logits.contiguous().float()
    row = torch.empty(3 * B, dtype=torch.float32, device=logits.device)
    out = torch.empty(3, dtype=torch.float32, device=logits.device)
    check(_L().cn_softmax_ce(ptr(logits), ptr(target.contiguous()), None, _lib.F32, ptr(row), ptr(out), None, B, K,
                             1.0, None, 0.0, stream_of(logits)), 'cn_softmax_ce')
    return out
