"""Simulated 8-bit training operators of BASELINE config 5 (``resnet(quantize=True)``) on the HIP kernels.

Same classes, constructor signatures, ``state_dict`` keys and arithmetic as /root/reference
models/modules/quantize.py (QuantMeasure :140-182, QConv2d :185-220, QLinear :223-253, RangeBN :256-330), which
models/resnet.py:387-391 swaps in for Conv2d / Linear / BatchNorm2d.  Like the reference this is *simulated*
integer training: activations, weights and gradients are snapped to 2^bits-level grids (cn_quantize*,
cn_rangebn_* in csrc/quant.hip) and stay floating point; the convolutions run on the ordinary MFMA kernels.

Semantics kept from the reference (restated and pinned in oracle/quant_oracle.py):
  * activations: per-tensor asymmetric 8 bit, zero point / range = batch mean of the per-sample min / (max-min),
    tracked as running = running*0.1 + new*0.9 for eval mode; straight-through gradient;
  * weights: 8 bit per output channel; classifier bias 16 bit over its global range;
  * "bi-precision" backward: the weight gradient sees the full-precision dy, the input gradient sees dy
    quantised to 8 bits over its global min / max with stochastic rounding;
  * RangeBN: statistics from 16 chunk-wise (max - min), gradient routed through the maxima / minima, output
    gradient quantised like a convolution's.
Deviations (both documented in oracle/make_golden_quant.py, the reference does not run without them on this
torch): a zero quantisation range makes the quantiser the identity instead of dividing by zero, and the
gradient quantiser's forward is a copy instead of an alias.

Stochastic rounding noise: by default the kernels' own counter-based generator (seeded per call from
``manual_seed``); tests install ``set_noise_source`` to feed the very U(-0.5, 0.5) stream the reference draws
from torch's CPU generator (one draw per gradient tensor, NCHW element order, in backward execution order).
"""
import math
import os

import torch
import torch.nn as tnn
from torch.autograd import Function

from . import _lib, flags, nn as cnn, ops
from ._lib import check, dtype_code, ptr, stream_of

# Producer-side fusions of the quantised chain (round 4; flag quant_fuse, bit-identical to the separate passes,
# tests/test_quant.py::test_producer_side_fusions_*):
#  * RangeBN's input quantiser runs inside its statistics pass, which stores the snapped tensor (cn_rangebn_fwd_q): the
#    separate quantise pass - one read of every convolution output - disappears.  (Round 3 snapped on load in all three
#    RangeBN kernels without storing: three divisions per element, VALU-bound, slower; removed.)
#  * the kernels that PRODUCE a tensor the next operator quantises also emit its per-sample min / max: RangeBN's apply
#    (-> the next activation quantiser), the residual add + ReLU (-> the next block's), RangeBN's backward apply with the
#    routing folded in (-> the convolution's gradient quantiser), the ReLU-mask pass (-> RangeBN's gradient quantiser).
#    The quantisers find them in a stash and skip their own min / max pass over the tensor.
FUSE_QUANT = flags.on('quant_fuse')
# the block-input gradient sum folded into the later data gradient's epilogue (QConv2dFunction.backward)
JUNCTION_ADD = flags.on('quant_junction_add')
# QConv2d's weight gradients on the side stream (0: on the chain, as through round 5)
WGRAD_SIDE = flags.on('quant_wgrad_side')
# 8-bit LEVEL storage (round 6): RangeBN's snapped input (saved for its backward pass) and the quantised gradient its
# backward kernels read are kept as one byte per element and de-quantised on load - bit-identical to storing the snapped
# values, 6 of the 16 bytes per element RangeBN's passes move.  0: values in the compute dtype, as through round 5.
STORE8 = flags.on('quant_store8')
# the producers of a gradient (ReLU-mask pass, RangeBN's backward apply) also reduce its per-sample extremes to the gradient
# quantiser's [zero_point, range]: ~105 cn_qparams launches per ResNet-50 step less (same values)
QP_FROM_PRODUCER = flags.on('quant_qp_from_producer')

# produced tensor -> its per-sample min / max ([rows][2] floats), keyed by storage address; the entry holds the tensor, so
# the address cannot be recycled while it lives; `uses` consumers may take it, tick() (next forward) drops the rest
_MM_STASH = {}


def _stash_minmax(t, rows, mm, uses=1, qp_extreme=None):
    """qp_extreme: [zero_point, range] of the 'extreme' reduction of mm (what a GRADIENT quantiser derives from it), when the
    producer's final kernel emitted it too."""
    _MM_STASH[t.data_ptr()] = [t, rows, mm, uses, qp_extreme]


_LAST_QP_EXTREME = [None]


def _take_minmax(x, rows):
    e = _MM_STASH.get(x.data_ptr())
    _LAST_QP_EXTREME[0] = None
    if e is None or e[1] != rows or e[0].shape != x.shape or e[0].dtype != x.dtype:
        return None
    e[3] -= 1
    if e[3] <= 0:
        del _MM_STASH[x.data_ptr()]
    _LAST_QP_EXTREME[0] = e[4]
    return e[2]
_NOISE_SOURCE = None
_SEED = [0x5EED5EED]


def set_noise_source(fn):
    """fn(shape) -> CPU fp32 tensor of U(-0.5, 0.5) noise in the reference's element order (NCHW for 4-d
    gradients); None restores the in-kernel generator."""
    global _NOISE_SOURCE
    _NOISE_SOURCE = fn


def manual_seed(seed):
    _SEED[0] = int(seed) & 0xFFFFFFFFFFFF


def _next_seed():
    _SEED[0] = (_SEED[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
    return _SEED[0]


def _L():
    return _lib.load()


# Device-resident step counter mixed into the generator seed of every stochastic quantiser launch (cn_quantize_s): the
# per-call seeds are host constants - frozen when the step is captured into a HIP graph - and the counter, advanced by
# one tiny launch at the start of every forward pass (tick, also captured), makes each replay draw fresh noise.
_STEP_COUNTER = {}


def _step_counter(device):
    dev = torch.device(device)
    key = str(dev)
    c = _STEP_COUNTER.get(key)
    if c is None:
        c = _STEP_COUNTER[key] = torch.zeros(1, dtype=torch.int64, device=dev)
    return c


def tick(device):
    """Advance the noise step counter (once per training step, on the compute stream)."""
    _MM_STASH.clear()
    c = _step_counter(device)
    check(_L().cn_counter_inc(ptr(c), stream_of(c)), 'cn_counter_inc')


def minmax_rows(x, rows):
    """[rows][2] = per-row {min, max} of a contiguous tensor viewed as [rows][numel/rows]."""
    mm = _take_minmax(x, rows)
    if mm is not None:         # the kernel that produced x already measured it
        return mm
    L = _L()
    row_len = x.numel() // rows
    out = torch.empty(rows * 2, dtype=torch.float32, device=x.device)
    ws = ops.workspace(L.cn_minmax_workspace(rows, row_len), x.device, 'quant')
    ops.PROFILER.run('quant: minmax_rows', 2, 0.0, x.numel() * x.element_size(),
                     lambda: check(L.cn_minmax_rows(ptr(x), rows, row_len, dtype_code(x.dtype), ptr(out), ptr(ws),
                                                    ws.numel() * 4, stream_of(x)), 'cn_minmax_rows'), x.device)
    return out


def qparams(minmax, rows, mode, running_zp=None, running_range=None, momentum=0.1):
    qp = torch.empty(2, dtype=torch.float32, device=minmax.device)
    check(_L().cn_qparams(ptr(minmax), rows, mode, ptr(qp), ptr(running_zp), ptr(running_range), momentum,
                          stream_of(minmax)), 'cn_qparams')
    return qp


def quantize(x, zp, rng, num_bits=8, noise=None, stochastic=False):
    """zp / rng: one-element fp32 device tensors (or views)."""
    y = torch.empty_like(x)
    seed = _next_seed() if (stochastic and noise is None) else 0
    step = _step_counter(x.device) if (stochastic and noise is None) else None
    ops.PROFILER.run('quant: quantize', 1, 0.0, 2 * x.numel() * x.element_size(),
                     lambda: check(_L().cn_quantize_s(ptr(x), ptr(y), x.numel(), dtype_code(x.dtype), ptr(zp), ptr(rng),
                                                      num_bits, ptr(noise), int(stochastic), seed, ptr(step), stream_of(x)),
                                   'cn_quantize_s'), x.device)
    return y


def eltwise_mm(op, b, c, rows, want_qp=False):
    """a = b * (c > 0) (op 2) or relu(b + c) (op 4) and [rows][2] per-row min / max of a (cn_eltwise_mm).
    want_qp: returns (a, mm, qp) with qp = the gradient quantiser's [zero_point, range] of a (None beyond 256 rows)."""
    L = _L()
    a = torch.empty_like(b)
    mm = torch.empty(rows * 2, dtype=torch.float32, device=b.device)
    code = dtype_code(b.dtype)
    ws = ops.workspace(L.cn_eltwise_mm_workspace(b.numel(), rows, code), b.device, 'quant')
    if want_qp and rows <= 256 and QP_FROM_PRODUCER:
        qp = torch.empty(2, dtype=torch.float32, device=b.device)
        ops.PROFILER.run('quant: eltwise+minmax', 2, 0.0, 3 * b.numel() * b.element_size(),
                         lambda: check(L.cn_eltwise_mm_qp(op, ptr(a), ptr(b), ptr(c), b.numel(), code, rows, ptr(mm), ptr(qp),
                                                          ptr(ws), ws.numel() * 4, stream_of(b)), 'cn_eltwise_mm_qp'), b.device)
        return a, mm, qp
    ops.PROFILER.run('quant: eltwise+minmax', 2, 0.0, 3 * b.numel() * b.element_size(),
                     lambda: check(L.cn_eltwise_mm(op, ptr(a), ptr(b), ptr(c), b.numel(), code, rows, ptr(mm), ptr(ws),
                                                   ws.numel() * 4, stream_of(b)), 'cn_eltwise_mm'), b.device)
    return (a, mm, None) if want_qp else (a, mm)


def _mm_ok(t):
    """The fused producers split a tensor into its batch samples: whole 16-byte chunks per sample, bf16 / fp32."""
    return (FUSE_QUANT and t.dim() >= 2 and t.dtype in (torch.bfloat16, torch.float32) and t.shape[0] <= 8192
            and (t.numel() // t.shape[0]) % _lib.chunk_elems(t.dtype) == 0)


def _noise_like(g):
    """Reference-ordered rounding noise for gradient tensor g (NHWC or 2-d), or None (in-kernel generator)."""
    if _NOISE_SOURCE is None:
        return None
    if g.dim() == 4:
        N, H, W, C = g.shape
        n = _NOISE_SOURCE((N, C, H, W)).to(device=g.device, dtype=torch.float32)
        return ops.nchw_to_nhwc(n, torch.float32, C)
    return _NOISE_SOURCE(tuple(g.shape)).to(device=g.device, dtype=torch.float32).contiguous()


def quantize_grad(g, num_bits=8, levels=False):
    """UniformQuantizeGrad.backward (quantize.py:101-112): global min / max, stochastic rounding.
    levels: return (uint8 levels, [zero_point, range]) instead of the snapped values (STORE8: the consumer - RangeBN's
    backward kernels - de-quantises on load; one byte per element written and read instead of two)."""
    g = g.contiguous()
    rows = g.shape[0]
    mm = minmax_rows(g, rows)
    qp = _LAST_QP_EXTREME[0]          # the producer of g measured it AND reduced it (cn_eltwise_mm_qp / cn_rangebn_bwd_q8) ...
    _LAST_QP_EXTREME[0] = None
    if qp is None:
        qp = qparams(mm, rows, 1)     # ... else one more tiny launch
    if not levels:
        return quantize(g, qp[0:1], qp[1:2], num_bits, noise=_noise_like(g), stochastic=True)
    noise = _noise_like(g)
    y8 = torch.empty(g.shape, dtype=torch.uint8, device=g.device)
    seed = _next_seed() if noise is None else 0
    step = _step_counter(g.device) if noise is None else None
    ops.PROFILER.run('quant: quantize (8-bit levels out)', 1, 0.0, g.numel() * (g.element_size() + 1),
                     lambda: check(_L().cn_quantize_levels(ptr(g), ptr(y8), g.numel(), dtype_code(g.dtype), ptr(qp[0:1]),
                                                           ptr(qp[1:2]), num_bits, ptr(noise), 1, seed, ptr(step), stream_of(g)),
                                   'cn_quantize_levels'), g.device)
    return y8, qp


def _store8_ok(t, bits):
    """8-bit level storage applies: flag on, <= 8 bits, whole 16-byte chunks of a 16 / 32-bit float tensor."""
    return (STORE8 and bits <= 8 and t.dtype in (torch.bfloat16, torch.float32)
            and t.numel() % _lib.chunk_elems(t.dtype) == 0)


class QuantMeasure(tnn.Module):
    """quantize.py:140-182 (measure=False): running zero point / range, quantises its input."""
    quantized_op = True

    @property
    def no_graph(self):
        """The step can be captured into a HIP graph with the kernels' own noise generator (device step counter in the
        seed); a host-side noise source (tests replaying the reference's stream) rules it out."""
        return _NOISE_SOURCE is not None

    def __init__(self, num_bits=8, shape_measure=(1,), flatten_dims=(1, -1), inplace=False, dequantize=True,
                 stochastic=False, momentum=0.1, measure=False):
        super().__init__()
        if measure or not dequantize or stochastic or tuple(flatten_dims) != (1, -1):
            raise NotImplementedError('QuantMeasure: only the configuration QConv2d / QLinear / RangeBN use')
        self.register_buffer('running_zero_point', torch.zeros(*shape_measure))
        self.register_buffer('running_range', torch.zeros(*shape_measure))
        self.num_bits, self.momentum = num_bits, momentum

    def params(self, x, training=None):
        """(zero_point, range) one-element device tensors for x; training: measured on x (and folded into the
        running buffers), else the running buffers."""
        training = self.training if training is None else training
        if training:
            rows = x.shape[0]
            qp = qparams(minmax_rows(x, rows), rows, 0, self.running_zero_point, self.running_range, self.momentum)
            return qp[0:1], qp[1:2]
        return self.running_zero_point, self.running_range

    def qparams_tensor(self, x, training=None):
        """[zero_point, range] as one 2-element device tensor (what the fused kernels take)."""
        training = self.training if training is None else training
        if training:
            rows = x.shape[0]
            return qparams(minmax_rows(x, rows), rows, 0, self.running_zero_point, self.running_range, self.momentum)
        return torch.cat([self.running_zero_point.reshape(1), self.running_range.reshape(1)]).float().contiguous()

    def forward(self, x, training=None):
        """x: contiguous activation whose leading dimension is the batch."""
        zp, rng = self.params(x, training)
        return quantize(x, zp, rng, self.num_bits)


def _quantize_filters(mod, num_bits):
    """Per-output-channel quantisation of the fp32 master filter into the compute-dtype KRSC / CRSK copies."""
    mod.ensure_prepared()
    if getattr(mod._arena, '_qrows', None) is not None:
        return      # every filter of the model was snapped and laid out by the arena's one refresh (engine.prepare_weights)
    K = mod.out_channels
    taps = mod.kernel_size[0] * mod.kernel_size[1]
    c_real = mod.in_channels
    master = mod.master_view('weight')
    tmp = torch.empty_like(master)
    check(_L().cn_quantize_rows(ptr(master), ptr(tmp), K, taps * c_real, num_bits, stream_of(master)),
          'cn_quantize_rows')
    c_pad = mod.w_krsc.numel() // (K * taps)
    ops.weight_prep(tmp, mod.w_krsc, mod.w_crsk, K, taps, c_real, c_pad)


# ---- true int8 MFMA forward (csrc/qconv_i8.hip) ------------------------------------------------------------
# flag quant_int8 (or QConv2d.int8_forward = True): the forward product of every eligible QConv2d
# (input channels a multiple of 16: everything but the 3-channel stem) runs on v_mfma_i32_32x32x32_i8 instead
# of the float kernels on dequantised operands.  Same result up to fp32 rounding; backward unchanged.
INT8_FORWARD = flags.on('quant_int8')
_I8_GEOM = {}


def _border_classes(n_in, k, stride, pad, n_out):
    """Per output index: which taps fall inside the image -> (class id per output index, [(lo, hi)] per class)."""
    ids, pats = [], []
    for o in range(n_out):
        lo = max(0, pad - o * stride)
        hi = min(k - 1, n_in - 1 + pad - o * stride)
        if (lo, hi) not in pats:
            pats.append((lo, hi))
        ids.append(pats.index((lo, hi)))
    return ids, pats


def _i8_geometry(H, W, R, S, stride, pad, device):
    key = (H, W, R, S, tuple(stride), tuple(pad), str(device))
    g = _I8_GEOM.get(key)
    if g is None:
        P, Q = ops.conv_out_hw(H, W, R, S, stride, pad)
        rid, rp = _border_classes(H, R, stride[0], pad[0], P)
        cid, cp = _border_classes(W, S, stride[1], pad[1], Q)
        mask = torch.zeros(len(rp) * len(cp), R * S, dtype=torch.uint8)
        for i, (rlo, rhi) in enumerate(rp):
            for j, (slo, shi) in enumerate(cp):
                for r in range(rlo, rhi + 1):
                    for c in range(slo, shi + 1):
                        mask[i * len(cp) + j, r * S + c] = 1
        g = (torch.tensor(rid, dtype=torch.uint8).to(device), torch.tensor(cid, dtype=torch.uint8).to(device),
             len(cp), mask.to(device), len(rp) * len(cp))
        _I8_GEOM[key] = g
    return g


def int8_eligible(mod, x):
    return ((INT8_FORWARD or getattr(mod, 'int8_forward', False)) and x.shape[-1] % 16 == 0
            and mod.in_channels == x.shape[-1] and mod.out_channels % 8 == 0)


def conv2d_fwd_int8(x, zp, rng, mod):
    """QConv2d's forward product on the int8 matrix cores.  x: the UNquantised NHWC input, (zp, rng) its
    quantisation parameters (device scalars)."""
    L = _L()
    N, H, W, C = x.shape
    K = mod.out_channels
    R, S = mod.kernel_size
    P, Q = ops.conv_out_hw(H, W, R, S, mod.stride, mod.padding)
    dev = x.device
    rowcls, colcls, ncolcls, clsmask, ncls = _i8_geometry(H, W, R, S, mod.stride, mod.padding, dev)
    st = stream_of(x)
    xq = torch.empty((N, H, W, C), dtype=torch.int8, device=dev)
    chansum = torch.empty(N * H * W, dtype=torch.int32, device=dev)
    A = torch.empty(N * P * Q, dtype=torch.int32, device=dev)
    cls = torch.empty(N * P * Q, dtype=torch.uint8, device=dev)
    ops.PROFILER.run('quant int8: levels+chansum+window', 3, 0.0, x.numel() * (x.element_size() + 2),
                     lambda: check(L.cn_i8_prepare_activation(ptr(x), ptr(xq), ptr(chansum), ptr(A), ptr(cls), N, H, W, C,
                                                              R, S, mod.stride[0], mod.stride[1], mod.padding[0],
                                                              mod.padding[1], dtype_code(x.dtype), ptr(zp), ptr(rng),
                                                              ptr(rowcls), ptr(colcls), ncolcls, st),
                                   'cn_i8_prepare_activation'), dev)
    master = mod.master_view('weight')
    wq = torch.empty(K * R * S * C, dtype=torch.int8, device=dev)
    wsum = torch.empty(K * R * S, dtype=torch.int32, device=dev)
    wpar = torch.empty(K * 2, dtype=torch.float32, device=dev)
    check(L.cn_i8_prepare_weight(ptr(master), ptr(wq), ptr(wsum), ptr(wpar), K, R * S, C, st), 'cn_i8_prepare_weight')
    tables = torch.empty((2 + ncls) * K, dtype=torch.float32, device=dev)
    y = torch.empty((N, P, Q, K), dtype=x.dtype, device=dev)
    ops.PROFILER.run(lambda: L.cn_last_kernel_name().decode(), 2, 2.0 * N * P * Q * K * C * R * S,
                     xq.numel() + wq.numel() + y.numel() * y.element_size(),
                     lambda: check(L.cn_conv2d_fwd_i8(ptr(xq), ptr(wq), ptr(y), ptr(A), ptr(cls), ptr(zp), ptr(rng),
                                                      ptr(wpar), ptr(wsum), ptr(clsmask), ncls, ptr(tables), N, H, W, C,
                                                      K, R, S, mod.stride[0], mod.stride[1], mod.padding[0],
                                                      mod.padding[1], dtype_code(x.dtype), st), 'cn_conv2d_fwd_i8'),
                     dev, detail='fwd-int8 %d,%d->%d %dx%d/%d' % (C, H, K, R, R, mod.stride[0]))
    return y


class QConv2dFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, mod, prequantized):
        if not prequantized and int8_eligible(mod, x):
            x = x.contiguous()
            zp, rng = mod.quantize_input.params(x)
            y = conv2d_fwd_int8(x, zp, rng, mod)
            qx = quantize(x, zp, rng, mod.num_bits)      # float copies of both operands: the backward pass
            _quantize_filters(mod, mod.num_bits_weight)  # (wgrad / dgrad) runs on the float kernels
            ctx.mod = mod
            ctx.save_for_backward(qx)
            return y
        if prequantized:
            qx = x
        else:
            # conv1 and the projection shortcut of a block quantise the SAME tensor with the same number of bits: the
            # shortcut (share_q_from = conv1) reuses conv1's per-sample min / max and its quantised copy and only
            # updates its own running range (identical values)
            qm = mod.quantize_input
            src = getattr(mod, 'share_q_from', None)
            st = src.__dict__.pop('_q_stash', None) if src is not None else None
            x = x.contiguous()
            if st is not None and qm.training and st[0] == x.data_ptr() and st[1] == tuple(x.shape) \
                    and st[2] == qm.num_bits:
                qparams(st[3], x.shape[0], 0, qm.running_zero_point, qm.running_range, qm.momentum)
                qx = st[4]
            elif qm.training and getattr(mod, 'share_q_out', False):
                # (QuantMeasure exists in ONE configuration - its constructor refuses stochastic / non-dequantising /
                # other flatten_dims - so the shared copy depends on num_bits only, which the stash records and matches)
                mm = minmax_rows(x, x.shape[0])
                qp = qparams(mm, x.shape[0], 0, qm.running_zero_point, qm.running_range, qm.momentum)
                qx = quantize(x, qp[0:1], qp[1:2], qm.num_bits)
                mod.__dict__['_q_stash'] = (x.data_ptr(), tuple(x.shape), qm.num_bits, mm, qx)
            else:
                qx = qm(x)
        _quantize_filters(mod, mod.num_bits_weight)
        y = ops.conv2d_fwd(qx, mod.w_krsc, None, mod.out_channels, mod.kernel_size[0], mod.kernel_size[1],
                           mod.stride, mod.padding)
        ctx.mod = mod
        ctx.save_for_backward(qx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (qx,) = ctx.saved_tensors
        mod = ctx.mod
        dy = dy.contiguous()
        R, S = mod.kernel_size
        # the weight gradient sees the full-precision dy (bi-precision, quantize.py:115-121) and nothing on the chain waits
        # for it: it goes to the weight-gradient side stream like the float convolutions' (ops.SIDE; dy's producer is the
        # last launch on the chain at this point, so the hand-off event waits for exactly that kernel), beside the
        # gradient quantiser and the data gradient
        if ops.SIDE.active(qx) and WGRAD_SIDE:
            def launch():
                ops.conv2d_wgrad(qx, dy, mod.grad_view('weight'), mod.in_channels, mod.out_channels, R, S, mod.stride,
                                 mod.padding, tag='side')
                return (qx, dy)
            ops.SIDE.submit(qx.device, launch, mod._notify_grad_ready, None)
        else:
            ops.conv2d_wgrad(qx, dy, mod.grad_view('weight'), mod.in_channels, mod.out_channels, R, S, mod.stride,
                             mod.padding)
            mod._notify_grad_ready()
        dx = None
        if ctx.needs_input_grad[0]:   # (the stem never gets here: no noise is drawn for it, as in the reference)
            gq = quantize_grad(dy, mod.num_bits_grad)
            # The two gradients that meet at a block's input (conv1's and the shortcut's: models/resnet.py:154-163) are
            # summed in the data gradient's epilogue of whichever of the two arrives second (ops.ResGradHolder, as in
            # the float blocks) instead of a separate add pass over the block input: fp32 identical, 16-bit storage
            # rounds the sum once instead of twice.
            holder = getattr(mod, '_res_holder', None) if JUNCTION_ADD else None
            addend = None
            if holder is not None and holder.dres is not None and holder.sub == 1 and holder.dres.dtype == gq.dtype \
                    and tuple(holder.dres.shape) == tuple(qx.shape):
                addend, holder.fused = holder.dres, True
            dx = ops.conv2d_dgrad(gq, mod.w_crsk, qx.shape, mod.out_channels, R, S, mod.stride, mod.padding,
                                  addend=addend)
            if holder is not None and addend is None:
                holder.dres, holder.sub, holder.fused = dx, 1, False
        return dx, None, None, None


class QConv2d(cnn.Conv2d):
    """quantize.py:185-220 (biprecision=True, the default and what models/resnet.py uses)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 num_bits=8, num_bits_weight=8, num_bits_grad=8, biprecision=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias)
        if bias or not biprecision or num_bits_grad is None:
            raise NotImplementedError('QConv2d: bias-free bi-precision configuration only (what ResNet uses)')
        self.num_bits, self.num_bits_weight, self.num_bits_grad = num_bits, num_bits_weight or num_bits, num_bits_grad
        self.quantize_input = QuantMeasure(num_bits, shape_measure=(1, 1, 1, 1), flatten_dims=(1, -1))
        self.biprecision = biprecision

    def pair_eligible(self, x_nchw):
        return False

    def forward(self, x):
        self._require_prepared()
        return QConv2dFunction.apply(x, self.weight, self, False)

    def forward_from_nchw(self, x_nchw):
        """The network input is quantised in its fp32 NCHW form (channel padding must stay zero), then laid out."""
        self._require_prepared()
        qx = self.quantize_input(x_nchw.contiguous())
        x = ops.nchw_to_nhwc(qx, self.compute_dtype, self.padded_in_channels())
        return QConv2dFunction.apply(x, self.weight, self, True)


class QLinearFunction(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mod):
        B = x.shape[0]
        qx = mod.quantize_input(x.contiguous())
        _quantize_filters(mod, mod.num_bits_weight)
        b32 = mod.master_view('bias')
        bq = qparams(minmax_rows(b32, 1), 1, 1)
        qb = quantize(b32, bq[0:1], bq[1:2], mod.num_bits_weight + mod.num_bits)
        y = ops.conv2d_fwd(qx.view(B, 1, 1, mod.in_features), mod.w_krsc, qb, mod.out_features, 1, 1, (1, 1), (0, 0),
                           out_f32=True)
        ctx.mod = mod
        ctx.save_for_backward(qx)
        return y.view(B, mod.out_features)

    @staticmethod
    def backward(ctx, dy):
        (qx,) = ctx.saved_tensors
        mod = ctx.mod
        B = qx.shape[0]
        dy = dy.contiguous()
        dyc = dy if dy.dtype == qx.dtype else ops.cast_from_f32(dy, qx.dtype)
        ops.conv2d_wgrad(qx.view(B, 1, 1, mod.in_features), dyc.view(B, 1, 1, mod.out_features),
                         mod.grad_view('weight'), mod.in_features, mod.out_features, 1, 1, (1, 1), (0, 0))
        ops.colsum(dy.view(B, mod.out_features), mod.grad_view('bias'))
        mod._notify_grad_ready()
        dx = None
        if ctx.needs_input_grad[0]:
            gq = quantize_grad(dy, mod.num_bits_grad)
            gqc = gq if gq.dtype == qx.dtype else ops.cast_from_f32(gq, qx.dtype)
            dx = ops.conv2d_dgrad(gqc.view(B, 1, 1, mod.out_features), mod.w_crsk, (B, 1, 1, mod.in_features),
                                  mod.out_features, 1, 1, (1, 1), (0, 0)).view(B, mod.in_features)
        return dx, None, None, None


class QLinear(cnn.Linear):
    """quantize.py:223-253."""

    def __init__(self, in_features, out_features, bias=True, num_bits=8, num_bits_weight=8, num_bits_grad=8,
                 biprecision=True):
        super().__init__(in_features, out_features, bias)
        if not bias or not biprecision or num_bits_grad is None:
            raise NotImplementedError('QLinear: biased bi-precision configuration only (what ResNet uses)')
        self.num_bits, self.num_bits_weight, self.num_bits_grad = num_bits, num_bits_weight or num_bits, num_bits_grad
        self.biprecision = biprecision
        self.quantize_input = QuantMeasure(num_bits)

    def forward(self, x):
        self._require_prepared()
        if self.out_features % _lib.chunk_elems(self.compute_dtype) != 0:
            raise NotImplementedError('QLinear: out_features must be a multiple of the 16-byte chunk')
        return QLinearFunction.apply(x.reshape(x.shape[0], self.in_features), self.weight, self.bias, self)


def _scale_fix(values_per_chunk):
    """quantize.py:295-296."""
    return (0.5 * 0.35) * (1 + (math.pi * math.log(4)) ** 0.5) / ((2 * math.log(values_per_chunk)) ** 0.5)


class RangeBNFunction(Function):
    @staticmethod
    def forward(ctx, y, weight, bias, mod, relu, train_graph=False):
        # train_graph: the caller saw grad mode ON (torch.is_grad_enabled() is always False in here: autograd runs a
        # Function's forward with grad mode off - round 4's forward-side min / max fusions tested it in here and never ran)
        N, H, W, C = y.shape
        M = N * H * W
        L = _L()
        code = dtype_code(y.dtype)
        fused = FUSE_QUANT and y.dtype in (torch.bfloat16, torch.float32)
        if M % mod.num_chunks != 0 or M // mod.num_chunks < 2:
            raise _lib.ConvNetHipError('RangeBN: %d values per channel do not split into %d chunks of >= 2'
                                       % (M, mod.num_chunks))
        fix = _scale_fix(M // mod.num_chunks)
        stats = torch.empty(2 * C, dtype=torch.float32, device=y.device)
        arg = torch.empty(C * 2 * mod.num_chunks, dtype=torch.int32, device=y.device)
        ws = ops.workspace(L.cn_rangebn_workspace(M, C, mod.num_chunks), y.device, 'quant')
        if fused:     # the statistics pass snaps the raw convolution output on load and stores the snapped tensor
            y = y.contiguous()
            qp = mod.quantize_input.qparams_tensor(y)
            store8 = _store8_ok(y, mod.quantize_input.num_bits) and _mm_ok(y)     # (_mm_ok: the fused backward pass will run)
            qy = torch.empty(y.shape, dtype=torch.uint8, device=y.device) if store8 else torch.empty_like(y)
            z = torch.empty_like(y)
            # z feeds an activation quantiser when a ReLU follows (bn1 / bn2 of a block): its per-sample extremes come along
            want_mm = relu and _mm_ok(z) and train_graph
            zmm = torch.empty(N * 2, dtype=torch.float32, device=y.device) if want_mm else None
            if store8:
                ops.PROFILER.run('quant: rangebn quantise(8-bit levels)+stats, finalize, apply%s' % ('+minmax' if want_mm else ''),
                                 4 if want_mm else 3, 0.0, y.numel() * (3 * y.element_size() + 2),
                                 lambda: check(L.cn_rangebn_fwd_q8(ptr(y), ptr(qp), mod.quantize_input.num_bits, ptr(qy),
                                                                   ptr(z), ptr(weight), ptr(bias), ptr(mod.running_mean),
                                                                   ptr(mod.running_var), mod.momentum, mod.eps, mod.num_chunks,
                                                                   fix, ptr(stats), ptr(arg), M, C, int(relu), code,
                                                                   N if want_mm else 0, ptr(zmm), ptr(ws), ws.numel() * 4,
                                                                   stream_of(y)), 'cn_rangebn_fwd_q8'), y.device)
            else:
                ops.PROFILER.run('quant: rangebn quantise+stats, finalize, apply%s' % ('+minmax' if want_mm else ''),
                                 4 if want_mm else 3, 0.0, 4 * y.numel() * y.element_size(),
                                 lambda: check(L.cn_rangebn_fwd_q(ptr(y), ptr(qp), mod.quantize_input.num_bits, ptr(qy), None,
                                                                  ptr(z), ptr(weight), ptr(bias), ptr(mod.running_mean),
                                                                  ptr(mod.running_var), mod.momentum, mod.eps, mod.num_chunks,
                                                                  fix, ptr(stats), ptr(arg), M, C, int(relu), code, N if want_mm else 0,
                                                                  ptr(zmm), ptr(ws), ws.numel() * 4, stream_of(y)),
                                               'cn_rangebn_fwd_q'), y.device)
            if want_mm:
                _stash_minmax(z, N, zmm)
            ctx.x_qp = qp if store8 else None
        else:
            qy = mod.quantize_input(y.contiguous())
            z = torch.empty_like(qy)
            ops.PROFILER.run('quant: rangebn_stats+finalize+apply', 3, 0.0, 3 * qy.numel() * qy.element_size(),
                             lambda: check(L.cn_rangebn_fwd(ptr(qy), None, ptr(z), ptr(weight), ptr(bias),
                                                            ptr(mod.running_mean), ptr(mod.running_var), mod.momentum,
                                                            mod.eps, mod.num_chunks, fix, ptr(stats), ptr(arg), M, C,
                                                            int(relu), 1, code, ptr(ws), ws.numel() * 4, stream_of(y)),
                                           'cn_rangebn_fwd'), y.device)
            ctx.x_qp = None
        ctx.mod, ctx.relu, ctx.fix = mod, relu, fix
        ctx.cdtype = y.dtype
        ctx.save_for_backward(qy, weight, stats, arg, *((z,) if relu else ()))
        return z

    @staticmethod
    def backward(ctx, dz):
        saved = ctx.saved_tensors
        qy, weight, stats, arg = saved[:4]
        mod = ctx.mod
        N, H, W, C = qy.shape
        M = N * H * W
        L = _L()
        dz = dz.contiguous()
        fuse = _mm_ok(dz)
        if ctx.relu:     # the ReLU that follows the output-gradient quantiser in the reference
            if fuse:     # ... masked and measured in one pass
                g0, mm, gqp = eltwise_mm(2, dz, saved[4], N, want_qp=True)
                _stash_minmax(g0, N, mm, qp_extreme=gqp)
            else:
                g0 = torch.empty_like(dz)
                check(L.cn_eltwise(2, ptr(g0), ptr(dz), ptr(saved[4]), dz.numel(), dtype_code(dz.dtype), stream_of(dz)),
                      'cn_eltwise')
        else:
            g0 = dz
        x_qp = ctx.x_qp                       # not None: qy holds 8-bit levels of that grid (STORE8)
        cdt = ctx.cdtype
        g8 = fuse and _store8_ok(g0, mod.num_bits_grad)
        if g8:
            gq, g_qp = quantize_grad(g0, mod.num_bits_grad, levels=True)
        else:
            gq, g_qp = quantize_grad(g0, mod.num_bits_grad), None
        dx = torch.empty(qy.shape, dtype=cdt, device=qy.device)
        ws = ops.workspace(L.cn_rangebn_workspace(M, C, mod.num_chunks), qy.device, 'quant')
        if x_qp is not None and not fuse:
            raise _lib.ConvNetHipError('RangeBN: the 8-bit level storage of the forward pass needs the fused backward pass')
        if fuse and (g8 or x_qp is not None):
            dxmm = torch.empty(N * 2, dtype=torch.float32, device=qy.device)
            dxqp = torch.empty(2, dtype=torch.float32, device=qy.device) if (N <= 256 and QP_FROM_PRODUCER) else None
            esz = dx.element_size()
            ops.PROFILER.run('quant: rangebn_bwd reduce+finalize+apply(route, minmax) on 8-bit levels', 4, 0.0,
                             qy.numel() * (esz + (2 if g8 else 2 * esz) + (1 if x_qp is not None else esz)),
                             lambda: check(L.cn_rangebn_bwd_q8(ptr(gq), ptr(g_qp), mod.num_bits_grad, ptr(qy), ptr(x_qp),
                                                               mod.quantize_input.num_bits, ptr(weight), ptr(stats), ptr(arg),
                                                               ptr(dx), ptr(mod.grad_view('weight')), ptr(mod.grad_view('bias')),
                                                               M, C, mod.num_chunks, ctx.fix, dtype_code(cdt), N, ptr(dxmm),
                                                               ptr(dxqp), ptr(ws), ws.numel() * 4, stream_of(qy)),
                                           'cn_rangebn_bwd_q8'),
                             qy.device)
            _stash_minmax(dx, N, dxmm, qp_extreme=dxqp)     # for the gradient quantiser of the convolution in front
        elif fuse:
            dxmm = torch.empty(N * 2, dtype=torch.float32, device=qy.device)
            ops.PROFILER.run('quant: rangebn_bwd reduce+finalize+apply(route, minmax)', 4, 0.0, 4 * qy.numel() * qy.element_size(),
                             lambda: check(L.cn_rangebn_bwd_mm(ptr(gq), ptr(qy), ptr(weight), ptr(stats), ptr(arg), ptr(dx),
                                                               ptr(mod.grad_view('weight')), ptr(mod.grad_view('bias')), M, C,
                                                               mod.num_chunks, ctx.fix, dtype_code(qy.dtype), N, ptr(dxmm),
                                                               ptr(ws), ws.numel() * 4, stream_of(qy)), 'cn_rangebn_bwd_mm'),
                             qy.device)
            _stash_minmax(dx, N, dxmm)     # for the gradient quantiser of the convolution in front (QConv2d.backward)
        else:
            ops.PROFILER.run('quant: rangebn_bwd reduce+finalize+apply+route', 4, 0.0, 4 * qy.numel() * qy.element_size(),
                             lambda: check(L.cn_rangebn_bwd(ptr(gq), ptr(qy), ptr(weight), ptr(stats), ptr(arg), ptr(dx),
                                                            ptr(mod.grad_view('weight')), ptr(mod.grad_view('bias')), M, C,
                                                            mod.num_chunks, ctx.fix, dtype_code(qy.dtype), ptr(ws),
                                                            ws.numel() * 4, stream_of(qy)), 'cn_rangebn_bwd'), qy.device)
        mod._notify_grad_ready()
        return dx, None, None, None, None, None


class RangeBN(cnn.BatchNorm2d):
    """quantize.py:256-330.  Subclasses the HIP BatchNorm2d so that the reference's `isinstance(m,
    nn.BatchNorm2d)` tests (init_model, weight-decay filter -- nn.BatchNorm2d *is* RangeBN there after the
    rebinding of models/resnet.py:391) keep their meaning; none of the parent's state is created."""

    def __init__(self, num_features, dim=1, momentum=0.1, affine=True, num_chunks=16, eps=1e-5, num_bits=8,
                 num_bits_grad=8):
        cnn._ArenaModule.__init__(self)
        if not affine or num_bits_grad is None:
            raise NotImplementedError('RangeBN: affine with a quantised output gradient (the reference default)')
        self.num_features = num_features
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.zeros(num_features))
        self.momentum, self.dim, self.eps, self.num_chunks = momentum, dim, eps, num_chunks
        self.bias = tnn.Parameter(torch.empty(num_features))
        self.weight = tnn.Parameter(torch.empty(num_features))
        self.num_bits, self.num_bits_grad = num_bits, num_bits_grad
        self.quantize_input = QuantMeasure(num_bits, inplace=True, shape_measure=(1, 1, 1, 1), flatten_dims=(1, -1))
        self.affine, self.track_running_stats, self.sync_group = True, True, None
        self.weight.data.uniform_()   # reset_params (quantize.py:277-281): same RNG consumption as the reference
        self.bias.data.zero_()

    def forward(self, y, residual=None, relu=False):
        self._require_prepared()
        if residual is not None:
            raise NotImplementedError('RangeBN: the residual junction is a separate add + ReLU (reference order)')
        if self.training:
            fn = RangeBNFunction.apply
            if torch.is_grad_enabled():
                return fn(y, self.weight, self.bias, self, relu, True)
            with torch.no_grad():
                return fn(y, self.weight, self.bias, self, relu, False)
        N, H, W, C = y.shape
        L = _L()
        qy = self.quantize_input(y.contiguous())
        z = torch.empty_like(qy)
        stats = torch.empty(2 * C, dtype=torch.float32, device=y.device)
        check(L.cn_rangebn_fwd(ptr(qy), None, ptr(z), ptr(self.weight), ptr(self.bias), ptr(self.running_mean),
                               ptr(self.running_var), self.momentum, self.eps, self.num_chunks, 0.0, ptr(stats), None,
                               N * H * W, C, int(relu), 0, dtype_code(y.dtype), None, 0, stream_of(y)), 'cn_rangebn_fwd')
        return z

    def reset_running_stats(self):
        # Trainer.calibrate_bn (trainer.py:277-285) calls this on every `nn.BatchNorm2d`; the reference's RangeBN has
        # no such method (AttributeError there) and no cumulative-average mode: say so instead of mis-calibrating
        raise NotImplementedError('calibrate_bn is not defined for RangeBN (the reference raises AttributeError here)')

    def extra_repr(self):
        return '{}, eps={}, momentum={}, num_chunks={}'.format(self.num_features, self.eps, self.momentum,
                                                               self.num_chunks)


class AddReLUFunction(Function):
    """relu(a + b): the residual junction of the quantised blocks (models/resnet.py:162-163), kept apart from
    the last RangeBN so that the backward pass visits the operators in the reference's order."""

    @staticmethod
    def forward(ctx, a, b, train_graph=False, holder=None):
        ctx.holder = holder
        a, b = a.contiguous(), b.contiguous()
        if _mm_ok(a) and train_graph:     # (no stash outside training: nothing would consume it; see RangeBNFunction)
            # the block output is what the next block's conv1 / projection quantise: measured here
            z, mm = eltwise_mm(4, a, b, a.shape[0])
            _stash_minmax(z, a.shape[0], mm)
        else:
            z = torch.empty_like(a)
            check(_L().cn_eltwise(4, ptr(z), ptr(a), ptr(b), a.numel(), dtype_code(a.dtype), stream_of(a)), 'cn_eltwise')
        ctx.save_for_backward(z)
        return z

    @staticmethod
    def backward(ctx, dz):
        (z,) = ctx.saved_tensors
        dz = dz.contiguous()
        if _mm_ok(dz):
            # g reaches the gradient quantiser of bn3 (and of the projection's RangeBN when the block has one)
            g, mm, gqp = eltwise_mm(2, dz, z, dz.shape[0], want_qp=True)
            _stash_minmax(g, dz.shape[0], mm, uses=2, qp_extreme=gqp)
        else:
            g = torch.empty_like(dz)
            check(_L().cn_eltwise(2, ptr(g), ptr(dz), ptr(z), dz.numel(), dtype_code(dz.dtype), stream_of(dz)), 'cn_eltwise')
        if ctx.holder is not None and JUNCTION_ADD:
            # identity shortcut: g IS the shortcut branch's gradient at the block input; conv1's data gradient adds it
            ctx.holder.dres, ctx.holder.sub, ctx.holder.fused = g, 1, False
        return g, g, None, None


def add_relu(a, b, holder=None):
    """holder: the block's ResGradHolder when the shortcut is the identity (b is the block input itself)."""
    return AddReLUFunction.apply(a, b, torch.is_grad_enabled(), holder)
