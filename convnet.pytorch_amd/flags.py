"""The engine's switches, in ONE table.

Every value below is the product configuration.  Two ways to change one:

* tests flip the module attribute that mirrors a flag (`ops.LAZY_DY = False`, `quant.INT8_FORWARD = True`, ...) to compare a
  fused path with the path it replaces - every fusion listed here has such a test;
* measurements set `CONVNET_AMD_FLAGS="name=value,name=value"` before the package is imported (tools/gpu_ab.sh) for an
  interleaved whole-step A/B.  Unknown names are an error, so a stale script cannot silently measure the default.

Kernel-variant knobs that live inside the library (tile shapes, workgroup counts) are a separate, smaller table:
`cn_set_option` / `CONVNET_AMD_OPTIONS` (csrc/runtime.hip); they never change results.

Switches that were measured neutral or negative in rounds 1-3 and removed in round 4 (code and knob): more than one
weight-gradient side stream, side-stream priority and CU masks, the minimum sizes of the statistics / BatchNorm-backward
epilogue fusions, the fused reductions at the inner BatchNorms, the BatchNorm apply folded into the tiled convolution's
operand load, the 256 x 256 weight-gradient tile, the halo data gradient's BatchNorm-backward epilogue
(profiles/README.md has the numbers)."""
import os

# 256 MB Infinity Cache (MI355X_MICROARCH.md).  A junction-sized tensor above ~0.6 of it is streamed from HBM by every
# pass that touches it, whatever ran before: that is where leaving a BatchNorm pass to its consumer ("lazy" forms) pays for
# the consumer's slower operand path; smaller tensors are partly cache hits and keep the plain kernels.
INFINITY_CACHE_BYTES = 256 * 2 ** 20
LAZY_MIN_BYTES = 0.6 * INFINITY_CACHE_BYTES

_DEFAULTS = {
    # ---- fusions (bit-identical to the passes they replace unless said otherwise; tests/test_ops.py, test_trajectory.py)
    'fuse_bn_stats': 1,        # BatchNorm statistics partials in the producing convolution's epilogue
    'fuse_bn_bwd': 1,          # junction BatchNorm-backward sums in the block-input dgrad's epilogue
    'fuse_stem_pool': 1,       # stem bn1 -> relu -> maxpool as one pooling pass over the pre-BN tensor
    'stem_xmax': 1,            # the stem's BatchNorm-backward sums over the pooled map
    'subsampled_shortcut_grad': 1,   # stride-2 projection's input gradient kept on its coarse grid
    'dual_bn': 1,              # projection shortcut's BatchNorm applied inside the junction's apply pass
    'lazy_dy': 1,              # junction BatchNorm-backward apply left to conv3's / the projection's dgrad + wgrad
    'lazy_min_mb': LAZY_MIN_BYTES / 2 ** 20,   # lazy dy / lazy z for junction tensors of at least this size
    'lazy_z': 1,               # junction apply left to the next block's conv1
    'lazy_a': '1',             # inner BatchNorm apply left to its streaming / halo consumer ('1x1': 1x1 consumers only)
    'jpair': 1,                # dgrad + wgrad of a lazy-dy 64 -> 256 convolution in one pass (fp32 summation order differs)
    'jdgrad': 1,               # junction data gradient as a persistent streaming kernel
    'centered_stats': '1',     # statistics centred on the running mean: 0 never, 1 fp32 models, all every dtype
    # ---- kernel families (0 = the tiled implicit-GEMM kernel serves the layer)
    'stem_pairs': 1, 'stem_halo': 1, 'conv3x3_halo': 1, 'conv1x1_stream': 1,
    # ---- schedule
    'wgrad_stream': 1,         # weight gradients on a side stream beside the backward chain
    'side_hold': 1,            # side-stream operands held until the step's join instead of Tensor.record_stream (allocator events)
    'dgrad_first': 1,          # a convolution's backward launches its data gradient before the side-stream hand-off of the weight gradient
    'marks': 1,                # the side stream waits for the producing kernel's completion mark, not a queued event
    'main_stream_prio': '-1',  # the step loop's own high-priority stream ('off': the default stream)
    'graph': 'auto',           # whole-step HIP graph: auto (when the host is the limit) / 0 / 1
    'graph_dp': 0,             # capture the RCCL bucket all-reduces too (opt-in until run on >= 2 devices)
    'plan': 1,                 # launch plan (csrc/plan.hip): the step re-issued from one C call on both streams, RCCL live
    # ---- config 5
    'quant_int8': 0,           # QConv2d forward on the int8 MFMA kernel
    'quant_fuse': 1,           # producer-side fusions of the quantised chain (quant.py: FUSE_QUANT)
    'quant_qp_from_producer': 1,   # gradient producers also emit the gradient quantiser's [zero_point, range]
    'quant_store8': 1,         # RangeBN's snapped input and its quantised output gradient kept as 8-bit levels (bit-identical)
    'quant_wgrad_side': 1,     # QConv2d weight gradients on the weight-gradient side stream
    'quant_junction_add': 1,   # block-input gradient sum in the later data gradient's epilogue (fp32 identical; 16-bit: one rounding)
}

_ENV = 'CONVNET_AMD_FLAGS'


def _parse():
    vals = dict(_DEFAULTS)
    for item in os.environ.get(_ENV, '').split(','):
        item = item.strip()
        if not item:
            continue
        name, _, v = item.partition('=')
        if name not in vals:
            raise ValueError('%s: unknown flag %r (known: %s)' % (_ENV, name, ', '.join(sorted(vals))))
        vals[name] = v
    return vals


_VALUES = _parse()


def get(name):
    """Raw value (the default's type when it was not overridden, else the override's string)."""
    return _VALUES[name]


def on(name):
    return str(_VALUES[name]) not in ('0', 'off', 'False', '')


def text(name):
    return str(_VALUES[name])


def graph_mode():
    """The `graph` flag as one of 'auto' / '0' / '1': every spelling `on()` counts as off means NEVER capture (it used to
    be compared as text, so graph=off meant 'always'); anything else is an error, not a silent A/B mistake."""
    v = str(_VALUES['graph'])
    if v == 'auto':
        return 'auto'
    if v in ('0', 'off', 'False', ''):
        return '0'
    if v in ('1', 'on', 'True'):
        return '1'
    raise ValueError("CONVNET_AMD_FLAGS graph=%r: expected auto, 0 / off / False or 1 / on / True" % v)

