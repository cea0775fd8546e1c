"""Device-resident model state: flat fp32 parameter / gradient arenas and data-parallel buckets.

MI355X-first re-design of what the reference gets from ``model.to(device, dtype)`` +
``nn.parallel.DistributedDataParallel`` (/root/reference main.py:236, trainer.py:79-82):

* every parameter is a view into ONE flat fp32 arena (master weights) and its ``.grad`` a view
  into a parallel flat gradient arena, so ``optimizer.step`` / ``zero_grad`` / grad-norm are single
  kernels and a gradient bucket is just a contiguous slice handed to RCCL;
* conv / linear weights keep the reference's ``[O, I, kh, kw]`` *shape* (state_dict compatible) but
  live in KRSC memory order (channels_last strides) - the order the kernels produce and consume;
* the arena is ordered by backward completion (classifier first, stem last) so buckets fill in
  launch order and their all-reduce overlaps the rest of backward;
* gradients are averaged by folding 1/world_size into the fused SGD kernel, BN statistics stay
  per-rank (DistributedDataParallel semantics without --sync-bn).
"""
import torch
import torch.distributed as dist

from . import ops

_ALIGN = 64  # floats; keeps every parameter segment 256-byte aligned


def _round_up(n, a):
    return (n + a - 1) // a * a


class _Slot(object):
    __slots__ = ('name', 'param', 'offset', 'numel', 'bucket', 'module', 'is_filter')


class ParamArena(object):
    def __init__(self, model, device, bucket_mb=25.0):
        from . import nn as cnn
        self.device = torch.device(device)
        filters, others = [], []
        seen = set()
        for mod_name, mod in model.named_modules():
            for pname, p in mod.named_parameters(recurse=False):
                if id(p) in seen:
                    continue
                seen.add(id(p))
                full = (mod_name + '.' if mod_name else '') + pname
                is_filter = isinstance(mod, (cnn.Conv2d, cnn.Linear)) and pname == 'weight'
                (filters if is_filter else others).append((full, mod, pname, p, is_filter))
        ordered = list(reversed(filters)) + list(reversed(others))
        self.slots = []
        off = 0
        for full, mod, pname, p, is_filter in ordered:
            s = _Slot()
            s.name, s.param, s.module, s.is_filter = full, p, mod, is_filter
            s.offset, s.numel = off, p.numel()
            off += _round_up(p.numel(), _ALIGN)
            self.slots.append(s)
        self.n_filter = sum(_round_up(s.numel, _ALIGN) for s in self.slots if s.is_filter)
        self.total = off
        self.params = torch.zeros(max(off, _ALIGN), dtype=torch.float32, device=self.device)
        self.grads = torch.zeros_like(self.params)
        for s in self.slots:
            p = s.param
            seg = self.params[s.offset:s.offset + s.numel]
            gseg = self.grads[s.offset:s.offset + s.numel]
            src = p.detach().to(device=self.device, dtype=torch.float32)
            if s.is_filter and p.dim() == 4:
                O, I, R, S_ = p.shape
                view = seg.view(O, R, S_, I).permute(0, 3, 1, 2)
                gview = gseg.view(O, R, S_, I).permute(0, 3, 1, 2)
            else:
                view = seg.view(p.shape)
                gview = gseg.view(p.shape)
            view.copy_(src)
            p.data = view
            p.grad = gview
            if hasattr(s.module, '_bind_arena'):
                s.module._bind_arena(self, s)
        # buckets: contiguous slot ranges of <= bucket_mb (reference DDP default 25 MB)
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []
        cur_start, cur_len, members = 0, 0, []
        for s in self.slots:
            seglen = _round_up(s.numel, _ALIGN)
            if members and cur_len + seglen > cap:
                self.buckets.append((cur_start, cur_len, members))
                cur_start, cur_len, members = s.offset, 0, []
            members.append(s)
            cur_len += seglen
        if members:
            self.buckets.append((cur_start, cur_len, members))
        for bi, (_, _, members) in enumerate(self.buckets):
            for s in members:
                s.bucket = bi
        # modules that own several slots (BN weight+bias) notify once per backward
        self._module_slots = {}
        for s in self.slots:
            self._module_slots.setdefault(id(s.module), []).append(s)
        self.version = 0          # bumped whenever master weights change (optimizer step / load)
        self.reducer = None
        self.wbuf = None          # compute-dtype filter copies of every conv / linear (one buffer)
        self._wdesc = None
        self._wdesc_reg = None
        self._wtiles = None
        self._wbytes = 0
        self._wtotal = 0
        self._wversion = -1

    # -- compute-dtype filter copies ------------------------------------------------------
    def build_weight_plan(self, dtype):
        """Lay out the KRSC (+ CRSK for layers that need dgrad) compute-dtype copies of all filters in
        one buffer and build the device descriptor table of cn_weight_prep_multi."""
        from . import nn as cnn
        rows, off, start = [], 0, 0
        mods = []
        for s in self.slots:
            mod = s.module
            if not s.is_filter or not isinstance(mod, (cnn.Conv2d, cnn.Linear)):
                continue
            if mod.out_channels % (4 if dtype == torch.float32 else 8) != 0:
                continue   # ragged dense head: computed from the fp32 master directly (cn_small_linear)
            taps = mod.kernel_size[0] * mod.kernel_size[1]
            co, creal = mod.out_channels, mod.in_channels
            ch = 4 if dtype == torch.float32 else 8
            cpad = _round_up(creal, ch) if isinstance(mod, cnn.Conv2d) else creal
            n = co * taps * cpad
            krsc_off = off
            off += _round_up(n, _ALIGN)
            want_crsk = isinstance(mod, cnn.Linear) or (getattr(mod, 'needs_dgrad', True) and cpad == creal)
            crsk_off = -1
            if want_crsk:
                crsk_off = off
                off += _round_up(n, _ALIGN)
            rows.append([s.offset, start, krsc_off, crsk_off, co, taps, creal, cpad])
            start += n
            mods.append((mod, krsc_off, crsk_off, n))
        self.wbuf = torch.zeros(max(off, _ALIGN), dtype=dtype, device=self.device)
        # config 5 (quant.QConv2d / QLinear: per-output-channel 8-bit filters, quantize.py:201-203): when EVERY filter
        # of the model is a quantised operator's, one launch snaps all of them into a shadow of the arena and the
        # compute-dtype copies below are made from the shadow (quant._quantize_filters then has nothing left to do)
        self._qrows, self._qshadow = None, None
        if mods and all(hasattr(m, 'num_bits_weight') for m, _, _, _ in mods) \
                and len(mods) == sum(1 for s in self.slots if s.is_filter):
            tab = []
            for r, (m, _, _, _) in zip(rows, mods):
                row_len = r[5] * r[6]
                qmax = (1 << int(m.num_bits_weight)) - 1
                tab.extend([r[0] + k * row_len, row_len, qmax] for k in range(r[4]))
            self._qrows = torch.tensor(tab, dtype=torch.int64, device=self.device)
            self._qshadow = torch.zeros(_round_up(self.n_filter, _ALIGN), dtype=torch.float32, device=self.device)
        # regular filters (no channel padding) go through the tiled, coalescing kernel; the few padded
        # ones (the 3 -> 8 channel stem) through the per-element kernel
        regular = [r for r in rows if r[6] == r[7]]
        ragged = [list(r) for r in rows if r[6] != r[7]]
        start = 0
        for r in ragged:
            r[1] = start
            start += r[4] * r[5] * r[7]
        self._wtotal = start
        self._wdesc = torch.tensor(ragged, dtype=torch.int64, device=self.device) if ragged else None
        self._wdesc_reg = torch.tensor(regular, dtype=torch.int64, device=self.device) if regular else None
        tiles = []
        for di, r in enumerate(regular):
            co, J = r[4], r[5] * r[6]
            for co0 in range(0, co, 64):
                for j0 in range(0, J, 64):
                    tiles.append((di, co0, j0, 0))
        self._wtiles = torch.tensor(tiles, dtype=torch.int32, device=self.device) if tiles else None
        self._wbytes = sum(r[4] * r[5] * (r[6] * 4 + r[7] * self.wbuf.element_size() * (2 if r[3] >= 0 else 1))
                           for r in rows)
        self._wversion = -1
        for mod, krsc_off, crsk_off, n in mods:
            mod.w_krsc = self.wbuf[krsc_off:krsc_off + n]
            mod.w_crsk = self.wbuf[crsk_off:crsk_off + n] if crsk_off >= 0 else None

    def prepare_weights(self):
        if self._wversion == self.version or (self._wdesc is None and self._wdesc_reg is None):
            return
        from . import _lib
        L = _lib.load()
        code, st = _lib.dtype_code(self.wbuf.dtype), _lib.stream_of(self.params)
        src = self.params if self._qrows is None else self._qshadow

        def run():
            if self._qrows is not None:
                _lib.check(L.cn_quantize_rows_multi(self.params.data_ptr(), self._qshadow.data_ptr(),
                                                    self._qrows.data_ptr(), self._qrows.shape[0], st),
                           'cn_quantize_rows_multi')
            if self._wdesc_reg is not None:
                _lib.check(L.cn_weight_prep_tiled(src.data_ptr(), self.wbuf.data_ptr(),
                                                  self._wdesc_reg.data_ptr(), self._wtiles.data_ptr(),
                                                  self._wtiles.shape[0], code, st), 'cn_weight_prep_tiled')
            if self._wdesc is not None:
                _lib.check(L.cn_weight_prep_multi(src.data_ptr(), self.wbuf.data_ptr(),
                                                  self._wdesc.data_ptr(), self._wdesc.shape[0], self._wtotal, code,
                                                  st), 'cn_weight_prep_multi')
        ops.PROFILER.run('weight_prep', 2, 0.0, float(self._wbytes), run, self.device)
        self._wversion = self.version

    # -- gradient lifecycle ---------------------------------------------------------------
    def zero_grad(self):
        ops.fill_f32_(self.grads, 0.0)

    def bump_version(self):
        self.version += 1

    def module_ready(self, mod):
        if self.reducer is not None:
            for s in self._module_slots.get(id(mod), ()):
                self.reducer.slot_ready(s)


class BucketReducer(object):
    """Bucketed gradient all-reduce overlapped with backward.  Transport: the direct-RCCL communicator
    (comm.RcclCommunicator over cn_comm_*: its own high-priority HIP stream, ordered behind the producer
    streams by events) when the ranks own HIP devices; torch.distributed collectives (gloo) in the CPU
    tests.  Sums only; the 1/world_size average is folded into the optimizer kernel."""

    def __init__(self, arena, process_group=None):
        from . import comm as _comm
        self.arena = arena
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.comm = _comm.create_default(arena.device, process_group) if _comm.wanted(arena.device, process_group) \
            else None
        self.enabled = True
        self._pending = None
        self._works = []
        self.reset()
        arena.reducer = self

    def describe(self):
        return self.comm.describe() if self.comm is not None else \
            'torch.distributed (%s), %d ranks' % (dist.get_backend(self.pg), self.world)

    def reset(self):
        self._pending = [len(members) for (_, _, members) in self.arena.buckets]
        self._works = []

    def broadcast_(self, t, src=0):
        if self.comm is not None and t.is_cuda and t.is_contiguous():
            self.comm.broadcast_(t, src)
        else:
            dist.broadcast(t, src=src, group=self.pg)

    def broadcast_parameters(self, src=0):
        self.broadcast_(self.arena.params, src)
        self.arena.bump_version()

    def _reduce(self, b):
        start, length, _ = self.arena.buckets[b]
        view = self.arena.grads[start:start + length]
        dev = self.arena.device
        side_used = dev.type == 'cuda' and ops.SIDE.enabled and ops.SIDE.used
        if self.comm is not None:
            # The bucket's weight gradients are queued on the wgrad side stream, its BN / bias gradients on
            # the main stream: the communicator's stream waits for both (events), neither is stalled.
            streams = [torch.cuda.current_stream(dev).cuda_stream]
            if side_used:
                streams.append(ops.SIDE.gather(dev).cuda_stream)
            self.comm.allreduce_bucket(view, streams)
        elif side_used:
            # torch.distributed: issue the collective from the side stream after making IT wait for the
            # main stream, so the main stream (dgrad / BN chain) is never stalled at a bucket boundary
            side = ops.SIDE.gather(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        else:
            self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def slot_ready(self, slot):
        if not self.enabled:
            return
        b = slot.bucket
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self._reduce(b)

    def finish(self):
        """Flush buckets that never filled (unused parameters) and wait for all reductions."""
        ops.SIDE.join(self.arena.device)
        if self.enabled:
            for b, left in enumerate(self._pending):
                if left > 0:
                    self._reduce(b)
            if self.comm is not None:
                self.comm.join(torch.cuda.current_stream(self.arena.device).cuda_stream)
            for w in self._works:
                w.wait()
        self.reset()


def prepare(model, device, dtype=torch.float32, bucket_mb=25.0):
    """Move `model` onto `device`, build its flat arenas and set the compute dtype.
    Idempotent; returns the ParamArena (also stored as model._cn_arena)."""
    arena = getattr(model, '_cn_arena', None)
    if arena is not None and arena.device == torch.device(device) and getattr(model, '_cn_dtype', None) == dtype:
        return arena
    device = torch.device(device)
    # buffers (BN running stats) move with a plain .to(); parameters are re-homed into the arena
    for mod in model.modules():
        for bname, buf in list(mod._buffers.items()):
            if buf is not None:
                mod._buffers[bname] = buf.to(device)
    arena = ParamArena(model, device, bucket_mb=bucket_mb)
    for mod in model.modules():
        if hasattr(mod, '_set_compute_dtype'):
            mod._set_compute_dtype(dtype)
    arena.build_weight_plan(dtype)
    # masters change behind our back when a checkpoint is loaded: refresh the compute copies
    model.register_load_state_dict_post_hook(lambda module, incompatible: arena.bump_version())
    model._cn_arena = arena
    model._cn_dtype = dtype
    return arena
