"""Trainer: the step engine, with the class / method surface of /root/reference trainer.py:54-285
(`Trainer(model, criterion, optimizer, device_ids, device, dtype, distributed, local_rank, ...)`,
`.train / .validate / .forward / .calibrate_bn / ._step`, result keys step, data, loss, prec1,
prec5, [grad], error1, error5; attributes `epoch`, `training_steps` poked by main.py:298,300).

What is different by design (MI355X-first):
* the model runs on the HIP kernels through a flat fp32 parameter arena (engine.prepare) instead
  of ATen + DistributedDataParallel; data parallelism is a bucketed RCCL all-reduce of the flat
  gradient arena overlapped with backward, with the 1/world average and the reference's
  `p.grad.div_(loss_scale)` loop (trainer.py:165-169) folded into the fused SGD kernel;
* no per-step device->host sync: the reference calls float(loss) / float(prec) every iteration
  (trainer.py:153,225-229); here loss / prec@1 / prec@5 / grad-norm meters accumulate on the device
  and are read back only when a report line is due (values at report points are identical).

Out of the hot path (raise if requested): mixup / cutmix, duplicates + adapt_grad_norm,
tensorwatch streams, nn.DataParallel.
"""
import ctypes
import logging
import os
import time
import weakref

import torch
import torch.distributed as dist

from . import _lib, engine, flags, ops
from ._lib import check, ptr, stream_of
from .cross_entropy import CrossEntropyLoss
from .meters import AverageMeter, accuracy


class DevicePrefetcher(object):
    """Overlaps the host->device copy of batch i+1 with the compute of batch i (the reference moves
    each batch synchronously inside the step, trainer.py:116-117).  Batches already on the device
    pass through untouched.  Copies run on a side stream; the compute stream waits on an event and
    `record_stream` keeps the caching allocator from recycling a batch still in use."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, device
        self.stream = torch.cuda.Stream(device) if device.type == 'cuda' else None
        # loaders built with `device_normalize` (data.DataRegime) hand over uint8 NHWC crops: ToTensor + Normalize happen
        # here, behind the copy (a quarter of the bytes over PCIe), and produce the very fp32 NCHW batch the host transforms
        # would have (ops.u8_nhwc_to_nchw: a per-channel table of the reference's own fp32 arithmetic)
        norm = getattr(loader, 'device_normalize', None)
        self._lut = ops.normalize_lut(norm['mean'], norm['std']).to(device) if norm else None

    def _finish(self, inputs):
        if isinstance(inputs, dict):      # device_resize loader: uint8 crops of any size + PIL's resampling tables
            inputs = ops.resize_crops(inputs)
        if inputs.dtype == torch.uint8:
            if self._lut is None:
                raise ValueError('uint8 batches need a loader built with device_normalize (mean / std travel with it)')
            return ops.u8_nhwc_to_nchw(inputs, self._lut)
        return inputs

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        inputs, target = batch
        if isinstance(inputs, dict):
            if self.stream is None:
                return self._finish(inputs), target, None
            with torch.cuda.stream(self.stream):
                dev_in = {k: (v if k == 'size' else (v if v.is_pinned() else v.pin_memory()).to(self.device, non_blocking=True))
                          for k, v in inputs.items()}
                x = self._finish(dev_in)
                t = target.to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            return x, t, ev
        if self.stream is None or (inputs.is_cuda and target.is_cuda):
            return self._finish(inputs), target, None
        with torch.cuda.stream(self.stream):
            if not inputs.is_pinned():
                inputs = inputs.pin_memory()
            if inputs.dtype == torch.uint8:
                x = self._finish(inputs.to(self.device, non_blocking=True))
            else:
                x = inputs.to(self.device, dtype=torch.float32, non_blocking=True)
            t = target.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return x, t, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            x, t, ev = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            if ev is not None:
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(ev)
                x.record_stream(cur)
                t.record_stream(cur)
            yield x, t


class EagerWatch(object):
    """Re-examines an 'eager launches' verdict of graph = auto while it is in force.  The verdict is taken once, on the
    fourth step; a host that slows down later (other tenants on its cores, a box whose first minutes are slow) turns the
    same step host-bound - measured on this pool: 7.4k and 11.1k img/s eager against 13.9k replayed in the same minute.
    One timing event per step is recorded behind the step (cn_step_timer_*: no system-scope fence, unlike torch's timing
    events); completed pairs give the step PERIOD on the device timeline without a synchronisation.  That period is
    max(loader wait + host launches, device time): the time the loop spent WAITING FOR THE LOADER before the step
    (measured on the host by Trainer.forward, handed to `step`) is subtracted, so a loader-bound run - the normal case
    for main.py with PIL workers - is not mistaken for a launch-bound one (a replayed graph cannot make the loader
    faster).  The verdict is withdrawn when the MEDIAN of the last `window` corrected periods exceeds
    `factor` x the step time it was based on (a replayed single-chain graph costs ~1.07 x that); the median, because
    single long periods are normal - the pause between two train() calls, a validation pass.

    The library's ring of timing marks is shared by every watch of the process: a mark carries (watch id, sequence
    number) as its tag, a polled period belongs to the watch whose id BOTH of its marks carry (two consecutive steps of
    that configuration) and is routed there whichever watch happened to poll it; periods between marks of different
    watches are dropped.  Nothing is reset globally."""

    _next_id = [1]
    _live = weakref.WeakValueDictionary()     # watch id -> watch (periods polled by another watch are routed here)

    def __init__(self, ref_ms, window=9, factor=1.2):
        self.ref_ms, self.window, self.factor = float(ref_ms), window, factor
        self.periods = []
        self.id = EagerWatch._next_id[0]
        EagerWatch._next_id[0] += 1
        EagerWatch._live[self.id] = self
        self._seq = 0
        self._waits = {}          # sequence number of a mark -> loader wait (ms) in front of the step it closes
        self._fire = False

    def add_period(self, ms, wait_ms=0.0):
        """Pure bookkeeping (unit-tested on the CPU): True when the verdict should be withdrawn."""
        # (the loader wait is host time: when the host runs ahead of a backlogged device it overlaps device work, and
        # subtracting it in full would under-report the step - never below the time the verdict was based on)
        ms, wait_ms = float(ms), float(wait_ms)
        self.periods.append(max(ms - wait_ms, min(ms, self.ref_ms)))
        if len(self.periods) > self.window:
            self.periods.pop(0)
        return len(self.periods) == self.window and self.recent_ms() > self.factor * self.ref_ms

    def recent_ms(self):
        p = sorted(self.periods)
        return p[len(p) // 2] if p else 0.0

    def _deliver(self, ms, seq):
        wait = self._waits.pop(seq, 0.0)
        for k in [k for k in self._waits if k < seq]:      # marks whose period never arrived (dropped by the ring)
            del self._waits[k]
        if ms >= 0.0:
            self._fire = self.add_period(ms, wait) or self._fire

    def step(self, stream, wait_ms=0.0):
        """One mark behind the step just queued on `stream` (`wait_ms`: host time the loop waited for the loader in front
        of this step); folds in every period that has completed since (no wait).  True: withdraw the verdict."""
        import ctypes
        L = _lib.load()
        self._seq += 1
        self._waits[self._seq] = float(wait_ms)
        if len(self._waits) > 64:        # periods that never arrive (every pair foreign: alternating configurations) must
            for k in [k for k in self._waits if k <= self._seq - 64]:      # not accumulate: the ring holds 64 marks
                del self._waits[k]
        check(L.cn_step_timer_mark(stream.cuda_stream, (self.id << 32) | self._seq), 'cn_step_timer_mark')
        ms, ta, tb = ctypes.c_float(0.0), ctypes.c_longlong(0), ctypes.c_longlong(0)
        while L.cn_step_timer_poll(ctypes.byref(ms), ctypes.byref(ta), ctypes.byref(tb)) == 1:
            self.route(ms.value, ta.value, tb.value)
        fire, self._fire = self._fire, False
        return fire

    @classmethod
    def route(cls, ms, tag_prev, tag_cur):
        """A polled period goes to the watch both of whose marks it lies between (unit-tested on the CPU)."""
        wid = tag_cur >> 32
        if wid != (tag_prev >> 32) or (tag_cur & 0xffffffff) != (tag_prev & 0xffffffff) + 1:
            return None           # the two marks close steps of different configurations (or a mark was dropped)
        w = cls._live.get(wid)
        if w is not None:
            w._deliver(ms, tag_cur & 0xffffffff)
        return w


class PlanRefused(RuntimeError):
    """The recorded step holds something a launch plan cannot re-issue (cn_plan_import_graph says what)."""


class LaunchPlan(object):
    """Handle of one recording of csrc/plan.hip (created = recording started)."""

    def __init__(self, L, main_stream):
        self._L = L
        self.handle = ctypes.c_void_p()
        check(L.cn_plan_begin(ctypes.byref(self.handle), ctypes.c_void_p(main_stream)), 'cn_plan_begin')

    def end(self):
        check(self._L.cn_plan_end(self.handle), 'cn_plan_end')

    def info(self):
        c = (ctypes.c_longlong * 8)()
        check(self._L.cn_plan_info(self.handle, c), 'cn_plan_info')
        return list(c)

    def describe(self):
        return self._L.cn_plan_describe(self.handle).decode()

    def destroy(self):
        if self.handle:
            self._L.cn_plan_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Trainer(object):

    def __init__(self, model, criterion, optimizer=None,
                 device_ids=[0], device='cuda', dtype=torch.float,
                 distributed=False, local_rank=-1, adapt_grad_norm=None,
                 mixup=None, cutmix=None, loss_scale=1., grad_clip=-1, print_freq=100,
                 bucket_mb=25.0, process_group=None):
        if mixup is not None or cutmix is not None or adapt_grad_norm is not None:
            raise NotImplementedError('mixup / cutmix / adapt_grad_norm are outside the MI355X hot path')
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise NotImplementedError('compute dtype %s: float32, bfloat16 and float16 (the reference\'s `half`: fp32 '
                                      'BatchNorm parameters / statistics and fp32 master weights, main.py:239-250) '
                                      'are built' % dtype)
        if dtype == torch.float16 and any(getattr(m, 'quantized_op', False) for m in model.modules()):
            raise NotImplementedError('resnet(quantize=True) runs in float32 or bfloat16 storage')
        self._model = model
        self.model = model
        self.criterion = criterion
        self.epoch = 0
        self.training_steps = 0
        self.optimizer = optimizer
        if not isinstance(device, (str, torch.device)):
            device = 'cuda'
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None and device_ids:
            self.device = torch.device('cuda', device_ids[0])
        self.dtype = dtype
        self.distributed = distributed
        self.local_rank = local_rank
        self.print_freq = print_freq
        self.grad_clip = grad_clip
        self.grad_scale = None
        self.loss_scale = loss_scale
        self.watcher = None
        self.arena = engine.prepare(model, self.device, dtype, bucket_mb=bucket_mb)
        self.reducer = None
        self._main_stream = None   # high-priority HIP stream of the step loop (created lazily on a GPU)
        # flag graph: 'auto' (default) captures only when the eager step is host-bound, 1 = always, 0 = never
        self._graph_mode = flags.graph_mode()
        self._use_graph = self._graph_mode != '0'
        self._data_wait_ms = 0.0         # host time Trainer.forward waited for the loader in front of the current step
        self._graph_dp = flags.on('graph_dp')   # capture RCCL buckets too (opt-in)
        # launch plans (csrc/plan.hip): the step recorded once, re-issued from one C call on BOTH streams.  Where a
        # plan can be built it is what runs after the warm-up steps, whatever the batch size: it costs the host ~1 ms
        # per step instead of 11-12 ms and keeps the eager step's two-stream schedule (a replayed HIP graph does not).
        self._plan = flags.on('plan')
        self._gstates = {}               # per (shapes, step options) key: {'seen': warm-up / timing bookkeeping, 'graph': capture}
        self._graph_eager_for = set()    # (shapes, chunking) for which auto mode settled on eager launches
        self._watch = {}                 # key -> EagerWatch: the eager verdict is re-examined while it is in force
        from . import nn as cnn
        # a captured step replays the SAME kernels: host-drawn Dropout masks (models/mnist.py) rule it out
        # modules whose capturability can change at run time (quant.QuantMeasure.no_graph follows the noise source) are
        # asked at every step (_graph_ok), the others once
        self._graph_dynamic = [m for m in model.modules() if isinstance(getattr(type(m), 'no_graph', None), property)]
        self._graph_model_ok = not any((isinstance(m, cnn.Dropout) and m.p > 0)
                                       or (not isinstance(getattr(type(m), 'no_graph', None), property)
                                           and getattr(m, 'no_graph', False)) for m in model.modules())
        self.world_size = 1
        if distributed:
            if not dist.is_initialized():
                raise RuntimeError('Trainer(distributed=True) needs an initialised process group')
            self.reducer = engine.BucketReducer(self.arena, process_group)
            self.world_size = self.reducer.world
            self.reducer.broadcast_parameters(0)   # DDP construction broadcast (trainer.py:80-82)
            self._broadcast_buffers()
        # device-side meters: [loss*B, prec1*B, prec5*B, B, gradnorm*w, w]
        self._meters = torch.zeros(8, dtype=torch.float32, device=self.device)
        self._norm_out = torch.zeros(2, dtype=torch.float32, device=self.device)
        self._norm_ws = torch.zeros(_lib.load().cn_grad_norm_workspace() // 4, dtype=torch.float32,
                                    device=self.device)

    # ------------------------------------------------------------------------------------
    def _broadcast_buffers(self):
        if self.reducer is None:
            return
        for buf in self._model.buffers():
            self.reducer.broadcast_(buf, 0)

    def _to_device(self, inputs, target):
        target = target.to(self.device, non_blocking=True)
        inputs = inputs.to(self.device, dtype=torch.float32, non_blocking=True)
        return inputs, target

    def _step(self, inputs_batch, target_batch, training=False, average_output=False, chunk_batch=1):
        if average_output:
            raise NotImplementedError('average_output (duplicates) is outside the MI355X hot path')
        if training:
            # host side of trainer.py:111-112 (the regime moves lr / momentum; nothing here touches the device
            # except a tiny H2D copy when the schedule changes)
            self.optimizer.update(self.epoch, self.training_steps)
            # configurations for which auto mode already settled on eager launches go straight to the eager body (the
            # bookkeeping of _graph_step costs a nearly host-bound step 1 %); the verdict is keyed like the graphs are
            key = self._graph_key(inputs_batch, target_batch, chunk_batch)
            if key not in self._graph_eager_for:
                if self._graph_ok(inputs_batch, target_batch):
                    return self._graph_step(inputs_batch, target_batch, chunk_batch)
            elif key in self._watch and not ops.PROFILER.enabled:
                res = self._body(inputs_batch, target_batch, training, chunk_batch)
                if self._watch[key].step(torch.cuda.current_stream(self.device), self._data_wait_ms):
                    self._eager_verdict_withdrawn(key)
                return res
        return self._body(inputs_batch, target_batch, training, chunk_batch)

    def _body(self, inputs_batch, target_batch, training, chunk_batch):
        """Device side of one step (trainer.py:106-177): everything here is a stream of kernel launches with
        no host synchronisation, which is what lets `_graph_step` capture it into one HIP graph."""
        outputs = []
        total_loss = None
        grad = None

        if training:
            self.optimizer.zero_grad()
            if self.reducer is not None:
                self.reducer.reset()

        in_chunks = inputs_batch.chunk(chunk_batch, dim=0)
        tg_chunks = target_batch.chunk(chunk_batch, dim=0)
        n_chunks = len(in_chunks)
        for i, (inputs, target) in enumerate(zip(in_chunks, tg_chunks)):
            inputs, target = self._to_device(inputs, target)
            if training:
                self.optimizer.pre_forward()

            output = self.model(inputs)
            loss = self.criterion(output, target)

            if chunk_batch > 1:
                loss = loss / chunk_batch
            if isinstance(output, (list, tuple)):
                output = output[0]
            outputs.append(output.detach())
            total_loss = loss.detach() if total_loss is None else total_loss + loss.detach()

            if training:
                if i == 0:
                    self.optimizer.pre_backward()
                if self.grad_scale is not None:
                    loss = loss * self.grad_scale
                if self.loss_scale is not None and self.loss_scale != 1:
                    loss = loss * self.loss_scale
                if self.reducer is not None:
                    self.reducer.enabled = (i == n_chunks - 1)   # reduce once, after accumulation
                loss.backward()

        if training:
            ops.SIDE.join(self.device)      # weight-gradient kernels queued on the side stream
            if self.reducer is not None:
                self.reducer.finish()
            gscale = 1.0 / float(self.world_size)
            if self.loss_scale is not None:
                gscale /= float(self.loss_scale)
            clip_coef = None
            if self.grad_clip > 0:
                a = self.arena
                check(_lib.load().cn_grad_norm_clip(ptr(a.grads), a.grads.numel(), gscale, float(self.grad_clip),
                                                    ptr(self._norm_out), ptr(self._meters[4:]),
                                                    float(inputs_batch.size(0)), ptr(self._norm_ws),
                                                    stream_of(a.grads)), 'cn_grad_norm_clip')
                grad = self._norm_out[0]
                clip_coef = self._norm_out[1:2]
            self.optimizer.grad_scale = gscale
            self.optimizer.clip_coef = clip_coef
            self.optimizer.step()
            self.training_steps += 1

        outputs = outputs[0] if len(outputs) == 1 else torch.cat(outputs, dim=0)
        return outputs, total_loss, grad

    # -- whole-step HIP graph ------------------------------------------------------------------
    # The eager step issues ~520 kernel launches through ctypes + the autograd tape (about 10 ms of host time
    # for ResNet-50, profiles/r01_bench_line_b8_host_overhead.json).  The launches never depend on host
    # values - meters, the clip coefficient and (lr, momentum) live in device memory, workspaces are
    # caller-owned - so after two eager warm-up steps of a given shape the whole device side of the step
    # (layout cast, forward, loss + meters, backward on both streams, bucket all-reduces, clip, SGD) is
    # captured once and replayed: one graph launch per step.  Flag graph = 0 keeps the eager path.
    def _graph_ok(self, inputs, target):
        if not self._use_graph or self.device.type != 'cuda' or ops.PROFILER.enabled:
            return False
        if not (inputs.is_cuda and target.is_cuda and inputs.dtype == torch.float32):
            return False
        if not isinstance(self.criterion, CrossEntropyLoss) or not self._graph_model_ok:
            return False
        if self._graph_dynamic and any(getattr(m, 'no_graph', False) for m in self._graph_dynamic):
            return False     # (a host-side noise source was installed after construction: quant.set_noise_source)
        if self.reducer is not None and (self.reducer.comm is None or not (self._graph_dp or self._plan)):
            return False     # (a plan re-issues the RCCL buckets live; a HIP graph captures them: opt-in)
        return True

    def _graph_key(self, inputs, target, chunk_batch):
        """Everything a captured step bakes in besides the tensors' contents."""
        return (tuple(inputs.shape), tuple(target.shape), chunk_batch, float(self.grad_clip), self.loss_scale,
                self.grad_scale, self.optimizer.runs_signature(), getattr(self.criterion, 'smooth_eps', None), self.world_size)

    def _graph_step(self, inputs, target, chunk_batch):
        """State is kept PER KEY (batch shapes + step options): the odd-shaped last batch of an epoch warms up /
        runs eagerly under its own key and leaves the captured full-batch graph alone (a single slot
        used to be discarded and rebuilt every epoch).  NOTE: the (output, loss, grad-norm) a replay returns are
        the graph's STATIC buffers - the next replay of the same graph overwrites them; Trainer.forward folds them
        into the device meters right away, other callers must clone what they keep."""
        opt = self.optimizer
        opt.push_hyper()
        key = self._graph_key(inputs, target, chunk_batch)
        gs = self._gstates.pop(key, None)
        if gs is None:
            if len(self._gstates) >= 4:      # bounded, least recently USED out (a hit re-inserts its key at the end, so
                self._gstates.pop(next(iter(self._gstates)))   # the hot full-batch graph is never the one evicted)
            gs = {'seen': {'n': 0}, 'graph': None}
        self._gstates[key] = gs
        seen = gs['seen']
        st = gs['graph']
        eager_key = key      # an 'eager' verdict holds for exactly the configuration it was measured on
        if st is None:
            # eager warm-up steps before the capture: 4 when the host-vs-device timing of the last one decides between eager
            # launches and a HIP graph (the first steps still grow workspaces and the allocator's pools); 2 when a launch plan
            # follows whatever the timing says - the step then runs as a plan from the THIRD step on, so that a benchmark's
            # warm-up of >= 3 steps leaves only replays in its timed region
            warm = 2 if (self._graph_mode != 'auto' or self._plan) else 4
            if seen['n'] < warm:       # eager warm-up (lazy workspace growth, allocator warm)
                seen['n'] += 1
                if seen['n'] < warm or self._graph_mode != 'auto':
                    return self._body(inputs, target, True, chunk_batch)
                # last warm-up step, mode 'auto' (the fourth: the first ones still grow workspaces and the allocator's
                # pools): is the eager step bound by the host (launch time ~ device
                # time) or by the device?  A replayed graph removes the host cost but measured 5 % SLOWER than the
                # eager two-stream schedule when the device is the limit (ResNet-50 b=256: 22.1 vs 21.0 ms), and
                # 1.55x faster when the host is (b=8: 5.5 vs 8.6 ms) - profiles/README.md.
                torch.cuda.synchronize(self.device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                res = self._body(inputs, target, True, chunk_batch)
                e1.record()
                host_ms = (time.perf_counter() - t0) * 1e3
                e1.synchronize()
                dev_ms = e0.elapsed_time(e1)
                seen['host_bound'] = host_ms > 0.75 * dev_ms
                seen['plan'] = self._plan
                seen['use'] = seen['plan'] or seen['host_bound']
                seen['eager_ms'] = dev_ms
                if not seen['use']:
                    self._graph_eager_for.add(eager_key)
                    self._watch[eager_key] = EagerWatch(dev_ms)
                logging.debug('step: host %.2f ms, device %.2f ms -> %s', host_ms, dev_ms,
                              'try a HIP graph' if seen['use'] else 'eager launches')
                return res
            if not seen.get('use', True):
                return self._body(inputs, target, True, chunk_batch)
            try:
                try:
                    st = gs['graph'] = self._capture(inputs, target, chunk_batch, key,
                                                     plan=seen.get('plan', self._plan))
                except PlanRefused as e:
                    # the step holds something a plan cannot re-issue: the HIP graph (when the host is the limit and
                    # nothing but this library's kernels would be captured) or eager launches serve it
                    logging.warning('launch plan refused (%s): %s', e, 'HIP graph / eager launches instead')
                    seen['plan'] = False
                    if self._graph_mode == 'auto':
                        seen['use'] = seen.get('host_bound', False)
                    if self.reducer is not None and not self._graph_dp:
                        seen['use'] = False
                    if not seen['use']:
                        self._graph_eager_for.add(eager_key)
                        if 'eager_ms' in seen:
                            self._watch[eager_key] = EagerWatch(seen['eager_ms'])
                        return self._body(inputs, target, True, chunk_batch)
                    st = gs['graph'] = self._capture(inputs, target, chunk_batch, key, plan=False)
            except RuntimeError as e:      # (torch.cuda.OutOfMemoryError is one)
                # A capture needs its own pool for the step's tensors, next to the blocks the eager steps keep cached.  When
                # the capture was the WATCH's idea (a verdict withdrawn after many eager steps) and it does not fit, the
                # configuration simply stays eager; a capture the configuration started with fails as it always did.
                # The same holds for a launch-plan recording in graph = auto: the plan is an optimisation of a step that already
                # ran eagerly twice - whatever made its capture fail (memory, a runtime that refuses the capture with the
                # communicators of a multi-rank job alive, ...) must not take the job down.  graph = 1 asked for a capture
                # and still gets the error.
                if not (seen.get('withdrawn') or (seen.get('plan') and self._graph_mode == 'auto')):
                    raise
                logging.warning('capture of the training step failed (%s): staying with eager launches',
                                str(e).split('\n')[0][:200])
                seen['plan'] = False
                if 'eager_ms' in seen and not seen.get('withdrawn'):
                    self._watch[eager_key] = EagerWatch(seen['eager_ms'])
                gs['graph'] = None
                seen['use'] = False
                self._graph_eager_for.add(eager_key)
                torch.cuda.empty_cache()
                return self._body(inputs, target, True, chunk_batch)
        self._feed(st, inputs, target)
        if self._graph_mode == 'auto' and 'graph_ms' not in seen and 'eager_ms' in seen:
            # the prediction above is checked once: a nearly host-bound eager step can still beat the replay
            # (ResNet-50 b=128: 12.2 ms eager vs 13.3 ms replayed), so the second replay is timed and the graph
            # dropped if it is not faster than the eager step it was meant to replace
            n = seen.get('replays', 0)
            seen['replays'] = n + 1
            if n == 1:
                torch.cuda.synchronize(self.device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._replay(st)
                e1.record()
                e1.synchronize()
                seen['graph_ms'] = e0.elapsed_time(e1)
                # (a plan is the eager schedule minus the host - the same launches on the same streams: it cannot be slower by
                # construction, and this ONE sample also sees whatever else is on the device at that moment, e.g. the loader's
                # host-to-device copy of the next batch (a plan was dropped that way in a --host-inputs run).  It is given up
                # only when it measures grossly slower; what it buys - independence from the host's load - no quiet-box
                # comparison can show)
                if seen['graph_ms'] > (1.25 if st.get('plan') else 0.98) * seen['eager_ms']:
                    seen['use'] = False
                    self._graph_eager_for.add(eager_key)
                    logging.debug('replayed step %.2f ms vs eager %.2f ms -> eager launches from now on',
                                  seen['graph_ms'], seen['eager_ms'])
                out, loss, grad = st['out'], st['loss'], st['grad']
                self.arena.bump_version()
                self.training_steps += 1
                if not seen['use']:
                    out = out.clone()
                    loss = loss.clone()
                    grad = grad.clone() if grad is not None else None
                    gs['graph'] = None
                return out, loss, grad
        self._replay(st)
        self.arena.bump_version()      # what optimizer.step() does on the host: master weights moved
        self.training_steps += 1
        return st['out'], st['loss'], st['grad']

    def _eager_verdict_withdrawn(self, key):
        """The eager step of this configuration has been running slower than a replayed graph would for a window of
        steps (EagerWatch): the host has become the limit after the verdict was taken (a loaded host, a slow box: the
        eager step needs ~11 ms of host time per ResNet-50 step, a replay none).  The next step captures; the usual
        check of the second replay against the eager time - now the recent one, loader wait excluded - still applies."""
        w = self._watch.pop(key)
        self._graph_eager_for.discard(key)
        gs = self._gstates.get(key)
        if gs is None:
            gs = self._gstates[key] = {'seen': {'n': 4}, 'graph': None}
        gs['seen'].update(use=True, withdrawn=True, eager_ms=w.recent_ms(), n=max(gs['seen'].get('n', 0), 4))
        gs['seen'].pop('graph_ms', None)
        gs['seen'].pop('replays', None)
        logging.info('eager step %.2f ms against %.2f ms when it was chosen: trying a HIP graph', w.recent_ms(), w.ref_ms)

    def _feed(self, st, inputs, target):
        """The batch of this step -> where the captured step reads it.  A HIP graph reads static buffers (two copies per
        step); a launch plan's own launches are re-pointed at the caller's input tensor instead (cn_plan_set_input: the
        154 MB read + write of a ResNet-50 b=256 batch, 0.06 ms, stays out of the step, as in the eager step)."""
        if st.get('x_sites', 0) > 0 and inputs.is_contiguous():
            check(_lib.load().cn_plan_set_input(st['plan'].handle, 0, ctypes.c_void_p(inputs.data_ptr())), 'cn_plan_set_input')
            st['x_live'] = inputs          # (held until the next step: the launches that read it are queued, not done)
        else:
            if st.get('x_sites', 0) > 0:
                check(_lib.load().cn_plan_set_input(st['plan'].handle, 0, ctypes.c_void_p(st['x'].data_ptr())), 'cn_plan_set_input')
            st['x'].copy_(inputs, non_blocking=True)
        # the targets likewise (slot 1).  The copy they replace is 2 KB, but a device-to-device memcpy between the optimizer
        # kernel of one step and the first kernel of the next left the queue idle for 155 + 20 us per step in the traced plan
        # (profiles/r06_trace_gaps_plan.txt): 1 % of a ResNet-50 step
        if st.get('t_sites', 0) > 0 and target.is_contiguous() and target.dtype == st['t'].dtype:
            check(_lib.load().cn_plan_set_input(st['plan'].handle, 1, ctypes.c_void_p(target.data_ptr())), 'cn_plan_set_input')
            st['t_live'] = target
        else:
            if st.get('t_sites', 0) > 0:
                check(_lib.load().cn_plan_set_input(st['plan'].handle, 1, ctypes.c_void_p(st['t'].data_ptr())), 'cn_plan_set_input')
            st['t'].copy_(target, non_blocking=True)

    def _replay(self, st):
        if st.get('plan') is not None:
            if torch.cuda.current_stream(self.device).cuda_stream != st['stream']:
                raise RuntimeError('a launch plan is tied to the stream it was recorded on')
            check(_lib.load().cn_plan_replay(st['plan'].handle), 'cn_plan_replay')
        else:
            st['graph'].replay()

    def _capture(self, inputs, target, chunk_batch, key, plan=False):
        """plan = False: the step as one HIP graph (single chain: ops.SIDE folds the side stream in).  plan = True: the
        same capture with the two-stream schedule kept and the library logging its launches, hand-offs and communicator
        calls (cn_plan_begin / _end); the captured graph is never launched - it is the cross-check (every logged launch
        must be a node of it), the source of whatever torch itself put on the stream, and the owner of the memory pool
        the step's tensors live in."""
        cur = torch.cuda.current_stream(self.device)
        x, t = torch.empty_like(inputs), torch.empty_like(target)
        x.copy_(inputs)
        t.copy_(target)
        steps_before = self.training_steps
        L = _lib.load()
        g = torch.cuda.CUDAGraph(keep_graph=True) if plan else torch.cuda.CUDAGraph()
        rec = LaunchPlan(L, cur.cuda_stream) if plan else None
        ops.SIDE.capturing = not plan
        try:
            with torch.cuda.graph(g, stream=cur, capture_error_mode='thread_local'):
                out, loss, grad = self._body(x, t, True, chunk_batch)
        except BaseException:
            if rec is not None:
                rec.destroy()
            # an aborted step body leaves per-step mailboxes half filled: empty them, the next (eager) step starts clean
            for m in self._model.modules():
                h = getattr(m, '_holder', None)
                if h is not None:
                    h.dres, h.fused, h.sub = None, False, 1
                m.__dict__.pop('_lazy_z', None)
                m.__dict__.pop('_lazy_a', None)
                if hasattr(m, '_lazy_dy'):
                    m._lazy_dy = None
            ops.SIDE._held.clear()
            ops.SIDE._held_bytes, ops.SIDE.used, ops.SIDE._mark = 0, False, None
            try:
                from . import quant as _q
                _q._MM_STASH.clear()
            except Exception:
                pass
            raise
        finally:
            ops.SIDE.capturing = False
            self.training_steps = steps_before     # capture executes nothing: the replay is the step
        if rec is not None:
            rec.end()
            n = L.cn_plan_import_graph(rec.handle, ctypes.c_void_p(g.raw_cuda_graph()))
            if n < 0:
                why = _lib.last_error()
                rec.destroy()
                raise PlanRefused(why)
            info = rec.info()
            logging.debug('recorded the training step as a launch plan (%s): %d launches + %d imported on %d streams, '
                          '%d hand-offs, %d communicator calls', key[0], info[1], info[2], info[6], info[4], info[5])
            # the input batch is read by this library's layout cast only (the model's first operator; _graph_ok admits
            # fp32 device inputs, so torch converts nothing): its launches can be re-pointed at the caller's tensor.  The
            # static copy is poisoned so that a reader the search missed shows up as NaN, not as a stale batch.
            sites = L.cn_plan_bind_input(rec.handle, 0, ctypes.c_void_p(x.data_ptr()), x.numel() * x.element_size())
            if sites > 0:
                x.fill_(float('nan'))
            t_sites = L.cn_plan_bind_input(rec.handle, 1, ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size())
            return {'key': key, 'graph': g, 'plan': rec, 'stream': cur.cuda_stream, 'x': x, 't': t, 'out': out,
                    'loss': loss, 'grad': grad, 'x_sites': max(sites, 0), 't_sites': max(t_sites, 0)}
        logging.debug('captured the training step as one HIP graph (%s)', (key[0],))
        return {'key': key, 'graph': g, 'x': x, 't': t, 'out': out, 'loss': loss, 'grad': grad}

    # ------------------------------------------------------------------------------------
    def forward(self, data_loader, num_steps=None, training=False, average_output=False, chunk_batch=1):
        """The per-batch loop of trainer.py:179-263.  On a GPU the whole loop runs on a high-priority HIP
        stream: the forward / dgrad / BatchNorm chain is the critical path, the weight-gradient kernels on
        the (normal-priority) side stream only have to fill its gaps (+0.6 % on ResNet-50; flag main_stream_prio =
        off keeps everything on the caller's stream)."""
        dev = torch.device(self.device)
        if dev.type != 'cuda' or flags.text('main_stream_prio') == 'off':
            return self._forward(data_loader, num_steps, training, average_output, chunk_batch)
        if self._main_stream is None:
            self._main_stream = torch.cuda.Stream(dev, priority=int(flags.text('main_stream_prio')))
        caller = torch.cuda.current_stream(dev)
        self._main_stream.wait_stream(caller)          # everything the caller queued (inputs, weights) is visible
        try:
            with torch.cuda.stream(self._main_stream):
                return self._forward(data_loader, num_steps, training, average_output, chunk_batch)
        finally:
            caller.wait_stream(self._main_stream)      # ... and what the loop produced is visible to the caller

    def _forward(self, data_loader, num_steps=None, training=False, average_output=False, chunk_batch=1):
        meters = {name: AverageMeter() for name in ['step', 'data', 'loss', 'prec1', 'prec5']}
        if training and self.grad_clip > 0:
            meters['grad'] = AverageMeter()

        device_meters = isinstance(self.criterion, CrossEntropyLoss)
        if device_meters:
            ops.fill_f32_(self._meters, 0.0)
            self.criterion.meters = self._meters
        host_acc = None

        def sync_meters(last_loss):
            """One D2H of the device accumulators -> reference AverageMeter state."""
            if device_meters:
                m = self._meters.tolist()
                last = self.criterion.last_step.tolist() if self.criterion.last_step is not None else [0, 0, 0]
                cnt = m[3] if m[3] > 0 else 0
                lv = float(last_loss) if last_loss is not None else last[0]
                meters['loss'].set(lv, m[0], cnt)
                meters['prec1'].set(last[1], m[1], cnt)
                meters['prec5'].set(last[2], m[2], cnt)
                if 'grad' in meters:
                    meters['grad'].set(float(self._norm_out[0]), m[4], m[5])

        def meter_results(meters):
            results = {name: meter.avg for name, meter in meters.items()}
            results['error1'] = 100. - results['prec1']
            results['error5'] = 100. - results['prec5']
            return results

        end = time.time()
        n_batches = len(data_loader)
        try:
            for i, (inputs, target) in enumerate(DevicePrefetcher(data_loader, self.device)):
                if inputs.dim() > 4:
                    raise NotImplementedError('duplicates (B x D x C x H x W inputs) are outside the hot path')
                meters['data'].update(time.time() - end)
                self._data_wait_ms = meters['data'].val * 1e3     # (EagerWatch: not the step's own time)

                output, loss, grad = self._step(inputs, target, training=training,
                                                average_output=average_output, chunk_batch=chunk_batch)

                if not device_meters:   # foreign criterion: reference behaviour, host meters
                    tgt = target.to(self.device)
                    prec1, prec5 = accuracy(output, tgt, topk=(1, 5))
                    meters['loss'].update(float(loss), inputs.size(0))
                    meters['prec1'].update(float(prec1), inputs.size(0))
                    meters['prec5'].update(float(prec5), inputs.size(0))
                    if grad is not None:
                        meters['grad'].update(float(grad), inputs.size(0))

                report = (i % self.print_freq == 0) or (i == n_batches - 1)
                if report:
                    sync_meters(loss)   # the only device->host sync of the loop
                meters['step'].update(time.time() - end)
                end = time.time()

                if report:
                    line = str('{phase} - Epoch: [{0}][{1}/{2}]\t'
                               'Time {meters[step].val:.3f} ({meters[step].avg:.3f})\t'
                               'Data {meters[data].val:.3f} ({meters[data].avg:.3f})\t'
                               'Loss {meters[loss].val:.4f} ({meters[loss].avg:.4f})\t'
                               'Prec@1 {meters[prec1].val:.3f} ({meters[prec1].avg:.3f})\t'
                               'Prec@5 {meters[prec5].val:.3f} ({meters[prec5].avg:.3f})\t'
                               .format(self.epoch, i, n_batches,
                                       phase='TRAINING' if training else 'EVALUATING', meters=meters))
                    if 'grad' in meters.keys():
                        line += 'Grad {meters[grad].val:.3f} ({meters[grad].avg:.3f})'.format(meters=meters)
                    logging.info(line)

                if num_steps is not None and i >= num_steps:
                    break
            sync_meters(None)
        finally:
            if device_meters:
                self.criterion.meters = None
        return meter_results(meters)

    def train(self, data_loader, average_output=False, chunk_batch=1):
        self.model.train()
        return self.forward(data_loader, training=True, average_output=average_output, chunk_batch=chunk_batch)

    def validate(self, data_loader, average_output=False):
        self.model.eval()
        self._broadcast_buffers()   # DDP broadcast_buffers: every rank evaluates rank 0's statistics
        with torch.no_grad():
            return self.forward(data_loader, average_output=average_output, training=False)

    def calibrate_bn(self, data_loader, num_steps=None):
        from . import nn as cnn
        for m in self.model.modules():
            if isinstance(m, cnn.BatchNorm2d):
                m.momentum = None
                m.track_running_stats = True
                m.reset_running_stats()
        self.model.train()
        with torch.no_grad():
            return self.forward(data_loader, num_steps=num_steps, training=False)

    # tensorwatch hooks of the reference (trainer.py:287-337) are observability extras; keep the
    # call surface as no-ops so main.py-style drivers keep working.
    def set_watcher(self, filename, port=0):
        return False

    def observe(self, **kwargs):
        return False

    def stream_meters(self, meters_dict, prefix=None):
        return False

    def write_stream(self, name, values):
        return False
