"""Regime / OptimRegime: the optimizer object Trainer drives (duck-typed API at /root/reference
trainer.py:111-112,121,157,173,258 and main.py:243-253: zero_grad, update(epoch, steps),
pre_forward, pre_backward, step, get_lr, state_dict / load_state_dict).

The classes live in the reference's un-vendored utils submodule; they are re-stated here from the
call sites and from the regime dictionaries the models attach (models/resnet.py:250-256):
a regime is a list of dicts keyed by 'epoch' (or 'step') whose later entries override earlier ones,
optionally with `step_lambda` / `epoch_lambda` strings that evaluate to a dict of overrides.

The step itself is the fused flat-arena SGD+momentum kernel (cn_sgd_momentum): weight decay is the
reference's WeightDecay regulariser (g += wd * p for the filtered parameters, applied right before
the SGD update, i.e. after gradient clipping), momentum buffers are fp32, master weights are fp32
(the reference's use_float_copy for half precision is therefore always on).
"""
from copy import deepcopy

import torch

from . import _lib, engine, ops
from ._lib import check, ptr, stream_of


def eval_func(f, x):
    if isinstance(f, str):
        f = eval(f)  # regime lambdas are strings in the reference (models/resnet.py:70-72)
    return f(x)


class Regime(object):
    """Epoch/step keyed settings with cumulative override semantics."""

    def __init__(self, regime, defaults=None):
        self.regime = regime
        self.defaults = dict(defaults or {})
        self.reset()

    def reset(self):
        self.current_regime_phase = None
        self.setting = dict(self.defaults)

    def update(self, epoch=None, train_steps=None):
        if self.regime is None:
            return False
        epoch = -1 if epoch is None else epoch
        train_steps = -1 if train_steps is None else train_steps
        setting = deepcopy(self.setting)
        if self.current_regime_phase is None:
            for phase, phase_setting in enumerate(self.regime):
                start_epoch = phase_setting.get('epoch', 0)
                start_step = phase_setting.get('step', 0)
                if epoch >= start_epoch or train_steps >= start_step:
                    self.current_regime_phase = phase
                    break
                setting.update(phase_setting)
            if self.current_regime_phase is None:
                self.current_regime_phase = 0
        while len(self.regime) > self.current_regime_phase + 1:
            nxt = self.regime[self.current_regime_phase + 1]
            if epoch >= nxt.get('epoch', float('inf')) or train_steps >= nxt.get('step', float('inf')):
                setting.update(self.regime[self.current_regime_phase])
                self.current_regime_phase += 1
            else:
                break
        setting.update(self.regime[self.current_regime_phase])
        if 'lr_decay_rate' in setting and 'lr' in setting:
            decay_steps = setting.pop('lr_decay_steps', 100)
            if train_steps % decay_steps == 0:
                setting['lr'] *= setting.pop('lr_decay_rate') ** (train_steps / decay_steps)
        elif 'step_lambda' in setting:
            setting.update(eval_func(setting.pop('step_lambda'), train_steps))
        elif 'epoch_lambda' in setting:
            setting.update(eval_func(setting.pop('epoch_lambda'), epoch))
        if 'execute' in setting:
            setting.pop('execute')()
        if _comparable(setting) == _comparable(self.setting):
            return False
        self.setting = setting
        return True

    def __repr__(self):
        return 'Current: %s\n Regime:%s' % (self.setting, self.regime)


def _comparable(setting):
    return {k: (v if not callable(v) else id(v)) for k, v in setting.items() if k != 'regularizer'} | \
        {'regularizer': repr(setting.get('regularizer'))}


class OptimRegime(Regime):
    def __init__(self, model, regime, defaults=None, filter=None, use_float_copy=False, log=True):
        super().__init__(regime, defaults)
        if filter is not None:
            raise NotImplementedError('OptimRegime(filter=...) is outside the hot path')
        self.model = model
        self.use_float_copy = use_float_copy
        self.hyper = {'lr': 0.0, 'momentum': 0.0, 'weight_decay': 0.0, 'dampening': 0.0, 'nesterov': False}
        self.regularizer_cfg = []
        self.arena = None
        self.momentum_buf = None
        self._runs = None
        # set by Trainer each step
        self.grad_scale = 1.0     # 1/loss_scale (and 1/world_size for data parallel)
        self.clip_coef = None     # device scalar written by cn_grad_norm_clip, or None
        self.hyper_dev = None     # device copy of (lr, momentum): read by the SGD kernel, so a captured step
        self._hyper_pushed = None  # (HIP graph) follows the schedule; refreshed by push_hyper() when it changes

    # -- binding to the device arena ------------------------------------------------------
    def _bind(self):
        if self.arena is not None:
            return
        arena = getattr(self.model, '_cn_arena', None)
        if arena is None:
            raise _lib.ConvNetHipError('OptimRegime: call engine.prepare(model, device, dtype) before stepping')
        self.arena = arena
        self.momentum_buf = torch.zeros_like(arena.params)
        self._runs = None

    def _build_runs(self):
        """Contiguous arena ranges sharing one weight-decay value -> one SGD launch each."""
        wd_default = float(self.hyper.get('weight_decay', 0.0) or 0.0)
        per_slot = []
        for s in self.arena.slots:
            wd = wd_default
            for reg in self.regularizer_cfg:
                if reg.get('name') != 'WeightDecay':
                    raise NotImplementedError('regularizer %r is outside the hot path' % reg.get('name'))
                flt = reg.get('filter') or {}
                ok = True
                if 'parameter_name' in flt and not flt['parameter_name'](s.name):
                    ok = False
                if 'module' in flt and not flt['module'](s.module):
                    ok = False
                if ok:
                    wd += float(reg.get('value', 0.0))
            per_slot.append(wd)
        runs = []
        for s, wd in zip(self.arena.slots, per_slot):
            end = s.offset + engine._round_up(s.numel, engine._ALIGN)
            if runs and runs[-1][2] == wd and runs[-1][1] == s.offset:
                runs[-1][1] = end
            else:
                runs.append([s.offset, end, wd])
        self._runs = [(a, b, wd) for a, b, wd in runs]

    # -- regime handling ------------------------------------------------------------------
    def update(self, epoch=None, train_steps=None, metrics=None):
        if super().update(epoch, train_steps):
            self.adjust(self.setting)
            return True
        return False

    def adjust(self, setting):
        opt = setting.get('optimizer', 'SGD')
        if not isinstance(opt, str):
            opt = getattr(opt, '__name__', str(opt))
        if opt != 'SGD':
            raise NotImplementedError('optimizer %r: the MI355X hot path implements SGD+momentum' % opt)
        for key in self.hyper:
            if key in setting and setting[key] != self.hyper[key]:
                if key == 'weight_decay':
                    self._runs = None
                self.hyper[key] = setting[key]
        if self.hyper.get('nesterov') or self.hyper.get('dampening'):
            raise NotImplementedError('nesterov / dampening are not used by the reference regimes')
        if 'regularizer' in setting:
            reg = deepcopy_regularizer(setting['regularizer'])
            self.regularizer_cfg = reg
            self._runs = None

    # -- the API Trainer calls ------------------------------------------------------------
    def zero_grad(self):
        self._bind()
        self.arena.zero_grad()

    def pre_forward(self):
        pass

    def pre_backward(self):
        pass

    def step(self, *args, **kwargs):
        self._bind()
        if self._runs is None:
            self._build_runs()
        L = _lib.load()
        a = self.arena
        lr, mu = float(self.hyper['lr']), float(self.hyper['momentum'])
        self.push_hyper()
        for start, end, wd in self._runs:
            n = end - start
            ops.PROFILER.run('sgd_momentum', 1, 0.0, 20.0 * n,
                             lambda: check(L.cn_sgd_momentum(ptr(a.params[start:]), ptr(a.grads[start:]),
                                                             ptr(self.momentum_buf[start:]), n, lr, mu, float(wd),
                                                             float(self.grad_scale), ptr(self.clip_coef),
                                                             ptr(self.hyper_dev), stream_of(a.params)),
                                               'cn_sgd_momentum'),
                             a.device)
        a.bump_version()

    def push_hyper(self):
        """Device copy of (lr, momentum) for the SGD kernel; one tiny H2D copy whenever the schedule moves."""
        self._bind()
        cur = (float(self.hyper['lr']), float(self.hyper['momentum']))
        if self.hyper_dev is None:
            self.hyper_dev = torch.zeros(2, dtype=torch.float32, device=self.arena.device)
            self._hyper_pushed = None
        if cur != self._hyper_pushed:
            self.hyper_dev.copy_(torch.tensor(cur, dtype=torch.float32), non_blocking=False)
            self._hyper_pushed = cur

    def runs_signature(self):
        """What a captured step bakes in besides lr / momentum (weight-decay runs)."""
        self._bind()
        if self._runs is None:
            self._build_runs()
        return tuple(self._runs)

    def get_value(self, key):
        return [self.hyper.get(key)]

    def get_lr(self):
        return self.get_value('lr')

    # -- checkpoint -----------------------------------------------------------------------
    def _import_torch_sgd_state(self, st):
        return _torch_sgd_state_as_named_buffers(self.model, st)

    def state_dict(self):
        self._bind()
        bufs = {}
        for s in self.arena.slots:
            seg = self.momentum_buf[s.offset:s.offset + s.numel]
            p = s.param
            if s.is_filter and p.dim() == 4:
                O, I, R, S_ = p.shape
                bufs[s.name] = seg.view(O, R, S_, I).permute(0, 3, 1, 2).contiguous().cpu()
            else:
                bufs[s.name] = seg.view(p.shape).clone().cpu()
        return {'momentum_buffer': bufs, 'hyper': dict(self.hyper),
                'regime_phase': self.current_regime_phase}

    def load_state_dict(self, state):
        """Restores the momentum buffers (and the hyper-parameters as a starting point).  The regime
        position is deliberately NOT restored: `setting` is cumulative over all phases passed so far
        (e.g. the WeightDecay regulariser only appears in phase 0 of the ResNet regime,
        models/resnet.py:250-256), so the first `update(epoch, steps)` after a resume replays the
        regime from the start exactly as a fresh OptimRegime at that epoch would.
        Two formats are understood: this engine's own ({'momentum_buffer': {name: tensor}, ...}) and the
        reference's, i.e. what its OptimRegime.state_dict() hands to torch.save (utils.pytorch optim.py: the
        torch.optim.SGD state_dict, bare or under 'optimizer_state'): {'state': {i: {'momentum_buffer': t}},
        'param_groups': [{'params': [i...]}]} with i enumerating model.parameters() - so a checkpoint written by
        the reference resumes WITH its momentum.  Anything else is refused loudly (never a silent restart from
        zero momentum)."""
        self._bind()
        bufs = state.get('momentum_buffer') if isinstance(state, dict) else None
        if bufs is None and isinstance(state, dict):
            bufs = self._import_torch_sgd_state(state.get('optimizer_state', state))
        if bufs is None:
            raise _lib.ConvNetHipError(
                "OptimRegime.load_state_dict: neither this engine's format ('momentum_buffer') nor a "
                "torch.optim.SGD state_dict ('state' + 'param_groups'): refusing to resume from zero momentum "
                "(main.py --drop-optim-state skips the optimizer state on purpose)")
        missing = [s.name for s in self.arena.slots if s.name not in bufs]
        if missing:
            raise _lib.ConvNetHipError('OptimRegime.load_state_dict: momentum buffers missing for %s%s'
                                       % (missing[:4], ' ...' if len(missing) > 4 else ''))
        for s in self.arena.slots:
            src = bufs[s.name].to(self.arena.device, torch.float32)
            seg = self.momentum_buf[s.offset:s.offset + s.numel]
            p = s.param
            if s.is_filter and p.dim() == 4:
                O, I, R, S_ = p.shape
                seg.view(O, R, S_, I).permute(0, 3, 1, 2).copy_(src)
            else:
                seg.view(p.shape).copy_(src)
        self.hyper.update(state.get('hyper', {}))
        self.reset()          # current_regime_phase = None, setting = defaults: next update() replays the regime
        self._runs = None


def _torch_sgd_state_as_named_buffers(model, st):
    """torch.optim.SGD state_dict -> {parameter name: momentum buffer (reference OIHW shape)} or None."""
    if not (isinstance(st, dict) and 'state' in st and 'param_groups' in st):
        return None
    order = [i for grp in st['param_groups'] for i in grp['params']]
    names = [n for n, _ in model.named_parameters()]
    if len(order) != len(names):
        raise _lib.ConvNetHipError('OptimRegime.load_state_dict: optimizer state covers %d parameters, the model '
                                   'has %d' % (len(order), len(names)))
    bufs = {}
    for name, idx in zip(names, order):
        ent = st['state'].get(idx, st['state'].get(str(idx)))
        mb = ent.get('momentum_buffer') if isinstance(ent, dict) else None
        # a parameter that never received a gradient has no entry yet: zero momentum is exactly its state
        bufs[name] = mb if mb is not None else torch.zeros_like(dict(model.named_parameters())[name], device='cpu')
    return bufs


def deepcopy_regularizer(reg):
    if isinstance(reg, dict):
        reg = [reg]
    return [dict(r) for r in reg]
