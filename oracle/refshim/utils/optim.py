"""utils.optim.OptimRegime re-stated from its call sites (main.py:243-253; trainer.py:111-112,121,
157,173,258) on top of torch.optim.SGD: epoch/step keyed regime, WeightDecay regulariser with the
(parameter_name, module) filter of models/resnet.py:34-40 applied as g += wd*p right before the
SGD update."""
from copy import deepcopy

import torch


class OptimRegime(object):
    def __init__(self, model, regime, defaults=None, use_float_copy=False, **kw):
        self.model = model
        self.regime = regime
        self.setting = dict(defaults or {})
        self.phase = None
        self.parameters = list(model.parameters())
        self.optimizer = torch.optim.SGD(self.parameters, lr=0)
        self.decayed = []  # (param, value)

    def _merge(self, epoch, steps):
        setting = deepcopy_keep_callables(self.setting)
        if self.phase is None:
            self.phase = 0
        while self.phase + 1 < len(self.regime):
            nxt = self.regime[self.phase + 1]
            if epoch >= nxt.get('epoch', float('inf')) or steps >= nxt.get('step', float('inf')):
                setting.update(self.regime[self.phase])
                self.phase += 1
            else:
                break
        setting.update(self.regime[self.phase])
        if 'step_lambda' in setting:
            f = setting.pop('step_lambda')
            setting.update((eval(f) if isinstance(f, str) else f)(steps))
        elif 'epoch_lambda' in setting:
            f = setting.pop('epoch_lambda')
            setting.update((eval(f) if isinstance(f, str) else f)(epoch))
        return setting

    def update(self, epoch=None, train_steps=None, metrics=None):
        epoch = -1 if epoch is None else epoch
        train_steps = -1 if train_steps is None else train_steps
        setting = self._merge(epoch, train_steps)
        self.setting = setting
        for group in self.optimizer.param_groups:
            for key in group.keys():
                if key in setting and not callable(setting[key]) and key != 'params':
                    group[key] = setting[key]
        if 'regularizer' in setting:
            regs = setting['regularizer']
            regs = [regs] if isinstance(regs, dict) else regs
            self.decayed = []
            named = [(n, m, pn, p) for n, m in self.model.named_modules()
                     for pn, p in m.named_parameters(recurse=False)]
            for reg in regs:
                assert reg['name'] == 'WeightDecay'
                flt = reg.get('filter', {})
                for n, m, pn, p in named:
                    full = (n + '.' if n else '') + pn
                    if 'parameter_name' in flt and not flt['parameter_name'](full):
                        continue
                    if 'module' in flt and not flt['module'](m):
                        continue
                    self.decayed.append((p, reg['value']))
        return True

    def zero_grad(self):
        self.optimizer.zero_grad()

    def pre_forward(self):
        pass

    def pre_backward(self):
        pass

    def step(self):
        with torch.no_grad():
            for p, value in self.decayed:
                if p.grad is not None:
                    p.grad.add_(p, alpha=value)
        self.optimizer.step()

    def get_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, sd):
        self.optimizer.load_state_dict(sd)


def deepcopy_keep_callables(d):
    return {k: (v if callable(v) or k == 'regularizer' else deepcopy(v)) for k, v in d.items()}
