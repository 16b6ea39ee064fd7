"""Lookahead wrapper with the reference's interface (models/optimiser/RAdam/lookahead.py:7-106, pullback_momentum
"none" -- the only mode the scripts use, pretrain_BreastPathQ.py:247) whose arithmetic runs in the engine.

``step()`` = one inner optimizer step with whatever gradients the engine currently holds (the reference calls it once
per EPOCH, as ``scheduler.step()``, with the last batch's stale gradients -- pretrain_BreastPathQ.py:293) followed,
every ``la_steps`` calls, by  p = alpha*p + (1-alpha)*cached ; cached = p  (lookahead.py:93-97).
"""
import torch

from .engine import get_engine


class Lookahead:
    def __init__(self, optimizer, la_steps=5, la_alpha=0.8, pullback_momentum="none"):
        if pullback_momentum.lower() != "none":
            raise NotImplementedError("the reference scripts use pullback_momentum='none'")
        self.optimizer = optimizer
        self._la_step = 0
        self.la_alpha = la_alpha
        self._total_la_steps = la_steps
        self.pullback_momentum = "none"
        self.state = {}
        for group in optimizer.param_groups:
            for p in group["params"]:
                self.state[p] = {"cached_params": p.detach().clone()}
        self._net = None

    def bind(self, model, classifier):
        """optional: name the (model, classifier) pair the inner optimizer updates.  Not needed at the reference's call site
        (``scheduler = Lookahead(optimizer, la_steps=5, la_alpha=0.5)``, pretrain_BreastPathQ.py:247): step() finds the engine
        binding that train() made for exactly this optimizer's parameters."""
        eng = get_engine(next(model.parameters()).device)
        self._net = eng.bind(model, classifier)
        return self

    def _find_net(self):
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        ids = {id(p) for p in params}
        eng = get_engine(params[0].device)
        for net in eng._bound.values():
            if net.still_valid() and {id(p) for p in net.params if p.requires_grad} == ids:
                return net
        raise RuntimeError("Lookahead.step(): no engine binding holds this optimizer's parameters yet -- the reference calls it "
                           "after train() (pretrain_BreastPathQ.py:286-293); or name the modules with .bind(model, classifier)")

    @property
    def param_groups(self):
        return self.optimizer.param_groups

    def zero_grad(self):
        pass                                   # gradients live in the engine and are rebuilt every backward

    def get_la_step(self):
        return self._la_step

    def state_dict(self):
        return self.optimizer.state_dict()

    def load_state_dict(self, state_dict):
        self.optimizer.load_state_dict(state_dict)

    def step(self, closure=None):
        if self._net is None or not self._net.still_valid():
            self._net = self._find_net()
        self._net.optimizer_step(self.optimizer)
        self._la_step += 1
        if self._la_step >= self._total_la_steps:
            self._la_step = 0
            cached = [self.state[p]["cached_params"] if p in self.state else None for p in self._net.params]
            self._net.lookahead(cached, self.la_alpha)
        return None
