"""Train-step pieces around the model, mirroring the reference call sites:
``build_optimizer`` (pythia/utils/build_utils.py:54-83), ``get_optimizer_parameters``
(pythia/utils/general.py:124-137), ``lr_lambda_update`` (general.py:20-29), ``clip_gradients``
(general.py:32-41) and the step order of ``BaseTrainer._backward`` (pythia/trainers/base_trainer.py:262-272)."""
from bisect import bisect

import torch
from torch import nn


def lr_lambda_update(i_iter, cfg):
    tp = cfg["training_parameters"]
    if tp["use_warmup"] is True and i_iter <= tp["warmup_iterations"]:
        alpha = float(i_iter) / float(tp["warmup_iterations"])
        return tp["warmup_factor"] * (1.0 - alpha) + alpha
    idx = bisect(tp["lr_steps"], i_iter)
    return pow(tp["lr_ratio"], idx)


def clip_gradients(model, config):
    tp = config["training_parameters"]
    max_norm = tp["max_grad_l2_norm"]
    if max_norm is None:
        return None
    if tp["clip_norm_mode"] != "all":
        raise NotImplementedError("Clip norm mode %s not implemented" % tp["clip_norm_mode"])
    return nn.utils.clip_grad_norm_(model.parameters(), max_norm, foreach=True)


def get_optimizer_parameters(model, config):
    has = hasattr(model, "get_optimizer_parameters")
    if not has and hasattr(model, "module"):
        model, has = model.module, hasattr(model.module, "get_optimizer_parameters")
    return model.get_optimizer_parameters(config) if has else [p for p in model.parameters() if p.requires_grad]


class FusedClipAdam(torch.optim.Adam):
    """torch.optim.Adam (weight_decay 0, no amsgrad) whose ``step`` runs this build's multi-tensor kernels, with the global-norm
    clip of ``clip_gradients`` folded in (``t2s_grad_sqnorm`` / ``t2s_clip_coef`` / ``t2s_adam_step``, include/t2s_hip.h): one read
    of the gradients for the norm, one pass for clip + moments + update, instead of the framework's norm / stack / scale / Adam
    launches.  A subclass, so ``param_groups``, ``state_dict()`` / ``load_state_dict()`` (``exp_avg``, ``exp_avg_sq``, ``step`` per
    parameter, the reference's checkpoint layout) and ``LambdaLR`` work unchanged.

    ``step_clipped(max_norm)`` = ``clip_grad_norm_(params, max_norm)`` + ``step()``; returns the total gradient norm (a device
    scalar; before clipping, like clip_grad_norm_).  Gradients are left clipped in ``p.grad``, as the reference leaves them."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise ValueError("FusedClipAdam covers the reference's recipe only: weight_decay 0, no amsgrad")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False, fused=False)
        self._tables = None

    def _build_tables(self, live):
        from . import hipext as X
        chunk = X.lib().t2s_optim_chunk_elems()
        dev = live[0][1].device
        desc, group_of, chunks = [], [], []
        for t, (gi, p) in enumerate(live):
            st = self.state[p]
            desc.append([p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()])
            group_of.append(gi)
            chunks += [[t, c] for c in range((p.numel() + chunk - 1) // chunk)]
        key = tuple(tuple(d) for d in desc)
        tb = dict(key=key, desc=torch.tensor(desc, dtype=torch.int64, device=dev), group_of=torch.tensor(group_of, dtype=torch.int32, device=dev),
                  chunks=torch.tensor(chunks, dtype=torch.int32, device=dev), n_chunks=len(chunks),
                  partials=torch.empty(len(chunks), dtype=torch.float32, device=dev), norm_coef=torch.zeros(2, dtype=torch.float32, device=dev))
        self._tables = tb
        return tb

    @torch.no_grad()
    def step_clipped(self, max_norm=None, closure=None):
        import ctypes
        from . import hipext as X
        if closure is not None:
            raise ValueError("FusedClipAdam does not take a closure")
        if len(self.param_groups) > 8:
            raise ValueError("FusedClipAdam supports at most 8 param groups")
        self._opt_called = True            # what the wrapper lr_scheduler puts around step() records (its call-order check)
        live = []
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_cuda:
                    raise RuntimeError("FusedClipAdam needs dense fp32 device parameters and gradients")
                if not (p.is_contiguous() and p.grad.is_contiguous()):
                    raise RuntimeError("FusedClipAdam needs contiguous parameters and gradients")
                st = self.state[p]
                if len(st) == 0:          # the layout torch.optim.Adam creates (state_dict compatible)
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                live.append((gi, p))
        if not live:
            return None
        steps = {int(self.state[p]["step"].item() if torch.is_tensor(self.state[p]["step"]) else self.state[p]["step"]) for _, p in live}
        if len(steps) != 1:
            # uneven step counts (a parameter that got no gradient on some iterations, or a resumed reference checkpoint): the
            # kernels carry ONE bias correction, torch's Adam keeps per-parameter steps (the reference) - take its path this step
            # (torch's multi-tensor "foreach" implementation, not its per-parameter loop; said once, since uneven counts stay uneven)
            if not getattr(self, "_warned_uneven", False):
                import warnings
                warnings.warn("FusedClipAdam: per-parameter step counts differ (%d distinct values, e.g. after loading a checkpoint whose "
                              "parameters were updated unevenly): taking torch.optim.Adam's foreach path with clip_grad_norm_ instead of "
                              "the fused clip + Adam kernels on this and every later step" % len(steps))
                self._warned_uneven = True
            norm = None
            if max_norm is not None:
                norm = torch.nn.utils.clip_grad_norm_([p for _, p in live], max_norm, foreach=True)
            for g in self.param_groups:
                g["foreach"] = True
            try:
                super().step()
            finally:
                for g in self.param_groups:
                    g["foreach"] = False
            return norm
        step = steps.pop() + 1
        tb = self._tables
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(), p.numel())
                    for _, p in live)
        if tb is None or tb["key"] != key:          # pointers moved (first step, load_state_dict, a new gradient buffer)
            tb = self._build_tables(live)
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        for g in self.param_groups:
            if tuple(g["betas"]) != (b1, b2) or g["eps"] != eps:
                raise RuntimeError("FusedClipAdam: betas / eps must agree across param groups")
        lrs = (ctypes.c_float * len(self.param_groups))(*[float(g["lr"]) for g in self.param_groups])
        L, st = X.lib(), X.stream()
        nc = None
        if max_norm is not None:
            X.check(L.t2s_grad_sqnorm(X.ptr(tb["desc"]), X.ptr(tb["chunks"]), tb["n_chunks"], X.ptr(tb["partials"]), st), "t2s_grad_sqnorm")
            X.check(L.t2s_clip_coef(X.ptr(tb["partials"]), tb["n_chunks"], float(max_norm), X.ptr(tb["norm_coef"]), st), "t2s_clip_coef")
            nc = tb["norm_coef"]
        X.check(L.t2s_adam_step(X.ptr(tb["desc"]), X.ptr(tb["chunks"]), tb["n_chunks"], X.ptr(tb["group_of"]), lrs, len(self.param_groups),
                                float(b1), float(b2), float(eps), step, X.ptr(nc), 1 if nc is not None else 0, st), "t2s_adam_step")
        for _, p in live:
            self.state[p]["step"] = torch.tensor(float(step), dtype=torch.float32)
        return nc[0].clone() if nc is not None else None

    def step(self, closure=None):
        self.step_clipped(None, closure)


def build_optimizer(model, config):
    oc = config.optimizer_attributes
    if "type" not in oc:
        raise ValueError("Optimizer attributes must have a 'type' key specifying the type of optimizer.")
    params = dict(oc.get("params", {}))
    if not hasattr(torch.optim, oc.type):
        raise ValueError("No optimizer class of type {} present in torch".format(oc.type))
    groups = get_optimizer_parameters(model, config)
    every = [p for g in (groups if isinstance(groups[0], dict) else [{"params": groups}]) for p in g["params"]]
    on_card = bool(every) and all(p.is_cuda and p.dtype == torch.float32 and not p.is_sparse for p in every)
    if oc.type == "Adam" and on_card and params.get("weight_decay", 0) == 0 and not params.get("amsgrad", False):
        return FusedClipAdam(groups, **params)          # the reference's recipe: own multi-tensor clip + Adam kernels
    kw = {}
    if oc.type in ("Adam", "AdamW") and on_card:
        kw["fused"] = True          # one multi-tensor kernel over all parameters (same arithmetic)
    return getattr(torch.optim, oc.type)(groups, **params, **kw)


def clip_and_step(model, optimizer, config):
    """``clip_gradients`` + ``optimizer.step()`` of BaseTrainer._backward (base_trainer.py:268-270); one fused pass when the
    optimizer is ``FusedClipAdam``.  Returns the gradient norm before clipping (or None)."""
    tp = config["training_parameters"]
    if isinstance(optimizer, FusedClipAdam) and tp["clip_gradients"] and tp["max_grad_l2_norm"] is not None:
        if tp["clip_norm_mode"] != "all":
            raise NotImplementedError("Clip norm mode %s not implemented" % tp["clip_norm_mode"])
        return optimizer.step_clipped(tp["max_grad_l2_norm"])
    norm = clip_gradients(model, config) if tp["clip_gradients"] else None
    optimizer.step()
    return norm


def train_step(model, optimizer, scheduler, sample_list, config):
    """BaseTrainer._forward_pass + _extract_loss + _backward (base_trainer.py:251-278)."""
    out = model(sample_list)
    loss = sum(l.mean() for l in out["losses"].values())
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    norm = clip_and_step(model, optimizer, config)
    if scheduler is not None:
        scheduler.step()
    return loss.detach(), norm, out
