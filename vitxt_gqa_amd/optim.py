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
        self._status_probe = None          # (pinned host copy of the device's sticky status words, event): checked at the next step

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
                  partials=torch.empty(len(chunks), dtype=torch.float32, device=dev), norm_coef=torch.zeros(2, dtype=torch.float32, device=dev),
                  unit=torch.tensor([0.0, 1.0], dtype=torch.float32, device=dev))
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
                              "the fused clip + Adam kernels while the counts stay uneven (decided anew on every step)" % len(steps))
                self._warned_uneven = True
            raise_if_handoff_failed(live[0][1].device)          # (this path has no device-side gate: look before the update)
            norm = None
            if max_norm is not None:
                norm = torch.nn.utils.clip_grad_norm_([p for _, p in live], max_norm, foreach=True)
            saved = [g.get("foreach") for g in self.param_groups]
            for g in self.param_groups:
                g["foreach"] = True
            try:
                super().step()
            finally:
                for g, v in zip(self.param_groups, saved):
                    g["foreach"] = v
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
        from . import ops
        dev = live[0][1].device
        # a hand-off wait of the fused attention backward that timed out in the PREVIOUS step gated that step off on the device
        # (below); its sticky word was copied to the host behind the step without a synchronisation: look at it now and raise
        self._check_status_probe()
        nc = None
        if max_norm is not None:
            X.check(L.t2s_grad_sqnorm(X.ptr(tb["desc"]), X.ptr(tb["chunks"]), tb["n_chunks"], X.ptr(tb["partials"]), st), "t2s_grad_sqnorm")
            X.check(L.t2s_clip_coef(X.ptr(tb["partials"]), tb["n_chunks"], float(max_norm), X.ptr(tb["norm_coef"]), st), "t2s_clip_coef")
            nc = tb["norm_coef"]
        else:
            nc = tb["norm_coef"]
            nc.copy_(tb["unit"])          # (norm 0, coefficient 1: no clipping)
        # the gate: with the sticky status word set (this step's gradients hold NaN rows) the clip coefficient becomes -1 and
        # t2s_adam_step returns without touching parameters, moments or gradients
        sticky = ops.fused_status_tensor(dev)
        X.check(L.t2s_status_gate(X.ptr(sticky), X.ptr(nc), st), "t2s_status_gate")
        X.check(L.t2s_adam_step(X.ptr(tb["desc"]), X.ptr(tb["chunks"]), tb["n_chunks"], X.ptr(tb["group_of"]), lrs, len(self.param_groups),
                                float(b1), float(b2), float(eps), step, X.ptr(nc), 1 if max_norm is not None else 0, st), "t2s_adam_step")
        for _, p in live:
            self.state[p]["step"] = torch.tensor(float(step), dtype=torch.float32)
        # the sticky word travels to a pinned host buffer behind the step (asynchronously); the next step - or
        # raise_if_handoff_failed() at a logging / checkpoint synchronisation point - reads it
        if self._status_probe is None:
            self._status_probe = [torch.zeros(4, dtype=torch.int32).pin_memory(), torch.cuda.Event(), dev, False]
        host, ev, _, _ = self._status_probe
        host.copy_(sticky, non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        self._status_probe[3] = True
        return nc[0].clone() if max_norm is not None else None

    def _check_status_probe(self):
        pr = self._status_probe
        if pr is None or not pr[3]:
            return
        host, ev, dev, _ = pr
        ev.synchronize()          # recorded a whole step ago: complete unless the host runs more than a step ahead
        pr[3] = False
        from . import ops
        if ops.decode_status_words(host.tolist()) != 0:
            # the step behind this word was gated off on the device (no parameter, moment or gradient touched), but the host had
            # already advanced the step counts: take that step back, so that a trainer which catches the error and carries on gets
            # the bias correction of the updates that were really applied (a LambdaLR stepped by the trainer is the trainer's to rewind)
            for st in self.state.values():
                if "step" in st:
                    st["step"] = st["step"] - 1
            raise_if_handoff_failed(dev)

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


def raise_if_handoff_failed(device=None):
    """Synchronising check of the sticky status of the fused attention backward (ops.fused_handoff_status): raises
    ``ops.HandoffTimeout`` when a bounded wait of the dQ hand-off has timed out since the last reset (``ops.HandoffPlacement`` when an
    XCD group of workgroups ran on two XCDs; both are ``ops.HandoffError``), and clears the word.  Call it
    where the trainer synchronises anyway (loss logging, checkpoints); ``FusedClipAdam`` also checks the previous step's word at
    every step without synchronising, and the step in which the timeout happened was gated off on the device."""
    from . import ops
    st = ops.fused_handoff_status(device)
    if st != 0:
        ops.reset_fused_status()
        if st & 2:
            raise ops.HandoffPlacement("fused attention backward: workgroups of one XCD group ran on different XCDs (status word %d) - the running "
                                       "dQ sums, kept in one XCD's L2, may have been read stale; the optimizer step of that iteration was skipped "
                                       "on the device.  Set T2S_FB_HANDOFF_SCOPE=agent (write-through sums) on this device" % st)
        raise ops.HandoffTimeout("fused attention backward: a hand-off wait timed out (status word %d) - the dQ rows behind it are NaN and the "
                                 "optimizer step of that iteration was skipped on the device; a workgroup died or the card is oversubscribed" % st)


def clip_and_step(model, optimizer, config):
    """``clip_gradients`` + ``optimizer.step()`` of BaseTrainer._backward (base_trainer.py:268-270); one fused pass when the
    optimizer is ``FusedClipAdam``.  Returns the gradient norm before clipping (or None)."""
    tp = config["training_parameters"]
    if isinstance(optimizer, FusedClipAdam) and tp["clip_gradients"] and tp["max_grad_l2_norm"] is not None:
        if tp["clip_norm_mode"] != "all":
            raise NotImplementedError("Clip norm mode %s not implemented" % tp["clip_norm_mode"])
        return optimizer.step_clipped(tp["max_grad_l2_norm"])
    if any(p.is_cuda for g in optimizer.param_groups for p in g["params"]):
        raise_if_handoff_failed()          # an optimizer without the device-side gate: look (and synchronise) before the update
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
