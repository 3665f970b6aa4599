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


def build_optimizer(model, config):
    oc = config.optimizer_attributes
    if "type" not in oc:
        raise ValueError("Optimizer attributes must have a 'type' key specifying the type of optimizer.")
    params = dict(oc.get("params", {}))
    if not hasattr(torch.optim, oc.type):
        raise ValueError("No optimizer class of type {} present in torch".format(oc.type))
    kw = {}
    if oc.type in ("Adam", "AdamW") and torch.cuda.is_available():
        kw["fused"] = True          # one multi-tensor kernel over all parameters (same arithmetic)
    return getattr(torch.optim, oc.type)(get_optimizer_parameters(model, config), **params, **kw)


def train_step(model, optimizer, scheduler, sample_list, config):
    """BaseTrainer._forward_pass + _extract_loss + _backward (base_trainer.py:251-278)."""
    out = model(sample_list)
    loss = sum(l.mean() for l in out["losses"].values())
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    norm = clip_gradients(model, config) if config["training_parameters"]["clip_gradients"] else None
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach(), norm, out
