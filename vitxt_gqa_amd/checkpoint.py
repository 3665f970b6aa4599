"""Checkpoint files interchangeable with the reference's.

The reference writes ``torch.save({"model": state_dict, "optimizer": ..., "best_iteration": ..., "best_metric_value": ...,
"config": ...})`` (``pythia/utils/checkpoint.py:226-240``) and, when loading, accepts a bare state_dict as well, strips one
leading ``module.`` left by DataParallel / DDP and renames ``fa_history`` -> ``fa_context`` (``checkpoint.py:90-111``).  The
parameter names of this build's T2S are the reference's (``vitxt_gqa_amd/schema.py``), so the tensors load unchanged; the
optimizer's param groups list every parameter in the reference's order (``T2S.get_optimizer_parameters``: the frozen dead
parameters included), so ``optimizer.state_dict()`` indices agree with the reference's too and its checkpoints resume.
"""
import torch


def normalize_state_dict(sd):
    """The reference's key normalisation for a non-wrapped model (checkpoint.py:100-111)."""
    out = {}
    for k, v in sd.items():
        if "fa_history" in k:
            k = k.replace("fa_history", "fa_context")
        elif k.startswith("module."):
            k = k.replace("module.", "", 1)
        out[k] = v
    return out


def load_checkpoint(path_or_obj, model=None, optimizer=None, strict=True, map_location="cpu"):
    """Read a reference-format checkpoint (file path or already loaded object).  Returns the dict with a normalised
    ``model`` entry; loads it into ``model`` / ``optimizer`` when given."""
    ck = torch.load(path_or_obj, map_location=map_location, weights_only=False) if isinstance(path_or_obj, (str, bytes)) or hasattr(path_or_obj, "read") \
        else path_or_obj
    if "model" not in ck:                       # a bare state_dict (checkpoint.py:88-91)
        ck = {"model": ck}
    ck = dict(ck)
    ck["model"] = normalize_state_dict(ck["model"])
    if model is not None:
        target = model.module if hasattr(model, "module") and not hasattr(model, "get_optimizer_parameters") else model
        target.load_state_dict(ck["model"], strict=strict)
    if optimizer is not None and ck.get("optimizer") is not None:     # save_checkpoint writes None when it was given no optimizer
        optimizer.load_state_dict(ck["optimizer"])
    return ck


def save_checkpoint(path, model, optimizer=None, best_iteration=0, best_metric_value=None, config=None):
    """Write the reference's layout (checkpoint.py:226-236; the VCS fields of :238 are omitted)."""
    target = model.module if hasattr(model, "module") and not hasattr(model, "get_optimizer_parameters") else model
    ck = {"model": target.state_dict(), "optimizer": optimizer.state_dict() if optimizer is not None else None,
          "best_iteration": best_iteration, "best_metric_value": best_metric_value, "config": config}
    torch.save(ck, path)
    return ck
