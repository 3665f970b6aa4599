"""vitxt_gqa_amd -- MI355X-native T2S-QA fusion path of zhousheng97/ViTXT-GQA.

(The directory ``vitxt-gqa_amd`` is a symlink to this package: a hyphen cannot be imported.)

Python host code on PyTorch-ROCm (device memory, streams, library GEMMs, torch.distributed/RCCL) over a
C-ABI shared library of hand-written gfx950 kernels (``include/t2s_hip.h`` -> ``libt2s_hip.so``).
Importing the package does not need a GPU; running the model does, and there is no CPU fallback.
"""
from .registry import registry  # noqa: F401
from .config import ConfigNode, t2s_model_config, training_config  # noqa: F401
from .sample import SampleList  # noqa: F401


def bind_reference(reference_registry, reference_base_model):
    """Register this build's T2S (and its two losses) with the REFERENCE's registry / BaseModel instead of the
    host-side mirrors.  Must run before ``vitxt_gqa_amd.t2s`` is first imported.
    Returns the T2S class."""
    import importlib
    import sys
    global registry
    if __name__ + ".t2s" in sys.modules:
        raise RuntimeError("bind_reference() must be called before vitxt_gqa_amd.t2s is imported")
    reg_mod = importlib.import_module(__name__ + ".registry")     # the MODULE (the package attribute is the instance)
    sys.modules[__name__ + ".registry"].registry = reference_registry
    reg_mod.registry = reference_registry
    bm = importlib.import_module(__name__ + ".base_model")
    bm.BaseModel, bm.registry = reference_base_model, reference_registry
    registry = reference_registry
    t2s = importlib.import_module(__name__ + ".t2s")              # decorator registers "t2s" with the reference registry
    importlib.import_module(__name__ + ".losses")                 # registers pos_bce_loss / InfoNCE (same names)
    return t2s.T2S


def build_model(config):
    """``pythia/utils/build_utils.py:38-51``: registry lookup, construct, build(), init_losses_and_metrics()."""
    from . import t2s as _t2s  # noqa: F401  (registers "t2s")
    from . import losses as _losses  # noqa: F401
    model_class = registry.get_model_class(config.model)
    if model_class is None:
        raise ValueError("No model registered for name: %s" % config.model)
    model = model_class(config)
    model.build()
    model.init_losses_and_metrics()
    return model
