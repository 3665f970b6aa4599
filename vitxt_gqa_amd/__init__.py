"""vitxt_gqa_amd -- MI355X-native T2S-QA fusion path of zhousheng97/ViTXT-GQA.

(The directory ``vitxt-gqa_amd`` is a symlink to this package: a hyphen cannot be imported.)

Python host code on PyTorch-ROCm (device memory, streams, library GEMMs, torch.distributed/RCCL) over a
C-ABI shared library of hand-written gfx950 kernels (``include/t2s_hip.h`` -> ``libt2s_hip.so``).
Importing the package does not need a GPU; running the model does, and there is no CPU fallback.
"""
from .registry import registry  # noqa: F401
from .config import ConfigNode, t2s_model_config, training_config  # noqa: F401
from .sample import SampleList  # noqa: F401


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
