"""``BaseModel`` contract of ``pythia/models/base_model.py:53-149``: ``build()``,
``init_losses_and_metrics()``, ``forward(sample_list) -> dict`` and ``__call__`` appending ``losses`` and ``metrics``
(the yml ``metrics`` list through ``metrics.Metrics``: host-side string / IoU evaluators, as in the reference)."""
import collections

from torch import nn

from .registry import registry


class BaseModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.writer = registry.get("writer")

    def build(self):
        raise NotImplementedError("Build method not implemented in the child model class.")

    def init_losses_and_metrics(self):
        from .losses import Losses
        self.losses = Losses(self.config.get("losses", []))
        from .metrics import Metrics
        self.metrics = Metrics(self.config.get("metrics", []))

    def forward(self, sample_list, *args, **kwargs):
        raise NotImplementedError("Forward of the child model class needs to be implemented.")

    def __call__(self, sample_list, *args, **kwargs):
        model_output = super().__call__(sample_list, *args, **kwargs)
        assert isinstance(model_output, collections.abc.Mapping), "A dict must be returned from the forward of the model."
        if "losses" not in model_output:
            model_output["losses"] = self.losses(sample_list, model_output)
        if "metrics" not in model_output:
            model_output["metrics"] = self.metrics(sample_list, model_output)
        return model_output
