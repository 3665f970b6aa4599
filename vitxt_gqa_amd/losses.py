"""Losses of the T2S path behind the reference's ``Losses`` / ``PythiaLoss`` wrapper surface
(``pythia/modules/losses.py:41-173``): ``pos_bce_loss`` (:323-343) and ``InfoNCE`` (:346-385)."""
import torch
import torch.nn.functional as F
from torch import nn

from .registry import registry


class Losses(nn.Module):
    """losses.py:70-111: dict ``{"<type>/<dataset>/<name>": weight * loss}``; empty if no targets."""

    def __init__(self, loss_list):
        super().__init__()
        self.losses = [PythiaLoss(l) for l in loss_list]

    def forward(self, sample_list, model_output, *args, **kwargs):
        output = {}
        if "targets" not in sample_list:
            return output
        for loss in self.losses:
            output.update(loss(sample_list, model_output, *args, **kwargs))
        key = "losses.{}.{}".format(sample_list.get("dataset_name", "vtextgqa"), sample_list.get("dataset_type", "train"))
        registry.register(key, output)
        return output


class PythiaLoss(nn.Module):
    """losses.py:134-173: weight, reshape to [1], key ``<dataset_type>/<dataset_name>/<loss name>``."""

    def __init__(self, params):
        super().__init__()
        if "type" not in params:
            raise ValueError("Parameters to loss must have 'type' field to specify type of loss to instantiate")
        self.name = params["type"]
        self.weight = params["weight"]
        loss_class = registry.get_loss_class(self.name)
        if loss_class is None:
            raise ValueError("No loss named {} is registered to registry".format(self.name))
        self.loss_criterion = loss_class(**params.get("params", {}))

    def forward(self, sample_list, model_output, *args, **kwargs):
        loss = self.weight * self.loss_criterion(sample_list, model_output, *args, **kwargs)
        if loss.dim() == 0:
            loss = loss.view(1)
        key = "{}/{}/{}".format(sample_list.get("dataset_type", "train"), sample_list.get("dataset_name", "vtextgqa"), self.name)
        return {key: loss}


@registry.register_loss("pos_bce_loss")
class POSBCEWithMaskLoss(nn.Module):
    """losses.py:329-343: sum(BCEWithLogits(pos_scores, targets) * mask) / max(sum(mask), 1)."""

    def forward(self, sample_list, model_output):
        scores = model_output["pos_scores"].float()
        targets = sample_list["targets"].to(scores.dtype)
        loss_mask = sample_list["train_loss_mask"].to(scores.dtype)
        assert scores.dim() == 3 and loss_mask.dim() == 2
        losses = F.binary_cross_entropy_with_logits(scores, targets, reduction="none")
        losses = losses * loss_mask.unsqueeze(-1)
        count = torch.clamp(loss_mask.sum(), min=1.0)
        return losses.sum() / count


@registry.register_loss("InfoNCE")
class InfoNCE(nn.Module):
    """losses.py:346-385: query = ref_scores, positive = pos_scores, negative = neg_scores; each
    L2-normalised over the last dim, flattened per sample, cosine similarities, CE over the two
    logits / 0.1 with label 0, mean over the batch."""

    def __init__(self, temperature=0.1, reduction="mean", negative_mode="unpaired"):
        super().__init__()
        self.temperature = temperature

    def forward(self, sample_list, model_output, temperature=0.1):
        q = F.normalize(model_output["ref_scores"].float(), dim=-1)
        p = F.normalize(model_output["pos_scores"].float(), dim=-1)
        n = F.normalize(model_output["neg_scores"].float(), dim=-1)
        B = q.size(0)
        q, p, n = q.reshape(B, -1), p.reshape(B, -1), n.reshape(B, -1)
        logits = torch.stack([F.cosine_similarity(q, p, dim=1), F.cosine_similarity(q, n, dim=1)], dim=1)
        labels = torch.zeros(B, dtype=torch.long, device=logits.device)
        return F.cross_entropy(logits / temperature, labels, reduction="mean")
