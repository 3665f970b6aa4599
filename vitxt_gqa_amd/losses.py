"""Losses of the T2S path behind the reference's ``Losses`` / ``PythiaLoss`` wrapper surface
(``pythia/modules/losses.py:41-173``): ``pos_bce_loss`` (:323-343) and ``InfoNCE`` (:346-385)."""
import torch
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


class _BCEMaskedFn(torch.autograd.Function):
    """sum_r mask[r] * sum_c BCEWithLogits(x[r, c], t[r, c]) / max(sum(mask), 1): loss and gradient in one HIP pass."""

    @staticmethod
    def forward(ctx, scores, targets, loss_mask):
        from . import ops
        row_loss, grad = ops.bce_masked(scores.contiguous(), targets.contiguous(), loss_mask.reshape(-1).contiguous())
        count = torch.clamp(loss_mask.sum(), min=1.0)
        ctx.save_for_backward(grad, count)
        return row_loss.sum() / count

    @staticmethod
    def backward(ctx, g):
        grad, count = ctx.saved_tensors
        return grad * (g / count), None, None


class _InfoNCEStatsFn(torch.autograd.Function):
    """[rows, 5] = (q.q, p.p, n.n, q.p, q.n) per logits row; backward is one fused pass over the three tensors."""

    @staticmethod
    def forward(ctx, q, p, n):
        from . import ops
        q, p, n = q.contiguous(), p.contiguous(), n.contiguous()
        ctx.save_for_backward(q, p, n)
        return ops.infonce_stats(q, p, n)

    @staticmethod
    def backward(ctx, g):
        from . import ops
        q, p, n = ctx.saved_tensors
        return ops.infonce_bwd(q, p, n, g.contiguous())


@registry.register_loss("pos_bce_loss")
class POSBCEWithMaskLoss(nn.Module):
    """losses.py:329-343: sum(BCEWithLogits(pos_scores, targets) * mask) / max(sum(mask), 1)."""

    def forward(self, sample_list, model_output):
        scores = model_output["pos_scores"].float()
        targets = sample_list["targets"].float()
        loss_mask = sample_list["train_loss_mask"].float()
        assert scores.dim() == 3 and loss_mask.dim() == 2
        return _BCEMaskedFn.apply(scores, targets, loss_mask)


@registry.register_loss("InfoNCE")
class InfoNCE(nn.Module):
    """losses.py:346-385: query = ref_scores, positive = pos_scores, negative = neg_scores; each
    L2-normalised over the last dim (eps 1e-12), flattened per sample, cosine similarities (eps 1e-8), CE over the
    two logits / 0.1 with label 0, mean over the batch.  All reductions over the [B, 12, V+N] logits come from the
    five per-row statistics of the HIP kernel; the remaining arithmetic is on [B, 12] tensors."""

    def __init__(self, temperature=0.1, reduction="mean", negative_mode="unpaired"):
        super().__init__()
        self.temperature = temperature

    def forward(self, sample_list, model_output, temperature=0.1):
        q, p, n = model_output["ref_scores"].float(), model_output["pos_scores"].float(), model_output["neg_scores"].float()
        B, D, _ = q.shape
        st = _InfoNCEStatsFn.apply(q, p, n).view(B, D, 5)
        nq, np_, nn_ = [st[..., i].sqrt().clamp_min(1e-12) for i in range(3)]      # F.normalize denominators per row
        # flattened normalised vectors: dot = sum_r (q_r.p_r)/(|q_r||p_r|), squared norm = sum_r |q_r|^2/|q_r|^2
        dot_qp = (st[..., 3] / (nq * np_)).sum(1)
        dot_qn = (st[..., 4] / (nq * nn_)).sum(1)
        Nq = (st[..., 0] / (nq * nq)).sum(1).sqrt().clamp_min(1e-8)
        Np = (st[..., 1] / (np_ * np_)).sum(1).sqrt().clamp_min(1e-8)
        Nn = (st[..., 2] / (nn_ * nn_)).sum(1).sqrt().clamp_min(1e-8)
        logits = torch.stack([dot_qp / (Nq * Np), dot_qn / (Nq * Nn)], dim=1) / temperature
        return (torch.logsumexp(logits, dim=1) - logits[:, 0]).mean()
