"""Data-parallel training step for a canonicalizer wrapping an unmodified prediction network.

The reference trains through PyTorch-Lightning with ``strategy="ddp"``
(examples/images/classification/train_utils.py:89-91); the Lightning/Hydra/wandb harness is out of scope, but its
STEP SEMANTICS are what a user switching over relies on, so they are reproduced here on plain torch:

* loss = task_weight * CE(prediction(canonicalizer(x)), y) + prior_weight * prior loss
         [+ 2 * group_contrast_weight * optimization-specific loss for "opt" canonicalizers -- the reference adds that
         term twice (examples/images/classification/model.py:76-89); kept, flagged by ``reference_double_contrast``]
* two parameter groups (prediction lr / canonicalization lr); SGD(0.9, wd 5e-4) + MultiStepLR or AdamW
  (model.py:184-239);
* one process per GPU, gradients averaged by DDP's bucketed all-reduce (backend "nccl" == RCCL over xGMI on ROCm),
  BatchNorm statistics per replica (no SyncBN), scalar metrics all-reduced (the reference's ``sync_dist=True``).

The canonicalize / invert forward needs no collective: every image is independent.
"""
from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F


@dataclass
class LossWeights:
    task_weight: float = 1.0
    prior_weight: float = 100.0
    group_contrast_weight: float = 0.0
    reference_double_contrast: bool = True  # the reference adds the contrast term twice


class CanonicalizedClassifier(torch.nn.Module):
    """``prediction_network(canonicalizer(x))`` with the reference's loss composition."""

    def __init__(self, canonicalizer: torch.nn.Module, prediction_network: torch.nn.Module,
                 weights: Optional[LossWeights] = None):
        super().__init__()
        self.canonicalizer = canonicalizer
        self.prediction_network = prediction_network
        self.weights = weights or LossWeights()

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> Dict[str, torch.Tensor]:
        w = self.weights
        out: Dict[str, torch.Tensor] = {}
        loss = x.new_zeros(())
        x_canon = self.canonicalizer(x)
        if w.group_contrast_weight and hasattr(self.canonicalizer, "get_optimization_specific_loss"):
            contrast = self.canonicalizer.get_optimization_specific_loss()
            loss = loss + contrast * w.group_contrast_weight * (2.0 if w.reference_double_contrast else 1.0)
            out["optimization_specific_loss"] = contrast.detach()
        if w.task_weight:
            logits = self.prediction_network(x_canon)
            task = F.cross_entropy(logits, y)
            loss = loss + w.task_weight * task
            out["task_loss"] = task.detach()
            out["acc"] = (logits.argmax(dim=-1) == y).float().mean()
        if w.prior_weight and hasattr(self.canonicalizer, "get_prior_regularization_loss"):
            prior = self.canonicalizer.get_prior_regularization_loss()
            loss = loss + prior * w.prior_weight
            out["prior_loss"] = prior.detach()
            out["identity_metric"] = self.canonicalizer.get_identity_metric().detach()
        out["loss"] = loss
        return out


def select_image_optimizer_kind(prediction_network_architecture: str, dataset_name: str) -> str:
    """The reference's selection rule for the image classification loop
    (examples/images/classification/model.py:184-191): SGD + MultiStepLR when the prediction network is a ResNet and the
    dataset is not an MNIST variant, AdamW otherwise."""
    return "sgd" if ("resnet" in prediction_network_architecture and "mnist" not in dataset_name) else "adamw"


def configure_optimizer(model: CanonicalizedClassifier, prediction_lr: float, canonicalization_lr: float,
                        kind: Optional[str] = "adamw", max_epochs: int = 100,
                        prediction_network_architecture: Optional[str] = None, dataset_name: Optional[str] = None):
    """Two parameter groups as in the reference; returns (optimizer, scheduler or None).

    ``kind=None`` applies the reference's own rule from the architecture / dataset names
    (``select_image_optimizer_kind``); SGD(0.9, wd 5e-4) + MultiStepLR(milestones //6,//3,//2 when max_epochs > 100 else
    //3,//2; gamma 0.1) or AdamW with defaults (model.py:184-239)."""
    if kind is None:
        if prediction_network_architecture is None or dataset_name is None:
            raise ValueError("kind=None needs prediction_network_architecture and dataset_name (the reference's rule)")
        kind = select_image_optimizer_kind(prediction_network_architecture, dataset_name)
    groups = [{"params": list(model.prediction_network.parameters()), "lr": prediction_lr},
              {"params": list(model.canonicalizer.parameters()), "lr": canonicalization_lr}]
    if kind == "sgd":
        opt = torch.optim.SGD(groups, momentum=0.9, weight_decay=5e-4)
        ms = [max_epochs // 6, max_epochs // 3, max_epochs // 2] if max_epochs > 100 else [max_epochs // 3, max_epochs // 2]
        return opt, torch.optim.lr_scheduler.MultiStepLR(opt, milestones=ms, gamma=0.1)
    if kind == "adamw":
        return torch.optim.AdamW(groups), None
    raise ValueError(f"unknown optimizer kind {kind}")


def configure_pointcloud_optimizer(model: CanonicalizedClassifier, prediction_lr: float, canonicalization_lr: float,
                                   optimizer: str = "SGD", lr_scheduler: str = "cosine", num_epochs: int = 250):
    """The point-cloud classification loop's rule (examples/pointcloud/classification/model.py:245-300):
    "Adam": Adam(wd 1e-4), no scheduler; "SGD": both learning rates x100, momentum 0.9, wd 1e-4, with
    CosineAnnealingLR(T_max=num_epochs, eta_min=1e-3) or StepLR(20, 0.7); anything else raises NotImplementedError."""
    params = lambda scale: [{"params": list(model.prediction_network.parameters()), "lr": prediction_lr * scale},  # noqa: E731
                            {"params": list(model.canonicalizer.parameters()), "lr": canonicalization_lr * scale}]
    if optimizer == "Adam":
        return torch.optim.Adam(params(1), weight_decay=1e-4), None
    if optimizer == "SGD":
        opt = torch.optim.SGD(params(100), momentum=0.9, weight_decay=1e-4)
        if lr_scheduler == "cosine":
            return opt, torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=num_epochs, eta_min=1e-3)
        if lr_scheduler == "step":
            return opt, torch.optim.lr_scheduler.StepLR(opt, step_size=20, gamma=0.7)
        raise NotImplementedError(f"Unknown learning rate decay schedule {lr_scheduler}")
    raise NotImplementedError(optimizer)


def shard_range(n: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n items for ``rank`` (first n % world ranks take one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def wrap_ddp(model: torch.nn.Module, device: Optional[torch.device] = None, bucket_cap_mb: int = 64,
             find_unused: Optional[bool] = None, force: bool = False) -> torch.nn.Module:
    """DistributedDataParallel over the default process group (no-op when not initialised / world size 1).

    xGMI is point-to-point, so a ring all-reduce is bound by one ~153 GB/s link; a 64 MB bucket keeps the ~100 MB of
    ResNet-50 gradients to two collectives whose latency overlaps the backward pass.
    ``find_unused_parameters`` defaults to "only when task_weight == 0" -- then the prediction network receives no
    gradient (the reference switches to ddp_find_unused_parameters_true in the analogous situation,
    examples/images/segmentation/train_utils.py:84-88).
    """
    if find_unused is None:
        w = getattr(model, "weights", None)
        find_unused = bool(w is not None and not w.task_weight)
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return model   # (`force`: wrap a single process too, to exercise DDP's hooks on one GPU)
    ids = [device.index] if device is not None and device.type == "cuda" else None
    # gradient_as_bucket_view: the gradients ARE the bucket slices -- no copy into the buckets before each all-reduce (and half
    # the gradient memory); train_step clears them with set_to_none=True, which DDP re-points at the buckets
    return torch.nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb,
                                                     find_unused_parameters=find_unused, gradient_as_bucket_view=True)


def reduce_metrics(metrics: Dict[str, torch.Tensor]) -> Dict[str, float]:
    """Mean over ranks of scalar metrics (the reference's ``sync_dist=True``); plain floats out."""
    keys = sorted(metrics)
    vec = torch.stack([metrics[k].detach().float().reshape(()) for k in keys])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        vec = vec / dist.get_world_size()
    return {k: v for k, v in zip(keys, vec.tolist())}


def train_step(ddp_model: torch.nn.Module, optimizer: torch.optim.Optimizer, x: torch.Tensor, y: torch.Tensor,
               check_nan: bool = True) -> Dict[str, torch.Tensor]:
    """One optimisation step on this rank's shard: forward, backward (DDP all-reduces the gradients), update.

    ``check_nan`` reproduces the reference's ``assert not torch.isnan(loss)`` (model.py:125), which reads the loss on the
    host (one device synchronisation between forward and backward); a throughput loop passes False and looks at the
    returned loss whenever it logs."""
    optimizer.zero_grad(set_to_none=True)
    out = ddp_model(x, y)
    if check_nan and torch.isnan(out["loss"]):
        raise FloatingPointError("Loss is NaN")
    out["loss"].backward()
    optimizer.step()
    return out
