"""Canonicalizer base classes: the drop-in boundary.

Same class names, constructor arguments, method names, state (``canonicalization_info_dict``, ``device``)
and error behaviour as the reference's ``equiadapt/common/basecanonicalization.py`` so that callers
(`examples/*/model.py` style training loops) switch by changing the import.  The group-element
bookkeeping here is a handful of (B, G)-sized torch ops on the device; the heavy arithmetic lives in the
HIP kernels the subclasses call.
"""
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.nn.functional as F

from equiadapt_amd import ops

CanonOut = Union[torch.Tensor, Tuple[torch.Tensor, List]]


class BaseCanonicalization(torch.nn.Module):
    """Interface of every canonicalizer (reference: basecanonicalization.py:29-93).

    ``forward(x, targets=None, **kw)`` is ``canonicalize``; subclasses implement ``canonicalize`` and
    ``invert_canonicalization``.  ``canonicalization_info_dict`` carries the state of the LAST call
    (one canonicalize -> its invert; not re-entrant, exactly like the reference).
    """

    def __init__(self, canonicalization_network: torch.nn.Module):
        super().__init__()
        self.canonicalization_network = canonicalization_network
        self.canonicalization_info_dict: Dict[str, Any] = {}

    def forward(self, x: torch.Tensor, targets: Optional[List] = None, **kwargs: Any) -> CanonOut:
        return self.canonicalize(x, targets, **kwargs)

    def canonicalize(self, x: torch.Tensor, targets: Optional[List] = None, **kwargs: Any) -> CanonOut:
        raise NotImplementedError()

    def invert_canonicalization(self, x_canonicalized_out: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        raise NotImplementedError()


class IdentityCanonicalization(BaseCanonicalization):
    """No-op canonicalizer (reference: basecanonicalization.py:96-179)."""

    def __init__(self, canonicalization_network: torch.nn.Module = torch.nn.Identity()):
        super().__init__(canonicalization_network)

    def canonicalize(self, x: torch.Tensor, targets: Optional[List] = None, **kwargs: Any) -> CanonOut:
        return (x, targets) if targets else x

    def invert_canonicalization(self, x_canonicalized_out: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        return x_canonicalized_out

    def get_prior_regularization_loss(self) -> torch.Tensor:
        return torch.tensor(0.0)

    def get_identity_metric(self) -> torch.Tensor:
        return torch.tensor(1.0)


class DiscreteGroupCanonicalization(BaseCanonicalization):
    """Discrete-group bookkeeping: activations -> (differentiable) one-hot, prior loss, identity metric.

    Reference: basecanonicalization.py:182-311.  ``num_group`` is set by subclasses.
    """

    def __init__(self, canonicalization_network: torch.nn.Module, beta: float = 1.0,
                 gradient_trick: str = "straight_through"):
        super().__init__(canonicalization_network)
        self.beta = beta
        self.gradient_trick = gradient_trick

    def group_index(self, group_activations: torch.Tensor) -> torch.Tensor:
        """Bit-exact orientation index (B,) int32: first maximum over the group axis.

        On the device this is the wavefront-shuffle argmax kernel (eqa_group_argmax); it replaces
        ``torch.argmax`` at basecanonicalization.py:233-235.
        """
        return ops.group_argmax(group_activations)

    def groupactivations_to_groupelementonehot(self, group_activations: torch.Tensor,
                                               group_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(B, G) activations -> (B, G) one-hot; straight-through / gumbel as the reference (:221-256)."""
        if group_index is None:
            group_index = self.group_index(group_activations)
        hard = F.one_hot(group_index.long(), self.num_group).float()
        if self.gradient_trick == "straight_through":
            if not self.training:
                return hard
            soft = F.softmax(self.beta * group_activations, dim=-1)
            return hard + soft - soft.detach()
        if self.gradient_trick == "gumbel_softmax":
            return F.gumbel_softmax(group_activations, tau=1, hard=True)
        raise ValueError(f"Gradient trick {self.gradient_trick} not implemented")

    def get_prior_regularization_loss(self) -> torch.Tensor:
        """Cross-entropy of the activations against the identity element (:290-301)."""
        acts = self.canonicalization_info_dict["group_activations"]
        target = torch.zeros((acts.shape[0],), dtype=torch.long, device=self.device)
        return F.cross_entropy(acts, target)

    def get_identity_metric(self) -> torch.Tensor:
        """Fraction of samples mapped to the identity element (:303-311)."""
        info = self.canonicalization_info_dict
        # argmax of the ACTIVATIONS like the reference (under gumbel_softmax "group_index" is the sampled element instead)
        idx = info.get("argmax_index", info.get("group_index"))
        if idx is None:
            idx = info["group_activations"].argmax(dim=-1)
        return (idx == 0).float().mean()


class ContinuousGroupCanonicalization(BaseCanonicalization):
    """Continuous-group bookkeeping (reference: basecanonicalization.py:314-430)."""

    def __init__(self, canonicalization_network: torch.nn.Module, beta: float = 1.0):
        super().__init__(canonicalization_network)
        self.beta = beta

    def canonicalizationnetworkout_to_groupelement(self, group_activations: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError()

    def _identity_like(self, rep: torch.Tensor) -> torch.Tensor:
        return torch.eye(rep.shape[-1], device=self.device).repeat(rep.shape[0], 1, 1)

    def get_prior_regularization_loss(self) -> torch.Tensor:
        """MSE between the predicted group element matrices and the identity (:390-408)."""
        rep = self.canonicalization_info_dict["group_element_matrix_representation"]
        return F.mse_loss(rep, self._identity_like(rep))

    def get_identity_metric(self) -> torch.Tensor:
        """1 - MSE(R, I) (:410-430)."""
        rep = self.canonicalization_info_dict["group_element_matrix_representation"]
        return 1.0 - F.mse_loss(rep, self._identity_like(rep)).mean()
