"""E(3) canonicalization of n-body systems.

Reference: equiadapt/nbody/canonicalization/euclidean_group.py:8-157.  Same API: ``forward(nodes, **kwargs)`` with the
keyword arguments ``loc, edges, vel, edge_attr, charges`` in that order, ``canonicalize`` returning
``(canonical_loc, canonical_vel)``, ``invert_canonicalization`` mapping predicted locations back (``x R + t``).
The canonicalization network (``VNDeepSets`` in the reference, which needs torch_scatter) is any module returning
``(rotation_vectors (M,3,3), translation_vectors (M,3))``; the rigid actions and the modified Gram-Schmidt run as HIP kernels
(forward only -- n-body training keeps the op-by-op path, selected automatically when autograd is recording).
"""
from typing import Any, Dict, List, Optional, Tuple, Union

import torch

from equiadapt_amd import ops
from equiadapt_amd.common.basecanonicalization import ContinuousGroupCanonicalization


class EuclideanGroupNBody(ContinuousGroupCanonicalization):
    def __init__(self, canonicalization_network: torch.nn.Module) -> None:
        super().__init__(canonicalization_network)

    def forward(self, x: torch.Tensor, targets: Optional[List] = None, **kwargs: Any):
        return self.canonicalize(x, None, **kwargs)

    @staticmethod
    def _fused(*tensors: torch.Tensor) -> bool:
        return all(t.is_cuda and t.dtype == torch.float32 for t in tensors) and not (
            torch.is_grad_enabled() and any(t.requires_grad for t in tensors))

    def modified_gram_schmidt(self, vectors: torch.Tensor) -> torch.Tensor:
        if self._fused(vectors):
            return ops.modified_gram_schmidt(vectors)
        v1 = vectors[:, 0] / torch.norm(vectors[:, 0], dim=1, keepdim=True)
        v2 = vectors[:, 1] - torch.sum(vectors[:, 1] * v1, dim=1, keepdim=True) * v1
        v2 = v2 / torch.norm(v2, dim=1, keepdim=True)
        v3 = vectors[:, 2] - torch.sum(vectors[:, 2] * v1, dim=1, keepdim=True) * v1
        v3 = v3 - torch.sum(v3 * v2, dim=1, keepdim=True) * v2
        v3 = v3 / torch.norm(v3, dim=1, keepdim=True)
        return torch.stack([v1, v2, v3], dim=1)

    def get_groupelement(self, nodes, loc, edges, vel, edge_attr, charges) -> Dict[str, torch.Tensor]:
        rotation_vectors, translation_vectors = self.canonicalization_network(nodes, loc, edges, vel, edge_attr, charges)
        rotation_matrix = self.modified_gram_schmidt(rotation_vectors)
        if not hasattr(self, "canonicalization_info_dict"):
            self.canonicalization_info_dict = {}
        element = {"rotation_matrix": rotation_matrix, "translation_vectors": translation_vectors,
                   "rotation_matrix_inverse": rotation_matrix.transpose(1, 2)}
        self.canonicalization_info_dict["group_element"] = element
        return element

    def canonicalize(self, x: torch.Tensor, targets: Optional[List] = None, **kwargs: Any
                     ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        self.device = x.device
        loc, edges, vel, edge_attr, charges = kwargs.values()
        el = self.get_groupelement(x, loc, edges, vel, edge_attr, charges)
        R, t = el["rotation_matrix"], el["translation_vectors"]
        if self._fused(loc, vel, R, t):
            return ops.rigid_rows(loc, R, t, inverse=True), ops.rigid_rows(vel, R, None, inverse=True)
        Rinv = el["rotation_matrix_inverse"]
        canonical_loc = torch.bmm(loc[:, None, :], Rinv).squeeze() - torch.bmm(t[:, None, :], Rinv).squeeze()
        return canonical_loc, torch.bmm(vel[:, None, :], Rinv).squeeze()

    def invert_canonicalization(self, x_canonicalized_out: torch.Tensor, **kwargs: Any) -> torch.Tensor:
        R, t, _ = self.canonicalization_info_dict["group_element"].values()
        if self._fused(x_canonicalized_out, R, t):
            return ops.rigid_rows(x_canonicalized_out, R, t, inverse=False)
        return torch.bmm(x_canonicalized_out[:, None, :], R).squeeze() + t
