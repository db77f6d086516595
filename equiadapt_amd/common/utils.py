"""Shared utilities on the hot path (reference: equiadapt/common/utils.py:22-51)."""
import torch

from equiadapt_amd import ops


class _GramSchmidtFn(torch.autograd.Function):
    """Forward: eqa_gram_schmidt kernel.  Backward: eqa_gram_schmidt_bwd, the analytic derivative of the same three steps."""

    @staticmethod
    def forward(ctx, vectors: torch.Tensor) -> torch.Tensor:
        ctx.save_for_backward(vectors)
        return ops.gram_schmidt(vectors)

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        (vectors,) = ctx.saved_tensors
        return ops.gram_schmidt_backward(vectors, grad_out)


def gram_schmidt(vectors: torch.Tensor) -> torch.Tensor:
    """Batched classical Gram-Schmidt of three 3-vectors: (B, 3, 3) -> orthonormal rows.

    No epsilon and no handedness fix: the result is in O(3), determinant may be -1, like the reference.
    """
    return _GramSchmidtFn.apply(vectors)


def update_running_stats(bn: torch.nn.modules.batchnorm._BatchNorm, mean: torch.Tensor, var_unbiased: torch.Tensor) -> None:
    """What nn.BatchNorm*d does to its buffers in a training-mode forward: num_batches_tracked += 1, then an exponential
    average with `momentum` (or the cumulative average 1/num_batches_tracked when momentum is None).  No-op when the module
    does not track running statistics."""
    if not bn.track_running_stats or bn.running_mean is None:
        return
    with torch.no_grad():
        bn.num_batches_tracked += 1
        m = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
        bn.running_mean.mul_(1 - m).add_(m * mean.to(bn.running_mean.dtype))
        bn.running_var.mul_(1 - m).add_(m * var_unbiased.to(bn.running_var.dtype))


def mark_written(*tensors: torch.Tensor) -> None:
    """Bump the autograd version counter of buffers that a kernel has written through their raw pointers (the running
    statistics that eqa_vn_bn_finalize updates).  An in-place torch op would do this by itself; without it, caches keyed on
    ``_version`` (VNSmall.packed_parameters) and autograd's saved-tensor checks would not see the write."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)
