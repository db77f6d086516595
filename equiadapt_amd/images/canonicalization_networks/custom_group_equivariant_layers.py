"""C_n / D_n lifting and group convolutions with the reference's parameter shapes and semantics.

Reference: equiadapt/images/canonicalization_networks/custom_group_equivariant_layers.py.
The reference rebuilds the expanded filter bank on every forward by *resampling* the k x k filters with
``kornia.geometry.rotate`` (bilinear!) and ``hflip``.  Rotating a k x k image bilinearly is a fixed
linear map on R^(k*k); here that map is computed ONCE per (k, group) by pushing the k*k unit images
through the HIP resampling kernel (``eqa_orbit_expand_fwd``), giving E matrices A_e (k^2 x k^2).  The
expanded bank is then one tiny einsum -- differentiable, no resampling in the training loop, and
bit-compatible with what the kernel (hence the reference op sequence) produces.
Parameter names/shapes (``weights``, ``bias``) match the reference so its checkpoints load.
"""
import math
from typing import Dict, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from equiadapt_amd import ops
from equiadapt_amd.images import geometry

_bank_cache: Dict[tuple, torch.Tensor] = {}


def filter_action_matrices(kernel_size: int, num_rotations: int, reflections: bool, device: torch.device) -> torch.Tensor:
    """(E, k*k, k*k): A_e maps a flattened filter to ``[hflip](rotate(filter, +angle_e))``.

    rotate = kornia semantics on a (k, k) frame, zero fill (custom_group_equivariant_layers.py:77-83,
    :184-190: rotation first, then hflip for the reflected half).
    """
    key = (kernel_size, num_rotations, reflections, str(device))
    hit = _bank_cache.get(key)
    if hit is not None:
        return hit
    k, N = kernel_size, num_rotations
    E = 2 * N if reflections else N
    if k == 1:  # a 1x1 filter is its own rotation / reflection
        mats = torch.ones(E, 1, 1, device=device)
    else:
        ang = torch.linspace(0.0, 360.0, steps=N + 1, dtype=torch.float32)[:N]
        theta = geometry.rotation_theta(ang, (k, k))
        flags = torch.zeros(N, dtype=torch.int32)
        if reflections:
            theta = torch.cat([theta, theta], dim=0)
            flags = torch.cat([flags, torch.full((N,), geometry.FLIP_DST, dtype=torch.int32)])
        basis = torch.eye(k * k, device=device).view(1, k * k, k, k)
        out = ops.orbit_expand(basis, theta.to(device), flags.to(device), 0)  # (E, k*k [q], k, k [p])
        mats = out.reshape(E, k * k, k * k).transpose(1, 2).contiguous()      # A[e, p, q]
    _bank_cache[key] = mats
    return mats


def _slot_table(num_rotations: int, reflections: bool) -> torch.Tensor:
    """src_slot[e, m]: which input group slot of the stored weight feeds slot m of output element e.

    rotation: (m - e) mod N (reference :283-293).  roto-reflection (reference :430-456):
      e <  N : [ (m - e) mod N | N + (m' + e) mod N ]
      e >= N : [ N + (m + n) mod N | (m' - n) mod N ]   with n = e - N.
    """
    N = num_rotations
    m = torch.arange(N)
    n = torch.arange(N)[:, None]
    fwd = (m[None, :] - n) % N
    if not reflections:
        return fwd
    inv = (m[None, :] + n) % N
    upper = torch.cat([fwd, inv + N], dim=1)
    lower = torch.cat([inv + N, fwd], dim=1)
    return torch.cat([upper, lower], dim=0)


class _GroupConvBase(nn.Module):
    reflections = False
    lifting = True

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, num_rotations: int = 4,
                 stride: int = 1, padding: int = 0, bias: bool = True, device: str = "cuda"):
        super().__init__()
        E = 2 * num_rotations if self.reflections else num_rotations
        shape = (out_channels, in_channels, kernel_size, kernel_size) if self.lifting else \
            (out_channels, in_channels, E, kernel_size, kernel_size)
        self.weights = nn.Parameter(torch.empty(*shape, device=device))
        nn.init.kaiming_uniform_(self.weights, a=math.sqrt(5))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels, device=device))
        else:
            self.bias = None  # type: ignore
        self.in_channels, self.out_channels = in_channels, out_channels
        self.stride, self.padding = stride, padding
        self.num_rotations, self.kernel_size = num_rotations, kernel_size
        self.num_group_elements = E
        if not self.lifting:
            self.register_buffer("src_slot", _slot_table(num_rotations, self.reflections), persistent=False)
        self._cached_bank: Tuple[int, torch.Tensor] = (-1, None)  # type: ignore

    def expanded_weights(self) -> torch.Tensor:
        """The dense conv filter bank: (O*E, I, k, k) for lifting, (O*E, I*E, k, k) otherwise."""
        w = self.weights
        if not self.training and not torch.is_grad_enabled():
            ver, bank = self._cached_bank
            if ver == w._version and bank is not None and bank.device == w.device:
                return bank
        O, I, E, k = self.out_channels, self.in_channels, self.num_group_elements, self.kernel_size
        A = filter_action_matrices(k, self.num_rotations, self.reflections, w.device).to(w.dtype)  # (E, p, q)
        if self.lifting:
            bank = torch.einsum("epq,oiq->oeip", A, w.reshape(O, I, k * k)).reshape(O * E, I, k, k)
        else:
            wp = w.reshape(O, I, E, k * k)[:, :, self.src_slot]            # (O, I, E[e], E[m], q)
            bank = torch.einsum("epq,oiemq->oeimp", A, wp).reshape(O * E, I * E, k, k)
        if not self.training and not torch.is_grad_enabled():
            self._cached_bank = (w._version, bank)
        return bank

    def mean_response_weights(self) -> torch.Tensor:
        """(E, Cin*k*k) fp64: the expanded bank summed over output fields (see pooling.conv_then_group_pool)."""
        w = self.weights
        ver, hit = getattr(self, "_cached_weff", (-1, None))
        if ver == w._version and hit is not None and hit.device == w.device:
            return hit
        bank = self.expanded_weights().detach()
        weff = bank.view(self.out_channels, self.num_group_elements, -1).double().sum(0)
        self._cached_weff = (w._version, weff)
        return weff

    def mean_bias_value(self) -> float:
        """mean(bias) as a Python float, cached per parameter version (inference: one host read when the weights change,
        then no launches at all for the bias of the linearised last layer)."""
        if self.bias is None:
            return 0.0
        ver, hit = getattr(self, "_cached_bias_mean", (-1, 0.0))
        if ver != self.bias._version:
            hit = float(self.bias.detach().double().mean().item())
            self._cached_bias_mean = (self.bias._version, hit)
        return hit

    def supports_linear_tail(self) -> bool:
        return self.stride == 1 and self.padding == 0 and self.kernel_size <= ops.MAX_WINDOW_K

    def mfma_lifting_ok(self, x: torch.Tensor) -> bool:
        """Inference on the hand-written fp32-MFMA lifting convolution (eqa_lift_conv_nhwc) instead of the framework's
        convolution: a Z2 -> G lifting layer with stride 1, no padding, on a shape the kernel takes (kernel rows of 9..15 floats,
        k in {3, 5}, O * |G| a multiple of 16)."""
        return (self.lifting and x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled() and self.stride == 1
                and self.padding == 0 and x.dim() == 4 and x.shape[-2] >= self.kernel_size and x.shape[-1] >= self.kernel_size
                and (ops.lift_conv_supported(self.in_channels, self.kernel_size, self.kernel_size, self.out_channels * self.num_group_elements)
                     or ops.lift_conv_wide_supported(self.in_channels, self.kernel_size, self.kernel_size, self.out_channels * self.num_group_elements)))

    def lift_nhwc(self, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
        """[relu](lifting convolution + bias) as a channels-last (B, O*|G|, H', W') tensor (channel = field * |G| + element): the
        layout the window-sum kernels read, without the 5-D reshape of ``forward``."""
        w, b = self.weights, self.bias
        key = (w._version, -1 if b is None else b._version, str(w.device))
        hit = getattr(self, "_cached_lift", None)
        if hit is None or hit[0] != key:
            narrow = ops.lift_conv_supported(self.in_channels, self.kernel_size, self.kernel_size, self.out_channels * self.num_group_elements)
            wpk = (ops.pack_lift_weights if narrow else ops.pack_lift_weights_wide)(self.expanded_weights().detach())
            be = None if b is None else b.detach().repeat_interleave(self.num_group_elements).contiguous()
            hit = (key, wpk, be, narrow)
            self._cached_lift = hit
        fn = ops.lift_conv_nhwc if hit[3] else ops.lift_conv_wide
        return fn(x.contiguous(memory_format=torch.channels_last), hit[1], hit[2], relu, self.kernel_size, self.kernel_size)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B = x.shape[0]
        if self.mfma_lifting_ok(x):
            y = self.lift_nhwc(x)
            return y.reshape(B, self.out_channels, self.num_group_elements, y.shape[2], y.shape[3])   # (contiguous 5-D copy)
        if not self.lifting:
            x = x.flatten(1, 2)
        x = F.conv2d(x, self.expanded_weights(), stride=self.stride, padding=self.padding)
        x = x.reshape(B, self.out_channels, self.num_group_elements, x.shape[2], x.shape[3])
        if self.bias is not None:
            x = x + self.bias[None, :, None, None, None]
        return x


class RotationEquivariantConvLift(_GroupConvBase):
    """Z2 -> C_n lifting convolution; weights (O, I, k, k) (reference :9-111)."""
    reflections, lifting = False, True


class RotoReflectionEquivariantConvLift(_GroupConvBase):
    """Z2 -> D_n lifting convolution; weights (O, I, k, k) (reference :114-226)."""
    reflections, lifting = True, True


class RotationEquivariantConv(_GroupConvBase):
    """C_n -> C_n group convolution; weights (O, I, N, k, k) (reference :229-361)."""
    reflections, lifting = False, False


class RotoReflectionEquivariantConv(_GroupConvBase):
    """D_n -> D_n group convolution; weights (O, I, 2N, k, k) (reference :364-538)."""
    reflections, lifting = True, False
