"""Oracle for the image canonicalization networks (rows I2b, I10).  TEST INFRASTRUCTURE ONLY.

Functional restatements driven by a ``state_dict`` (keys as in the reference modules):
the expanded filter banks are built the reference's way -- by RESAMPLING the stored filters with
``kornia_rotate`` / ``kornia_hflip`` on every call -- then one ``F.conv2d`` per layer.
Parity unpinned at the kornia boundary (see oracle/__init__.py).
"""
from typing import Dict

import torch
import torch.nn.functional as F

from oracle.image_ops import kornia_hflip, kornia_rotate


def _angles(n: int) -> torch.Tensor:
    return torch.linspace(0.0, 360.0, steps=n + 1, dtype=torch.float32)[:n]


def rotation_lift_bank(w: torch.Tensor, N: int) -> torch.Tensor:
    """custom_group_equivariant_layers.py:62-90: (O,I,k,k) -> (O*N, I, k, k), out channel o*N + n."""
    O, I, k, _ = w.shape
    stack = w.flatten(0, 1).unsqueeze(0).repeat(N, 1, 1, 1)
    rot = kornia_rotate(stack, _angles(N))
    return rot.reshape(N, O, I, k, k).transpose(0, 1).flatten(0, 1)


def rotoreflection_lift_bank(w: torch.Tensor, N: int) -> torch.Tensor:
    """:168-199: rotations, then their h-flips; (O,I,k,k) -> (O*2N, I, k, k)."""
    O, I, k, _ = w.shape
    stack = w.flatten(0, 1).unsqueeze(0).repeat(N, 1, 1, 1)
    rot = kornia_rotate(stack, _angles(N))
    both = torch.cat([rot, kornia_hflip(rot)], dim=0)
    return both.reshape(2 * N, O, I, k, k).transpose(0, 1).flatten(0, 1)


def rotation_conv_bank(w: torch.Tensor, N: int) -> torch.Tensor:
    """:298-334: W[(o,n),(i,m)] = rot_n(w[o,i,(m-n) mod N]); (O,I,N,k,k) -> (O*N, I*N, k, k)."""
    O, I, _, k, _ = w.shape
    idx = torch.arange(N).view(1, 1, N, 1, 1).repeat(N, O * I, 1, k, k)
    idx = (idx - torch.arange(N)[:, None, None, None, None]) % N
    stack = w.flatten(0, 1).unsqueeze(0).repeat(N, 1, 1, 1, 1)
    perm = torch.gather(stack, 2, idx)
    rot = kornia_rotate(perm.flatten(1, 2), _angles(N))
    return rot.reshape(N, O, I, N, k, k).transpose(0, 1).reshape(O * N, I * N, k, k)


def rotoreflection_conv_bank(w: torch.Tensor, N: int) -> torch.Tensor:
    """:430-516: (O,I,2N,k,k) -> (O*2N, I*2N, k, k); lower half h-flipped after the rotation."""
    O, I, E, k, _ = w.shape
    base = torch.arange(N).view(1, 1, N, 1, 1).repeat(N, O * I, 1, k, k)
    step = torch.arange(N)[:, None, None, None, None]
    fwd, inv = (base - step) % N, (base + step) % N
    upper = torch.cat([fwd, inv + N], dim=2)
    lower = torch.cat([inv + N, fwd], dim=2)
    idx = torch.cat([upper, lower], dim=0)
    stack = w.flatten(0, 1).unsqueeze(0).repeat(E, 1, 1, 1, 1)
    perm = torch.gather(stack, 2, idx)
    rot = kornia_rotate(perm.flatten(1, 2), torch.cat([_angles(N), _angles(N)]))
    both = torch.cat([rot[:N], kornia_hflip(rot[N:])])
    return both.reshape(E, O, I, E, k, k).transpose(0, 1).reshape(O * E, I * E, k, k)


def custom_equivariant_network(x: torch.Tensor, sd: Dict[str, torch.Tensor], group_type: str, N: int,
                               num_layers: int) -> torch.Tensor:
    """CustomEquivariantNetwork.forward (custom_equivariant_networks.py:80-93) -> (B, G).

    Layer keys: eqv_network.0.{weights,bias}, then eqv_network.{2,4,...} for the 1x1 group convs.
    """
    refl = group_type == "roto-reflection"
    E = 2 * N if refl else N
    B = x.shape[0]
    w, b = sd["eqv_network.0.weights"], sd["eqv_network.0.bias"]
    bank = rotoreflection_lift_bank(w, N) if refl else rotation_lift_bank(w, N)
    h = F.conv2d(x, bank)
    h = h.reshape(B, w.shape[0], E, h.shape[2], h.shape[3]) + b[None, :, None, None, None]
    for layer in range(1, num_layers):
        h = F.relu(h)
        w, b = sd[f"eqv_network.{2 * layer}.weights"], sd[f"eqv_network.{2 * layer}.bias"]
        bank = rotoreflection_conv_bank(w, N) if refl else rotation_conv_bank(w, N)
        h = F.conv2d(h.flatten(1, 2), bank)
        h = h.reshape(B, w.shape[0], E, h.shape[2], h.shape[3]) + b[None, :, None, None, None]
    return torch.mean(h, dim=(1, 3, 4))


def escnn_like_network(x: torch.Tensor, sd: Dict[str, torch.Tensor], group_type: str, N: int, num_layers: int,
                       out_channels: int) -> torch.Tensor:
    """Eval-mode forward of this repo's ESCNNEquivariantNetwork stand-in (see its docstring: e2cnn's basis
    expansion is NOT restated; the layer sequence / shapes follow escnn_networks.py:67-117).

    conv(k) -> [InnerBN -> ReLU -> (dropout: identity in eval) -> conv(k)] x (L-1) -> mean over (fields, H, W).
    Filter banks are rebuilt by resampling on every call, like e2cnn re-expands its basis in training.
    """
    refl = group_type == "roto-reflection"
    E = 2 * N if refl else N
    B = x.shape[0]
    conv_keys = sorted({int(k.split(".")[1]) for k in sd if k.endswith(".weights")})
    h = x
    for li, idx in enumerate(conv_keys):
        w, b = sd[f"eqv_network.{idx}.weights"], sd[f"eqv_network.{idx}.bias"]
        if li == 0:
            bank = rotoreflection_lift_bank(w, N) if refl else rotation_lift_bank(w, N)
            h = F.conv2d(h, bank)
        else:
            bank = rotoreflection_conv_bank(w, N) if refl else rotation_conv_bank(w, N)
            h = F.conv2d(h.flatten(1, 2), bank)
        h = h.reshape(B, out_channels, E, h.shape[2], h.shape[3]) + b[None, :, None, None, None]
        if li < len(conv_keys) - 1:
            p = f"eqv_network.{idx + 1}."
            h = F.batch_norm(h, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                             False, 0.9, 1e-5)
            h = F.relu(h)
    return torch.mean(h, dim=(1, 3, 4))


def conv_network(x: torch.Tensor, sd: Dict[str, torch.Tensor], num_layers: int, training: bool = False,
                 dropout_p: float = 0.5, momentum: float = 0.1, eps: float = 1e-5) -> torch.Tensor:
    """ConvNetwork.forward (custom_nonequivariant_networks.py:19-80), functional on a reference-named ``state_dict``
    (enc_network.{3i}.{weight,bias}, enc_network.{3i+1}.{weight,bias,running_mean,running_var}, final_fc.0.*, final_fc.3.*).

    :44-57  layer i: Conv2d(stride 2; padding 1 exactly when i % 3 == 2, else 0) -> BatchNorm2d -> GELU (erf form);
    :70-80  reshape(B, -1) -> BatchNorm1d -> Dropout1d(0.5) -> ReLU -> Linear.
    ``training`` uses batch statistics and updates the running statistics IN ``sd`` (momentum 0.1, unbiased variance), like
    the modules; Dropout1d on a 2-D (B, D) input is the reference's call as written: torch treats it as one unbatched
    (C, L) sample and zeroes whole ROWS (samples) -- pass ``dropout_p=0`` for a deterministic train-mode comparison."""
    h = x
    for i in range(num_layers):
        c, b = f"enc_network.{3 * i}.", f"enc_network.{3 * i + 1}."
        h = F.conv2d(h, sd[c + "weight"], sd.get(c + "bias"), stride=2, padding=1 if i % 3 == 2 else 0)
        h = F.batch_norm(h, sd[b + "running_mean"], sd[b + "running_var"], sd[b + "weight"], sd[b + "bias"], training, momentum, eps)
        if training and b + "num_batches_tracked" in sd:
            sd[b + "num_batches_tracked"] += 1
        h = F.gelu(h)
    h = h.reshape(x.shape[0], -1)
    h = F.batch_norm(h, sd["final_fc.0.running_mean"], sd["final_fc.0.running_var"], sd["final_fc.0.weight"], sd["final_fc.0.bias"],
                     training, momentum, eps)
    if training and "final_fc.0.num_batches_tracked" in sd:
        sd["final_fc.0.num_batches_tracked"] += 1
    h = F.dropout1d(h, dropout_p, training)
    h = F.relu(h)
    return F.linear(h, sd["final_fc.3.weight"], sd["final_fc.3.bias"])
