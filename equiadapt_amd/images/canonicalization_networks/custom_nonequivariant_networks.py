"""ConvNetwork: the plain strided encoder the optimized canonicalizer scores group views with.

Reference: equiadapt/images/canonicalization_networks/custom_nonequivariant_networks.py:8-80.
Small MIOpen convolutions (PyTorch-ROCm); module / parameter names match the reference's state_dict.
The pretrained torchvision wrappers (ResNet18Network, WideResNet*) are out of scope.
"""
from typing import List

import torch
from torch import nn


class ConvNetwork(nn.Module):
    def __init__(self, in_shape: tuple, out_channels: int, kernel_size: int, num_layers: int = 2,
                 out_vector_size: int = 128):
        super().__init__()
        chans = in_shape[0]
        blocks: List[nn.Module] = []
        for i in range(num_layers):
            widen = i > 0 and i % 3 == 2          # every third layer doubles the width (and pads by 1)
            nxt = 2 * out_channels if widen else out_channels
            blocks.append(nn.Conv2d(chans, nxt, kernel_size, 2, 1 if widen else 0))
            blocks += [nn.BatchNorm2d(nxt), nn.GELU()]
            chans = out_channels = nxt
        self.enc_network = nn.Sequential(*blocks)
        with torch.no_grad():
            was = self.enc_network.training
            probe = self.enc_network(torch.zeros(1, *in_shape)).shape   # same dry run as the reference ctor
            self.enc_network.train(was)
        out_dim = probe[1] * probe[2] * probe[3]
        self.final_fc = nn.Sequential(nn.BatchNorm1d(out_dim), nn.Dropout1d(0.5), nn.ReLU(),
                                      nn.Linear(out_dim, out_vector_size))
        self.out_vector_size = out_vector_size

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.final_fc(self.enc_network(x).reshape(x.shape[0], -1))
